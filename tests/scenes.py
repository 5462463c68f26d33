"""Seeded synthetic scenes / cameras for the parity tests and a helper that
runs the whole hot path through the CPU oracle.  numpy only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
# The scene / camera generators are also used by bench.py and the tools for their workloads: the oracle is
# imported only inside oracle_geometry (the checker), never by merely building a scene.


def look_at(pos, at=(0.0, 0.0, 0.0), up=(0.0, 0.0, 1.0)):
    """data/__init__.py:14-29 get_c2w_from_up_and_look_at"""
    pos, at, up = (np.asarray(v, np.float64) for v in (pos, at, up))
    up = up / np.linalg.norm(up)
    z = at - pos
    z = z / np.linalg.norm(z)
    y = -up
    x = np.cross(y, z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    c2w = np.zeros((3, 4), np.float32)
    c2w[:, 0], c2w[:, 1], c2w[:, 2], c2w[:, 3] = x, y, z, pos
    return c2w


def orbit(dist, elev_deg, azim_deg):
    e, a = np.deg2rad(elev_deg), np.deg2rad(azim_deg)
    return look_at((dist * np.cos(e) * np.cos(a), dist * np.cos(e) * np.sin(a), dist * np.sin(e)))


class Camera:
    def __init__(self, W, H, fx=None, fy=None, cx=None, cy=None, near=0.01, far=100.0, c2w=None):
        self.w, self.h = int(W), int(H)
        self.fx = float(fx if fx is not None else W)
        self.fy = float(fy if fy is not None else (fx if fx is not None else W))
        self.cx = float(cx if cx is not None else W / 2.0)
        self.cy = float(cy if cy is not None else H / 2.0)
        self.near, self.far = float(near), float(far)
        self.c2w = c2w if c2w is not None else look_at((2.5, 0.0, 0.0))

    @property
    def intr(self):
        return (self.fx, self.fy, self.cx, self.cy, self.w, self.h, self.near, self.far)

    @property
    def topleft(self):  # gs/gaussian_splatting.py:1274-1276
        return np.array([-self.cx / self.fx, -self.cy / self.fy], np.float32)

    @property
    def tiles(self):
        return (self.h + 15) // 16, (self.w + 15) // 16


def random_scene(N, seed=0, svec=0.02, svec_sigma=0.3, spread=0.8, C=1):
    """cfg1-style random Gaussians (post-activation parameters)."""
    rng = np.random.default_rng(seed)
    s = {}
    s["mean"] = rng.normal(0, spread, (N, 3)).astype(np.float32)
    q = rng.normal(0, 1, (N, 4))
    s["qvec"] = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    s["svec"] = np.exp(rng.normal(np.log(svec), svec_sigma, (N, 3))).astype(np.float32)
    s["alpha"] = rng.uniform(0.1, 0.99, N).astype(np.float32)
    s["color"] = rng.uniform(0, 1, (N, 3)).astype(np.float32)
    sh = rng.normal(0, 0.3, (N, 3, C * C)).astype(np.float32)
    col = np.clip(s["color"], 1e-3, 1 - 1e-3)
    sh[:, :, 0] = (np.log(col / (1 - col)) / 0.28209479177387814).astype(np.float32)
    s["sh"] = np.ascontiguousarray(sh)
    s["C"] = C
    return s


def pointe_scene(N, seed=0, svec=0.02, C=4):
    """cfg2-style "Point-E init" cloud: 4096 points uniform in a 0.8 ball + a
    N(0,0.8^2) halo, isotropic svec, near-identity rotations, alpha 0.8, degree-3 SH."""
    rng = np.random.default_rng(seed)
    nb = min(4096, N)
    d = rng.normal(size=(nb, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ball = d * (rng.uniform(size=(nb, 1)) ** (1 / 3.0)) * 0.8
    halo = rng.normal(0, 0.8, (N - nb, 3))
    mean = np.concatenate([ball, halo]).astype(np.float32)
    mean -= mean.mean(0, keepdims=True)
    s = {"mean": mean}
    q = np.array([1.0, 0, 0, 0]) + rng.normal(0, 0.1, (N, 4))
    s["qvec"] = q.astype(np.float32)
    s["svec"] = np.full((N, 3), svec, np.float32)
    s["alpha"] = np.full(N, 0.8, np.float32)
    col = rng.uniform(0.02, 0.98, (N, 3))
    s["color"] = col.astype(np.float32)
    sh = rng.normal(0, 0.1, (N, 3, C * C)).astype(np.float32)
    sh[:, :, 0] = (np.log(col / (1 - col)) / 0.28209479177387814).astype(np.float32)
    s["sh"] = np.ascontiguousarray(sh)
    s["C"] = C
    return s


def densified_scene(N, seed=0, C=4):
    """cfg3-style post-densify cloud: svec = 0.01*exp(N(0,0.5^2)), alpha ~ U(0.05,1)."""
    s = pointe_scene(N, seed, 0.01, C)
    rng = np.random.default_rng(seed + 1)
    s["svec"] = (0.01 * np.exp(rng.normal(0, 0.5, (N, 3)))).astype(np.float32)
    s["alpha"] = rng.uniform(0.05, 1.0, N).astype(np.float32)
    return s


def oracle_geometry(scene, cam, frustum_radius=6.0, tile_radius=6.0):
    """cull -> (gather) -> project -> aabb -> bin/sort through the oracle, on the COMPACTED
    (post-mask) Gaussians exactly as gs/gaussian_splatting.py:1208-1295 does."""
    from oracle import oracle as O
    normals, pts = O.frustum(cam.c2w, *cam.intr)
    if frustum_radius > 0:
        mask = O.cull_bsphere(scene["mean"], scene["svec"], normals, pts, frustum_radius)
    else:
        mask = np.ones(scene["mean"].shape[0], bool)
    g = {"mask": mask, "normals": normals, "pts": pts}
    m, q, s = scene["mean"][mask], scene["qvec"][mask], scene["svec"][mask]
    g["mean2d"], g["cov2d"], g["JW"], g["depth"] = O.project(m, q, s, cam.c2w)
    D, tl, br = O.aabb_count(g["mean2d"], g["cov2d"], 16, cam.fx, cam.fy, cam.cx, cam.cy, cam.w, cam.h,
                             tile_radius)
    nth, ntw = cam.tiles
    g["D"], g["tl"], g["br"] = D, tl, br
    g["ids"], g["start"], g["end"] = O.bin_sort(tl, br, g["depth"], nth, ntw, D)
    return g


# what the parity helpers saw, printed at the end of the run (tests/conftest.py: pytest_terminal_summary) -- e.g. how many
# threshold-adjacent pixels a frame really had (the docs say "none observed": this records it in the GPU log)
PARITY_LOG = []


def assert_sh_image_parity(img, ref, mean2d, cov2d, alpha, start, end, ids, topleft, psx, psy, tol=1e-4,
                           max_exceptions=2, what=""):
    """north_star: every pixel within `tol` of the oracle.  The one legitimate exception is named, not waved
    through: a pixel whose a*G came within a few fp32 ulps of the 1/255 skip threshold (or whose T came that close
    to the stop threshold) in the ORACLE's own evaluation -- there two correct fp32 evaluations of the same Gaussian
    (different exp implementations) may decide differently, and the decision moves the pixel by up to 1/255.
    Returns the number of such pixels (normally 0)."""
    from oracle import oracle as O
    img, ref = np.asarray(img), np.asarray(ref)
    err = np.abs(img - ref).max(-1)
    bad = np.argwhere(err > tol)
    PARITY_LOG.append(f"{what or 'frame'} {err.shape[1]}x{err.shape[0]}: max |pixel - oracle| = {err.max():.2e}, "
                      f"threshold-adjacent exception pixels = {len(bad)}")
    if len(bad) == 0:
        return 0
    H, W = err.shape
    margin = O.sh_decision_margin(mean2d, cov2d, alpha, start, end, ids, topleft, psx, psy, H, W)
    for y, x in bad:
        assert margin[y, x].min() <= 4e-7, (f"{what}: pixel ({y},{x}) off by {err[y, x]:.3e} with no decision within a "
                                            f"few ulps of its threshold (skip margin {margin[y, x, 0]:.2e}, stop margin "
                                            f"{margin[y, x, 1]:.2e})")
        assert err[y, x] <= 1.0 / 255.0 + tol, (what, y, x, err[y, x])
    assert len(bad) <= max_exceptions, f"{what}: {len(bad)} threshold-adjacent pixels: {bad[:8].tolist()}"
    return len(bad)


def per_gaussian_grad_error(got, want):
    """Row-wise (per Gaussian) error of a gradient tensor in units of its tolerance (rtol 1e-3, atol 1e-5 per
    Gaussian):  max over rows i of  max_j |got_ij - want_ij| / (1e-3 max_j |want_ij| + 1e-5 max |want|).
    <= 1 passes.  A Gaussian whose gradient is a thousandth of the largest is still held to 1 % of ITS OWN magnitude --
    normalising by the tensor's global maximum would let it be entirely wrong."""
    a = np.asarray(got, np.float64).reshape(len(want), -1)
    b = np.asarray(want, np.float64).reshape(len(want), -1)
    tol = 1e-3 * np.abs(b).max(1) + 1e-5 * np.abs(b).max()
    ratio = np.abs(a - b).max(1) / np.maximum(tol, 1e-300)
    return float(ratio.max()), int(ratio.argmax())
