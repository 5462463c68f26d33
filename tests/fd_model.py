"""An fp64 numpy statement of the differentiable path for FINITE-DIFFERENCE gradient checks, tiny scenes only.

What is differentiated is what the reference's autograd differentiates (gs/renderer.py:366-421, vol_render_sh.h:171-455):
  * projection: pm = R^T (mean - t); M = Rq(qvec) * svec (columns scaled, utils/transforms.py:34-46); Sigma = M M^T;
    cov2d = (J W Sigma W^T J^T)[:2,:2] with the Jacobian J a CONSTANT (gs/renderer.py:366-378 is @torch.no_grad, :405);
    mean2d = pm.xy / pm.z, the denominator detached iff detach_depth (:413-418);
  * compositing: front to back over the tile's depth-sorted list, G = exp(-0.5 d^T cov2d^-1 d) (kernels.h:172-193),
    a = min(alpha, 0.99), skip where a G < 1/255, stop once T < thresh, colour = sigmoid(sh . Y(dir)) with
    dir = normalize(R (qx, qy, 1)) per pixel (vol_render_sh.h:48-65, shencoder.h:13-62), out = sum + bg T.
Every DISCRETE decision -- the tile lists and their order, "skip", "stopped" -- is frozen at the base point (`Frozen`), exactly
as the analytic backward treats it; `margins` reports how far the base point is from each discontinuity so that a test can
insist on a smooth neighbourhood.  Nothing here is used by the product."""
import numpy as np

MIN_ALPHA = 1.0 / 255.0
ALPHA_CLAMP = 0.99


def quat_to_rot(q):
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((q.shape[0], 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def jacobian(u):
    l = np.linalg.norm(u, axis=-1)
    J = np.zeros((u.shape[0], 3, 3))
    J[:, 0, 0] = 1 / u[:, 2]; J[:, 2, 0] = u[:, 0] / l
    J[:, 1, 1] = 1 / u[:, 2]; J[:, 2, 1] = u[:, 1] / l
    J[:, 0, 2] = -u[:, 0] / u[:, 2] ** 2; J[:, 1, 2] = -u[:, 1] / u[:, 2] ** 2; J[:, 2, 2] = u[:, 2] / l
    return J


def sh_basis(d, C):
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    Y = [np.full(x.shape, 0.28209479177387814)]
    if C >= 2:
        Y += [-0.48860251190291987 * y, 0.48860251190291987 * z, -0.48860251190291987 * x]
    if C >= 3:
        Y += [1.0925484305920792 * x * y, -1.0925484305920792 * y * z, 0.94617469575755997 * z * z - 0.31539156525251999,
              -1.0925484305920792 * x * z, 0.54627421529603959 * (x * x - y * y)]
    if C >= 4:
        Y += [0.59004358992664352 * y * (-3 * x * x + y * y), 2.8906114426405538 * x * y * z,
              0.45704579946446572 * y * (1 - 5 * z * z), 0.3731763325901154 * z * (5 * z * z - 3),
              0.45704579946446572 * x * (1 - 5 * z * z), 1.4453057213202769 * z * (x * x - y * y),
              0.59004358992664352 * x * (-x * x + 3 * y * y)]
    return np.stack(Y, -1)


class Frozen:
    """the discrete state of the base point: Jacobians, detached depths, per-tile lists, per-(pixel, entry) masks"""

    def __init__(self):
        self.J = self.depth = None
        self.lists = None      # per tile: int array of Gaussian indices, front to back
        self.take = {}         # tile -> bool [pixels, entries]: alive and not skipped
        self.margins = {"skip": np.inf, "stop": np.inf}


def forward(P, cam, C, go, bg=None, thresh=1e-4, detach_depth=False, frozen=None, lists=None):
    """P: dict of fp64 arrays mean [N,3], qvec [N,4], svec [N,3], alpha [N], and sh [N,3,C*C] (C >= 1) or color [N,3]
    (C == 0).  go: [H,W,3].  First call: frozen=None, lists = (start, end, ids) of the geometry stage -> (loss, image,
    Frozen).  Later calls: frozen=<that> -> (loss, image, frozen)."""
    c2w = np.asarray(cam.c2w, np.float64)
    R, t = c2w[:, :3], c2w[:, 3]
    mean, qvec, svec, alpha = P["mean"], P["qvec"], P["svec"], P["alpha"]
    pm = (mean - t) @ R
    first = frozen is None
    if first:
        frozen = Frozen()
        frozen.J = jacobian(pm)
        frozen.depth = pm[:, 2].copy()
        start, end, ids = lists
        frozen.lists = [np.asarray(ids[s:e], np.int64) if s >= 0 else np.zeros(0, np.int64) for s, e in zip(start, end)]
    M = quat_to_rot(qvec) * svec[:, None, :]
    sigma = M @ M.transpose(0, 2, 1)
    JW = frozen.J @ R.T
    cov = (JW @ sigma @ JW.transpose(0, 2, 1))[:, :2, :2]
    den = frozen.depth if detach_depth else pm[:, 2]
    m2 = pm[:, :2] / den[:, None]
    H, W = cam.h, cam.w
    nth, ntw = cam.tiles
    tl = np.array([-cam.cx / cam.fx, -cam.cy / cam.fy])
    psx, psy = 1.0 / cam.fx, 1.0 / cam.fy
    img = np.zeros((H, W, 3))
    a = np.minimum(alpha, ALPHA_CLAMP)
    for tile, ids_t in enumerate(frozen.lists):
        ty, tx = divmod(tile, ntw)
        ys, xs = np.meshgrid(np.arange(ty * 16, min(ty * 16 + 16, H)), np.arange(tx * 16, min(tx * 16 + 16, W)), indexing="ij")
        ys, xs = ys.reshape(-1), xs.reshape(-1)
        px, py = tl[0] + xs * psx, tl[1] + ys * psy
        if C > 0:
            d = np.stack([R[0, 0] * px + R[0, 1] * py + R[0, 2], R[1, 0] * px + R[1, 1] * py + R[1, 2],
                          R[2, 0] * px + R[2, 1] * py + R[2, 2]], -1)
            Y = sh_basis(d / np.linalg.norm(d, axis=-1, keepdims=True), C)   # [pix, C*C]
        T = np.ones(px.shape)
        acc = np.zeros(px.shape + (3,))
        if first:
            frozen.take[tile] = np.zeros((px.size, len(ids_t)), bool)
        for e, g in enumerate(ids_t):
            c0, c1, c2, c3 = cov[g, 0, 0], cov[g, 0, 1], cov[g, 1, 0], cov[g, 1, 1]
            det = c0 * c3 - c1 * c2
            x, y = px - m2[g, 0], py - m2[g, 1]
            radial = ((x * c3 - y * c2) * x + (-x * c1 + y * c0) * y) / det
            G = np.exp(-0.5 * np.where(radial < 0, 1000.0, radial))
            ag = a[g] * G
            if first:
                alive = ~(T < thresh)
                take = alive & ~(ag < MIN_ALPHA)
                frozen.take[tile][:, e] = take
                if alive.any():
                    frozen.margins["skip"] = min(frozen.margins["skip"], float(np.abs(ag[alive] / MIN_ALPHA - 1).min()))
                    frozen.margins["stop"] = min(frozen.margins["stop"], float(np.abs(T[alive] / thresh - 1).min()))
            else:
                take = frozen.take[tile][:, e]
            col = (1 / (1 + np.exp(-(Y @ P["sh"][g].T)))) if C > 0 else np.broadcast_to(P["color"][g], px.shape + (3,))
            w = np.where(take, ag * T, 0.0)
            acc += w[:, None] * col
            T = T * np.where(take, 1 - ag, 1.0)
        out = acc + (T[:, None] * np.asarray(bg, np.float64)[None] if bg is not None else 0.0)
        img[ys, xs] = out
    return float((img * go).sum()), img, frozen


def fd_gradients(P, cam, C, go, frozen, names, bg=None, thresh=1e-4, detach_depth=False, rel_step=1e-6):
    """central differences of the loss, one scalar parameter at a time, discrete state frozen"""
    out = {}
    for k in names:
        base = P[k]
        g = np.zeros(base.shape)
        it = np.nditer(base, flags=["multi_index"])
        for v in it:
            idx = it.multi_index
            h = rel_step * max(1.0, abs(float(v))) if k != "svec" else rel_step * abs(float(v))
            lo_hi = []
            for sgn in (-1.0, 1.0):
                Q = dict(P)
                arr = base.copy()
                arr[idx] = float(v) + sgn * h
                Q[k] = arr
                lo_hi.append(forward(Q, cam, C, go, bg, thresh, detach_depth, frozen)[0])
            g[idx] = (lo_hi[1] - lo_hi[0]) / (2 * h)
        out[k] = g
    return out


def tiny_scene(n, seed, C, cam, spread=0.03, svec=0.012):
    """n Gaussians inside the view of `cam` (origin-centred), translucent enough that no pixel saturates"""
    rng = np.random.default_rng(seed)
    s = {"mean": rng.normal(0, spread, (n, 3)).astype(np.float32)}
    q = rng.normal(0, 1, (n, 4))
    s["qvec"] = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    s["svec"] = np.exp(rng.normal(np.log(svec), 0.35, (n, 3))).astype(np.float32)
    s["alpha"] = rng.uniform(0.15, 0.6, n).astype(np.float32)
    s["color"] = rng.uniform(0.05, 0.95, (n, 3)).astype(np.float32)
    cc = max(C, 1) ** 2
    sh = rng.normal(0, 0.4, (n, 3, cc)).astype(np.float32)
    sh[:, :, 0] = rng.normal(0, 1.5, (n, 3))
    s["sh"], s["C"] = np.ascontiguousarray(sh), C
    return s
