"""Pins the CPU oracle (oracle/gs_oracle.c) against golden vectors produced by the REFERENCE
ITSELF (tests/golden/make_golden.py: the reference's Python imported from /root/reference and
its CUDA kernels compiled for the CPU), and -- when oracle/_ref/libgs_ref.so is present --
against that library live on further scenes.

Bars: everything the reference computes without atomics is BIT-EXACT (cull mask, projection,
tile rectangles, pair count, sorted per-tile lists, RGB / scalar / SH images, transmittance);
gradients (fp32 atomics in the reference, summed in thread order on the emulator; fp64 sums in
the oracle) agree to 2e-5 of the largest gradient of the tensor.
"""
import glob
import os

import numpy as np
import pytest

import scenes
from oracle import oracle as O
from oracle import ref as Rf

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
GTOL = 2e-5


def close(a, b, tol=GTOL):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() <= tol * (np.abs(b).max() + 1e-30)


def test_golden_files_present():
    assert len(GOLD) >= 5 and any(p.endswith('long_lists.npz') for p in GOLD)


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_oracle_reproduces_reference_golden(path):
    g = np.load(path)
    fx, fy, cx, cy, w, h, near, far = g["cam_intr"]
    w, h = int(w), int(h)
    c2w = g["c2w"]
    normals, pts = O.frustum(c2w, fx, fy, cx, cy, w, h, near, far)
    # torch's cross/normalize round one component differently now and then: planes agree to 1 ulp,
    # the cull decisions below are identical
    assert np.abs(normals - g["frustum_normals"]).max() <= 6e-8 and np.array_equal(pts, g["frustum_pts"])
    mask = O.cull_bsphere(g["in_mean"], g["in_svec"], normals, pts, 6.0)
    assert np.array_equal(mask, g["mask"])
    m = mask
    mean2d, cov2d, JW, depth = O.project(g["in_mean"][m], g["in_qvec"][m], g["in_svec"][m], c2w)
    # The reference projects with torch einsum/bmm, whose BLAS kernels sum the 3-term dot
    # products in their own order (and differently on CPU and GPU): agreement to a few ulp, not
    # bit-exact (it IS bit-exact for axis-aligned poses, e.g. mock2).  Everything downstream is
    # checked from the reference's own mean2d / cov2d / depth so that it can be exact.
    for a, b in ((mean2d, g["mean2d"]), (cov2d, g["cov2d"]), (depth, g["depth"]), (JW, g["JW"])):
        assert np.abs(a - b).max() <= 4e-6 * max(1e-3, np.abs(b).max())
    mean2d, cov2d, depth = g["mean2d"], g["cov2d"], g["depth"]
    D, tl, br = O.aabb_count(mean2d, cov2d, 16, fx, fy, cx, cy, w, h, 6.0)
    assert D == int(g["D"]) and np.array_equal(tl, g["tl"]) and np.array_equal(br, g["br"])
    nth, ntw = (h + 15) // 16, (w + 15) // 16
    ids, start, end = O.bin_sort(tl, br, depth, nth, ntw, D)
    assert np.array_equal(ids, g["ids"]) and np.array_equal(start, g["start"]) and np.array_equal(end, g["end"])
    topleft = np.array([-cx / fx, -cy / fy], np.float32)
    geo = (start, end, ids, topleft, 1 / fx, 1 / fy, h, w)
    col, al = g["in_color"][m], g["in_alpha"][m]
    rgb, T = O.render_rgb_fwd(mean2d, cov2d, col, al, *geo)
    assert np.array_equal(rgb, g["rgb"]) and np.array_equal(T, g["T"])
    final = (rgb + T * g["bg_img"]).astype(np.float32)
    gr = O.render_rgb_bwd(mean2d, cov2d, col, al, start, end, ids, final, g["grad_out"], topleft, 1 / fx, 1 / fy, h, w)
    for a, k in zip(gr, ("rgb_gmean", "rgb_gcov", "rgb_gcol", "rgb_galpha")):
        assert close(a, g[k]), k
    pg = O.project_bwd(g["in_mean"][m], g["in_qvec"][m], g["in_svec"][m], c2w, g["rgb_gmean"], g["rgb_gcov"], None, True)
    for a, k in zip(pg, ("proj_gmean", "proj_gqvec", "proj_gsvec")):
        assert close(a, g[k], 1e-4), k  # torch fp32 autograd on the reference side
    s_img, sT = O.render_scalar_fwd(mean2d, cov2d, depth.ravel(), al, *geo)
    assert np.array_equal(s_img, g["depth_img"]) and np.array_equal(sT, g["depth_T"])
    gr = O.render_scalar_bwd(mean2d, cov2d, depth.ravel(), al, start, end, ids, s_img, g["grad_out"][..., 0].copy(),
                             topleft, 1 / fx, 1 / fy, h, w)
    for a, k in zip(gr, ("sc_gmean", "sc_gcov", "sc_gscalar", "sc_galpha")):
        assert close(a, g[k]), k
    C = int(g["C"])
    rot = c2w[:3, :3].reshape(-1)
    sh = g["in_sh"][m]
    for tag, bg in (("sh", None), ("shbg", g["bg_rgb"])):
        img = O.render_sh_fwd(mean2d, cov2d, sh, al, start, end, ids, topleft, rot, C, 1 / fx, 1 / fy, h, w, bg=bg)
        assert np.array_equal(img, g[tag + "_img"]), tag
        gr = O.render_sh_bwd(mean2d, cov2d, sh, al, start, end, ids, img, g["grad_out"], topleft, rot, C, 1 / fx,
                             1 / fy, h, w)
        for a, k in zip(gr, ("_gmean", "_gcov", "_gsh", "_galpha")):
            assert close(a, g[tag + k]), tag + k


needs_ref = pytest.mark.skipif(not Rf.available(), reason="oracle/_ref/libgs_ref.so not built")


@needs_ref
@pytest.mark.parametrize("seed,W,H,C", [(21, 100, 37, 2), (22, 48, 48, 4), (23, 33, 70, 1)])
def test_oracle_vs_reference_kernels_live(seed, W, H, C):
    cam = scenes.Camera(W, H, fx=0.9 * W, fy=1.1 * W, cx=W / 2 + 0.7, cy=H / 2 - 1.3, c2w=scenes.orbit(2.4, 10 + seed, 40 * seed))
    sc = scenes.random_scene(500, seed=seed, svec=0.06, svec_sigma=0.5, C=C)
    g = scenes.oracle_geometry(sc, cam)
    m = g["mask"]
    assert np.array_equal(Rf.cull_bsphere(sc["mean"], sc["qvec"], sc["svec"], g["normals"], g["pts"], 6.0), m)
    nth, ntw = cam.tiles
    ids, st, en = Rf.bin_sort(g["tl"], g["br"], g["depth"], nth, ntw, g["D"])
    assert np.array_equal(ids, g["ids"]) and np.array_equal(st, g["start"]) and np.array_equal(en, g["end"])
    a = (g["mean2d"], g["cov2d"])
    al = sc["alpha"][m]
    geo = (st, en, ids, cam.topleft, 1 / cam.fx, 1 / cam.fy, H, W)
    r, rT = Rf.render_rgb_fwd(*a, sc["color"][m], al, *geo)
    o, oT = O.render_rgb_fwd(*a, sc["color"][m], al, *geo)
    assert np.array_equal(r, o) and np.array_equal(rT, oT)
    go = np.random.default_rng(seed).normal(size=(H, W, 3)).astype(np.float32)
    rot = cam.c2w[:3, :3].reshape(-1)
    bg = np.array([0.3, 0.1, 0.8], np.float32)
    sh = np.ascontiguousarray(sc["sh"][m])
    r = Rf.render_sh_fwd(*a, sh, al, st, en, ids, cam.topleft, rot, C, 1 / cam.fx, 1 / cam.fy, H, W, bg=bg)
    o = O.render_sh_fwd(*a, sh, al, st, en, ids, cam.topleft, rot, C, 1 / cam.fx, 1 / cam.fy, H, W, bg=bg)
    assert np.array_equal(r, o)
    rg = Rf.render_sh_bwd(*a, sh, al, st, en, ids, o, go, cam.topleft, rot, C, 1 / cam.fx, 1 / cam.fy, H, W, bg=bg)
    og = O.render_sh_bwd(*a, sh, al, st, en, ids, o, go, cam.topleft, rot, C, 1 / cam.fx, 1 / cam.fy, H, W)
    for x, y in zip(rg, og):
        assert close(x, y)


@needs_ref
@pytest.mark.parametrize("C", [4, 2])
def test_oracle_vs_reference_kernels_live_on_long_lists(C):
    """the oracle-vs-reference pin on lists that cross every LDS batch boundary of the reference's kernels
    (RGB forward 1 200 entries, RGB backward 472: vol_render.h:501, :442; SH degree-3 forward 218, backward 109:
    vol_render_sh.h:171-248, :353-455): 4 000 splats on a 32 x 32 image, the longest list beyond 1 400 entries, opacities scaled
    so that the pixels of one half walk their whole list and those of the other saturate early.  Images and T bit for bit,
    gradients to the summation-order tolerance."""
    W = H = 32
    cam = scenes.Camera(W, H, fx=26.0, c2w=scenes.look_at((2.6, 0.0, 0.3)))
    sc = scenes.random_scene(4000, seed=31 + C, svec=0.09, spread=0.5, C=C)
    sc["alpha"] = np.where(sc["mean"][:, 1] < 0.0, sc["alpha"] * 0.02, sc["alpha"]).astype(np.float32)
    g = scenes.oracle_geometry(sc, cam)
    m = g["mask"]
    nth, ntw = cam.tiles
    ids, st, en = Rf.bin_sort(g["tl"], g["br"], g["depth"], nth, ntw, g["D"])
    assert np.array_equal(ids, g["ids"]) and np.array_equal(st, g["start"]) and np.array_equal(en, g["end"])
    assert int((en - st).max()) > 1400
    a = (g["mean2d"], g["cov2d"])
    al = sc["alpha"][m]
    geo = (st, en, ids, cam.topleft, 1 / cam.fx, 1 / cam.fy, H, W)
    col = sc["color"][m]
    r, rT = Rf.render_rgb_fwd(*a, col, al, *geo)
    o, oT = O.render_rgb_fwd(*a, col, al, *geo)
    assert np.array_equal(r, o) and np.array_equal(rT, oT)
    assert float(oT.min()) < 1e-4 and float(oT.max()) > 0.05  # early termination AND deep walks
    go = np.random.default_rng(C).normal(size=(H, W, 3)).astype(np.float32)
    rg = Rf.render_rgb_bwd(*a, col, al, st, en, ids, o, go, cam.topleft, 1 / cam.fx, 1 / cam.fy, H, W)
    og = O.render_rgb_bwd(*a, col, al, st, en, ids, o, go, cam.topleft, 1 / cam.fx, 1 / cam.fy, H, W)
    for x, y in zip(rg, og):
        assert close(x, y)
    sv = g["depth"].ravel()
    r, rT = Rf.render_scalar_fwd(*a, sv, al, *geo)
    o, oT = O.render_scalar_fwd(*a, sv, al, *geo)
    assert np.array_equal(r, o) and np.array_equal(rT, oT)
    rot = cam.c2w[:3, :3].reshape(-1)
    sh = np.ascontiguousarray(sc["sh"][m])
    for bg in (None, np.array([0.3, 0.1, 0.8], np.float32)):
        r = Rf.render_sh_fwd(*a, sh, al, st, en, ids, cam.topleft, rot, C, 1 / cam.fx, 1 / cam.fy, H, W, bg=bg)
        o = O.render_sh_fwd(*a, sh, al, st, en, ids, cam.topleft, rot, C, 1 / cam.fx, 1 / cam.fy, H, W, bg=bg)
        assert np.array_equal(r, o)
        rg = Rf.render_sh_bwd(*a, sh, al, st, en, ids, o, go, cam.topleft, rot, C, 1 / cam.fx, 1 / cam.fy, H, W, bg=bg)
        og = O.render_sh_bwd(*a, sh, al, st, en, ids, o, go, cam.topleft, rot, C, 1 / cam.fx, 1 / cam.fy, H, W)
        for x, y in zip(rg, og):
            assert close(x, y)


@needs_ref
def test_reference_sort_semantics_negative_depth_and_ties():
    """aabb_culling.h:36-37,235-241: key = (tile << 32) | float_bits(depth) sorted on all 64 bits
    -> negative depths come after positive ones, in reverse; equal keys keep emission order."""
    rng = np.random.default_rng(3)
    N = 300
    depth = rng.choice(np.array([0.5, 1.0, 2.0, -1.0, -0.25, 3.5], np.float32), size=N).astype(np.float32)
    tl = np.zeros((N, 2), np.int32); br = np.zeros((N, 2), np.int32)
    br[::3, 0] = 1
    D = int(((br[:, 0] - tl[:, 0] + 1)).sum())
    r = Rf.bin_sort(tl, br, depth, 1, 2, D)
    o = O.bin_sort(tl, br, depth, 1, 2, D)
    for x, y in zip(r, o):
        assert np.array_equal(x, y)
    first = depth[o[0][: o[2][0]]]
    pos = first[first > 0]
    assert np.all(np.diff(pos) >= 0) and np.all(first[len(pos):] < 0)


@pytest.mark.parametrize("mode", [0, 1])
def test_legacy_binning_oracle_equals_reference(mode):
    """oracle gso_legacy_* == the reference's tile_ops.h / culling.h kernels compiled for the CPU
    (oracle/_ref), bit for bit: counts, offsets, unsorted {tile, depth} keys, sorted ids"""
    from oracle import ref as Rf
    if not Rf.available():
        pytest.skip("oracle/_ref/libgs_ref.so not built")
    if not hasattr(Rf.lib(), "ref_image_sort"):
        pytest.skip("oracle/_ref predates the legacy-binning entry points (rebuild needs /root/reference)")
    import scenes
    sc = scenes.random_scene(400, seed=7, svec=0.06)
    cam = scenes.Camera(96, 64, fx=80.0)
    g = scenes.oracle_geometry(sc, cam)
    m2, c2, dep = g["mean2d"], g["cov2d"].reshape(-1, 4), g["depth"].ravel()
    nth, ntw = cam.tiles
    shape = c2 if mode == 0 else np.sqrt(6.0 * np.maximum(c2[:, 0], c2[:, 3])).astype(np.float32)
    args = (cam.topleft, 16, nth, ntw, 1 / cam.fx, 1 / cam.fy, 0.01)
    a = O.legacy_count(mode, m2, shape, *args)
    b = Rf.legacy_count(mode, m2, shape, *args)
    assert np.array_equal(a, b) and a.sum() > 0
    for x, y in zip(O.legacy_image_sort(mode, dep, a, m2, shape, *args), Rf.legacy_image_sort(mode, dep, b, m2, shape, *args)):
        assert np.array_equal(x, y)


def test_decision_margin_reports_threshold_adjacent_pixels():
    """oracle.sh_decision_margin (the test aid behind scenes.assert_sh_image_parity): a splat whose opacity is tuned so
    that a*G at one pixel sits exactly on the 1/255 skip threshold gives that pixel a zero margin and leaves the
    others well away from it; empty tiles report 'no decision'."""
    cam = scenes.Camera(32, 32, fx=32.0)
    sc = scenes.random_scene(1, seed=0, svec=0.2, C=1)
    sc["mean"][:] = 0.0
    g = scenes.oracle_geometry(sc, cam)
    assert g["mask"].all() and g["D"] >= 1
    a = (g["mean2d"], g["cov2d"])
    geo = (g["start"], g["end"], g["ids"], cam.topleft, 1 / cam.fx, 1 / cam.fy, cam.h, cam.w)
    m0 = O.sh_decision_margin(*a, np.array([0.5], np.float32), *geo)
    assert m0.shape == (32, 32, 2) and np.isfinite(m0[m0 < 1e29]).all() and (m0 >= 0).all()
    # G at pixel (20, 13) from the margin at alpha = 0.5: |0.5 G - t| / t = m  =>  G = (1 +- m) t / 0.5
    t = np.float32(1.0 / 255.0)
    y, x = 20, 13
    cands = [np.float32((1 + s * m0[y, x, 0]) * t / np.float32(0.5)) for s in (+1, -1)]
    hit = False
    for G in cands:
        if not (0 < G <= 1):
            continue
        alpha = np.float32(t / G)
        m1 = O.sh_decision_margin(*a, np.array([alpha], np.float32), *geo)
        hit |= bool(m1[y, x, 0] <= 3e-7)
    assert hit
    assert np.median(m0[..., 0][m0[..., 0] < 1e29]) > 1e-3


def test_torch_port_is_the_references_torch_code():
    """oracle/torch_port.py (bench.py's CPU-baseline leg (1): the two stages the reference runs in PyTorch) against the
    reference's OWN functions, imported from /root/reference behind the shims: project_gaussians (+ its autograd backward) and
    tile_culling_aabb_count, bit for bit.  Authoring container only; the golden vectors carry the same outputs to the GPU box
    (tests/golden/*.npz: mean2d / cov2d / depth / D were produced by the reference's functions)."""
    import refshim
    if not refshim.available():
        pytest.skip("/root/reference is not present")
    import torch
    refshim.install()
    import gs.renderer as GR
    import gs.culling as GC
    from utils.camera import CameraInfo
    from oracle import torch_port as TP
    sc = scenes.random_scene(2000, seed=3, svec=0.03, C=1)
    cam = scenes.Camera(200, 144, fx=190.0, c2w=scenes.orbit(2.5, 25.0, 70.0))
    c2w = torch.from_numpy(cam.c2w)
    ins = [torch.from_numpy(sc[k]).clone().requires_grad_(True) for k in ("mean", "qvec", "svec")]
    ins2 = [t.detach().clone().requires_grad_(True) for t in ins]
    a = GR.project_gaussians(*ins, c2w, detach_depth=True)
    b = TP.project_gaussians(*ins2, c2w, detach_depth=True)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    w = [torch.randn_like(x) for x in (a[0], a[1], a[3])]
    (a[0] * w[0]).sum().add((a[1] * w[1]).sum()).add((a[3] * w[2]).sum()).backward()
    (b[0] * w[0]).sum().add((b[1] * w[1]).sum()).add((b[3] * w[2]).sum()).backward()
    for x, y in zip(ins, ins2):
        assert torch.equal(x.grad, y.grad)
    ci = CameraInfo(cam.fx, cam.fy, cam.cx, cam.cy, cam.w, cam.h, cam.near, cam.far)
    n1, tl1, br1 = GC.tile_culling_aabb_count(a[0].detach(), a[1].detach(), 16, ci, 6.0)
    n2, tl2, br2 = TP.tile_culling_aabb_count(b[0].detach(), b[1].detach(), 16, cam.fx, cam.fy, cam.cx, cam.cy, cam.w, cam.h, 6.0)
    assert n1 == n2 and torch.equal(tl1, tl2) and torch.equal(br1, br2)
