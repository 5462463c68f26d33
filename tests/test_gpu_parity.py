"""GPU parity tests: the HIP path, called through the C ABI (gsgen_amd._capi -> libgsgen_hip.so),
against the CPU oracle on identical seeded inputs.

Bars: integer outputs (masks, tile rectangles, pair counts, start/end, sorted
ids) bit-exact; projection outputs bit-exact (same IEEE op order, no contraction); images
max|err| <= 1e-4 (north_star); gradients within rtol 1e-3 of the oracle's fp64-summed
per-pair contributions, measured against the largest gradient magnitude of the tensor
(fp32 atomics reorder the sums).
"""
import numpy as np
import pytest
import torch

import scenes
from oracle import oracle as O

pytestmark = pytest.mark.gpu

IMG_TOL = 1e-4
GRAD_RTOL = 1e-3


def dev():
    return torch.device("cuda:0")


_KEEP = []  # p(T_(x)) hands a raw address to the C ABI: keep the tensor alive past the call


def T_(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    t = t.to(dtype) if dtype is not None else t
    _KEEP.append(t)
    if len(_KEEP) > 256:
        torch.cuda.synchronize()
        del _KEEP[:128]
    return t


def lib():
    from gsgen_amd import _capi
    return _capi.load()


def stream():
    return torch.cuda.current_stream().cuda_stream


def p(t):
    return t.data_ptr() if t is not None else None


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def hip_geometry(sc, cam, g):
    """cull / project / count / bin through the compat entry points, on the oracle's mask."""
    L = lib()
    m = g["mask"]
    mean, q, s = (T_(sc[k][m]) for k in ("mean", "qvec", "svec"))
    N = mean.shape[0]
    c2w = T_(cam.c2w)
    m2 = torch.empty(N, 2, device=dev()); c2 = torch.empty(N, 2, 2, device=dev())
    JW = torch.empty(N, 3, 3, device=dev()); dep = torch.empty(N, 1, device=dev())
    L.project_gaussians(N, p(mean), p(q), p(s), p(c2w), p(m2), p(c2), p(JW), p(dep), stream())
    tl = torch.empty(N, 2, dtype=torch.int32, device=dev()); br = torch.empty_like(tl)
    tot = torch.zeros(1, dtype=torch.int32, device=dev())
    L.tile_culling_aabb_count(N, p(m2), p(c2), 16, cam.fx, cam.fy, cam.cx, cam.cy, cam.w, cam.h, 6.0, p(tl),
                              p(br), p(tot), stream())
    D = int(tot.item())
    nth, ntw = cam.tiles
    ids = torch.zeros(max(D, 1), dtype=torch.int32, device=dev())[:D]
    st = -torch.ones(nth * ntw, dtype=torch.int32, device=dev()); en = -torch.ones_like(st)
    nb = L.tile_culling_workspace_bytes(N, D, nth * ntw)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev())
    L.tile_culling_aabb_start_end(N, D, nth, ntw, p(tl), p(br), p(dep), p(ids), p(st), p(en), p(ws), nb, stream())
    torch.cuda.synchronize()
    return dict(N=N, D=D, mean2d=m2, cov2d=c2, JW=JW, depth=dep, tl=tl, br=br, ids=ids, start=st, end=en)


SCENES = {
    "cfg1": lambda: (scenes.random_scene(1000, seed=0, C=1), scenes.Camera(256, 256, fx=256.0)),
    "ragged": lambda: (scenes.random_scene(600, seed=1, svec=0.05, C=3), scenes.Camera(150, 70, fx=120.0, fy=110.0, cx=70.3, cy=33.1)),
    "dense": lambda: (scenes.random_scene(4000, seed=2, svec=0.05, spread=0.5, C=4), scenes.Camera(96, 96, fx=96.0)),
    "aniso": lambda: (scenes.random_scene(800, seed=3, svec=0.04, svec_sigma=0.9, C=2), scenes.Camera(128, 128, fx=128.0, c2w=scenes.orbit(2.2, 40, 100))),
}


@pytest.fixture(scope="module", params=list(SCENES))
def case(request):
    sc, cam = SCENES[request.param]()
    g = scenes.oracle_geometry(sc, cam)
    h = hip_geometry(sc, cam, g)
    return request.param, sc, cam, g, h


def test_cull_mask_exact(case):
    _, sc, cam, g, _ = case
    mean, q, s = (T_(sc[k]) for k in ("mean", "qvec", "svec"))
    mask = torch.zeros(mean.shape[0], dtype=torch.bool, device=dev())
    lib().culling_gaussian_bsphere(mean.shape[0], p(mean), p(q), p(s), p(T_(g["normals"])), p(T_(g["pts"])),
                                   p(mask), 6.0, stream())
    assert np.array_equal(mask.cpu().numpy(), g["mask"])
    assert 0 < g["mask"].sum()


def test_projection_bit_exact(case):
    _, sc, cam, g, h = case
    assert np.array_equal(h["mean2d"].cpu().numpy(), g["mean2d"])
    assert np.array_equal(h["cov2d"].cpu().numpy(), g["cov2d"])
    assert np.array_equal(h["depth"].cpu().numpy(), g["depth"])
    assert np.array_equal(h["JW"].cpu().numpy(), g["JW"])


def test_binning_exact(case):
    _, sc, cam, g, h = case
    assert h["D"] == g["D"]
    assert np.array_equal(h["tl"].cpu().numpy(), g["tl"])
    assert np.array_equal(h["br"].cpu().numpy(), g["br"])
    assert np.array_equal(h["start"].cpu().numpy(), g["start"])
    assert np.array_equal(h["end"].cpu().numpy(), g["end"])
    assert np.array_equal(h["ids"].cpu().numpy(), g["ids"])


def test_projection_backward(case):
    _, sc, cam, g, h = case
    m = g["mask"]
    rng = np.random.default_rng(5)
    N = h["N"]
    gm2 = rng.normal(size=(N, 2)).astype(np.float32); gc2 = rng.normal(size=(N, 2, 2)).astype(np.float32)
    gd = rng.normal(size=(N, 1)).astype(np.float32)
    mean, q, s = (T_(sc[k][m]) for k in ("mean", "qvec", "svec"))
    for detach in (1, 0):
        gm, gq, gs = torch.empty_like(mean), torch.empty_like(q), torch.empty_like(s)
        lib().project_gaussians_backward(N, p(mean), p(q), p(s), p(T_(cam.c2w)), detach, p(T_(gm2)), p(T_(gc2)),
                                         p(T_(gd)), p(gm), p(gq), p(gs), stream())
        om, oq, os_ = O.project_bwd(sc["mean"][m], sc["qvec"][m], sc["svec"][m], cam.c2w, gm2, gc2, gd, bool(detach))
        # per-row relative error: the gradients span many orders of magnitude across Gaussians
        for a, b in ((gm, om), (gq, oq), (gs, os_)):
            a = a.cpu().numpy().astype(np.float64)
            row = np.abs(b).max(axis=1, keepdims=True) + 1e-20
            assert float((np.abs(a - b) / row).max()) < 2e-3


def _comp_inputs(sc, g, h):
    m = g["mask"]
    return dict(col=T_(sc["color"][m]), al=T_(sc["alpha"][m]), sh=T_(sc["sh"][m]))


def test_rgb_forward_backward(case):
    name, sc, cam, g, h = case
    L = lib(); m = g["mask"]; ci = _comp_inputs(sc, g, h)
    H, W = cam.h, cam.w; nth, ntw = cam.tiles
    tl = T_(cam.topleft)
    out = torch.zeros(H, W, 3, device=dev()); T = torch.ones(H, W, 1, device=dev())
    L.vol_render_start_end_with_T(h["N"], h["D"], p(h["mean2d"]), p(h["cov2d"]), p(ci["col"]), p(ci["al"]),
                                  p(h["start"]), p(h["end"]), p(h["ids"]), p(out), p(tl), 16, nth, ntw,
                                  1 / cam.fx, 1 / cam.fy, H, W, 1e-4, p(T), stream())
    ref, refT = O.render_rgb_fwd(g["mean2d"], g["cov2d"], sc["color"][m], sc["alpha"][m], g["start"], g["end"],
                                 g["ids"], cam.topleft, 1 / cam.fx, 1 / cam.fy, H, W)
    assert np.abs(out.cpu().numpy() - ref).max() <= IMG_TOL
    assert np.abs(T.cpu().numpy() - refT).max() <= IMG_TOL
    # backward, with a background folded into `final` as gs/renderer.py:1182 does
    bg = np.random.default_rng(7).uniform(size=(H, W, 3)).astype(np.float32)
    final = ref + refT * bg
    go = np.random.default_rng(8).normal(size=(H, W, 3)).astype(np.float32)
    N = h["N"]
    gm = torch.zeros(N, 2, device=dev()); gc = torch.zeros(N, 2, 2, device=dev())
    gcol = torch.zeros(N, 3, device=dev()); ga = torch.zeros(N, device=dev())
    L.vol_render_backward_start_end(N, h["D"], p(h["mean2d"]), p(h["cov2d"]), p(ci["col"]), p(ci["al"]),
                                    p(h["start"]), p(h["end"]), p(h["ids"]), p(T_(final)), p(gm), p(gc), p(gcol),
                                    p(ga), p(T_(go)), p(tl), 16, nth, ntw, 1 / cam.fx, 1 / cam.fy, H, W, 1e-4, stream())
    om, oc, ocol, oa = O.render_rgb_bwd(g["mean2d"], g["cov2d"], sc["color"][m], sc["alpha"][m], g["start"], g["end"],
                                        g["ids"], final, go, cam.topleft, 1 / cam.fx, 1 / cam.fy, H, W)
    for a, b in ((gm, om), (gc, oc), (gcol, ocol), (ga, oa)):
        assert rel_err(a.cpu().numpy(), b) < GRAD_RTOL


def test_scalar_forward_backward(case):
    name, sc, cam, g, h = case
    L = lib(); m = g["mask"]; ci = _comp_inputs(sc, g, h)
    H, W = cam.h, cam.w; nth, ntw = cam.tiles
    tl = T_(cam.topleft)
    scal = h["depth"].reshape(-1).contiguous()
    out = torch.zeros(H * W, device=dev()); T = torch.ones(H, W, 1, device=dev())
    L.vol_render_scalar(h["N"], h["D"], p(h["mean2d"]), p(h["cov2d"]), p(scal), p(ci["al"]), p(h["start"]),
                        p(h["end"]), p(h["ids"]), p(out), p(tl), 16, nth, ntw, 1 / cam.fx, 1 / cam.fy, H, W, 1e-4,
                        p(T), stream())
    ref, refT = O.render_scalar_fwd(g["mean2d"], g["cov2d"], g["depth"].ravel(), sc["alpha"][m], g["start"], g["end"],
                                    g["ids"], cam.topleft, 1 / cam.fx, 1 / cam.fy, H, W)
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.abs(out.cpu().numpy().reshape(H, W) - ref).max() <= IMG_TOL * scale
    assert np.abs(T.cpu().numpy() - refT).max() <= IMG_TOL
    go = np.random.default_rng(9).normal(size=(H, W)).astype(np.float32)
    N = h["N"]
    gm = torch.zeros(N, 2, device=dev()); gc = torch.zeros(N, 2, 2, device=dev())
    gs = torch.zeros(N, device=dev()); ga = torch.zeros(N, device=dev())
    L.vol_render_scalar_backward(N, h["D"], p(h["mean2d"]), p(h["cov2d"]), p(scal), p(ci["al"]), p(h["start"]),
                                 p(h["end"]), p(h["ids"]), p(T_(ref)), p(gm), p(gc), p(gs), p(ga), p(T_(go)), p(tl),
                                 16, nth, ntw, 1 / cam.fx, 1 / cam.fy, H, W, 1e-4, stream())
    om, oc, os_, oa = O.render_scalar_bwd(g["mean2d"], g["cov2d"], g["depth"].ravel(), sc["alpha"][m], g["start"],
                                          g["end"], g["ids"], ref, go, cam.topleft, 1 / cam.fx, 1 / cam.fy, H, W)
    for a, b in ((gm, om), (gc, oc), (gs, os_), (ga, oa)):
        assert rel_err(a.cpu().numpy(), b) < GRAD_RTOL


@pytest.mark.parametrize("use_bg", [False, True])
def test_sh_forward_backward(case, use_bg):
    name, sc, cam, g, h = case
    L = lib(); m = g["mask"]; ci = _comp_inputs(sc, g, h)
    C = sc["C"]
    H, W = cam.h, cam.w; nth, ntw = cam.tiles
    tl = T_(cam.topleft)
    rot = np.ascontiguousarray(cam.c2w[:3, :3]).reshape(-1).copy()
    bg = np.array([0.2, 0.5, 0.7], np.float32) if use_bg else None
    bgt = T_(bg) if use_bg else None
    out = torch.zeros(H, W, 3, device=dev())
    L.vol_render_sh(h["N"], h["D"], p(h["mean2d"]), p(h["cov2d"]), p(ci["sh"]), p(ci["al"]), p(h["start"]),
                    p(h["end"]), p(h["ids"]), p(out), p(tl), p(T_(rot)), 16, nth, ntw, 1 / cam.fx, 1 / cam.fy, H, W,
                    C, 1e-4, p(bgt), None, stream())
    ref = O.render_sh_fwd(g["mean2d"], g["cov2d"], sc["sh"][m], sc["alpha"][m], g["start"], g["end"], g["ids"],
                          cam.topleft, rot, C, 1 / cam.fx, 1 / cam.fy, H, W, bg=bg)
    err = np.abs(out.cpu().numpy() - ref)
    # north_star: pixel for pixel within 1e-4.  The reference evaluates this path in fp32 and skips a splat when
    # a*G < 1/255 -- a discontinuity of up to 1/255 in the image; the kernel re-evaluates any a*G within 2e-4
    # (relative) of the threshold with the reference's own arithmetic, so the decision is the reference's: no
    # pixel may exceed the tolerance.
    img_bg = out.cpu().numpy()
    scenes.assert_sh_image_parity(img_bg, ref, g["mean2d"], g["cov2d"], sc["alpha"][m], g["start"], g["end"], g["ids"],
                                  cam.topleft, 1 / cam.fx, 1 / cam.fy, tol=IMG_TOL, what=name)
    go = np.random.default_rng(10).normal(size=(H, W, 3)).astype(np.float32)
    N = h["N"]
    gm = torch.zeros(N, 2, device=dev()); gc = torch.zeros(N, 2, 2, device=dev())
    gsh = torch.zeros(N, 3, C * C, device=dev()); ga = torch.zeros(N, device=dev())
    L.vol_render_backward_sh(N, h["D"], p(h["mean2d"]), p(h["cov2d"]), p(ci["sh"]), p(ci["al"]), p(h["start"]),
                             p(h["end"]), p(h["ids"]), p(out), p(gm), p(gc), p(gsh), p(ga), p(T_(go)), p(tl),
                             p(T_(rot)), 16, nth, ntw, 1 / cam.fx, 1 / cam.fy, H, W, C, 1e-4, p(bgt), stream())
    om, oc, osh, oa = O.render_sh_bwd(g["mean2d"], g["cov2d"], sc["sh"][m], sc["alpha"][m], g["start"], g["end"],
                                      g["ids"], ref, go, cam.topleft, rot, C, 1 / cam.fx, 1 / cam.fy, H, W)
    for a, b in ((gm, om), (gc, oc), (gsh, osh), (ga, oa)):
        assert rel_err(a.cpu().numpy(), b) < GRAD_RTOL


# ---- edge cases -------------------------------------------------------------------------------
def test_empty_inputs():
    """N = 0 / D = 0: undefined in the reference (SURVEY.md 8a trap 7); here: clean no-op."""
    L = lib()
    H = W = 32
    z = torch.zeros(0, device=dev())
    zi = torch.zeros(0, dtype=torch.int32, device=dev())
    st = -torch.ones(4, dtype=torch.int32, device=dev()); en = -torch.ones_like(st)
    out = torch.zeros(H, W, 3, device=dev()); T = torch.ones(H, W, 1, device=dev())
    tl = torch.tensor([-0.5, -0.5], device=dev())
    L.vol_render_start_end_with_T(0, 0, p(z), p(z), p(z), p(z), p(st), p(en), p(zi), p(out), p(tl), 16, 2, 2,
                                  1 / 32, 1 / 32, H, W, 1e-4, p(T), stream())
    nb = L.tile_culling_workspace_bytes(0, 0, 4)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev())
    L.tile_culling_aabb_start_end(0, 0, 2, 2, p(zi), p(zi), p(z), p(zi), p(st), p(en), p(ws), nb, stream())
    torch.cuda.synchronize()
    assert float(out.abs().max()) == 0.0 and float((T - 1).abs().max()) == 0.0
    assert (st == -1).all() and (en == -1).all()
    # SH with background: every pixel shows the background
    bg = torch.tensor([0.3, 0.6, 0.9], device=dev())
    L.vol_render_sh(0, 0, p(z), p(z), p(z), p(z), p(st), p(en), p(zi), p(out), p(tl), p(torch.eye(3, device=dev())),
                    16, 2, 2, 1 / 32, 1 / 32, H, W, 2, 1e-4, p(bg), None, stream())
    torch.cuda.synchronize()
    assert torch.allclose(out, bg.expand(H, W, 3))


def test_empty_inputs_of_the_additive_entry_points():
    """N = 0 / D = 0 through the fused frame, the segmented SH pair, the densify statistics, the
    accumulate-form projection backward and the Adam step: clean no-ops, bad arguments -> EINVAL"""
    from gsgen_amd._capi import GsgenError
    L = lib()
    H = W = 32
    z = torch.zeros(0, device=dev()); zi = torch.zeros(0, dtype=torch.int32, device=dev())
    st = -torch.ones(4, dtype=torch.int32, device=dev()); en = -torch.ones_like(st)
    out = torch.zeros(H, W, 3, device=dev())
    tl = torch.tensor([-0.5, -0.5], device=dev()); rot = torch.eye(3, device=dev())
    ws = torch.zeros(L.segment_workspace_bytes(4, 4), dtype=torch.uint8, device=dev())
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev())
    L.vol_render_sh_segmented(0, 0, p(z), p(z), p(z), p(z), p(st), p(en), p(zi), p(out), p(tl), p(rot), 16, 2, 2, 1 / 32,
                              1 / 32, H, W, 3, 1e-4, p(bg), None, None, p(ws), 4, stream())
    torch.cuda.synchronize()
    assert torch.allclose(out, bg.expand(H, W, 3))
    L.vol_render_backward_sh_segmented(0, 0, p(z), p(z), p(z), p(z), p(st), p(en), p(zi), p(out), p(z), p(z), p(z), p(z),
                                       p(out), p(tl), p(rot), 16, 2, 2, 1 / 32, 1 / 32, H, W, 3, 1e-4, None, None, p(ws), 4,
                                       stream())
    with pytest.raises(GsgenError, match="invalid"):
        L.vol_render_backward_sh_segmented(0, 0, p(z), p(z), p(z), p(z), p(st), p(en), p(zi), p(out), p(z), p(z), p(z),
                                           p(z), p(out), p(tl), p(rot), 16, 2, 2, 1 / 32, 1 / 32, H, W, 3, 1e-4, None, None,
                                           None, 4, stream())
    L.densify_update(0, None, None, None, None, None, None, stream())
    L.project_gaussians_backward_accum(0, p(z), p(z), p(z), p(rot), 1, None, p(z), p(z), None, p(z), p(z), p(z), stream())
    ends = np.array([0], np.uint64); lr = np.array([1e-3], np.float32)
    L.adam_step(0, p(z), p(z), p(z), p(z), 1, ends.ctypes.data, lr.ctypes.data, 0.9, 0.999, 1e-15, 1, stream())
    x = torch.ones(8, device=dev())
    ends = np.array([8], np.uint64)
    with pytest.raises(GsgenError, match="invalid"):  # steps count from 1
        L.adam_step(8, p(x), p(x), p(x), p(x), 1, ends.ctypes.data, lr.ctypes.data, 0.9, 0.999, 1e-15, 0, stream())
    with pytest.raises(GsgenError, match="invalid"):  # groups must end at n
        L.adam_step(8, p(x), p(x), p(x), p(x), 1, np.array([7], np.uint64).ctypes.data, lr.ctypes.data, 0.9, 0.999,
                    1e-15, 1, stream())
    # a frame with nothing in view: all background, zero pairs, zero gradients
    from gsgen_amd import renderer as R
    sc = scenes.random_scene(50, seed=1, svec=0.02, C=2)
    cam = scenes.Camera(48, 32, fx=40.0, c2w=scenes.look_at((50.0, 0.0, 0.0), at=(100.0, 0.0, 0.0)))
    ci = R.CameraInfo(*cam.intr)
    P_ = {k: T_(sc[k]).requires_grad_(True) for k in ("mean", "qvec", "svec", "alpha", "sh")}
    buf = R.FrameBuffers(50, cam.w, cam.h, dev(), segments=4)
    rgb, T = R.render_frame(P_["mean"], P_["qvec"], P_["svec"], P_["alpha"], P_["sh"], ci, cam.c2w, buf, C=2, bg_rgb=bg)
    rgb.sum().backward()
    assert int(buf.total.item()) == 0 and not bool(buf.mask.any())
    assert torch.allclose(rgb, bg.expand(cam.h, cam.w, 3)) and float(P_["sh"].grad.abs().max()) == 0.0


def test_unsupported_configs_raise():
    from gsgen_amd._capi import GsgenError
    L = lib()
    z = torch.zeros(4, device=dev()); zi = torch.zeros(4, dtype=torch.int32, device=dev())
    for bad in (0, 33, 64):  # tile sizes: 1 .. 32 (the reference's limit: tile_size^2 <= 1024 threads per tile)
        with pytest.raises(GsgenError):
            L.vol_render_start_end_with_T(1, 1, p(z), p(z), p(z), p(z), p(zi), p(zi), p(zi), p(z), p(z), bad, 1, 1, 1.0, 1.0,
                                          8, 8, 1e-4, p(z), stream())


def test_long_tile_list_and_ties():
    """One tile holding more pairs than one register block of the sort (2048: block sorts + merge passes) and more
    than one LDS staging batch; many exactly equal depths (ties resolve by Gaussian id, the oracle's
    emission order); negative depths sort after positive ones (unsigned key order)."""
    L = lib()
    N = 5000
    rng = np.random.default_rng(11)
    depth = rng.choice(np.array([0.5, 1.0, 1.5, 2.0, -1.0, -0.25], np.float32), size=N).astype(np.float32)
    depth[::7] = rng.uniform(0.1, 3.0, size=depth[::7].shape).astype(np.float32)
    tl = np.zeros((N, 2), np.int32); br = np.zeros((N, 2), np.int32)
    br[: N // 2, 0] = 1  # half of them also cover tile (1,0)
    nth, ntw = 1, 2
    D = int(((br[:, 0] - tl[:, 0] + 1) * (br[:, 1] - tl[:, 1] + 1)).sum())
    oi, os_, oe = O.bin_sort(tl, br, depth, nth, ntw, D)
    ids = torch.zeros(D, dtype=torch.int32, device=dev())
    st = -torch.ones(2, dtype=torch.int32, device=dev()); en = -torch.ones_like(st)
    nb = L.tile_culling_workspace_bytes(N, D, 2)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev())
    L.tile_culling_aabb_start_end(N, D, nth, ntw, p(T_(tl)), p(T_(br)), p(T_(depth)), p(ids), p(st), p(en), p(ws), nb,
                                  stream())
    torch.cuda.synchronize()
    assert np.array_equal(st.cpu().numpy(), os_) and np.array_equal(en.cpu().numpy(), oe)
    assert np.array_equal(ids.cpu().numpy(), oi)


@pytest.mark.parametrize("sizes", [(1, 40, 64, 65, 100, 128, 129, 200, 256, 300, 511, 512, 513, 700, 1024, 1025, 1500, 2039, 2040,
                                    2047, 2048, 2049, 3000, 4096, 4097, 4100, 6200, 9000)])
def test_sort_every_register_width(sizes):
    """Per-tile segments of every size class of the sort (one wavefront up to 256 entries, four up to 2048, block sort + merge
    passes beyond) and their boundaries, with duplicate depths."""
    L = lib()
    ntw, nth = len(sizes), 1
    rng = np.random.default_rng(12)
    tl_l, br_l, dep_l = [], [], []
    for t_, n in enumerate(sizes):
        tl_l.append(np.tile(np.array([[t_, 0]], np.int32), (n, 1))); br_l.append(tl_l[-1].copy())
        d = rng.uniform(0.1, 5.0, n).astype(np.float32)
        d[rng.integers(0, n, n // 3)] = 1.25  # ties
        dep_l.append(d)
    tl = np.concatenate(tl_l); br = np.concatenate(br_l); depth = np.concatenate(dep_l)
    perm = rng.permutation(len(depth))
    tl, br, depth = tl[perm], br[perm], depth[perm]
    N = D = len(depth)
    oi, os_, oe = O.bin_sort(tl, br, depth, nth, ntw, D)
    ids = torch.zeros(D, dtype=torch.int32, device=dev())
    st = -torch.ones(ntw, dtype=torch.int32, device=dev()); en = -torch.ones_like(st)
    nb = L.tile_culling_workspace_bytes(N, D, ntw)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev())
    L.tile_culling_aabb_start_end(N, D, nth, ntw, p(T_(tl)), p(T_(br)), p(T_(depth)), p(ids), p(st), p(en), p(ws), nb,
                                  stream())
    torch.cuda.synchronize()
    assert np.array_equal(st.cpu().numpy(), os_) and np.array_equal(en.cpu().numpy(), oe)
    assert np.array_equal(ids.cpu().numpy(), oi)


def test_sort_fuzz_segment_lengths():
    """hypothesis over the per-tile sort: 1 .. 10 tiles with list lengths anywhere in 0 .. 6000 (every register width, the
    four-wavefront path between 257 and 2048 entries, block sort + merge beyond), many equal depths,
    negative depths; lists and start / end against the oracle"""
    import os
    from hypothesis import given, settings, strategies as st, HealthCheck
    L = lib()
    n_ex = int(os.environ.get("GSGEN_FUZZ_EXAMPLES", "40"))

    @settings(max_examples=n_ex, deadline=None, derandomize=(n_ex == 40), suppress_health_check=list(HealthCheck))
    @given(sizes=st.lists(st.one_of(st.integers(0, 300), st.integers(0, 2100), st.integers(0, 6000)), min_size=1, max_size=10),
           seed=st.integers(0, 10_000), ties=st.booleans())
    def run(sizes, seed, ties):
        if sum(sizes) == 0:
            sizes = sizes + [1]
        ntw, nth = len(sizes), 1
        rng = np.random.default_rng(seed)
        tl = np.concatenate([np.tile(np.array([[t_, 0]], np.int32), (n, 1)) for t_, n in enumerate(sizes)])
        depth = rng.uniform(-1.0, 5.0, len(tl)).astype(np.float32)
        if ties:
            depth[rng.integers(0, len(tl), len(tl) // 2)] = 1.25
        perm = rng.permutation(len(depth))
        tl, depth = np.ascontiguousarray(tl[perm]), np.ascontiguousarray(depth[perm])
        br = tl.copy()
        N = D = len(depth)
        oi, os_, oe = O.bin_sort(tl, br, depth, nth, ntw, D)
        ids = torch.zeros(D, dtype=torch.int32, device=dev())
        st_ = -torch.ones(ntw, dtype=torch.int32, device=dev()); en = -torch.ones_like(st_)
        nb = L.tile_culling_workspace_bytes(N, D, ntw)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev())
        L.tile_culling_aabb_start_end(N, D, nth, ntw, p(T_(tl)), p(T_(br)), p(T_(depth)), p(ids), p(st_), p(en), p(ws), nb, stream())
        torch.cuda.synchronize()
        assert np.array_equal(st_.cpu().numpy(), os_) and np.array_equal(en.cpu().numpy(), oe), sizes
        assert np.array_equal(ids.cpu().numpy(), oi), sizes
    run()


def test_run_to_run_determinism_forward(case):
    """The forward has no atomics on the image path and the per-tile order is unique:
    two runs are bit-identical."""
    name, sc, cam, g, h = case
    g2 = hip_geometry(sc, cam, g)
    assert torch.equal(g2["ids"], h["ids"])


@pytest.mark.parametrize("Pc", [8, 16, 32, 64])
def test_wave_reduce_scatter_primitive(Pc):
    """v_permlane32/16_swap + DPP reduce-scatter of the backward against a plain sum."""
    x = np.random.default_rng(Pc).normal(size=(64, Pc)).astype(np.float32)
    xin = T_(x)
    out = torch.zeros(128, device=dev())
    lib().selftest_reduce_scatter(Pc, p(xin), p(out), stream())
    o = out.cpu().numpy()
    tot = x.astype(np.float64).sum(0)
    owner = o[64:].astype(int)
    assert sorted(owner[owner >= 0].tolist()) == list(range(Pc))  # every component owned exactly once
    for lane in range(64):
        if owner[lane] >= 0:
            assert abs(o[lane] - tot[owner[lane]]) < 1e-4


class _DeviceArrays:
    """tile_chain.other_tile_size_chain on the GPU: arrays are torch tensors on the device"""

    class Arr:
        def __init__(self, a):
            self.t = torch.from_numpy(np.ascontiguousarray(a).copy()).to(dev()); self.p = self.t.data_ptr(); self.n = self.t.numel()

        def get(self):
            return self.t.cpu().numpy()

    def __init__(self):
        self.lib, self.stream = lib(), stream()

    def to_dev(self, a):
        return self.Arr(a)


@pytest.mark.parametrize("ts,C,W,H", [(8, 4, 133, 90), (32, 4, 200, 120), (8, 1, 64, 48), (32, 2, 97, 65), (8, 3, 80, 80), (12, 4, 131, 77),
                                      (20, 2, 99, 64), (4, 1, 50, 33), (27, 3, 140, 90), (1, 1, 9, 7)])
def test_other_tile_sizes(ts, C, W, H):
    """tile sizes other than 16 (8, 32, and sides that are no power of two) through the C ABI: count, bin / sort, RGB / scalar / SH forward and backward against the
    oracle at that tile size (the reference takes the tile size as a parameter, conf/base.yaml:132)"""
    from tile_chain import other_tile_size_chain
    other_tile_size_chain(_DeviceArrays(), ts, C, W, H, sync=torch.cuda.synchronize)


def test_chain_fuzz():
    """hypothesis over the per-camera chain on the GPU (see tests/test_cpu_host.py::test_emulated_chain_fuzz): image
    shapes from one pixel to several ragged tiles, 1 .. 3000 Gaussians of any size, every SH degree and tile size,
    opaque scenes"""
    import os
    from hypothesis import given, settings, strategies as st, HealthCheck
    from tile_chain import other_tile_size_chain, FUZZ_ATOL, KNOWN_WORST
    n_ex = int(os.environ.get("GSGEN_FUZZ_EXAMPLES", "40"))

    @settings(max_examples=n_ex, deadline=None, derandomize=(n_ex == 40), suppress_health_check=list(HealthCheck))
    @given(ts=st.sampled_from([8, 16, 32, 5, 12, 27]), C=st.integers(1, 4), W=st.integers(1, 200), H=st.integers(1, 150),
           n=st.integers(1, 3000), seed=st.integers(0, 10_000), svec=st.sampled_from([0.01, 0.05, 0.2]),
           opaque=st.booleans())
    def run(ts, C, W, H, n, seed, svec, opaque):
        other_tile_size_chain(_DeviceArrays(), ts, C, W, H, sync=torch.cuda.synchronize, n=n, seed=seed, svec=svec,
                              opaque=opaque, rtol=1e-3, atol=FUZZ_ATOL, ftol=1e-4)  # a one-pixel image of opaque
        # image-sized splats is ill-conditioned ((final - prefix) / (1 - a G) with a G -> 0.99): tile_chain.FUZZ_ATOL
    run()
    for ts, C, W, H, n, seed, svec, opaque in KNOWN_WORST:
        other_tile_size_chain(_DeviceArrays(), ts, C, W, H, sync=torch.cuda.synchronize, n=n, seed=seed, svec=svec,
                              opaque=opaque, rtol=1e-3, atol=FUZZ_ATOL, ftol=1e-4)
