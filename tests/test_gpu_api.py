"""GPU tests of the Python surface: the `_gs` mirror + autograd Functions (drop-in for
gs/renderer.py), the fused render_frame, and full-size (BASELINE configs[1]) checks."""
import numpy as np
import pytest
import torch

import scenes
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def T_(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def test_reference_call_sequence_render_one():
    """The exact call sequence of GaussianSplattingRenderer.render_one
    (gs/gaussian_splatting.py:1208-1330) on the mirrored API, rgb + bg, forward and backward."""
    from gsgen_amd import _gs as _backend, renderer as R
    sc = scenes.random_scene(1000, seed=0, C=1)
    cam = scenes.Camera(256, 256, fx=256.0)
    ci = R.CameraInfo(*cam.intr)
    g = scenes.oracle_geometry(sc, cam)
    P = {k: T_(sc[k]).requires_grad_(True) for k in ("mean", "qvec", "svec", "color", "alpha")}
    c2w = T_(cam.c2w)
    f_normals, f_pts = (T_(a) for a in ci.get_frustum(cam.c2w))
    mask = torch.zeros(P["mean"].shape[0], dtype=torch.bool, device=dev())
    with torch.no_grad():
        _backend.culling_gaussian_bsphere(P["mean"], P["qvec"], P["svec"], f_normals, f_pts, mask, 6.0)
    assert np.array_equal(mask.cpu().numpy(), g["mask"])
    mean, qvec, svec, color, alpha = (P[k][mask].contiguous() for k in ("mean", "qvec", "svec", "color", "alpha"))
    mean2d, cov, JW, depth = R.project_gaussians(mean, qvec, svec, c2w, True)
    N_with_dub, tl, br = R.tile_culling_aabb_count(mean2d, cov, 16, ci, 6.0)
    assert N_with_dub == g["D"]
    H, W = ci.h, ci.w
    nth, ntw = R.n_tiles(H, W)
    start = -torch.ones([nth * ntw], dtype=torch.int32, device=dev())
    end = -torch.ones([nth * ntw], dtype=torch.int32, device=dev())
    gaussian_ids = torch.zeros([N_with_dub], dtype=torch.int32, device=dev())
    _backend.tile_culling_aabb_start_end(tl, br, gaussian_ids, start, end, depth, nth, ntw)
    assert np.array_equal(gaussian_ids.cpu().numpy(), g["ids"])
    img_topleft = torch.FloatTensor([-ci.cx / ci.fx, -ci.cy / ci.fy]).to(dev())
    bg = torch.rand(H, W, 3, device=dev(), requires_grad=True)
    import ref_autograd
    out = ref_autograd.render_with_T(mean2d, cov, color, alpha, start, end, gaussian_ids, img_topleft, 16, nth, ntw,
                          1.0 / ci.fx, 1.0 / ci.fy, H, W, 1e-4, bg)
    m = g["mask"]
    ref, refT = O.render_rgb_fwd(g["mean2d"], g["cov2d"], sc["color"][m], sc["alpha"][m], g["start"], g["end"],
                                 g["ids"], cam.topleft, 1 / cam.fx, 1 / cam.fy, H, W)
    final = ref + refT * bg.detach().cpu().numpy()
    assert np.abs(out.detach().cpu().numpy() - final).max() <= 1e-4
    go = torch.randn_like(out)
    (out * go).sum().backward()
    gm2, gc2, gcol, ga = O.render_rgb_bwd(g["mean2d"], g["cov2d"], sc["color"][m], sc["alpha"][m], g["start"],
                                          g["end"], g["ids"], final, go.cpu().numpy(), cam.topleft, 1 / cam.fx,
                                          1 / cam.fy, H, W)
    omean, oq, os_ = O.project_bwd(sc["mean"][m], sc["qvec"][m], sc["svec"][m], cam.c2w, gm2, gc2, None, True)
    full = lambda a: a  # noqa: E731
    assert rel_err(P["color"].grad.cpu().numpy()[m], gcol) < 1e-3
    assert rel_err(P["alpha"].grad.cpu().numpy()[m], ga) < 1e-3
    assert rel_err(P["mean"].grad.cpu().numpy()[m], omean) < 2e-3
    assert rel_err(P["qvec"].grad.cpu().numpy()[m], oq) < 2e-3
    assert rel_err(P["svec"].grad.cpu().numpy()[m], os_) < 2e-3
    assert float(P["mean"].grad[~mask].abs().max()) == 0.0
    assert np.abs(bg.grad.cpu().numpy() - go.cpu().numpy() * refT).max() <= 1e-4
    # scalar heads exactly as render_one calls them (depth, opacity, z^2)
    T = torch.ones([H, W, 1], device=dev())
    d_img = ref_autograd.render_scalar(mean2d, cov, depth, alpha, start, end, gaussian_ids, img_topleft, 16, nth, ntw,
                            1.0 / ci.fx, 1.0 / ci.fy, H, W, 1e-4, T).reshape(H, W)
    rd, _ = O.render_scalar_fwd(g["mean2d"], g["cov2d"], g["depth"].ravel(), sc["alpha"][m], g["start"], g["end"],
                                g["ids"], cam.topleft, 1 / cam.fx, 1 / cam.fy, H, W)
    assert np.abs(d_img.detach().cpu().numpy() - rd).max() <= 1e-4 * max(1.0, np.abs(rd).max())


def test_gs_mirror_rejects_bad_tensors():
    from gsgen_amd import _gs
    a = torch.zeros(4, 3, device=dev())
    with pytest.raises(RuntimeError, match="must be a contiguous tensor"):
        _gs.culling_gaussian_bsphere(a.t(), a, a, a, a, torch.zeros(4, dtype=torch.bool, device=dev()), 6.0)
    with pytest.raises(RuntimeError, match="must be an bool tensor"):
        _gs.culling_gaussian_bsphere(a, a, a, a, a, torch.zeros(4, device=dev()), 6.0)
    with pytest.raises(RuntimeError, match="must be a floating tensor"):
        _gs.culling_gaussian_bsphere(a.double(), a, a, a, a, torch.zeros(4, dtype=torch.bool, device=dev()), 6.0)


@pytest.mark.parametrize("C", [0, 1, 4])
def test_fused_frame_matches_oracle(C):
    from gsgen_amd import renderer as R
    sc = scenes.random_scene(3000, seed=4, svec=0.03, C=max(C, 1))
    cam = scenes.Camera(200, 136, fx=180.0, c2w=scenes.orbit(2.4, 20, 60))
    ci = R.CameraInfo(*cam.intr)
    g = scenes.oracle_geometry(sc, cam)
    m = g["mask"]
    colkey = "sh" if C > 0 else "color"
    P = {k: T_(sc[k]).requires_grad_(True) for k in ("mean", "qvec", "svec", "alpha", colkey)}
    buf = R.FrameBuffers(sc["mean"].shape[0], cam.w, cam.h, dev(), D_cap=1024)  # forces the overflow path once
    rgb, T = R.render_frame(P["mean"], P["qvec"], P["svec"], P["alpha"], P[colkey], ci, cam.c2w, buf, C=C)
    if not buf.ensure_capacity():
        assert bool(torch.isnan(rgb.detach()).all()) and bool(torch.isnan(T).all())  # nothing was binned: NaN, never a finite blank image
        rgb, T = R.render_frame(P["mean"], P["qvec"], P["svec"], P["alpha"], P[colkey], ci, cam.c2w, buf, C=C)
        assert buf.ensure_capacity()
    assert int(buf.total.item()) == g["D"]
    assert np.array_equal(buf.mask.cpu().numpy(), m)
    H, W = cam.h, cam.w
    rot = cam.c2w[:3, :3].reshape(-1)
    if C > 0:
        ref = O.render_sh_fwd(g["mean2d"], g["cov2d"], sc["sh"][m], sc["alpha"][m], g["start"], g["end"], g["ids"],
                              cam.topleft, rot, C, 1 / cam.fx, 1 / cam.fy, H, W)
    else:
        ref, _ = O.render_rgb_fwd(g["mean2d"], g["cov2d"], sc["color"][m], sc["alpha"][m], g["start"], g["end"],
                                  g["ids"], cam.topleft, 1 / cam.fx, 1 / cam.fy, H, W)
    if C > 0:
        scenes.assert_sh_image_parity(rgb.detach().cpu().numpy(), ref, g["mean2d"], g["cov2d"], sc["alpha"][m], g["start"],
                                      g["end"], g["ids"], cam.topleft, 1 / cam.fx, 1 / cam.fy, what="fused frame")
    else:
        assert np.abs(rgb.detach().cpu().numpy() - ref).max() <= 1e-4  # every pixel
    go = torch.randn_like(rgb)
    (rgb * go).sum().backward()
    if C > 0:
        gm2, gc2, gcol, ga = O.render_sh_bwd(g["mean2d"], g["cov2d"], sc["sh"][m], sc["alpha"][m], g["start"], g["end"],
                                             g["ids"], ref, go.cpu().numpy(), cam.topleft, rot, C, 1 / cam.fx,
                                             1 / cam.fy, H, W)
    else:
        gm2, gc2, gcol, ga = O.render_rgb_bwd(g["mean2d"], g["cov2d"], sc["color"][m], sc["alpha"][m], g["start"],
                                              g["end"], g["ids"], ref, go.cpu().numpy(), cam.topleft, 1 / cam.fx,
                                              1 / cam.fy, H, W)
    omean, oq, os_ = O.project_bwd(sc["mean"][m], sc["qvec"][m], sc["svec"][m], cam.c2w, gm2, gc2, None, True)
    tol = 2e-3
    assert rel_err(P[colkey].grad.cpu().numpy()[m], gcol) < tol
    assert rel_err(P["alpha"].grad.cpu().numpy()[m], ga) < tol
    assert rel_err(P["mean"].grad.cpu().numpy()[m], omean) < tol
    assert rel_err(P["svec"].grad.cpu().numpy()[m], os_) < tol
    assert rel_err(P["qvec"].grad.cpu().numpy()[m], oq) < tol
    for k in ("mean", "qvec", "svec", "alpha", colkey):
        assert float(P[k].grad[~buf.mask].abs().max()) == 0.0


def test_densify_statistics_over_two_cameras():
    """max_radii2d / mean2d-grad accumulation / cnt (gs/gaussian_splatting.py:1240-1245, :464-469)
    updated inside render_frame, against the oracle's statements on the oracle's own geometry."""
    from gsgen_amd import renderer as R
    sc = scenes.random_scene(4000, seed=9, svec=0.03, C=2)
    N = sc["mean"].shape[0]
    Pm = {k: T_(sc[k]).requires_grad_(True) for k in ("mean", "qvec", "svec", "alpha", "sh")}
    stats = R.DensifyStats(N, dev())
    want = [np.zeros(N, np.float32) for _ in range(3)]
    for az in (40.0, 200.0):
        cam = scenes.Camera(160, 120, fx=150.0, c2w=scenes.orbit(2.4, 10, az))
        ci = R.CameraInfo(*cam.intr)
        buf = R.FrameBuffers(N, cam.w, cam.h, dev())
        rgb, _ = R.render_frame(Pm["mean"], Pm["qvec"], Pm["svec"], Pm["alpha"], Pm["sh"], ci, cam.c2w, buf, C=2,
                                stats=stats)
        go = torch.randn_like(rgb)
        (rgb * go).sum().backward()
        g = scenes.oracle_geometry(sc, cam)
        m = g["mask"]
        assert np.array_equal(buf.mask.cpu().numpy(), m)
        ref = O.render_sh_fwd(g["mean2d"], g["cov2d"], sc["sh"][m], sc["alpha"][m], g["start"], g["end"], g["ids"],
                              cam.topleft, cam.c2w[:3, :3].reshape(-1), 2, 1 / cam.fx, 1 / cam.fy, cam.h, cam.w)
        gm2 = O.render_sh_bwd(g["mean2d"], g["cov2d"], sc["sh"][m], sc["alpha"][m], g["start"], g["end"], g["ids"],
                              ref, go.cpu().numpy(), cam.topleft, cam.c2w[:3, :3].reshape(-1), 2, 1 / cam.fx,
                              1 / cam.fy, cam.h, cam.w)[0]
        cov_full = np.zeros((N, 4), np.float32); cov_full[m] = g["cov2d"].reshape(-1, 4)
        gm_full = np.zeros((N, 2), np.float32); gm_full[m] = gm2
        O.densify_update(cov_full, gm_full, m, *want)
    assert np.array_equal(stats.max_radii2d.cpu().numpy(), want[0])  # same fp32 expression, contraction off
    assert np.array_equal(stats.cnt.cpu().numpy(), want[2])
    assert rel_err(stats.grad_accum.cpu().numpy(), want[1]) < 2e-3
    assert want[2].max() == 2.0 and want[2].min() == 0.0


@pytest.mark.parametrize("pipeline,C,fused,ncam", [(False, 3, 1, 5), ("auto", 3, 1, 5), (True, 0, 1, 5), ("auto", 4, 1, 5), (True, 2, 3, 5),
                                                   ("auto", 4, 1, 11), (False, 0, 1, 5), (True, 4, 1, 2), ("auto", 4, 1, 3)])
def test_batched_cameras_match_one_at_a_time(pipeline, C, fused, ncam):
    """BatchRenderer (one enqueue per stage for the whole batch, or -- pipeline -- for each of two half-batches on two
    streams) == a loop of render_frame: identical images, the gradient of the summed loss, and the same
    densify statistics."""
    from gsgen_amd import renderer as R
    from gsgen_amd.batch import BatchRenderer
    sc = scenes.random_scene(5000, seed=12, svec=0.03, C=max(C, 1))
    N = sc["mean"].shape[0]
    W, H = 176, 128
    cams = [scenes.Camera(W, H, fx=150.0 + 10 * i, c2w=scenes.orbit(2.3 + 0.1 * i, 5 + 10 * i, 70.0 * i)) for i in range(ncam)]
    cis = [R.CameraInfo(*c.intr) for c in cams]
    ck = "sh" if C > 0 else "color"
    keys = ("mean", "qvec", "svec", "alpha", ck)
    gos = [torch.randn(H, W, 3, device=dev(), generator=torch.Generator(device=dev()).manual_seed(i)) for i in range(ncam)]
    bg = torch.tensor([0.2, 0.4, 0.6], device=dev())

    Pa = {k: T_(sc[k]).requires_grad_(True) for k in keys}
    sa = R.DensifyStats(N, dev())
    imgs = []
    for c, ci, go in zip(cams, cis, gos):
        buf = R.FrameBuffers(N, W, H, dev())
        rgb, _ = R.render_frame(Pa["mean"], Pa["qvec"], Pa["svec"], Pa["alpha"], Pa[ck], ci, c.c2w, buf, C=C,
                                bg_rgb=bg, stats=sa)
        (rgb * go).sum().backward()
        imgs.append(rgb.detach())

    Pb = {k: T_(sc[k]).requires_grad_(True) for k in keys}
    sb = R.DensifyStats(N, dev())
    # fused: one compositing launch per batch and direction (gridDim.y = cameras); fused > 1: that many
    # backward segments per tile as well
    br = BatchRenderer(N, W, H, dev(), max_batch=ncam, pipeline=pipeline, segments=max(fused, 1))
    for _ in range(2):  # second pass: the slots and the pinned camera block are reused
        for k in keys:
            Pb[k].grad = None
        sb = R.DensifyStats(N, dev())
        rgb_b, T_b = br.render(Pb["mean"], Pb["qvec"], Pb["svec"], Pb["alpha"], Pb[ck], cis, [c.c2w for c in cams],
                               C=C, bg_rgb=bg, stats=sb)
        assert br.ensure_capacity(ncam)
        (rgb_b * torch.stack(gos, 0)).sum().backward()
        torch.cuda.synchronize()
        assert rgb_b.shape == (ncam, H, W, 3) and T_b.shape == (ncam, H, W, 1)
        for i in range(ncam):
            assert torch.equal(rgb_b[i].detach(), imgs[i])
        for k in keys:
            assert rel_err(Pb[k].grad.cpu().numpy(), Pa[k].grad.cpu().numpy()) < 1e-4, k
        assert torch.equal(sb.max_radii2d, sa.max_radii2d) and torch.equal(sb.cnt, sa.cnt)
        assert rel_err(sb.grad_accum.cpu().numpy(), sa.grad_accum.cpu().numpy()) < 1e-5
    with pytest.raises(ValueError):
        br.render(Pb["mean"], Pb["qvec"], Pb["svec"], Pb["alpha"], Pb[ck], cis * 2, [c.c2w for c in cams] * 2, C=C)


@pytest.mark.parametrize("B", [8, 5])
def test_batched_binning_push_and_pull_forms_against_the_oracle(B):
    """gsgen_frame_geometry_batch through both binning forms -- 8 views of 5 chunks are 40 workgroups: the push kernels
    (per-tile counters in LDS; a rectangle of more than 12 tiles is finished by the whole wavefront); 5 views are 25: the pull
    kernels -- on a scene with two dozen giant, near Gaussians whose rectangles cover hundreds of tiles: every view's lists are
    the oracle's, bit for bit."""
    from gsgen_amd import _capi, renderer as R
    lib = _capi.load()
    W, H = 304, 208
    sc = scenes.random_scene(9000, seed=35, svec=0.03)
    rng = np.random.default_rng(2)
    big = rng.choice(9000, 24, replace=False)
    sc["svec"][big] *= 25.0          # rectangles of up to the whole 19 x 13 tile grid
    N = sc["mean"].shape[0]
    cams = [scenes.Camera(W, H, fx=230.0 + 9 * i, c2w=scenes.orbit(2.3 + 0.05 * i, 25 - 8 * i, 40.0 + 45 * i)) for i in range(B)]
    gs_ = [scenes.oracle_geometry(sc, c) for c in cams]
    assert max(int(((g["br"] - g["tl"] + 1).clip(min=0).prod(-1)).max()) for g in gs_) > 150   # tiles of the largest rectangle
    cam_dev = [T_(R.CameraInfo(*c.intr).pack(c.c2w)) for c in cams]
    mean, qvec, svec = T_(sc["mean"]), T_(sc["qvec"]), T_(sc["svec"])
    p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    bufs = [R.FrameBuffers(N, W, H, dev(), D_cap=g["D"] + 16) for g in gs_]
    arr = (_capi.GeometryView * B)()
    for a, b_, cd in zip(arr, bufs, cam_dev):
        b_.ids.fill_(-3)
        a.cam, a.mean2d, a.cov2d, a.depth, a.mask = p(cd), p(b_.mean2d), p(b_.cov2d), p(b_.depth), p(b_.mask)
        a.gaussian_ids, a.start, a.end, a.total = p(b_.ids), p(b_.start), p(b_.end), p(b_.total)
        a.workspace, a.workspace_bytes, a.D_cap = p(b_.ws), b_.ws.numel(), b_.D_cap
    bws = torch.empty(lib.frame_batch_workspace_bytes(B), device=dev(), dtype=torch.uint8)
    lib.frame_geometry_batch(B, arr, N, p(mean), p(qvec), p(svec), W, H, p(bws), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    longest = 0
    for g, b_ in zip(gs_, bufs):
        assert int(b_.total.item()) == g["D"]
        full = np.nonzero(g["mask"])[0]
        assert np.array_equal(b_.start.cpu().numpy().ravel(), g["start"]) and np.array_equal(b_.end.cpu().numpy().ravel(), g["end"])
        assert np.array_equal(b_.ids.cpu().numpy()[:g["D"]], full[g["ids"]])
        longest = max(longest, int((g["end"] - g["start"]).max()))
    assert longest > 64


def test_batched_binning_forms_agree_on_random_batches():
    """the batched binning takes the push form (per-tile counters in LDS, slots in arrival order) from 32 (chunk, view)
    workgroups on, the per-camera entry point the pull form (no atomics, ascending id): the same lists on a dozen random
    batches -- ragged image sizes, 1 .. 9 views, scales from sub-pixel to half the image, pair buffers with and without room"""
    from gsgen_amd import _capi, renderer as R
    lib = _capi.load()
    rng = np.random.default_rng(11)
    p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    n_push = n_pull = 0
    for trial in range(12):
        W, H = int(rng.integers(40, 400)), int(rng.integers(40, 300))
        B, N = int(rng.integers(1, 10)), int(rng.integers(1, 24000))
        if (N + 2047) // 2048 * B >= 32:
            n_push += 1
        else:
            n_pull += 1
        sc = scenes.random_scene(N, seed=100 + trial, svec=float(rng.choice([0.004, 0.03, 0.3])), spread=float(rng.uniform(0.3, 1.5)))
        cams = [scenes.Camera(W, H, fx=float(rng.uniform(0.5, 1.6) * W), c2w=scenes.orbit(float(rng.uniform(1.5, 3.0)),
                float(rng.uniform(-40, 60)), float(rng.uniform(0, 360)))) for _ in range(B)]
        cam_dev = [T_(R.CameraInfo(*c.intr).pack(c.c2w)) for c in cams]
        mean, qvec, svec = T_(sc["mean"]), T_(sc["qvec"]), T_(sc["svec"])
        s_ = torch.cuda.current_stream().cuda_stream
        outs = {}
        for form in ("batch", "per view"):
            cap = 400_000 if trial % 4 else 50   # (every fourth batch: too small a pair buffer -- nothing binned, sizes reported)
            bufs = [R.FrameBuffers(N, W, H, dev(), D_cap=cap) for _ in cams]
            for b_ in bufs:
                b_.ids.fill_(-3)
            if form == "batch":
                arr = (_capi.GeometryView * B)()
                for a, b_, cd in zip(arr, bufs, cam_dev):
                    a.cam, a.mean2d, a.cov2d, a.depth, a.mask = p(cd), p(b_.mean2d), p(b_.cov2d), p(b_.depth), p(b_.mask)
                    a.gaussian_ids, a.start, a.end, a.total = p(b_.ids), p(b_.start), p(b_.end), p(b_.total)
                    a.workspace, a.workspace_bytes, a.D_cap = p(b_.ws), b_.ws.numel(), b_.D_cap
                bws = torch.empty(lib.frame_batch_workspace_bytes(B), device=dev(), dtype=torch.uint8)
                lib.frame_geometry_batch(B, arr, N, p(mean), p(qvec), p(svec), W, H, p(bws), s_)
            else:
                for b_, cd in zip(bufs, cam_dev):
                    lib.frame_geometry(N, p(mean), p(qvec), p(svec), p(cd), W, H, b_.D_cap, p(b_.mean2d), p(b_.cov2d),
                                       p(b_.depth), p(b_.mask), p(b_.ids), p(b_.start), p(b_.end), p(b_.total), p(b_.ws),
                                       b_.ws.numel(), s_)
            torch.cuda.synchronize()
            outs[form] = [(int(b_.total.item()), b_.start.cpu().numpy().copy(), b_.end.cpu().numpy().copy(),
                           b_.ids.cpu().numpy().copy()) for b_ in bufs]
        for v, (a_, b_) in enumerate(zip(outs["batch"], outs["per view"])):
            assert a_[0] == b_[0], (trial, v)
            assert np.array_equal(a_[1], b_[1]) and np.array_equal(a_[2], b_[2]), (trial, v)
            assert np.array_equal(a_[3], b_[3]), (trial, v)
    assert n_push >= 3 and n_pull >= 3, (n_push, n_pull)


def test_batched_geometry_and_projection_backward_c_abi():
    """gsgen_frame_geometry_batch leaves bit for bit what one gsgen_frame_geometry call per view leaves (lists,
    records, mask, pair count, launch order, one view overflowing its pair buffer), and
    gsgen_project_gaussians_backward_batch equals the sum of the masked per-view backwards; both against
    the oracle through the per-view entry points they are compared with (test_fused_frame_matches_oracle)."""
    import ctypes
    from gsgen_amd import _capi, renderer as R
    lib = _capi.load()
    W, H, B = 208, 144, 5
    sc = scenes.random_scene(9000, seed=33, svec=0.03)
    N = sc["mean"].shape[0]
    cams = [scenes.Camera(W, H, fx=170.0 + 11 * i, c2w=scenes.orbit(2.2 + 0.1 * i, 30 - 12 * i, 50.0 + 67 * i)) for i in range(B)]
    cam_dev = [T_(R.CameraInfo(*c.intr).pack(c.c2w)) for c in cams]
    mean, qvec, svec = T_(sc["mean"]), T_(sc["qvec"]), T_(sc["svec"])
    p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    s = torch.cuda.current_stream().cuda_stream
    Ds = [scenes.oracle_geometry(sc, c)["D"] for c in cams]
    caps = [d + 64 for d in Ds]
    caps[3] = Ds[3] - 1  # this view overflows: nothing binned, required size reported

    def run(batched):
        bufs = [R.FrameBuffers(N, W, H, dev(), D_cap=c) for c in caps]
        for b_ in bufs:
            b_.ids.fill_(-3); b_.ws.zero_()
        if batched:
            arr = (_capi.GeometryView * B)()
            for a, b_, cd in zip(arr, bufs, cam_dev):
                a.cam, a.mean2d, a.cov2d, a.depth, a.mask = p(cd), p(b_.mean2d), p(b_.cov2d), p(b_.depth), p(b_.mask)
                a.gaussian_ids, a.start, a.end, a.total = p(b_.ids), p(b_.start), p(b_.end), p(b_.total)
                a.workspace, a.workspace_bytes, a.D_cap = p(b_.ws), b_.ws.numel(), b_.D_cap
            bws = torch.empty(lib.frame_batch_workspace_bytes(B), device=dev(), dtype=torch.uint8)
            lib.frame_geometry_batch(B, arr, N, p(mean), p(qvec), p(svec), W, H, p(bws), s)
        else:
            for b_, cd in zip(bufs, cam_dev):
                lib.frame_geometry(N, p(mean), p(qvec), p(svec), p(cd), W, H, b_.D_cap, p(b_.mean2d), p(b_.cov2d),
                                   p(b_.depth), p(b_.mask), p(b_.ids), p(b_.start), p(b_.end), p(b_.total), p(b_.ws),
                                   b_.ws.numel(), s)
        torch.cuda.synchronize()
        return bufs
    ref, got = run(False), run(True)
    T = ref[0].nth * ref[0].ntw
    for i, (r, g) in enumerate(zip(ref, got)):
        assert int(g.total.item()) == Ds[i]
        for k in ("mean2d", "cov2d", "depth", "mask", "ids", "start", "end"):
            assert torch.equal(getattr(r, k), getattr(g, k)), (i, k)
        # the workspace's public part too (list lengths + control words, offsets), except the launch order: tiles of one
        # length bucket are placed by LDS atomics, so only "same buckets, a permutation" is defined.  (Behind it the batch's
        # push binning fills the key segments in no particular order within a chunk: the sort makes the lists.)
        wr, wg = r.ws.clone(), g.ws.clone()
        o = r.tile_order() - r.ws.data_ptr()
        orders = []
        for w in (wr, wg):
            orders.append(w[o:o + 4 * T].view(torch.int32).clone())
        assert torch.equal(wr[:o], wg[:o]), i
        cnt = (r.end - r.start).clamp(min=0)
        for od in orders:
            assert torch.equal(od.sort().values, torch.arange(T, device=dev(), dtype=torch.int32))
            b = cnt[od.long()] >> 3
            assert bool((b[1:] <= b[:-1]).all())  # longest lists first, in buckets of 8 entries
    assert bool((got[3].start == -2).all()) and bool((got[0].start >= 0).any())

    gen = torch.Generator(device=dev()).manual_seed(1)
    g2d = torch.randn(B, 6 * N, device=dev(), generator=gen)
    gdp = torch.randn(B, N, device=dev(), generator=gen)
    want = torch.zeros(10 * N, device=dev(), dtype=torch.float64)
    for i in range(B):
        o = torch.empty(10 * N, device=dev())
        lib.project_gaussians_backward_masked(N, p(mean), p(qvec), p(svec), p(cam_dev[i]), 0, p(got[i].mask),
                                              p(g2d[i]), p(g2d[i]) + 8 * N, p(gdp[i]), p(o), p(o) + 12 * N, p(o) + 28 * N, s)
        want += o.double()
    tab = lambda vals: (ctypes.c_void_p * B)(*vals)  # noqa: E731
    o = torch.full((10 * N,), 5.0, device=dev())
    lib.project_gaussians_backward_batch(B, N, p(mean), p(qvec), p(svec), tab([p(c) for c in cam_dev]), 0,
                                         tab([p(b_.mask) for b_ in got]), tab([p(g2d[i]) for i in range(B)]),
                                         tab([p(g2d[i]) + 8 * N for i in range(B)]), tab([p(gdp[i]) for i in range(B)]),
                                         p(o), p(o) + 12 * N, p(o) + 28 * N, s)
    torch.cuda.synchronize()
    assert rel_err(o.cpu().numpy(), want.cpu().numpy()) < 2e-6


def test_fused_adam_tracks_torch_adam_through_a_render():
    """optim.FusedAdam: parameters are views of one flat buffer, autograd accumulates straight into
    the flat gradient, one kernel updates every field; against torch.optim.Adam on the same grads"""
    from gsgen_amd import renderer as R
    from gsgen_amd.optim import FusedAdam
    sc = scenes.random_scene(2000, seed=3, svec=0.04, C=2)
    cam = scenes.Camera(96, 80, fx=90.0, c2w=scenes.orbit(2.4, 10, 30))
    ci = R.CameraInfo(*cam.intr)
    keys = ("mean", "qvec", "svec", "alpha", "sh")
    lrs = {"mean": 5e-3, "qvec": 1e-3, "svec": 5e-3, "alpha": 3e-2, "sh": 1e-2}
    fa = FusedAdam({k: T_(sc[k]) for k in keys}, lrs)
    ref = {k: T_(sc[k]).requires_grad_(True) for k in keys}
    opt = torch.optim.Adam([{"params": [ref[k]], "lr": lrs[k]} for k in keys], lr=0.0, eps=1e-15)
    buf = R.FrameBuffers(sc["mean"].shape[0], cam.w, cam.h, dev())
    target = torch.rand(cam.h, cam.w, 3, device=dev())
    for it in range(4):
        fa.zero_grad(); opt.zero_grad()
        for P_ in (fa.params, ref):
            rgb, _ = R.render_frame(P_["mean"], P_["qvec"], P_["svec"], P_["alpha"], P_["sh"], ci, cam.c2w, buf, C=2)
            ((rgb - target) ** 2).sum().backward()
        assert fa.params["sh"].grad.data_ptr() == fa.grad[fa.n - fa.params["sh"].numel():].data_ptr()  # still the view
        for k in keys:
            # identical parameters in the first iteration: the two backward passes differ by the order of their atomics only.
            # Afterwards the two optimisers' parameters differ in the last bits, and a pixel whose a*G sits on the 1/255
            # threshold may decide differently in the two renders (a 1e-4-class change of a few Gaussians' gradients)
            assert rel_err(fa.params[k].grad.cpu().numpy(), ref[k].grad.cpu().numpy()) < (1e-5 if it == 0 else 1e-3), (it, k)  # (later steps: two trajectories, atomics summed in different orders)
        new_lrs = {k: v / (it + 1) for k, v in lrs.items()}
        for grp, k in zip(opt.param_groups, keys):
            grp["lr"] = new_lrs[k]
        fa.all_reduce_grad()  # no process group: a no-op
        fa.step(new_lrs); opt.step()
        for k in keys:
            # (the two backward passes sum their atomics in different orders; Adam's first steps are sign-like, so a gradient
            # that differs in its last bits moves a parameter by up to ~lr * 1e-3: seen between 0.3e-5 and 2.1e-5 over runs)
            assert rel_err(fa.params[k].detach().cpu().numpy(), ref[k].detach().cpu().numpy()) < 4e-5, (it, k)


@pytest.mark.parametrize("C", [2, 4])
def test_segmented_backward_matches_per_tile_backward(C):
    """FrameBuffers(segments=6): forward checkpoints + one backward workgroup per (tile, 32-entry
    segment) against the per-tile backward, long lists (low opacity keeps pixels alive)"""
    from gsgen_amd import renderer as R
    sc = scenes.random_scene(6000, seed=17, svec=0.06, C=C)
    sc["alpha"] = (sc["alpha"] * 0.2).astype(np.float32)
    cam = scenes.Camera(160, 112, fx=120.0, c2w=scenes.orbit(2.3, 15, 80))
    ci = R.CameraInfo(*cam.intr)
    keys = ("mean", "qvec", "svec", "alpha", "sh")
    go = None
    res = []
    for segments in (1, 6):
        P_ = {k: T_(sc[k]).requires_grad_(True) for k in keys}
        buf = R.FrameBuffers(sc["mean"].shape[0], cam.w, cam.h, dev(), segments=segments)
        rgb, T = R.render_frame(P_["mean"], P_["qvec"], P_["svec"], P_["alpha"], P_["sh"], ci, cam.c2w, buf, C=C,
                                bg_rgb=torch.tensor([0.2, 0.3, 0.1], device=dev()))
        assert buf.ensure_capacity()
        if go is None:
            go = torch.randn_like(rgb)
            assert int((buf.end - buf.start).max()) > 32 * 5 + 10  # every segment and a long last one are exercised
        (rgb * go).sum().backward()
        res.append((rgb.detach(), {k: P_[k].grad.clone() for k in keys}))
    assert torch.equal(res[0][0], res[1][0])
    for k in keys:
        if k == "qvec":
            continue
        assert rel_err(res[1][1][k].cpu().numpy(), res[0][1][k].cpu().numpy()) < 2e-5, k


@pytest.mark.parametrize("kind", ["bcircle", "prob"])
def test_legacy_renderer_call_sequence(kind):
    """gs/renderer.py:1395-1520 (GaussianRenderer.render, tile_culling_type bcircle / prob) call for call through
    the `_gs` mirror: count -> .sum().item() -> prepare_image_sort / image_sort -> offset[-1] = total ->
    tile_based_vol_rendering (CSR form), against the oracle (itself pinned to tile_ops.h bit for bit)."""
    from gsgen_amd import _gs
    sc = scenes.random_scene(1500, seed=14, svec=0.05)
    cam = scenes.Camera(112, 80, fx=100.0, c2w=scenes.orbit(2.4, 12, 200))
    g = scenes.oracle_geometry(sc, cam)
    m = g["mask"]
    m2, c2, dep = g["mean2d"], g["cov2d"].reshape(-1, 4), g["depth"].ravel()
    radius = np.sqrt(np.maximum(c2[:, 0], c2[:, 3])).astype(np.float32)
    mean, cov, depth = T_(m2), T_(c2.reshape(-1, 2, 2)), T_(dep.reshape(-1, 1))
    color, alpha = T_(sc["color"][m]), T_(sc["alpha"][m])
    nth, ntw = cam.tiles; n_tiles = nth * ntw; H, W = cam.h, cam.w
    topleft = T_(cam.topleft); psx, psy = 1 / cam.fx, 1 / cam.fy
    num = torch.zeros(n_tiles, dtype=torch.int32, device=dev())
    if kind == "bcircle":
        shape_np = (radius * np.float32(2.5)).astype(np.float32)
        _gs.count_num_gaussians_each_tile_bcircle(mean, T_(shape_np), topleft, 16, nth, ntw, psx, psy, num)
        want_n = O.legacy_count(1, m2, shape_np, cam.topleft, 16, nth, ntw, psx, psy)
    else:
        shape_np = c2
        _gs.count_num_gaussians_each_tile(mean, cov, topleft, 16, nth, ntw, psx, psy, num, 0.01)
        want_n = O.legacy_count(0, m2, c2, cam.topleft, 16, nth, ntw, psx, psy, 0.01)
    assert np.array_equal(num.cpu().numpy(), want_n)
    total = int(num.sum().item())
    tiledepth = torch.zeros(total, dtype=torch.float64, device=dev())
    offset = torch.zeros(n_tiles + 1, dtype=torch.int32, device=dev())
    ids = torch.zeros(total, dtype=torch.int32, device=dev())
    if kind == "bcircle":
        _gs.prepare_image_sort(ids, tiledepth, depth, num, offset, mean, T_(shape_np), topleft, 16, nth, ntw, psx, psy)
    else:
        _gs.image_sort(ids, tiledepth, depth, num, offset, mean, cov, topleft, 16, nth, ntw, psx, psy, 0.01)
    offset[-1] = total
    w_ids, w_td, w_n, w_off = O.legacy_image_sort(1 if kind == "bcircle" else 0, dep, want_n, m2, shape_np, cam.topleft, 16,
                                                  nth, ntw, psx, psy, 0.01)
    assert np.array_equal(ids.cpu().numpy(), w_ids)
    assert np.array_equal(tiledepth.cpu().numpy().view(np.uint64), w_td)
    assert np.array_equal(num.cpu().numpy(), w_n) and np.array_equal(offset.cpu().numpy()[:-1], w_off)
    _gs.debug_check_tiledepth(offset.cpu(), torch.from_numpy(np.sort(w_td).view(np.float64)))  # sorted keys pass the reference's check
    out = torch.zeros(H, W, 3, device=dev())
    _gs.tile_based_vol_rendering(mean, cov, color, alpha, offset, ids, out, topleft, 16, nth, ntw, psx, psy, H, W, 1e-4)
    off = offset.cpu().numpy()
    st, en = off[:-1].copy(), off[1:].copy()
    st[en == st] = -1; en[st == -1] = -1
    ref, _ = O.render_rgb_fwd(m2, g["cov2d"], sc["color"][m], sc["alpha"][m], st, en, w_ids, cam.topleft, psx, psy, H, W)
    assert np.abs(out.cpu().numpy() - ref).max() <= 1e-5


@pytest.mark.parametrize("detach,pipeline", [(True, False), (False, "auto"), (False, True)])
def test_batched_fused_heads_match_oracle(detach, pipeline):
    """BatchRenderer.render_heads: rgb + depth + opacity + depth^2 of 3 cameras in one autograd node against the
    oracle's four separate passes and its projection backward (with the depth gradient of the two depth heads)"""
    from gsgen_amd import renderer as R
    from gsgen_amd.batch import BatchRenderer
    sc = scenes.random_scene(2500, seed=23, svec=0.04)
    N = sc["mean"].shape[0]
    W, H = 128, 96
    cams = [scenes.Camera(W, H, fx=110.0 + 15 * i, c2w=scenes.orbit(2.3 + 0.15 * i, 8 + 12 * i, 100.0 * i)) for i in range(3)]
    cis = [R.CameraInfo(*c.intr) for c in cams]
    keys = ("mean", "qvec", "svec", "alpha", "color")
    P_ = {k: T_(sc[k]).requires_grad_(True) for k in keys}
    # one enqueue per stage for the batch (gsgen_vol_render_rgbd_batch ...), or for each of two half-batches
    br = BatchRenderer(N, W, H, dev(), max_batch=3, pipeline=pipeline)
    rgb, dpt, opa, z2, T = br.render_heads(P_["mean"], P_["qvec"], P_["svec"], P_["alpha"], P_["color"], cis,
                                           [c.c2w for c in cams], detach_depth=detach)
    assert br.ensure_capacity(3)
    gen = torch.Generator(device=dev()).manual_seed(5)
    gos = [torch.randn(3, H, W, c, device=dev(), generator=gen) for c in (3, 1, 1, 1)]
    ((rgb * gos[0]).sum() + (dpt * gos[1]).sum() + (opa * gos[2]).sum() + (z2 * gos[3]).sum()).backward()
    want = {k: np.zeros_like(sc[k], dtype=np.float64) for k in keys}
    for i, cam in enumerate(cams):
        g = scenes.oracle_geometry(sc, cam)
        m = g["mask"]
        geo = (g["start"], g["end"], g["ids"], cam.topleft, 1 / cam.fx, 1 / cam.fy, H, W)
        dv = np.ascontiguousarray(g["depth"].ravel()); al = sc["alpha"][m]
        o_rgb, _ = O.render_rgb_fwd(g["mean2d"], g["cov2d"], sc["color"][m], al, *geo)
        heads = [(dv, dpt), (np.ones_like(dv), opa), (dv * dv, z2)]
        outs = [O.render_scalar_fwd(g["mean2d"], g["cov2d"], v, al, *geo)[0] for v, _ in heads]
        assert np.abs(rgb[i].detach().cpu().numpy() - o_rgb).max() <= 1e-5
        for (v, got), o_ in zip(heads, outs):
            assert np.abs(got[i, ..., 0].detach().cpu().numpy() - o_).max() <= 1e-5 * max(1.0, np.abs(o_).max())
        r = O.render_rgb_bwd(g["mean2d"], g["cov2d"], sc["color"][m], al, g["start"], g["end"], g["ids"], o_rgb,
                             gos[0][i].cpu().numpy(), cam.topleft, 1 / cam.fx, 1 / cam.fy, H, W)
        ss = [O.render_scalar_bwd(g["mean2d"], g["cov2d"], v, al, g["start"], g["end"], g["ids"], o_,
                                  np.ascontiguousarray(gos[1 + k][i, ..., 0].cpu().numpy()), cam.topleft, 1 / cam.fx, 1 / cam.fy, H, W)
              for k, ((v, _), o_) in enumerate(zip(heads, outs))]
        gm2 = r[0] + sum(x[0] for x in ss); gc2 = r[1] + sum(x[1] for x in ss); ga = r[3] + sum(x[3] for x in ss)
        gdepth = ss[0][2] + 2.0 * dv * ss[2][2]  # d/d(depth) through the depth and the depth^2 heads
        om, oq, os_ = O.project_bwd(sc["mean"][m], sc["qvec"][m], sc["svec"][m], cam.c2w, gm2, gc2, gdepth.reshape(-1, 1), detach)
        want["mean"][m] += om; want["qvec"][m] += oq; want["svec"][m] += os_
        want["alpha"][m] += ga; want["color"][m] += r[2]
    for k in keys:
        assert rel_err(P_[k].grad.cpu().numpy(), want[k]) < 2e-3, k


def test_full_size_cfg2():
    """BASELINE configs[1] (100k Gaussians, 800x800, SH degree 3) through the fused path:
    pair count and per-tile lists exact, image within 1e-4 of the oracle, per-tile lists
    sorted, backward linear in grad_out."""
    from gsgen_amd import renderer as R, _capi
    sc = scenes.pointe_scene(100_000, seed=0, C=4)
    cam = scenes.Camera(800, 800, fx=800.0, c2w=scenes.orbit(2.5, 15, 30))
    ci = R.CameraInfo(*cam.intr)
    g = scenes.oracle_geometry(sc, cam)
    m = g["mask"]
    P = {k: T_(sc[k]).requires_grad_(True) for k in ("mean", "qvec", "svec", "alpha", "sh")}
    buf = R.FrameBuffers(100_000, 800, 800, dev())
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev())
    rgb, T = R.render_frame(P["mean"], P["qvec"], P["svec"], P["alpha"], P["sh"], ci, cam.c2w, buf, C=4, bg_rgb=bg)
    assert buf.ensure_capacity()
    D = int(buf.total.item())
    assert D == g["D"]
    start, end, ids = buf.start.cpu().numpy(), buf.end.cpu().numpy(), buf.ids.cpu().numpy()[:D]
    assert np.array_equal(start, g["start"]) and np.array_equal(end, g["end"])
    # ids index the UNculled array here; map the oracle's compacted ids back
    full_idx = np.nonzero(m)[0]
    assert np.array_equal(ids, full_idx[g["ids"]])
    # sortedness property: (depth bits, id) ascending inside every tile
    dep = buf.depth.cpu().numpy().ravel().view(np.uint32).astype(np.uint64)
    key = (dep[ids] << np.uint64(32)) | ids.astype(np.uint64)
    for t in np.nonzero(start >= 0)[0][::37]:
        seg = key[start[t]:end[t]]
        assert np.all(seg[1:] > seg[:-1])
    rot = cam.c2w[:3, :3].reshape(-1)
    ref = O.render_sh_fwd(g["mean2d"], g["cov2d"], sc["sh"][m], sc["alpha"][m], g["start"], g["end"], g["ids"],
                          cam.topleft, rot, 4, 1 / 800, 1 / 800, 800, 800, bg=bg.cpu().numpy())
    scenes.assert_sh_image_parity(rgb.detach().cpu().numpy(), ref, g["mean2d"], g["cov2d"], sc["alpha"][m], g["start"],
                                  g["end"], g["ids"], cam.topleft, 1 / 800, 1 / 800, what="cfg2")  # every pixel (north_star)
    go = torch.randn_like(rgb)
    (rgb * go).sum().backward()
    g1 = {k: P[k].grad.clone() for k in P}
    for k in P:
        P[k].grad = None
    rgb2, _ = R.render_frame(P["mean"], P["qvec"], P["svec"], P["alpha"], P["sh"], ci, cam.c2w, buf, C=4, bg_rgb=bg)
    assert torch.equal(rgb2, rgb)  # forward is run-to-run bit-identical
    (rgb2 * (2.0 * go)).sum().backward()
    for k in P:
        if k == "qvec":
            continue  # isotropic svec: d/dq is pure rounding noise (|grad| ~ 1e-5), nothing to compare
        a, b = P[k].grad, 2.0 * g1[k]
        assert float((a - b).abs().max() / (b.abs().max() + 1e-30)) < 1e-3  # linearity (atomics reorder sums)
    gm2, gc2, gsh, ga = O.render_sh_bwd(g["mean2d"], g["cov2d"], sc["sh"][m], sc["alpha"][m], g["start"], g["end"],
                                        g["ids"], ref, go.cpu().numpy(), cam.topleft, rot, 4, 1 / 800, 1 / 800, 800, 800)
    assert rel_err(g1["sh"].cpu().numpy()[m], gsh) < 2e-3
    assert rel_err(g1["alpha"].cpu().numpy()[m], ga) < 2e-3


def test_full_size_cfg2_batched_launches():
    """BASELINE configs[1] as the bench runs it by default: 8 cameras per launch and stage through BatchRenderer
    against the same 8 cameras one at a time through render_frame (itself checked against the oracle at this
    size in test_full_size_cfg2): images bit-identical, gradient of the summed loss within atomics reordering,
    and the batch's backward linear in grad_out."""
    from gsgen_amd import renderer as R
    from gsgen_amd.batch import BatchRenderer
    N, W, H, B = 100_000, 800, 800, 8
    sc = scenes.pointe_scene(N, seed=0, C=4)
    cams = [scenes.Camera(W, H, fx=800.0, c2w=scenes.orbit(2.5, 15.0, 30.0 + 45.0 * i)) for i in range(B)]
    cis = [R.CameraInfo(*c.intr) for c in cams]
    keys = ("mean", "qvec", "svec", "alpha", "sh")
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev())
    go = torch.randn(B, H, W, 3, device=dev(), generator=torch.Generator(device=dev()).manual_seed(7))
    Pa = {k: T_(sc[k]).requires_grad_(True) for k in keys}
    buf = R.FrameBuffers(N, W, H, dev())
    imgs = []
    for i in range(B):
        rgb, _ = R.render_frame(Pa["mean"], Pa["qvec"], Pa["svec"], Pa["alpha"], Pa["sh"], cis[i], cams[i].c2w, buf, C=4,
                                bg_rgb=bg)
        assert buf.ensure_capacity()
        (rgb * go[i]).sum().backward()
        imgs.append(rgb.detach())
    Pb = {k: T_(sc[k]).requires_grad_(True) for k in keys}
    br = BatchRenderer(N, W, H, dev(), max_batch=B)
    rgb_b, T_b = br.render(Pb["mean"], Pb["qvec"], Pb["svec"], Pb["alpha"], Pb["sh"], cis, [c.c2w for c in cams], C=4,
                           bg_rgb=bg)
    assert br.ensure_capacity(B)
    for i in range(B):
        assert torch.equal(rgb_b[i].detach(), imgs[i]), i
    (rgb_b * go).sum().backward()
    g1 = {k: Pb[k].grad.clone() for k in keys}
    for k in keys:
        if k == "qvec":
            continue  # isotropic svec: d/dq is rounding noise
        assert rel_err(g1[k].cpu().numpy(), Pa[k].grad.cpu().numpy()) < 1e-4, k
    for k in keys:
        Pb[k].grad = None
    rgb_c, _ = br.render(Pb["mean"], Pb["qvec"], Pb["svec"], Pb["alpha"], Pb["sh"], cis, [c.c2w for c in cams], C=4,
                         bg_rgb=bg)
    assert torch.equal(rgb_c, rgb_b)  # run-to-run bit-identical
    (rgb_c * (2.0 * go)).sum().backward()
    for k in keys:
        if k != "qvec":
            assert rel_err(Pb[k].grad.cpu().numpy(), 2.0 * g1[k].cpu().numpy()) < 1e-3, k


def test_fused_rgb_heads_match_four_reference_passes():
    """render_rgb_heads == render_with_T + render_scalar x3 of render_one, forward and backward."""
    from gsgen_amd import renderer as R
    sc = scenes.random_scene(1500, seed=8, svec=0.04, C=1)
    cam = scenes.Camera(176, 120, fx=150.0, c2w=scenes.orbit(2.4, 12, 200))
    g = scenes.oracle_geometry(sc, cam)
    m = g["mask"]
    H, W = cam.h, cam.w
    nth, ntw = cam.tiles
    leaf = lambda a: T_(a).requires_grad_(True)  # noqa: E731
    mean2d, cov2d, color, depth, alpha = leaf(g["mean2d"]), leaf(g["cov2d"]), leaf(sc["color"][m]), leaf(g["depth"]), leaf(sc["alpha"][m])
    bg = torch.rand(H, W, 3, device=dev(), requires_grad=True)
    rgb, dimg, opac, z2, T = R.render_rgb_heads(mean2d, cov2d, color, depth, alpha, T_(g["start"]), T_(g["end"]),
                                                T_(g["ids"]), T_(cam.topleft), nth, ntw, 1 / cam.fx, 1 / cam.fy, H, W,
                                                1e-4, bg)
    geo = (g["start"], g["end"], g["ids"], cam.topleft, 1 / cam.fx, 1 / cam.fy, H, W)
    a = (g["mean2d"], g["cov2d"])
    al = sc["alpha"][m]
    o_rgb, o_T = O.render_rgb_fwd(*a, sc["color"][m], al, *geo)
    dv = g["depth"].ravel()
    o_d, _ = O.render_scalar_fwd(*a, dv, al, *geo)
    o_o, _ = O.render_scalar_fwd(*a, np.ones_like(dv), al, *geo)
    o_z, _ = O.render_scalar_fwd(*a, dv * dv, al, *geo)
    final = o_rgb + o_T * bg.detach().cpu().numpy()
    assert np.abs(rgb.detach().cpu().numpy() - final).max() <= 1e-4
    for x, y in ((dimg, o_d), (opac, o_o), (z2, o_z)):
        assert np.abs(x.detach().cpu().numpy()[..., 0] - y).max() <= 1e-4 * max(1.0, np.abs(y).max())
    rng = np.random.default_rng(4)
    go = [rng.normal(size=s).astype(np.float32) for s in ((H, W, 3), (H, W), (H, W), (H, W))]
    loss = (rgb * T_(go[0])).sum() + (dimg[..., 0] * T_(go[1])).sum() + (opac[..., 0] * T_(go[2])).sum() + (z2[..., 0] * T_(go[3])).sum()
    loss.backward()
    r = O.render_rgb_bwd(*a, sc["color"][m], al, g["start"], g["end"], g["ids"], final, go[0], cam.topleft, 1 / cam.fx, 1 / cam.fy, H, W)
    s1 = O.render_scalar_bwd(*a, dv, al, g["start"], g["end"], g["ids"], o_d, go[1], cam.topleft, 1 / cam.fx, 1 / cam.fy, H, W)
    s2 = O.render_scalar_bwd(*a, np.ones_like(dv), al, g["start"], g["end"], g["ids"], o_o, go[2], cam.topleft, 1 / cam.fx, 1 / cam.fy, H, W)
    s3 = O.render_scalar_bwd(*a, dv * dv, al, g["start"], g["end"], g["ids"], o_z, go[3], cam.topleft, 1 / cam.fx, 1 / cam.fy, H, W)
    want_mean = r[0] + s1[0] + s2[0] + s3[0]
    want_cov = r[1] + s1[1] + s2[1] + s3[1]
    want_alpha = r[3] + s1[3] + s2[3] + s3[3]
    want_depth = s1[2] + 2.0 * dv * s3[2]
    assert rel_err(mean2d.grad.cpu().numpy(), want_mean) < 1e-3
    assert rel_err(cov2d.grad.cpu().numpy(), want_cov) < 1e-3
    assert rel_err(alpha.grad.cpu().numpy(), want_alpha) < 1e-3
    assert rel_err(color.grad.cpu().numpy(), r[2]) < 1e-3
    assert rel_err(depth.grad.cpu().numpy().ravel(), want_depth) < 1e-3
    assert np.abs(bg.grad.cpu().numpy() - go[0] * o_T).max() <= 1e-4


def test_legacy_csr_entry_points():
    """tile_culling_aabb + the offset (CSR) forms of the RGB forward/backward give the same image
    and gradients as the start/end forms."""
    from gsgen_amd import _gs
    sc = scenes.random_scene(800, seed=3, svec=0.05)
    cam = scenes.Camera(96, 80, fx=90.0)
    g = scenes.oracle_geometry(sc, cam)
    m = g["mask"]
    nth, ntw = cam.tiles
    H, W = cam.h, cam.w
    tl, br, depth = T_(g["tl"]), T_(g["br"]), T_(g["depth"])
    ids = torch.zeros(g["D"], dtype=torch.int32, device=dev())
    offset = torch.zeros(nth * ntw + 1, dtype=torch.int32, device=dev())
    _gs.tile_culling_aabb(tl, br, ids, offset, depth, nth, ntw)
    assert np.array_equal(ids.cpu().numpy(), g["ids"])
    cnt = np.where(g["start"] >= 0, g["end"] - g["start"], 0)
    assert np.array_equal(offset.cpu().numpy(), np.concatenate([[0], np.cumsum(cnt)]))
    mean2d, cov2d, col, al = T_(g["mean2d"]), T_(g["cov2d"]), T_(sc["color"][m]), T_(sc["alpha"][m])
    out = torch.zeros(H * W * 3, device=dev())
    topleft = T_(cam.topleft)
    for fn in (_gs.tile_based_vol_rendering, _gs.tile_based_vol_rendering_v1, _gs.tile_based_vol_rendering_v2):
        out.zero_()
        fn(mean2d, cov2d, col, al, offset, ids, out, topleft, 16, nth, ntw, 1 / cam.fx, 1 / cam.fy, H, W, 1e-4)
        ref, _ = O.render_rgb_fwd(g["mean2d"], g["cov2d"], sc["color"][m], sc["alpha"][m], g["start"], g["end"], g["ids"],
                                  cam.topleft, 1 / cam.fx, 1 / cam.fy, H, W)
        assert np.abs(out.cpu().numpy().reshape(H, W, 3) - ref).max() <= 1e-4
    go = torch.randn(H, W, 3, device=dev())
    gm, gc = torch.zeros_like(mean2d), torch.zeros_like(cov2d)
    gcol, ga = torch.zeros_like(col), torch.zeros_like(al)
    _gs.tile_based_vol_rendering_backward(mean2d, cov2d, col, al, offset, ids, out, gm, gc, gcol, ga, go, topleft, 16, nth,
                                          ntw, 1 / cam.fx, 1 / cam.fy, H, W, 1e-4)
    r = O.render_rgb_bwd(g["mean2d"], g["cov2d"], sc["color"][m], sc["alpha"][m], g["start"], g["end"], g["ids"], ref,
                         go.cpu().numpy(), cam.topleft, 1 / cam.fx, 1 / cam.fy, H, W)
    assert rel_err(gm.cpu().numpy(), r[0]) < 1e-3 and rel_err(gcol.cpu().numpy(), r[2]) < 1e-3


def test_frame_is_hip_graph_capturable():
    """The whole frame (cull -> project -> bin/sort -> SH composite -> backward -> projection backward)
    is capture-safe: no allocation, no synchronisation, no host round trip.  A captured graph replays
    bit-identically to the eager launch sequence (forward) and within atomics noise (gradients)."""
    from gsgen_amd import renderer as R, _capi
    L = _capi.load()
    sc = scenes.random_scene(2000, seed=2, svec=0.04, C=2)
    cam = scenes.Camera(160, 112, fx=140.0, c2w=scenes.orbit(2.4, 15, 75))
    ci = R.CameraInfo(*cam.intr)
    N, C = 2000, 2
    t = {k: T_(sc[k]) for k in ("mean", "qvec", "svec", "alpha", "sh")}
    cam_dev = T_(ci.pack(cam.c2w)); rot = T_(np.ascontiguousarray(cam.c2w[:3, :3]).reshape(-1).copy())
    topleft = T_(cam.topleft); go = torch.randn(cam.h, cam.w, 3, device=dev())
    buf = R.FrameBuffers(N, cam.w, cam.h, dev())
    out = torch.zeros(cam.h, cam.w, 3, device=dev())
    gflat = torch.zeros(N * (7 + 3 * C * C), device=dev())
    g3 = [torch.zeros(N, k, device=dev()) for k in (3, 4, 3)]
    nth, ntw = buf.nth, buf.ntw
    p = lambda x: x.data_ptr()  # noqa: E731

    def frame():
        s = torch.cuda.current_stream().cuda_stream
        L.frame_geometry(N, p(t["mean"]), p(t["qvec"]), p(t["svec"]), p(cam_dev), cam.w, cam.h, buf.D_cap, p(buf.mean2d),
                         p(buf.cov2d), p(buf.depth), p(buf.mask), p(buf.ids), p(buf.start), p(buf.end), p(buf.total),
                         p(buf.ws), buf.ws.numel(), s)
        L.vol_render_sh_ordered(N, buf.D_cap, p(buf.mean2d), p(buf.cov2d), p(t["sh"]), p(t["alpha"]), p(buf.start),
                                p(buf.end), p(buf.ids), p(out), p(topleft), p(rot), 16, nth, ntw, 1 / ci.fx, 1 / ci.fy,
                                cam.h, cam.w, C, 1e-4, None, None, buf.tile_order(), s)
        gflat.zero_()
        L.vol_render_backward_sh_ordered(N, buf.D_cap, p(buf.mean2d), p(buf.cov2d), p(t["sh"]), p(t["alpha"]),
                                         p(buf.start), p(buf.end), p(buf.ids), p(out), p(gflat[:2 * N]),
                                         p(gflat[2 * N:6 * N]), p(gflat[7 * N:]), p(gflat[6 * N:7 * N]), p(go), p(topleft),
                                         p(rot), 16, nth, ntw, 1 / ci.fx, 1 / ci.fy, cam.h, cam.w, C, 1e-4, None,
                                         buf.tile_order(), s)
        L.project_gaussians_backward_masked(N, p(t["mean"]), p(t["qvec"]), p(t["svec"]), p(cam_dev), 1, p(buf.mask),
                                            p(gflat[:2 * N]), p(gflat[2 * N:6 * N]), None, p(g3[0]), p(g3[1]), p(g3[2]), s)

    out.zero_(); frame(); torch.cuda.synchronize()
    assert buf.ensure_capacity()
    ref_out, ref_g = out.clone(), [g.clone() for g in g3]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        out.zero_(); frame()  # warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        frame()
    out.zero_()
    for g in g3:
        g.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref_out)
    for a, b in zip(g3, ref_g):
        assert float((a - b).abs().max() / (b.abs().max() + 1e-30)) < 1e-4


def test_stale_backward_and_background_gradient():
    """(1) a backward whose forward state was overwritten by a later render raises instead of returning
    another frame's gradients; (3) a trainable background gets nan_to_num(grad * T) through the fused
    paths (gs/renderer.py:1283), reduced to its shape."""
    from gsgen_amd import renderer as R
    from gsgen_amd.batch import BatchRenderer
    sc = scenes.random_scene(3000, seed=4, svec=0.03, C=2)
    N = sc["mean"].shape[0]
    W, H = 96, 64
    cams = [scenes.Camera(W, H, fx=90.0, c2w=scenes.orbit(2.4, 10 + 5 * i, 60.0 * i)) for i in range(3)]
    cis = [R.CameraInfo(*c.intr) for c in cams]
    P = {k: T_(sc[k]).requires_grad_(True) for k in ("mean", "qvec", "svec", "alpha", "sh")}
    # (1) FrameBuffers
    buf = R.FrameBuffers(N, W, H, dev())
    rgb1, _ = R.render_frame(P["mean"], P["qvec"], P["svec"], P["alpha"], P["sh"], cis[0], cams[0].c2w, buf, C=2)
    rgb2, _ = R.render_frame(P["mean"], P["qvec"], P["svec"], P["alpha"], P["sh"], cis[1], cams[1].c2w, buf, C=2)
    with pytest.raises(RuntimeError, match="later render"):
        rgb1.sum().backward()
    rgb2.sum().backward()  # the latest frame is fine
    # (1) BatchRenderer
    br = BatchRenderer(N, W, H, dev(), max_batch=3)
    a, _ = br.render(P["mean"], P["qvec"], P["svec"], P["alpha"], P["sh"], cis, [c.c2w for c in cams], C=2)
    b, _ = br.render(P["mean"], P["qvec"], P["svec"], P["alpha"], P["sh"], cis, [c.c2w for c in cams], C=2)
    with pytest.raises(RuntimeError, match="between this batch's forward and its backward"):
        a.sum().backward()
    b.sum().backward()
    # (2) pair-list overflow: tests/test_gpu_overflow.py
    # (3) background gradients: SH batch with an rgb triple, post-activation colours with a full background image, heads
    go = torch.randn(3, H, W, 3, device=dev())
    bg3 = torch.tensor([0.2, 0.4, 0.6], device=dev(), requires_grad=True)
    out, T = br.render(P["mean"], P["qvec"], P["svec"], P["alpha"], P["sh"], cis, [c.c2w for c in cams], C=2, bg_rgb=bg3)
    (out * go).sum().backward()
    assert torch.allclose(bg3.grad, (go * T).sum((0, 1, 2)), rtol=1e-5, atol=1e-5)
    col = torch.sigmoid(T_(sc["sh"][:, :, 0])).requires_grad_(True)
    bgi = torch.rand(3, H, W, 3, device=dev(), requires_grad=True)
    out, T = br.render(P["mean"], P["qvec"], P["svec"], P["alpha"], col, cis, [c.c2w for c in cams], C=0, bg_rgb=bgi)
    (out * go).sum().backward()
    assert torch.allclose(bgi.grad, go * T, rtol=1e-6, atol=1e-6)
    bg3.grad = None
    rgb, dpt, opa, z2, T = br.render_heads(P["mean"], P["qvec"], P["svec"], P["alpha"], col, cis, [c.c2w for c in cams], bg_rgb=bg3)
    ((rgb * go).sum() + dpt.sum()).backward()
    assert torch.allclose(bg3.grad, (go * T).sum((0, 1, 2)), rtol=1e-5, atol=1e-5)
    bg1 = torch.tensor([0.1, 0.5, 0.9], device=dev(), requires_grad=True)
    f = R.FrameBuffers(N, W, H, dev())
    out, T = R.render_frame(P["mean"], P["qvec"], P["svec"], P["alpha"], P["sh"], cis[0], cams[0].c2w, f, C=2, bg_rgb=bg1)
    (out * go[0]).sum().backward()
    assert torch.allclose(bg1.grad, (go[0] * T).sum((0, 1)), rtol=1e-5, atol=1e-5)


def test_upload_small_does_not_wait_for_the_stream_and_is_capturable():
    """gsgen_upload_small: per-render constants travel as kernel arguments -- the call returns while the stream is
    still busy (a pageable `.to(device)` would wait), the host buffer may be reused at once, any multiple of 4 bytes
    arrives intact, and the launch can be captured into a hipGraph (a memcpy from pageable memory cannot)."""
    import time
    from gsgen_amd import _capi
    lib = _capi.load()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(11)
    s = torch.cuda.current_stream(dev).cuda_stream
    for words in (2, 68, 8 * 68, 896, 897, 2500):
        src = rng.standard_normal(words).astype(np.float32)
        keep = src.copy()
        dst = torch.zeros(words + 4, device=dev)
        lib.upload_small(dst.data_ptr(), src.ctypes.data, words * 4, s)
        src[:] = -1.0
        assert np.array_equal(dst[:words].cpu().numpy(), keep) and not dst[words:].any().item()
    # the stream is kept busy for tens of milliseconds; the upload must be enqueued behind it without waiting
    a = torch.randn(8192, 8192, device=dev)
    torch.cuda.synchronize()
    t_busy = time.perf_counter()
    for _ in range(20):
        a = a @ a
        a = a / a.abs().max()
    src = rng.standard_normal(8 * 68).astype(np.float32)
    dst = torch.zeros(8 * 68, device=dev)
    t0 = time.perf_counter()
    lib.upload_small(dst.data_ptr(), src.ctypes.data, src.nbytes, s)
    t_call = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_total = time.perf_counter() - t_busy
    assert np.array_equal(dst.cpu().numpy(), src)
    assert t_total > 5e-3, "the stream was not busy: the test proves nothing"
    assert t_call < 0.2 * t_total and t_call < 2e-3, f"upload waited for the stream: {t_call * 1e3:.2f} ms of {t_total * 1e3:.1f}"
    # graph capture
    side = torch.cuda.Stream(device=dev)
    g = torch.cuda.CUDAGraph()
    dst2 = torch.zeros(68, device=dev)
    src2 = np.arange(68, dtype=np.float32)
    with torch.cuda.graph(g, stream=side):
        lib.upload_small(dst2.data_ptr(), src2.ctypes.data, 272, torch.cuda.current_stream(dev).cuda_stream)
    src2[:] = 0          # the graph carries the bytes, not the address
    dst2.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(dst2.cpu().numpy(), np.arange(68, dtype=np.float32))


def test_rccl_process_group_of_one_runs_the_gather_and_its_backward():
    """The multi-GPU code path cannot meet a second GPU here; what CAN be checked on one is that RCCL initialises under
    this environment (HSA_ENABLE_IPC_MODE_LEGACY=0), that the differentiable image all_gather of gsgen_amd.dist runs on
    the device through it and that its backward hands the rank its own slice.  In a subprocess: a process group is
    process-wide state."""
    import subprocess, sys, os, socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    code = f"""
import os, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="{port}")
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from gsgen_amd import dist as D
x = torch.randn(3, 16, 24, 3, device=dev, requires_grad=True)
y = D._AllGatherImages.apply(x, None)
assert y.shape == x.shape and torch.equal(y.detach(), x.detach())
w = torch.randn_like(y)
(y * w).sum().backward()
assert torch.equal(x.grad, w)
t = torch.ones(5, device=dev)
dist.all_reduce(t)
assert torch.equal(t, torch.ones(5, device=dev))
dist.barrier()
dist.destroy_process_group()
print("ok")
"""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_batch_renderer_step_is_hip_graph_capturable():
    """A whole autograd step through the public batched path -- activations, BatchRenderer.render_heads forward and
    backward (camera blocks through kernel arguments, one enqueue per stage), densify statistics -- captured into ONE
    hipGraph ("one hipGraph per (B, H, W) bucket") and replayed: images bit-identical to the eager step,
    gradients within atomics noise.  (Overflow reporting under capture and replay: tests/test_gpu_overflow.py.)"""
    from gsgen_amd import renderer as R
    from gsgen_amd.batch import BatchRenderer
    sc = scenes.random_scene(3000, seed=31, svec=0.04)
    N, W, H, B = sc["mean"].shape[0], 112, 80, 2
    cams = [scenes.Camera(W, H, fx=100.0 + 20 * i, c2w=scenes.orbit(2.4, 10 + 15 * i, 60.0 + 140 * i)) for i in range(B)]
    cis, c2ws = [R.CameraInfo(*c.intr) for c in cams], [c.c2w for c in cams]
    keys = ("mean", "qvec", "svec", "alpha", "color")
    P_ = {k: T_(sc[k]).requires_grad_(True) for k in keys}
    gos = [torch.randn(B, H, W, c, device=dev()) for c in (3, 1, 1, 1)]
    br = BatchRenderer(N, W, H, dev(), max_batch=B)
    stats = R.DensifyStats(N, dev())

    def step():
        for v in P_.values():
            v.grad = None
        outs = br.render_heads(P_["mean"], P_["qvec"], P_["svec"] * 1.0, torch.clamp(P_["alpha"], 0, 1), P_["color"], cis, c2ws,
                               stats=stats)[:4]
        sum((o * g).sum() for o, g in zip(outs, gos)).backward()
        return [o.detach() for o in outs], [P_[k].grad for k in keys]

    outs_e, grads_e = step()
    torch.cuda.synchronize()
    assert br.ensure_capacity(B)
    outs_e, grads_e = step()
    outs_e, grads_e = [o.clone() for o in outs_e], [g.clone() for g in grads_e]
    cnt_e = stats.cnt.clone()
    torch.cuda.synchronize()
    assert br.ensure_capacity(B)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()  # warm-up on the capture stream
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            outs_g, grads_g = step()
        for o in outs_g:
            o.zero_()
        graph.replay()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for a, b in zip(outs_g, outs_e):
        assert torch.equal(a, b)
    for a, b, k in zip(grads_g, grads_e, keys):
        assert float((a - b).abs().max() / (b.abs().max() + 1e-30)) < 1e-4, k
    assert torch.equal(stats.cnt, cnt_e + 2 * (cnt_e / 2))  # two more visits per visible Gaussian per step: warm-up + replay
    # ... and the renderer is still usable eagerly afterwards
    outs_a, _ = step()
    torch.cuda.synchronize()
    assert torch.equal(outs_a[0], outs_e[0])


def test_pair_count_beyond_32_bits_reads_as_overflow_not_as_a_small_number():
    """A diverged scene -- 300 k Gaussians each covering every tile of a 2048^2 image: 4.9e9 (tile, Gaussian) pairs -- must
    come back as "does not fit" (the saturated count 2^32 - 1, every list empty), never as the count modulo 2^32, which
    would fit the buffer and send the emit pass past its end.  And a capacity beyond int32 is refused: list positions are
    int32 in the reference's layout."""
    from gsgen_amd import renderer as R, _capi
    L = _capi.load()
    N, W, H = 300_000, 2048, 2048
    rng = np.random.default_rng(0)
    mean = (rng.normal(size=(N, 3)) * 0.01).astype(np.float32)
    qvec = np.tile(np.array([1, 0, 0, 0], np.float32), (N, 1))
    svec = np.full((N, 3), 50.0, np.float32)  # far larger than the view: every tile
    cam = scenes.Camera(W, H, fx=float(W), c2w=scenes.orbit(2.5, 10, 20))
    ci = R.CameraInfo(*cam.intr)
    buf = R.FrameBuffers(N, W, H, dev(), D_cap=1 << 20)
    R.frame_geometry(T_(mean), T_(qvec), T_(svec), T_(ci.pack(cam.c2w)), buf)
    torch.cuda.synchronize()
    assert N * buf.nth * buf.ntw > 2 ** 32
    assert int(buf.mask.sum().item()) == N
    assert int(buf.total.cpu().numpy().view(np.uint32)[0]) == 0xFFFFFFFF  # (2^32 - 1 on the host side whatever dtype holds it)
    assert int((buf.start >= 0).sum().item()) == 0 and int((buf.end >= 0).sum().item()) == 0
    with pytest.raises(RuntimeError, match="diverged"):  # the host side does not try to "regrow" for it
        buf.ensure_capacity()
    p = lambda t: t.data_ptr()  # noqa: E731
    ws = torch.empty(1 << 20, dtype=torch.uint8, device=dev())
    with pytest.raises(Exception, match="invalid argument"):
        L.tile_culling_aabb_start_end(4, 0x80000000, 2, 2, p(buf.ids), p(buf.ids), p(buf.depth), p(buf.ids), p(buf.start), p(buf.end),
                                      p(ws), ws.numel(), torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("kind", ["heads", "sh", "rgb"])
def test_cpp_autograd_node_equals_the_python_functions(kind):
    """BatchRenderer's fast path -- the camera batch as ONE C++ autograd node (csrc/torch_batch.cpp -> gsgen_amd/ext/_gsbatch) --
    against its Python autograd Functions (use_ext = False): the same launches, so images bit for bit, gradients to the order
    of the atomics, densify statistics, background gradient; a second backward through a retained graph; the stale-backward
    error; and a backward that runs after the renderer itself has been dropped (the node keeps what it needs alive)."""
    import gc
    from gsgen_amd import renderer as R
    from gsgen_amd import batch as Bm
    assert Bm._batch_ext() is not None, "gsgen_amd/ext/_gsbatch.*.so is not built (python -m gsgen_amd.build --ext)"
    C = {"heads": 0, "sh": 4, "rgb": 0}[kind]
    sc = scenes.random_scene(4000, seed=8, svec=0.03, C=max(C, 1))
    if C == 4:
        sc["sh"][:, :, 1:] *= 0.3
    N, W, H, B = sc["mean"].shape[0], 144, 96, 3
    cams = [scenes.Camera(W, H, fx=230.0 + 20 * i, c2w=scenes.orbit(2.4, 5 + 12 * i, 50.0 + 100 * i)) for i in range(B)]
    cis, c2ws = [R.CameraInfo(*c.intr) for c in cams], [c.c2w for c in cams]
    ck = "sh" if C > 0 else "color"
    keys = ("mean", "qvec", "svec", "alpha", ck)
    gen = torch.Generator(device=dev()).manual_seed(3)
    gos = [torch.randn(B, H, W, c, device=dev(), generator=gen) for c in ((3, 1, 1, 1) if kind == "heads" else (3,))]

    def run(use_ext):
        P_ = {k: T_(sc[k]).requires_grad_(True) for k in keys}
        bg = torch.tensor([0.2, 0.5, 0.7], device=dev(), requires_grad=True)
        br = Bm.BatchRenderer(N, W, H, dev(), max_batch=B)
        br.use_ext = use_ext
        stats = R.DensifyStats(N, dev())
        outs = None
        for _ in range(2):  # (the first batch sizes the lists synchronously: Python path either way; the second is the one compared)
            for v in list(P_.values()) + [bg]:
                v.grad = None
            stats = R.DensifyStats(N, dev())
            if kind == "heads":
                outs = br.render_heads(P_["mean"], P_["qvec"], P_["svec"], P_["alpha"], P_[ck], cis, c2ws, bg_rgb=bg, stats=stats)
            else:
                outs = br.render(P_["mean"], P_["qvec"], P_["svec"], P_["alpha"], P_[ck], cis, c2ws, C=C, bg_rgb=bg, stats=stats)
            if _ > 0:  # which autograd node rendered this batch
                name = outs[0].grad_fn.name()
                assert (("HeadsFn" if kind == "heads" else "RenderFn") in name) == use_ext, name
            loss = sum((o * g_).sum() for o, g_ in zip(outs, gos))
            loss.backward(retain_graph=True)
        g1 = {k: P_[k].grad.clone() for k in keys}
        for v in list(P_.values()) + [bg]:
            v.grad = None
        loss.backward()  # a second backward through the retained graph: fresh accumulators, the same gradients
        for k in keys:
            assert rel_err(P_[k].grad.cpu().numpy(), g1[k].cpu().numpy()) < 1e-4, k
        return br, P_, bg, stats, [o.detach().clone() for o in outs], g1

    br_e, Pe, bge, se, oe, ge = run(True)
    br_p, Pp, bgp, sp, op, gp = run(False)
    for a, b in zip(oe, op):
        assert torch.equal(a, b)
    for k in keys:
        assert rel_err(ge[k].cpu().numpy(), gp[k].cpu().numpy()) < 1e-4, k
    assert torch.allclose(bge.grad, bgp.grad, rtol=1e-5, atol=1e-5)
    assert torch.equal(se.max_radii2d, sp.max_radii2d) and torch.equal(se.cnt, sp.cnt)
    assert rel_err(se.grad_accum.cpu().numpy(), sp.grad_accum.cpu().numpy()) < 1e-5
    # stale backward: a later render through the same renderer invalidates the pending one (the C++ node raises the same error)
    f = (lambda: br_e.render_heads(Pe["mean"], Pe["qvec"], Pe["svec"], Pe["alpha"], Pe[ck], cis, c2ws)[0]) if kind == "heads" else \
        (lambda: br_e.render(Pe["mean"], Pe["qvec"], Pe["svec"], Pe["alpha"], Pe[ck], cis, c2ws, C=C)[0])
    a, b = f(), f()
    with pytest.raises(RuntimeError, match="between this batch's forward and its backward"):
        a.sum().backward()
    b.sum().backward()
    # the renderer dropped before the backward: the node holds the tables and buffers it needs
    for v in Pe.values():
        v.grad = None
    f().sum().backward()
    want = {k: Pe[k].grad.clone() for k in keys}
    for v in Pe.values():
        v.grad = None
    c = f()
    del br_e, f, a, b
    gc.collect()
    junk = [torch.randn(1 << 20, device=dev()) for _ in range(8)]  # (whatever was freed would be handed out again here)
    c.sum().backward()
    torch.cuda.synchronize()
    for k in keys:
        assert rel_err(Pe[k].grad.cpu().numpy(), want[k].cpu().numpy()) < 1e-4, k
    del junk


def test_raw_parameters_depth_variance_and_background_inside_the_launches():
    """Round 6: BatchRenderer.render_heads(..., activations=(...), z_var=True, bg_rgb=...) -- the model's three parameter activations as
    one launch each way inside the batch's autograd node (gsgen_activate_fields), z_var = depth2 - depth^2 and rgb + T bg formed by the
    forward's epilogue, their chain rules by the backward's prologue -- against the same render with the activations and z_var in torch
    (activated leaves, torch's z_var arithmetic and autograd): images within 2e-6, every raw gradient and the background's within 2e-4
    of its tensor's largest entry."""
    from gsgen_amd import renderer as R
    from gsgen_amd import batch as Bm
    sc = scenes.random_scene(5000, seed=12, svec=0.03, C=1)
    N, W, H, B = sc["mean"].shape[0], 160, 112, 3
    cams = [scenes.Camera(W, H, fx=250.0 + 20 * i, c2w=scenes.orbit(2.4, 5 + 12 * i, 50.0 + 100 * i)) for i in range(B)]
    cis, c2ws = [R.CameraInfo(*c.intr) for c in cams], [c.c2w for c in cams]
    raw0 = {"mean": sc["mean"], "qvec": sc["qvec"], "svec": np.log(sc["svec"]),
            "alpha": np.log(np.clip(sc["alpha"], 1e-3, 1 - 1e-3) / (1 - np.clip(sc["alpha"], 1e-3, 1 - 1e-3))),
            "color": np.log(np.clip(sc["color"], 1e-3, 1 - 1e-3) / (1 - np.clip(sc["color"], 1e-3, 1 - 1e-3)))}
    gen = torch.Generator(device=dev()).manual_seed(5)
    gos = [torch.randn(B, H, W, c, device=dev(), generator=gen) for c in (3, 1, 1, 1)]
    keys = ("mean", "qvec", "svec", "alpha", "color")

    def run(inside, use_ext=True):
        P_ = {k: T_(raw0[k].astype(np.float32)).requires_grad_(True) for k in keys}
        bg = torch.tensor([0.2, 0.5, 0.7], device=dev(), requires_grad=True)
        br = Bm.BatchRenderer(N, W, H, dev(), max_batch=B)
        br.use_ext = use_ext
        for _ in range(2):
            for v in list(P_.values()) + [bg]:
                v.grad = None
            if inside:
                rgb, dep, opa, zv, _T = br.render_heads(P_["mean"], P_["qvec"], P_["svec"], P_["alpha"], P_["color"], cis, c2ws, bg_rgb=bg,
                                                        z_var=True, activations=("exp", "sigmoid", "sigmoid"))
            else:
                # (the background stays an argument: T is not a differentiable output -- composited outside, the T bg term's
                # dependence on the splats would be lost; its in-launch form against torch's arithmetic: tests/test_cpu_host.py)
                rgb, dep, opa, z2, T = br.render_heads(P_["mean"], P_["qvec"], torch.exp(P_["svec"]), torch.sigmoid(P_["alpha"]),
                                                       torch.sigmoid(P_["color"]), cis, c2ws, bg_rgb=bg)
                zv = z2 - dep * dep
            outs = (rgb, dep, opa, zv)
            sum((o * g_).sum() for o, g_ in zip(outs, gos)).backward()
        torch.cuda.synchronize()
        return [o.detach().cpu().numpy() for o in outs], {k: P_[k].grad.cpu().numpy() for k in keys}, bg.grad.cpu().numpy()

    o_in, g_in, b_in = run(True)
    o_py, g_py, b_py = run(True, use_ext=False)   # the Python Functions: torch's activations in front of the same launches
    o_t, g_t, b_t = run(False)
    for a, b, c in zip(o_in, o_t, o_py):
        sc_ = max(1.0, float(np.abs(b).max()))
        assert np.abs(a - b).max() <= 2e-6 * sc_ and np.abs(c - b).max() <= 2e-6 * sc_
    for k in keys:
        assert rel_err(g_in[k], g_t[k]) <= 2e-4 and rel_err(g_py[k], g_t[k]) <= 2e-4, k
    assert rel_err(b_in, b_t) <= 2e-4 and rel_err(b_py, b_t) <= 2e-4


@pytest.mark.gpu
@pytest.mark.parametrize("use_ext", [True, False])
def test_captured_step_replays_other_poses_and_intrinsics(use_ext):
    """gsgen_amd.graph.CapturedStep (opt-in hipGraph): a whole step -- render_heads with raw parameters, a loss, backward,
    FusedAdam -- captured once with cameras A on a device_cameras renderer, replayed for cameras B (other poses AND other focal
    lengths: the reference draws both per step, data/__init__.py:194) and again for A: images, gradients and parameters follow the eager trajectory on the same cameras (mean difference <= 2e-6, a few
    threshold pixels apart), one capture for all replays."""
    from gsgen_amd import renderer as R
    from gsgen_amd import batch as Bm
    from gsgen_amd.graph import CapturedStep
    from gsgen_amd.optim import FusedAdam
    sc = scenes.random_scene(6000, seed=31, svec=0.1, C=1)
    N, W, H, B = sc["mean"].shape[0], 160, 112, 2
    mk = lambda fx, el, az: scenes.Camera(W, H, fx=fx, c2w=scenes.orbit(2.4, el, az))  # noqa: E731
    sets = {"A": [mk(250.0, 5, 50), mk(300.0, 20, 170)], "B": [mk(190.0, 35, -60), mk(340.0, -10, 260)], "C": [mk(275.0, 50, 10), mk(225.0, 0, 95)]}
    cam = {k: ([R.CameraInfo(*c.intr) for c in v], np.stack([c.c2w for c in v])) for k, v in sets.items()}
    logit = lambda x: np.log(np.clip(x, 1e-3, 1 - 1e-3) / (1 - np.clip(x, 1e-3, 1 - 1e-3)))  # noqa: E731
    raw0 = {"mean": sc["mean"], "qvec": sc["qvec"], "svec": np.log(sc["svec"]), "alpha": logit(sc["alpha"]), "color": logit(sc["color"])}
    keys = ("mean", "qvec", "svec", "alpha", "color")
    gen = torch.Generator(device=dev()).manual_seed(9)
    gos = [torch.randn(B, H, W, c, device=dev(), generator=gen) * 1e-3 for c in (3, 1, 1, 1)]
    bg = torch.tensor([0.3, 0.4, 0.5], device=dev())
    order = ["A", "B", "A", "C", "B"]

    # pair lists sized for the three camera sets with 35 % to spare (an explicit D_cap: the default minimum of 65 536 pairs would
    # hold anything this scene can produce) -- the near cameras further down do not fit
    near = [scenes.Camera(W, H, fx=520.0, c2w=scenes.orbit(1.15, 10, 30)), scenes.Camera(W, H, fx=480.0, c2w=scenes.orbit(1.2, 25, 200))]
    need = {k: max(scenes.oracle_geometry(sc, c)["D"] for c in v) for k, v in {**sets, "near": near}.items()}
    D_cap = int(1.35 * max(need[k] for k in sets))
    assert need["near"] > D_cap, need

    def make(device_cameras):
        opt = FusedAdam({k: T_(raw0[k].astype(np.float32)) for k in keys}, {k: 1e-3 for k in keys}, eps=1e-15, capturable=device_cameras)
        br = Bm.BatchRenderer(N, W, H, dev(), max_batch=B, device_cameras=device_cameras, D_cap=D_cap)
        br.use_ext = use_ext
        P_ = opt.params

        def step(cis, c2ws):
            opt.zero_grad()
            outs = br.render_heads(P_["mean"], P_["qvec"], P_["svec"], P_["alpha"], P_["color"], cis, c2ws, bg_rgb=bg, z_var=True,
                                   activations=("exp", "sigmoid", "sigmoid"))[:4]
            sum((o * g_).sum() for o, g_ in zip(outs, gos)).backward()
            grads = [P_[k].grad for k in keys]
            opt.step()
            return outs, grads
        return opt, br, step

    def snap(outs, grads, opt):
        torch.cuda.synchronize()
        return ([o.detach().cpu().numpy().copy() for o in outs], [g.detach().cpu().numpy().copy() for g in grads],
                {k: opt.params[k].detach().cpu().numpy().copy() for k in keys})

    # eager reference trajectory on a default renderer: the warm-up steps CapturedStep takes on A (2 + 1 on the capture stream; the
    # capture itself records, it does not execute), then `order`
    opt_e, br_e, step_e = make(False)
    for _ in range(3):
        step_e(*cam["A"])
    eager = [snap(*step_e(*cam[k]), opt_e) for k in order]
    opt_g, br_g, step_g = make(True)
    cs = CapturedStep(br_g, step_g, *cam["A"], optimizers=[opt_g])
    got = []
    for k in order:
        outs, grads = cs(*cam[k])
        got.append(snap(outs, grads, opt_g))
    assert cs.captures == 1 and cs.replays == len(order) and opt_g.step_count == opt_e.step_count == 3 + len(order)
    for k, (eo, eg, ep), (go_, gg, gp) in zip(order, eager, got):
        # (two trajectories: the order of the gradients' atomics differs from run to run, Adam turns a sign flip of a near-zero
        # gradient into a step of its own, and a splat at an alpha / transmittance threshold flips a pixel -- hence mean and
        # outlier-fraction bounds, not a maximum; a replay that rendered the wrong cameras is off by tenths everywhere, below)
        for a, b in zip(eo, go_):
            assert np.isfinite(b).all() and np.abs(b).max() > 0
            d = np.abs(a - b) / max(1.0, float(np.abs(a).max()))
            assert d.mean() <= 2e-6 and (d > 2e-4).mean() <= 5e-4, (k, float(d.mean()), float((d > 2e-4).mean()))
        for name, a, b in zip(keys, eg, gg):
            assert rel_err(b, a) <= 2e-3, (k, name)
        for name in keys:
            d = np.abs(ep[name] - gp[name])
            assert d.mean() <= 2e-6 and d.max() <= 1e-2, (k, name)
    # the two focal lengths of a set really differ from the captured ones: a replay that ignored the uploaded intrinsics would not match
    assert np.abs(eager[0][0][0] - eager[1][0][0]).mean() > 0.02 and np.abs(eager[0][0][0] - eager[3][0][0]).mean() > 0.02
    with pytest.raises(ValueError, match="device_cameras"):
        CapturedStep(br_e, step_e, *cam["A"])
    # cameras that need far more (tile, Gaussian) pairs than the lists hold: the replay renders them as NaN and the report words say so
    # (PairListOverflow, lists regrown -- exactly the eager behaviour); the next call runs its step eagerly and records the graph again;
    # the one after replays the new graph.  One optimiser step per call throughout.
    import gsgen_amd
    near = ([R.CameraInfo(*c.intr) for c in near], np.stack([c.c2w for c in near]))
    cap0, calls = br_g.slots[0].D_cap, 0
    try:
        cs(*near); calls += 1
        torch.cuda.synchronize()
        cs(*near); calls += 1   # (the report of the first replay is certainly in by now)
        overflowed = False
    except gsgen_amd.PairListOverflow:
        calls += 1
        overflowed = True
    assert overflowed or br_g.slots[0].D_cap > cap0, "the near cameras were meant to outgrow the lists"
    outs, _ = cs(*near); calls += 1
    assert cs.captures == 2
    outs2, _ = cs(*near); calls += 1
    torch.cuda.synchronize()
    assert cs.captures == 2 and opt_g.step_count == 3 + len(order) + calls
    for o in list(outs) + list(outs2):
        assert torch.isfinite(o).all()
    assert float((outs2[2] > 0.5).float().mean()) > 0.2  # (opacity: the near views are mostly covered)


@pytest.mark.gpu
def test_device_cameras_renderer_renders_what_the_default_one_does():
    """BatchRenderer(device_cameras=True) outside any capture: camera rows and pixel sizes reach the kernels through device memory
    instead of kernel arguments -- SH degree 3 (pixel sizes still in the view tables), post-activation RGB and RGB + heads batches give the
    default renderer's images bit for bit and its gradients to atomics' order, also when a second batch with other intrinsics follows
    on the same renderer (the device block is rewritten, nothing stale is read)."""
    from gsgen_amd import renderer as R
    from gsgen_amd import batch as Bm
    sc = scenes.random_scene(4000, seed=77, svec=0.04, C=4)
    N, W, H, B = sc["mean"].shape[0], 144, 96, 3
    batches = [[scenes.Camera(W, H, fx=f, c2w=scenes.orbit(r_, el, az)) for f, r_, el, az in row] for row in (
        ((200.0, 2.4, 10, 30), (260.0, 2.2, 35, 150), (180.0, 2.6, -5, 260)), ((320.0, 2.0, 50, -40), (150.0, 2.8, 0, 90), (240.0, 2.3, 20, 200)))]
    col = np.ascontiguousarray(1 / (1 + np.exp(-sc["sh"][:, :, 0])), np.float32)  # some post-activation colour
    gen = torch.Generator(device=dev()).manual_seed(3)
    go3, go1 = torch.randn(B, H, W, 3, device=dev(), generator=gen), torch.randn(B, H, W, 1, device=dev(), generator=gen)

    def run(device_cameras):
        res = []
        br = Bm.BatchRenderer(N, W, H, dev(), max_batch=B, device_cameras=device_cameras)
        for cams in batches:
            cis, c2ws = [R.CameraInfo(*c.intr) for c in cams], [c.c2w for c in cams]
            P_ = {k: T_(sc[k]).requires_grad_(True) for k in ("mean", "qvec", "svec", "alpha", "sh")}
            c_ = T_(col).requires_grad_(True)
            for kind in ("sh", "rgb", "heads"):
                for q in list(P_.values()) + [c_]:
                    q.grad = None
                if kind == "sh":
                    outs = br.render(P_["mean"], P_["qvec"], P_["svec"], P_["alpha"], P_["sh"], cis, c2ws, C=4)[:1]
                    gos = [go3]
                elif kind == "rgb":
                    outs = br.render(P_["mean"], P_["qvec"], P_["svec"], P_["alpha"], c_, cis, c2ws, C=0)[:1]
                    gos = [go3]
                else:
                    outs = br.render_heads(P_["mean"], P_["qvec"], P_["svec"], P_["alpha"], c_, cis, c2ws)[:4]
                    gos = [go3, go1, go1, go1]
                torch.autograd.backward(list(outs), gos)
                torch.cuda.synchronize()
                res.append(([o.detach().cpu().numpy() for o in outs], P_["mean"].grad.cpu().numpy().copy(), P_["svec"].grad.cpu().numpy().copy()))
        return res

    want, got = run(False), run(True)
    for (wo, wm, ws), (go_, gm, gs) in zip(want, got):
        for a, b in zip(wo, go_):
            assert np.array_equal(a, b) and np.abs(a).max() > 0
        assert rel_err(gm, wm) <= 1e-4 and rel_err(gs, ws) <= 1e-4
