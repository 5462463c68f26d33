"""The SH coefficient bounds live on the device and are part of every step (-m gpu).

The reference evaluates the SH basis per pixel unconditionally (vol_render_sh.h:48-65).  This library's fast form -- a
tile-local polynomial fit of that basis -- is only legal while a scene-dependent bound on the coefficients holds, so the
bounds are MEASURED by every forward on the device (gsgen_sh_l1_bound_rows: one number per splat) and the kernels route on
them PER TILE (round 4; per view on the global maximum in round 3); no caller supplies them, nothing syncs with the host.
These tests move the coefficients between steps, mix narrow and wide cameras in one batch, plant outlier splats, fuzz the
routed launches against the exact ones, and pin the per-camera `_gs` SH names to the same path."""
import os
import sys

import numpy as np
import pytest
import torch

import scenes
from oracle import oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
KEYS = ("mean", "qvec", "svec", "alpha", "sh")


def dev():
    return torch.device("cuda:0")


def T_(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def _oracle_image(sc, sh, cam, bg):
    g = scenes.oracle_geometry(sc, cam)
    m = g["mask"]
    rot = cam.c2w[:3, :3].reshape(-1)
    ref = O.render_sh_fwd(g["mean2d"], g["cov2d"], sh[m], sc["alpha"][m], g["start"], g["end"], g["ids"], cam.topleft, rot, 4,
                          1 / cam.fx, 1 / cam.fy, cam.h, cam.w, bg=bg)
    return g, ref


def test_coefficients_moving_between_steps_keep_the_image_contract():
    """Three optimiser-like steps through ONE BatchRenderer with nothing but the default arguments: small higher-order
    coefficients (the polynomial kernels take the views), then coefficients 60 x larger (their bound fails: the same call
    renders with the exact kernels, bit for bit), then small again.  Every image within 1e-4 of the oracle on every pixel;
    the caller never computed, passed or refreshed a bound."""
    from gsgen_amd import renderer as R, _capi
    from gsgen_amd.batch import BatchRenderer
    L = _capi.load()
    N, W, H, B = 20_000, 320, 240, 2
    sc = scenes.pointe_scene(N, seed=4, C=4)
    cams = [scenes.Camera(W, H, fx=430.0 + 40 * i, c2w=scenes.orbit(2.5, 10.0 + 25 * i, 20.0 + 80 * i)) for i in range(B)]
    cis = [R.CameraInfo(*c.intr) for c in cams]
    bg = np.array([0.1, 0.2, 0.3], np.float32)
    br = BatchRenderer(N, W, H, dev(), max_batch=B)
    P = {k: T_(sc[k]) for k in KEYS}
    big = sc["sh"].copy()
    big[:, :, 1:] *= 60.0
    seen = []
    for step, sh_np in enumerate((sc["sh"], big, sc["sh"])):
        sh = T_(sh_np).requires_grad_(True)
        S = R.sh_l1_bound(sh)  # for the assertions below only
        applies = [L.sh_poly_applies(S, 1.0 / c.fx, 4) for c in cams]
        assert all(applies) == (step != 1) and any(applies) == (step != 1), (step, S, applies)
        out = {}
        for basis in ("auto", "exact"):
            for _ in range(2):
                rgb, T = br.render(P["mean"], P["qvec"], P["svec"], P["alpha"], sh, cis, [c.c2w for c in cams], C=4, bg_rgb=T_(bg),
                                   sh_basis=basis)
                if br.ensure_capacity(B):
                    break
            g, = torch.autograd.grad((rgb * rgb).sum(), [sh])
            out[basis] = (rgb.detach().cpu().numpy(), T.cpu().numpy(), g.cpu().numpy())
        d = float(np.abs(out["auto"][0] - out["exact"][0]).max())
        if step == 1:
            assert d == 0.0                                          # routed to the exact form: the same pixels, bit for bit
            assert rel_err(out["auto"][2], out["exact"][2]) <= 2e-6  # (gradient sums differ by atomics order only)
        else:
            assert 0.0 < d <= 1e-5, (step, d)                                    # the polynomial kernels, fit error only
            assert rel_err(out["auto"][2], out["exact"][2]) <= 1e-4
        assert np.array_equal(out["auto"][1], out["exact"][1])
        for i, cam in enumerate(cams):
            g_, ref = _oracle_image(sc, sh_np, cam, bg)
            m = g_["mask"]
            scenes.assert_sh_image_parity(out["auto"][0][i], ref, g_["mean2d"], g_["cov2d"], sc["alpha"][m], g_["start"], g_["end"],
                                          g_["ids"], cam.topleft, 1 / cam.fx, 1 / cam.fy, what=f"step {step} camera {i}")
        seen.append(d)
    assert seen[0] == seen[2]  # the third step is the first one again: no state survived the detour


def test_one_batch_narrow_and_wide_cameras_are_routed_per_view():
    """a 0.7 x focal camera (cfg4's widest) and a 1.35 x one in the same launch, the coefficients scaled so that the bound
    separates them: the narrow view comes from the polynomial kernel (differs from the exact render by the fit error), the
    wide view from the exact kernel (bit-identical) -- and the fused render_frame path agrees per camera"""
    from gsgen_amd import renderer as R, _capi
    from gsgen_amd.batch import BatchRenderer
    L = _capi.load()
    N, W, H = 30_000, 512, 512
    sc = scenes.pointe_scene(N, seed=2, C=4)
    cams = [scenes.Camera(W, H, fx=0.7 * W, c2w=scenes.orbit(2.2, 20.0, 10.0)), scenes.Camera(W, H, fx=1.35 * W, c2w=scenes.orbit(2.4, 35.0, 140.0))]
    cis = [R.CameraInfo(*c.intr) for c in cams]
    sh = sc["sh"].copy()
    rows0 = np.abs(sh[:, :, 1:]).sum(-1).max(-1)   # per splat: the largest of its three channels' sums
    sh[:, :, 1:] *= 2.5 / float(rows0.min())       # EVERY splat beyond the wide camera's limit (2.2) ...
    rows = np.abs(sh[:, :, 1:]).sum(-1).max(-1)
    assert rows.min() > 2.4 and rows.max() < 15.0  # ... and inside the narrow one's (15.8)
    P = {k: T_(sc[k]) for k in KEYS}
    P["sh"] = T_(sh)
    S = R.sh_l1_bound(P["sh"])
    assert not L.sh_poly_applies(S, 1 / cams[0].fx, 4) and L.sh_poly_applies(S, 1 / cams[1].fx, 4)
    br = BatchRenderer(N, W, H, dev(), max_batch=2)
    img = {}
    for basis in ("auto", "exact"):
        for _ in range(2):
            rgb, _ = br.render(P["mean"], P["qvec"], P["svec"], P["alpha"], P["sh"], cis, [c.c2w for c in cams], C=4, sh_basis=basis)
            if br.ensure_capacity(2):
                break
        img[basis] = rgb.cpu().numpy()
    assert np.array_equal(img["auto"][0], img["exact"][0]) and np.abs(img["exact"][0]).max() > 0.1
    d = float(np.abs(img["auto"][1] - img["exact"][1]).max())
    assert 0.0 < d <= 1e-5, d
    br.render(P["mean"], P["qvec"], P["svec"], P["alpha"], P["sh"], cis, [c.c2w for c in cams], C=4)
    fl = br.routing_flags(2).cpu().numpy()
    nonempty = np.stack([(br.slots[i].end > br.slots[i].start).cpu().numpy() for i in range(2)])
    assert (fl[0].astype(bool) == nonempty[0]).all() and not fl[1].any()  # wide view: every tile exact; narrow view: none
    for i, cam in enumerate(cams):  # one camera at a time through render_frame: the same routing, the same pixels
        buf = R.FrameBuffers(N, W, H, dev())
        for _ in range(2):
            one, _ = R.render_frame(P["mean"], P["qvec"], P["svec"], P["alpha"], P["sh"], cis[i], cam.c2w, buf, C=4)
            if buf.ensure_capacity():
                break
        assert np.array_equal(one.cpu().numpy(), img["auto"][i]), i


def test_outlier_splats_cost_their_entries_not_the_view():
    """Round 4, per-tile routing with a per-entry exact tier: 0.2 % of the splats carry higher-band coefficients 60 x larger than
    the rest.  Round 3's per-view rule sends both views to the exact kernels; now those splats alone are evaluated exactly,
    entry by entry, inside the polynomial kernel -- a tile only goes to the exact kernel when more than a quarter of a staged
    batch is such splats (here: almost none) --, the images stay within 1e-4 of the oracle on every pixel and the gradients
    within 1e-4 of the exact launch's.  Tiles the polynomial forward did flag are bit-identical to the exact launch's."""
    from gsgen_amd import renderer as R, _capi
    from gsgen_amd.batch import BatchRenderer
    L = _capi.load()
    N, W, H, B = 20_000, 320, 240, 2
    sc = scenes.pointe_scene(N, seed=4, C=4)
    rng = np.random.default_rng(5)
    outl = rng.choice(N, 40, replace=False)
    sh_np = sc["sh"].copy()
    sh_np[outl, :, 1:] *= 60.0
    cams = [scenes.Camera(W, H, fx=400.0 + 40 * i, c2w=scenes.orbit(2.5, 10.0 + 25 * i, 20.0 + 80 * i)) for i in range(B)]
    cis = [R.CameraInfo(*c.intr) for c in cams]
    bg = np.array([0.1, 0.2, 0.3], np.float32)
    P = {k: T_(sc[k]) for k in KEYS}
    sh = T_(sh_np).requires_grad_(True)
    assert not any(L.sh_poly_applies(R.sh_l1_bound(sh), 1.0 / c.fx, 4) for c in cams)   # the per-view rule: all exact
    br = BatchRenderer(N, W, H, dev(), max_batch=B)
    out = {}
    for basis in ("auto", "exact"):
        for _ in range(2):
            rgb, T = br.render(P["mean"], P["qvec"], P["svec"], P["alpha"], sh, cis, [c.c2w for c in cams], C=4, bg_rgb=T_(bg), sh_basis=basis)
            if br.ensure_capacity(B):
                break
        if basis == "auto":
            flags = br.routing_flags(B).cpu().numpy().astype(bool)
            nonempty = np.stack([(br.slots[i].end > br.slots[i].start).cpu().numpy() for i in range(B)])
        g, = torch.autograd.grad((rgb * rgb).sum(), [sh])
        out[basis] = (rgb.detach().cpu().numpy(), T.cpu().numpy(), g.cpu().numpy())
    frac = flags.sum() / max(1, nonempty.sum())
    scenes.PARITY_LOG.append(f"per-tile routing, 0.2 % outlier splats: {int(flags.sum())} of {int(nonempty.sum())} non-empty tiles exact = 0")
    assert frac < 0.02, frac
    assert np.array_equal(out["auto"][1], out["exact"][1])
    ntw = (W + 15) // 16
    for i in range(B):
        d = np.abs(out["auto"][0][i] - out["exact"][0][i]).max(-1)
        assert 0.0 < d.max() <= 1e-5
        for t in np.nonzero(flags[i])[0]:
            ty, tx = divmod(int(t), ntw)
            assert d[16 * ty:16 * ty + 16, 16 * tx:16 * tx + 16].max() == 0.0, (i, t)   # rendered by the exact kernel
        g_, ref = _oracle_image(sc, sh_np, cams[i], bg)
        m = g_["mask"]
        scenes.assert_sh_image_parity(out["auto"][0][i], ref, g_["mean2d"], g_["cov2d"], sc["alpha"][m], g_["start"], g_["end"],
                                      g_["ids"], cams[i].topleft, 1 / cams[i].fx, 1 / cams[i].fy, what=f"outlier splats, camera {i}")
    assert rel_err(out["auto"][2], out["exact"][2]) <= 1e-4


def test_exact_fallbacks_are_enqueued_only_while_crowded_tiles_are_reported():
    """Round 6: the polynomial forward reports tiles crowded with splats beyond the bound into a host-visible word
    (gsgen_sh_view::route_report); BatchRenderer reads it without a sync and, after three clean reports in a row, stops enqueueing the
    two persistent exact fallback launches (no_fallback) -- on a clean scene the images do not change by a bit.  A cluster of splats
    with large higher bands is reported, brings the fallbacks back within a batch or two, and -- rendered in either mode -- stays
    within 1e-5 of the exact kernels' image."""
    from gsgen_amd import renderer as R
    from gsgen_amd.batch import BatchRenderer
    N, W, H, B = 20_000, 320, 240, 2
    sc = scenes.pointe_scene(N, seed=4, C=4)
    cams = [scenes.Camera(W, H, fx=400.0 + 40 * i, c2w=scenes.orbit(2.5, 10.0 + 25 * i, 20.0 + 80 * i)) for i in range(B)]
    cis, c2ws = [R.CameraInfo(*c.intr) for c in cams], [c.c2w for c in cams]
    P = {k: T_(sc[k]) for k in KEYS}
    br = BatchRenderer(N, W, H, dev(), max_batch=B)
    assert br._route.ptr(0) is not None
    imgs, modes = [], []
    for step in range(6):
        sh = P["sh"].clone().requires_grad_(True)
        rgb, _ = br.render(P["mean"], P["qvec"], P["svec"], P["alpha"], sh, cis, c2ws, C=4)
        modes.append(br._route_args[1])
        (rgb * rgb).sum().backward()
        torch.cuda.synchronize()
        assert bool(torch.isfinite(sh.grad).all())
        imgs.append(rgb.detach().cpu().numpy())
    assert modes[0] == 0 and modes[-1] == 1, modes          # the first batches carry the fallbacks, a clean scene drops them
    for im in imgs[1:]:
        assert np.array_equal(im, imgs[0])
    # a crowded cluster appears: 300 neighbouring splats with higher bands 60 x larger
    centre = sc["mean"][17]
    near = np.argsort(np.linalg.norm(sc["mean"] - centre, axis=1))[:300]
    sh_np = sc["sh"].copy()
    sh_np[near, :, 1:] *= 60.0
    sh_d = T_(sh_np)
    exact = br.render(P["mean"], P["qvec"], P["svec"], P["alpha"], sh_d, cis, c2ws, C=4, sh_basis="exact")[0].cpu().numpy()
    seen_modes = []
    for step in range(5):
        rgb = br.render(P["mean"], P["qvec"], P["svec"], P["alpha"], sh_d, cis, c2ws, C=4)[0]
        seen_modes.append(br._route_args[1])
        torch.cuda.synchronize()
        assert np.abs(rgb.cpu().numpy() - exact).max() <= 1e-5, (step, seen_modes)
    assert seen_modes[0] == 1 and seen_modes[-1] == 0, seen_modes   # reported by the first such batch: the fallbacks are back
    assert bool(br.routing_flags(B).any())


def test_routed_launches_fuzz_against_the_exact_kernels():
    """hypothesis on the GPU over the launches BatchRenderer runs by default at SH degree 3: 1 .. 4 cameras of ragged shapes and
    focal lengths on both sides of the bound, 1 .. 4000 splats of any size, coefficient magnitudes over two decades, opaque
    scenes, 1 or 4 backward segments -- against the exact kernels of the same call: transmittance identical, images within
    2e-5, every gradient within 1e-4 of its largest entry; views in which not even the smallest splat passes the per-splat bound
    must come back bit-identical (every tile routed to the exact kernel)."""
    from hypothesis import given, settings, strategies as st, HealthCheck
    from gsgen_amd import renderer as R, _capi
    from gsgen_amd.batch import BatchRenderer
    L = _capi.load()
    n_ex = int(os.environ.get("GSGEN_FUZZ_EXAMPLES", "25"))

    @settings(max_examples=n_ex, deadline=None, derandomize=(n_ex == 25), suppress_health_check=list(HealthCheck))
    @given(B=st.integers(1, 4), W=st.integers(1, 150), H=st.integers(1, 120), n=st.integers(1, 4000), seed=st.integers(0, 10_000),
           svec=st.sampled_from([0.003, 0.02, 0.08]), opaque=st.booleans(), nseg=st.sampled_from([1, 4]),
           gain=st.sampled_from([0.2, 1.0, 8.0, 40.0]), fscale=st.sampled_from([0.4, 1.0, 2.5]), dc=st.sampled_from([1.0, 1.0, 150.0]))
    def run(B, W, H, n, seed, svec, opaque, nseg, gain, fscale, dc):
        sc = scenes.random_scene(n, seed=seed, svec=svec, spread=0.25, C=4)
        sc["sh"][:, :, 1:] *= gain
        sc["sh"][:, :, 0] *= dc  # 150: saturated colours (|sh . Y| in the hundreds)
        if opaque:
            sc["alpha"][:] = 0.999
        cams = [scenes.Camera(W, H, fx=fscale * (180.0 + 70 * i), c2w=scenes.orbit(2.5 + 0.1 * i, 12.0 * i, 50.0 + 95.0 * i)) for i in range(B)]
        cis = [R.CameraInfo(*c.intr) for c in cams]
        P = {k: T_(sc[k]).requires_grad_(True) for k in KEYS}
        rows = np.abs(sc["sh"][:, :, 1:]).sum(-1).max(-1)
        # per view: can NO splat take the polynomial form (then every tile must come back from the exact kernel, bit for bit)?
        none_ok = [not any(L.sh_poly_applies(float(r), max(1 / c.fx, 1 / c.fy), 4) for r in np.unique(rows)[:1]) for c in cams]
        br = BatchRenderer(n, W, H, dev(), max_batch=B, segments=nseg)
        go = torch.randn(B, H, W, 3, device=dev(), generator=torch.Generator(device=dev()).manual_seed(seed))
        res = {}
        for basis in ("exact", "auto"):
            for _ in range(2):
                rgb, T = br.render(P["mean"], P["qvec"], P["svec"], P["alpha"], P["sh"], cis, [c.c2w for c in cams], C=4, sh_basis=basis)
                if br.ensure_capacity(B):
                    break
            grads = torch.autograd.grad([rgb], [P[k] for k in KEYS], [go])
            res[basis] = (rgb.detach().cpu().numpy(), T.cpu().numpy(), [g.cpu().numpy() for g in grads])
        tag = (B, W, H, n, seed, svec, opaque, nseg, gain, fscale, dc)
        assert np.array_equal(res["auto"][1], res["exact"][1]), tag
        for i in range(B):
            d = float(np.abs(res["auto"][0][i] - res["exact"][0][i]).max())
            assert d <= 2e-5 and (not none_ok[i] or d == 0.0), (tag, i, d, none_ok)
        for k, a, e in zip(KEYS, res["auto"][2], res["exact"][2]):
            assert np.abs(a - e).max() <= 1e-4 * np.abs(e).max() + 1e-6, (tag, k)  # (floor: round 6's 500-example hunt -- saturated colours, dc = 150: every sh gradient ~1e-6, the two bases 1.05e-7 apart)
    run()


def test_gs_sh_names_take_the_routed_kernels():
    """the reference's own call shape (tile_based_vol_rendering_sh / _backward_sh and their _with_bg forms,
    gs/src/render.h:83-127; caller gs/sh_renderer.py:315-357) through the compiled `_gs` module and the ctypes mirror: the
    binding measures the bound in front of each call, a narrow camera is rendered by the polynomial kernels (within 1e-5 of
    the exact entry points, every pixel within 1e-4 of the oracle), a wide one by the exact kernel bit for bit"""
    import gsgen_amd
    from gsgen_amd import _gs as mirror, renderer as R, _capi
    L = _capi.load()
    compiled = gsgen_amd.compiled_gs()
    N, W, H, C = 20_000, 256, 192, 4
    sc = scenes.pointe_scene(N, seed=6, C=C)
    for fx, expect_poly in ((500.0, True), (60.0, False)):
        cam = scenes.Camera(W, H, fx=fx, c2w=scenes.orbit(2.5, 15.0, 60.0))
        g = scenes.oracle_geometry(sc, cam)
        m = g["mask"]
        nth, ntw = cam.tiles
        t = {k: T_(v) for k, v in dict(mean=g["mean2d"], cov=g["cov2d"], sh=sc["sh"][m], alpha=sc["alpha"][m], start=g["start"],
                                        end=g["end"], ids=g["ids"], topleft=cam.topleft, rot=np.ascontiguousarray(cam.c2w[:3, :3]).reshape(-1)).items()}
        assert L.sh_poly_applies(R.sh_l1_bound(t["sh"]), 1 / fx, 4) == expect_poly
        bg = T_(np.array([0.2, 0.3, 0.1], np.float32))
        go = torch.randn(H, W, 3, device=dev(), generator=torch.Generator(device=dev()).manual_seed(1))
        n_vis, D = int(m.sum()), g["D"]
        # the exact entry points of the C ABI (no bound) as the yardstick
        ex_out = torch.zeros(H, W, 3, device=dev())
        L.vol_render_sh(n_vis, D, *(t[k].data_ptr() for k in ("mean", "cov", "sh", "alpha", "start", "end", "ids")), ex_out.data_ptr(),
                        t["topleft"].data_ptr(), t["rot"].data_ptr(), 16, nth, ntw, 1 / cam.fx, 1 / cam.fy, H, W, C, 1e-4,
                        bg.data_ptr(), None, None)
        ex_g = [torch.zeros(n_vis, 2, device=dev()), torch.zeros(n_vis, 2, 2, device=dev()), torch.zeros(n_vis, 3, 16, device=dev()),
                torch.zeros(n_vis, device=dev())]
        L.vol_render_backward_sh(n_vis, D, *(t[k].data_ptr() for k in ("mean", "cov", "sh", "alpha", "start", "end", "ids")),
                                 ex_out.data_ptr(), *(x.data_ptr() for x in ex_g), go.data_ptr(), t["topleft"].data_ptr(),
                                 t["rot"].data_ptr(), 16, nth, ntw, 1 / cam.fx, 1 / cam.fy, H, W, C, 1e-4, bg.data_ptr(), None)
        ref = O.render_sh_fwd(g["mean2d"], g["cov2d"], sc["sh"][m], sc["alpha"][m], g["start"], g["end"], g["ids"], cam.topleft,
                              cam.c2w[:3, :3].reshape(-1), C, 1 / cam.fx, 1 / cam.fy, H, W, bg=bg.cpu().numpy())
        for mod in [x for x in (compiled, mirror) if x is not None]:
            out = torch.zeros(H, W, 3, device=dev())
            mod.tile_based_vol_rendering_sh_with_bg(t["mean"], t["cov"], t["sh"], t["alpha"], t["start"], t["end"], t["ids"], out,
                                                    t["topleft"], t["rot"], 16, nth, ntw, 1 / cam.fx, 1 / cam.fy, H, W, C, 1e-4, bg)
            gr = [torch.zeros_like(x) for x in ex_g]
            mod.tile_based_vol_rendering_backward_sh_with_bg(t["mean"], t["cov"], t["sh"], t["alpha"], t["start"], t["end"], t["ids"],
                                                             out, gr[0], gr[1], gr[2], gr[3], go, t["topleft"], t["rot"], 16, nth,
                                                             ntw, 1 / cam.fx, 1 / cam.fy, H, W, C, 1e-4, bg)
            d = float((out - ex_out).abs().max())
            if expect_poly:
                assert 0.0 < d <= 1e-5, (mod.__name__, d)
            else:
                assert d == 0.0, (mod.__name__, d)
            for a, e in zip(gr, ex_g):
                assert rel_err(a.cpu().numpy(), e.cpu().numpy()) <= 1e-4
            scenes.assert_sh_image_parity(out.cpu().numpy(), ref, g["mean2d"], g["cov2d"], sc["alpha"][m], g["start"], g["end"],
                                          g["ids"], cam.topleft, 1 / cam.fx, 1 / cam.fy, what=f"_gs SH names, fx {fx}")


def test_a_callers_own_bound_tensor_is_verified_on_request():
    """BatchRenderer.render(sh_l1_bound=<device tensor>, verify_bound=True): a tensor that holds the true bound passes, a stale
    one (the coefficients grew since) raises instead of silently breaking the 1e-4 contract"""
    from gsgen_amd import renderer as R
    from gsgen_amd.batch import BatchRenderer
    N, W, H = 3000, 96, 64
    sc = scenes.pointe_scene(N, seed=8, C=4)
    cam = scenes.Camera(W, H, fx=300.0)
    P = {k: T_(sc[k]) for k in KEYS}
    br = BatchRenderer(N, W, H, dev(), max_batch=1)
    ci = [R.CameraInfo(*cam.intr)]
    bound = R.sh_l1_bound_device(P["sh"])
    a, _ = br.render(P["mean"], P["qvec"], P["svec"], P["alpha"], P["sh"], ci, [cam.c2w], C=4, sh_l1_bound=bound, verify_bound=True)
    b, _ = br.render(P["mean"], P["qvec"], P["svec"], P["alpha"], P["sh"], ci, [cam.c2w], C=4)
    assert torch.equal(a, b)
    grown = P["sh"] * 1.5
    with pytest.raises(RuntimeError, match="stale or wrong"):
        br.render(P["mean"], P["qvec"], P["svec"], P["alpha"], grown, ci, [cam.c2w], C=4, sh_l1_bound=bound, verify_bound=True)
    with pytest.raises(ValueError, match="1-float CUDA tensor"):
        br.render(P["mean"], P["qvec"], P["svec"], P["alpha"], P["sh"], ci, [cam.c2w], C=4, sh_l1_bound=3.0)
