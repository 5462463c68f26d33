"""The reference's OWN autograd classes (gs/renderer.py, imported from /root/reference -- never copied) running on
this repo's `_gs` mirror, checked against the golden vectors the reference itself produced.

`gs/renderer.py:20-24` does `import _gs as _backend`; here `_backend` is gsgen_amd._gs, exactly what a user of the
reference gets after gsgen_amd.install_as_gs() / shim/_gs.py.  In this GPU-less container the mirror is bound to
the SIMT-emulator build of the same kernels (oracle/_build/libgsgen_emu.so) and fed host tensors
(the test rebinds gsgen_amd._gs._load to that build and lets it accept host tensors; the product binds the HIP library only).  Forward AND backward of
_render_with_T (gs/renderer.py:1135-1283), _render_scalar (:999-1132), _render_sh (:674-830), _render_sh_bg
(:833-996) and _render_start_end (:541-672), plus the reference's PyTorch projection chained in front of
_render_sh so that gradients flow to mean / qvec / svec through the reference's whole Python graph.

The last test runs the reference's MODEL class (gs/gaussian_splatting.py GaussianSplattingRenderer: forward over a camera
batch, backward, post_backward) unmodified on the mirror.

Skipped where /root/reference does not exist (the GPU box); tests/test_gpu_golden.py holds the HIP library to
the same golden vectors there."""
import os
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

import refshim

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLD = os.path.join(ROOT, "tests", "golden")
pytestmark = pytest.mark.skipif(not refshim.available(), reason="/root/reference is not present")


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.fixture(scope="module")
def ref():
    """gs.renderer of the reference with `_backend` = this repo's mirror on the emulator"""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "emu"])
    from gsgen_amd import _capi, _gs
    emu = _capi.Lib(os.path.join(ROOT, "oracle", "_build", "libgsgen_emu.so"))
    saved = (_gs._load, _gs._ACCEPT_HOST_TENSORS)
    _gs._load, _gs._ACCEPT_HOST_TENSORS = (lambda: emu), True  # the emulator build of the same kernels, host tensors
    refshim.install()
    sys.modules["_gs"] = _gs  # what `import _gs as _backend` finds (gs/renderer.py:20-24)
    import gs.renderer as GR
    GR._backend = _gs  # ... also when another test imported gs.renderer behind refshim's empty stand-in first
    # the SH classes bracket their kernels with cudaProfilerStart/Stop (gs/renderer.py:698,720,...): no CUDA here
    stub = types.SimpleNamespace(cudaProfilerStart=lambda: 0, cudaProfilerStop=lambda: 0)
    orig = torch.cuda.profiler.cudart
    torch.cuda.profiler.cudart = lambda: stub
    yield GR
    torch.cuda.profiler.cudart = orig
    _gs._load, _gs._ACCEPT_HOST_TENSORS = saved


def load(name):
    g = dict(np.load(os.path.join(GOLD, name + ".npz")))
    m = g["mask"].astype(bool)
    fx, fy, cx, cy, w, h = g["cam_intr"][:6]
    t = lambda a, **k: torch.tensor(np.ascontiguousarray(a), **k)  # noqa: E731
    c = {"g": g, "m": m, "H": int(h), "W": int(w), "fx": float(fx), "fy": float(fy),
         "topleft": t(np.array([-cx / fx, -cy / fy], np.float32)),
         "start": t(g["start"]), "end": t(g["end"]), "ids": t(g["ids"]),
         "nth": (int(h) + 15) // 16, "ntw": (int(w) + 15) // 16, "t": t}
    return c


def geo_args(c):
    return (c["start"], c["end"], c["ids"], c["topleft"], 16, c["nth"], c["ntw"], 1 / c["fx"], 1 / c["fy"], c["H"],
            c["W"], 1e-4)


@pytest.mark.parametrize("name", ["mock2", "rand_c1", "rand_c3", "rand_c4"])
def test_reference_render_with_T_and_start_end(ref, name):
    c = load(name); g, m, t = c["g"], c["m"], c["t"]
    P = [t(g["mean2d"], requires_grad=True), t(g["cov2d"], requires_grad=True),
         t(g["in_color"][m], requires_grad=True), t(g["in_alpha"][m], requires_grad=True)]
    bg = t(g["bg_img"], requires_grad=True)
    out = ref._render_with_T.apply(*P, *geo_args(c), bg)
    want = g["rgb"] + g["T"] * g["bg_img"]
    assert out.shape == (c["H"], c["W"], 3)
    assert np.abs(out.detach().numpy() - want).max() <= 1e-4
    (out * t(g["grad_out"])).sum().backward()
    for a, k in zip(P, ("rgb_gmean", "rgb_gcov", "rgb_gcol", "rgb_galpha")):
        assert rel(a.grad.numpy(), g[k]) < 1e-3, k
    assert np.abs(bg.grad.numpy() - g["grad_out"] * g["T"]).max() <= 1e-5  # gs/renderer.py:1283
    # render_start_end: flat image, no background (gs/renderer.py:541-672)
    Q = [t(g["mean2d"], requires_grad=True), t(g["cov2d"], requires_grad=True),
         t(g["in_color"][m], requires_grad=True), t(g["in_alpha"][m], requires_grad=True)]
    flat = ref.render_start_end(*Q, *geo_args(c))
    assert flat.shape == (c["H"] * c["W"] * 3,)
    assert np.abs(flat.detach().numpy().reshape(c["H"], c["W"], 3) - g["rgb"]).max() <= 1e-4


@pytest.mark.parametrize("name", ["mock2", "rand_c3"])
def test_reference_render_scalar(ref, name):
    c = load(name); g, m, t = c["g"], c["m"], c["t"]
    P = [t(g["mean2d"], requires_grad=True), t(g["cov2d"], requires_grad=True),
         t(g["depth"], requires_grad=True), t(g["in_alpha"][m], requires_grad=True)]  # depth as [N,1], as render_one passes it
    T = torch.ones(c["H"], c["W"], 1)
    out = ref.render_scalar(*P, *geo_args(c), T)
    assert out.shape == (c["H"] * c["W"],)
    assert np.abs(out.detach().numpy().reshape(c["H"], c["W"]) - g["depth_img"]).max() <= 1e-4 * max(1.0, np.abs(g["depth_img"]).max())
    assert np.abs(T.numpy() - g["depth_T"]).max() <= 1e-5  # the caller's T is overwritten in place
    (out * t(np.ascontiguousarray(g["grad_out"][..., 0])).reshape(-1)).sum().backward()
    for a, k in zip(P, ("sc_gmean", "sc_gcov", "sc_gscalar", "sc_galpha")):
        assert rel(a.grad.numpy().reshape(g[k].shape), g[k]) < 1e-3, k


@pytest.mark.parametrize("name", ["mock2", "rand_c1", "rand_c3", "rand_c4"])
@pytest.mark.parametrize("with_bg", [False, True])
def test_reference_render_sh(ref, name, with_bg):
    c = load(name); g, m, t = c["g"], c["m"], c["t"]
    C = int(g["C"])
    tag = "shbg" if with_bg else "sh"
    P = [t(g["mean2d"], requires_grad=True), t(g["cov2d"], requires_grad=True),
         t(g["in_sh"][m], requires_grad=True), t(g["in_alpha"][m], requires_grad=True)]
    c2w = t(g["c2w"][:3, :3])  # contiguous [3,3]: the kernels read 9 packed floats (vol_render_sh.h:48-55)
    a = (*P, c["start"], c["end"], c["ids"], c["topleft"], c2w, 16, c["nth"], c["ntw"], 1 / c["fx"], 1 / c["fy"],
         c["H"], c["W"], C, 1e-4)
    out = ref.render_sh_bg(*a, t(g["bg_rgb"])) if with_bg else ref.render_sh(*a)
    assert out.shape == (c["H"] * c["W"] * 3,)
    assert np.abs(out.detach().numpy().reshape(c["H"], c["W"], 3) - g[tag + "_img"]).max() <= 1e-4
    (out.reshape(c["H"], c["W"], 3) * t(g["grad_out"])).sum().backward()
    for p_, k in zip(P, ("_gmean", "_gcov", "_gsh", "_galpha")):
        assert rel(p_.grad.numpy(), g[tag + k]) < 1e-3, tag + k


def test_reference_projection_chained_into_reference_render_sh(ref):
    """project_gaussians (the reference's PyTorch, gs/renderer.py:391-421) -> _render_sh on the mirror: one autograd
    graph, all of it the reference's Python; gradients reach mean / qvec / svec"""
    c = load("rand_c4"); g, m, t = c["g"], c["m"], c["t"]
    C = int(g["C"])
    mean, qvec, svec = (t(g["in_" + k][m], requires_grad=True) for k in ("mean", "qvec", "svec"))
    c2w_full = t(g["c2w"])
    mean2d, cov2d, JW, depth = ref.project_gaussians(mean, qvec, svec, c2w_full, True)
    assert np.abs(mean2d.detach().numpy() - g["mean2d"]).max() <= 1e-6
    sh, al = t(g["in_sh"][m], requires_grad=True), t(g["in_alpha"][m], requires_grad=True)
    out = ref.render_sh(mean2d.contiguous(), cov2d.contiguous(), sh, al, c["start"], c["end"], c["ids"], c["topleft"],
                        t(g["c2w"][:3, :3]), 16, c["nth"], c["ntw"], 1 / c["fx"], 1 / c["fy"], c["H"], c["W"], C, 1e-4)
    (out.reshape(c["H"], c["W"], 3) * t(g["grad_out"])).sum().backward()
    assert rel(sh.grad.numpy(), g["sh_gsh"]) < 1e-3
    # the projection backward of the golden was fed the RGB path's 2-D gradients; rebuild the expectation for the SH
    # path with the reference's own autograd on the golden 2-D gradients
    m2, q2, s2 = (t(g["in_" + k][m], requires_grad=True) for k in ("mean", "qvec", "svec"))
    a2, b2, _, _ = ref.project_gaussians(m2, q2, s2, c2w_full, True)
    ((a2 * t(g["sh_gmean"])).sum() + (b2 * t(g["sh_gcov"])).sum()).backward()
    for got, want, k in ((mean, m2, "mean"), (qvec, q2, "qvec"), (svec, s2, "svec")):
        assert rel(got.grad.numpy(), want.grad.numpy()) < 2e-3, k


# ---------------------------------------------------------------------------------------------------------------
# The reference's MODEL class on the mirror: gs/gaussian_splatting.py imported from /root/reference, its own
# GaussianSplattingRenderer constructed from a config and run unmodified -- forward() over a camera batch (render_one
# per camera: culling_gaussian_bsphere -> project_gaussians -> tile_culling_aabb_count -> tile_culling_aabb_start_end ->
# render_with_T + three render_scalar passes, :1198-1466), backward of a loss on all four outputs, post_backward()
# (update_densify_info, :464-469).  Every `_backend.*` call in that file lands in this repo's `_gs`.
# ---------------------------------------------------------------------------------------------------------------
class _Cfg(dict):
    """what the reference reads its OmegaConf node through: attribute access, .get, hasattr"""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None


@pytest.fixture(scope="module")
def refmodel(ref):
    """-> (gs.gaussian_splatting module with _backend = the mirror, gs.renderer module)"""
    from gsgen_amd import _gs
    dm = types.ModuleType("kornia.geometry.depth")  # utils/ops.py:5 imports depth_to_3d (unused on this path)
    dm.depth_to_3d = None
    sys.modules["kornia.geometry.depth"] = dm
    sys.modules["kornia"].__path__ = []
    sys.modules["kornia.geometry"].__path__ = []
    import gs.gaussian_splatting as M
    M._backend = _gs
    return M, ref


def test_reference_model_forward_backward_and_densify_info_on_the_mirror(refmodel):
    import scenes
    from oracle import oracle as O
    from utils.camera import CameraInfo
    M, GR = refmodel
    sys.path.insert(0, GOLD)
    import make_golden_model as MG  # the scene, cameras and config the committed fixture was generated from
    bg = MG.BG
    sc, cams = MG.case()
    gold = np.load(os.path.join(GOLD, "model", "model_batch.npz"))
    t = lambda a, **k: torch.tensor(np.ascontiguousarray(a), **k)  # noqa: E731
    model = M.GaussianSplattingRenderer(_Cfg(MG.model_cfg()), {k: t(sc[k]) for k in ("mean", "qvec", "svec", "color", "alpha")})
    model.train()
    N = sc["mean"].shape[0]
    out = model({"c2w": torch.stack([t(c.c2w) for c in cams]), "camera_info": [CameraInfo(*c.intr) for c in cams]})
    assert {k: tuple(v.shape) for k, v in out.items()} == {"rgb": (2, 56, 72, 3), "depth": (2, 56, 72, 1),
                                                          "opacity": (2, 56, 72, 1), "z_var": (2, 56, 72, 1)}
    rng = np.random.default_rng(5)
    go = {k: rng.normal(size=tuple(v.shape)).astype(np.float32) for k, v in out.items()}
    sum((out[k] * t(go[k])).sum() for k in out).backward()
    masks = [m_.numpy().copy() for m_ in model.masks]
    g_mean2d = [m_.grad.numpy().copy() for m_ in model.mean_2ds]  # retained by render_one for update_densify_info
    model.post_backward()

    # ---- the expectation: the reference's torch projection (same bits as inside the model) in front of the oracle's
    # cull / count / bin / sort / four compositing passes and their backward, chained to the raw parameters by autograd
    raw = {"mean": model.mean, "qvec": model.qvec, "svec": model.svec_before_activation,
           "color": model.color_before_activation, "alpha": model.alpha_before_activation}
    P = {k: v.detach().clone().requires_grad_(True) for k, v in raw.items()}
    svec_a, color_a, alpha_a = torch.exp(P["svec"]), torch.sigmoid(P["color"]), torch.sigmoid(P["alpha"])
    want_maxr, want_acc, want_cnt = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros(N, np.float32)
    some_culled = False
    for b, cam in enumerate(cams):
        H, W = cam.h, cam.w
        normals, pts = O.frustum(cam.c2w, *cam.intr)
        mask = O.cull_bsphere(sc["mean"], svec_a.detach().numpy(), normals, pts, 6.0)
        assert np.array_equal(mask, masks[b])
        some_culled |= bool(0 < mask.sum() < N)
        mt = t(mask)
        mean2d, cov2d, _, depth = GR.project_gaussians(P["mean"][mt].contiguous(), P["qvec"][mt].contiguous(),
                                                       svec_a[mt].contiguous(), t(cam.c2w), True)
        col, al = color_a[mt].contiguous(), alpha_a[mt].contiguous()
        m2, c2, dv = mean2d.detach().numpy(), cov2d.detach().numpy(), depth.detach().numpy().ravel().copy()
        D, tl, br = O.aabb_count(m2, c2, 16, cam.fx, cam.fy, cam.cx, cam.cy, W, H, 6.0)
        ids, start, end = O.bin_sort(tl, br, dv, *cam.tiles, D)
        geo = (start, end, ids, cam.topleft, 1 / cam.fx, 1 / cam.fy, H, W)
        cn, an = col.detach().numpy(), al.detach().numpy()
        o_rgb, T = O.render_rgb_fwd(m2, c2, cn, an, *geo)
        o_d, _ = O.render_scalar_fwd(m2, c2, dv, an, *geo)
        o_o, _ = O.render_scalar_fwd(m2, c2, np.ones_like(dv), an, *geo)
        o_z, _ = O.render_scalar_fwd(m2, c2, dv * dv, an, *geo)
        img = o_rgb + T.reshape(H, W, 1) * np.array(bg, np.float32)
        zmax = max(1.0, float(np.abs(o_z).max()))
        assert np.abs(out["rgb"][b].detach().numpy() - img).max() <= 1e-4
        assert np.abs(out["depth"][b, ..., 0].detach().numpy() - o_d).max() <= 1e-4 * max(1.0, float(np.abs(o_d).max()))
        assert np.abs(out["opacity"][b, ..., 0].detach().numpy() - o_o).max() <= 1e-4
        assert np.abs(out["z_var"][b, ..., 0].detach().numpy() - (o_z - o_d * o_d)).max() <= 2e-4 * zmax
        # backward: z_var = z2 - depth^2 (gs/gaussian_splatting.py:1397); the RGB backward is handed the image INCLUDING
        # the background (gs/renderer.py:1182,1213: the suffix colour behind a splat contains T_final * bg)
        g_rgb = np.ascontiguousarray(go["rgb"][b])
        g_d = np.ascontiguousarray(go["depth"][b, ..., 0] - 2.0 * o_d * go["z_var"][b, ..., 0])
        g_o, g_z = np.ascontiguousarray(go["opacity"][b, ..., 0]), np.ascontiguousarray(go["z_var"][b, ..., 0])
        r = O.render_rgb_bwd(m2, c2, cn, an, start, end, ids, np.ascontiguousarray(img, np.float32), g_rgb, cam.topleft, 1 / cam.fx, 1 / cam.fy, H, W)
        ss = [O.render_scalar_bwd(m2, c2, v, an, start, end, ids, f, g_, cam.topleft, 1 / cam.fx, 1 / cam.fy, H, W)
              for v, f, g_ in ((dv, o_d, g_d), (np.ones_like(dv), o_o, g_o), (dv * dv, o_z, g_z))]
        gm2 = r[0] + sum(s[0] for s in ss)
        assert rel(g_mean2d[b], gm2) < 1e-3
        gc2 = (r[1] + sum(s[1] for s in ss)).reshape(-1, 2, 2)
        gal = r[3] + sum(s[3] for s in ss)
        gdv = (ss[0][2] + 2.0 * dv * ss[2][2]).reshape(-1, 1)  # depth is a scalar channel twice: z and z^2
        torch.autograd.backward([mean2d, cov2d, depth, col, al],
                                [t(gm2), t(gc2), t(gdv.astype(np.float32)), t(r[2]), t(gal)], retain_graph=True)
        # the densify statistics of this camera (:1240-1245, :464-469)
        mm = (c2[:, 0, 0] + c2[:, 1, 1]) / 2
        radii = mm + np.sqrt(np.clip(mm * mm - (c2[:, 0, 0] * c2[:, 1, 1] - c2[:, 0, 1] * c2[:, 1, 0]), 0, None))
        want_maxr[mask] = np.maximum(want_maxr[mask], radii)
        want_acc[mask] += np.linalg.norm(gm2, axis=-1)
        want_cnt[mask] += 1
    assert some_culled  # the cull does something on at least one camera
    for k in raw:
        assert rel(raw[k].grad.numpy(), P[k].grad.numpy()) < 1e-3, k
    assert np.abs(model.max_radii2d.numpy() - want_maxr).max() <= 1e-6 * max(1.0, float(want_maxr.max()))
    assert np.array_equal(model.cnt.numpy(), want_cnt)
    assert rel(model.mean_2d_grad_accum.numpy(), want_acc) < 1e-3
    # ... and against the fixture the reference's own kernels produced under the same model class
    # (tests/golden/make_golden_model.py): images, raw-parameter gradients, densify statistics
    assert np.array_equal(np.stack(masks), gold["masks"])
    for k in out:
        scale = max(1.0, float(np.abs(gold["out_" + k]).max()))
        assert np.abs(out[k].detach().numpy() - gold["out_" + k]).max() <= 1e-4 * scale, k
    for k in raw:
        assert rel(raw[k].grad.numpy(), gold["grad_" + k]) < 1e-3, k
    assert np.array_equal(model.cnt.numpy(), gold["cnt"])
    assert rel(model.max_radii2d.numpy(), gold["max_radii2d"]) < 1e-6
    assert rel(model.mean_2d_grad_accum.numpy(), gold["grad_accum"]) < 1e-3
