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
pytestmark = pytest.mark.skipif(refshim.root() is None, reason="neither /root/reference nor tests/_refpy.zip is present")


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


import refpy_cases as RC
from refpy_cases import rel  # noqa: F401,E402  (re-exported: the model test below uses it)


@pytest.mark.parametrize("name", ["mock2", "rand_c1", "rand_c3", "rand_c4", "long_lists"])
def test_reference_render_with_T_and_start_end(ref, name):
    RC.case_render_with_T_and_start_end(ref, "cpu", name)


@pytest.mark.parametrize("name", ["mock2", "rand_c3"])
def test_reference_render_scalar(ref, name):
    RC.case_render_scalar(ref, "cpu", name)


@pytest.mark.parametrize("name,with_bg", [(n_, b_) for n_ in ("mock2", "rand_c1", "rand_c3", "rand_c4") for b_ in (False, True)] +
                         [("long_lists", True)])  # (the emulator walks 1 500-entry lists slowly: one variant here, both on the GPU)
def test_reference_render_sh(ref, name, with_bg):
    RC.case_render_sh(ref, "cpu", name, with_bg)


def test_reference_projection_chained_into_reference_render_sh(ref):
    """project_gaussians (the reference's PyTorch, gs/renderer.py:391-421) -> _render_sh on the mirror: one autograd
    graph, all of it the reference's Python; gradients reach mean / qvec / svec"""
    RC.case_projection_chained_into_reference_render_sh(ref, "cpu")


# ---------------------------------------------------------------------------------------------------------------
# The reference's MODEL class on the mirror: gs/gaussian_splatting.py imported from /root/reference, its own
# GaussianSplattingRenderer constructed from a config and run unmodified -- forward() over a camera batch (render_one
# per camera: culling_gaussian_bsphere -> project_gaussians -> tile_culling_aabb_count -> tile_culling_aabb_start_end ->
# render_with_T + three render_scalar passes, :1198-1466), backward of a loss on all four outputs, post_backward()
# (update_densify_info, :464-469).  Every `_backend.*` call in that file lands in this repo's `_gs`.
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def refmodel(ref):
    """-> (gs.gaussian_splatting module with _backend = the mirror, gs.renderer module)"""
    from gsgen_amd import _gs
    return RC.import_reference_model(_gs), ref


def test_reference_model_forward_backward_and_densify_info_on_the_mirror(refmodel):
    from oracle import oracle as O
    M, GR = refmodel
    run = RC.run_reference_model(M, "cpu")
    model, out, go, masks, g_mean2d, raw, sc, cams, bg = (run[k] for k in ("model", "out", "go", "masks", "g_mean2d", "raw", "sc", "cams", "bg"))
    N = sc["mean"].shape[0]
    t = lambda a, **k: torch.tensor(np.ascontiguousarray(a), **k)  # noqa: E731

    # ---- the expectation: the reference's torch projection (same bits as inside the model) in front of the oracle's
    # cull / count / bin / sort / four compositing passes and their backward, chained to the raw parameters by autograd
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
    RC.check_model_against_fixture(run)
