"""The reference's OWN autograd classes (gs/renderer.py, imported from /root/reference -- never copied) running on
this repo's `_gs` mirror, checked against the golden vectors the reference itself produced.

`gs/renderer.py:20-24` does `import _gs as _backend`; here `_backend` is gsgen_amd._gs, exactly what a user of the
reference gets after gsgen_amd.install_as_gs() / shim/_gs.py.  In this GPU-less container the mirror is bound to
the SIMT-emulator build of the same kernels (oracle/_build/libgsgen_emu.so) and fed host tensors
(gsgen_amd._gs._bind_library_for_tests -- a hook nothing in the product uses).  Forward AND backward of
_render_with_T (gs/renderer.py:1135-1283), _render_scalar (:999-1132), _render_sh (:674-830), _render_sh_bg
(:833-996) and _render_start_end (:541-672), plus the reference's PyTorch projection chained in front of
_render_sh so that gradients flow to mean / qvec / svec through the reference's whole Python graph.

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
    _gs._bind_library_for_tests(emu, host_tensors=True)
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
    _gs._bind_library_for_tests(None)


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
