"""Finite-difference gradient checks ("finite-difference checks on tiny scenes"): the analytic backward of the
whole chain -- compositing backward (vol_render_sh.h:268-455, vol_render.h:318-418) chained into the projection backward
(autograd through gs/renderer.py:391-421) -- against central differences of an independent fp64 statement of the forward
(tests/fd_model.py), on scenes of <= 8 Gaussians and 32 x 32 pixels kept away from the 1/255 skip and the T < 1e-4 stop
discontinuities.  Per ENTRY of every parameter tensor: |analytic - fd| <= 1e-3 |fd| + 1e-5 max|fd|.

CPU: the oracle's backward (what the GPU kernels are held to everywhere else).  GPU (-m gpu): the HIP path itself, through
render_frame with the exact and with the routed (polynomial where its bound allows) SH basis, through BatchRenderer, and the
post-activation RGB path."""
import numpy as np
import pytest
import torch

import fd_model as FD
import scenes
from oracle import oracle as O

KEYS_SH = ("mean", "qvec", "svec", "alpha", "sh")
KEYS_RGB = ("mean", "qvec", "svec", "alpha", "color")


def _case(seed, C, n=8, fx=600.0):
    cam = scenes.Camera(32, 32, fx=fx, c2w=scenes.orbit(2.5, 20.0 + 7 * seed, 40.0 + 50 * seed))
    sc = FD.tiny_scene(n, seed, C, cam)
    g = scenes.oracle_geometry(sc, cam)
    assert g["mask"].all() and g["D"] >= n  # every Gaussian is visible and lands in a tile list
    go = np.random.default_rng(100 + seed).normal(size=(32, 32, 3))
    return cam, sc, g, go


def _fd(sc, cam, C, go, g, names, bg):
    P = {k: np.asarray(sc[k], np.float64) for k in names}
    loss, img, frozen = FD.forward(P, cam, C, go, bg, lists=(g["start"], g["end"], g["ids"]))
    # The decisions are frozen in the finite differences, as they are in the analytic backward; what the base point must
    # guarantee is that an fp32 evaluation takes the SAME decisions as this fp64 one: no (pixel, Gaussian) within 1e-5
    # (~170 fp32 ulps) of the skip threshold, no pixel anywhere near saturation.
    assert frozen.margins["skip"] > 1e-5 and frozen.margins["stop"] > 0.5, frozen.margins
    return P, img, FD.fd_gradients(P, cam, C, go, frozen, names, bg)


def _assert_entrywise(got, fd, what):
    for k, want in fd.items():
        a, b = np.asarray(got[k], np.float64).reshape(want.shape), want
        tol = 1e-3 * np.abs(b) + 1e-5 * np.abs(b).max()
        bad = np.abs(a - b) > tol
        assert not bad.any(), (what, k, int(bad.sum()), float((np.abs(a - b) / tol).max()),
                               a[bad][:4].tolist(), b[bad][:4].tolist())
        assert np.abs(b).max() > 0, (what, k)


@pytest.mark.parametrize("seed,C", [(0, 4), (1, 2), (2, 3), (3, 1), (4, 4)])
def test_oracle_sh_chain_against_finite_differences(seed, C):
    cam, sc, g, go = _case(seed, C)
    bg = np.array([0.3, 0.1, 0.2], np.float32)
    P, img, fd = _fd(sc, cam, C, go, g, KEYS_SH, bg)
    rot = cam.c2w[:3, :3].reshape(-1)
    ref = O.render_sh_fwd(g["mean2d"], g["cov2d"], sc["sh"], sc["alpha"], g["start"], g["end"], g["ids"], cam.topleft, rot, C,
                          1 / cam.fx, 1 / cam.fy, cam.h, cam.w, bg=bg)
    assert np.abs(ref - img).max() <= 1e-5  # the fp64 model states the same forward (the oracle rounds mean2d / cov2d to fp32)
    gm2, gc2, gsh, ga = O.render_sh_bwd(g["mean2d"], g["cov2d"], sc["sh"], sc["alpha"], g["start"], g["end"], g["ids"], ref,
                                        go.astype(np.float32), cam.topleft, rot, C, 1 / cam.fx, 1 / cam.fy, cam.h, cam.w)
    gm, gq, gs = O.project_bwd(sc["mean"], sc["qvec"], sc["svec"], cam.c2w, gm2, gc2, None, False)
    _assert_entrywise({"mean": gm, "qvec": gq, "svec": gs, "alpha": ga, "sh": gsh}, fd, ("oracle", seed, C))


@pytest.mark.parametrize("seed", [5, 6])
def test_oracle_rgb_chain_against_finite_differences(seed):
    cam, sc, g, go = _case(seed, 0)
    P, img, fd = _fd(sc, cam, 0, go, g, KEYS_RGB, None)
    ref, T = O.render_rgb_fwd(g["mean2d"], g["cov2d"], sc["color"], sc["alpha"], g["start"], g["end"], g["ids"], cam.topleft,
                              1 / cam.fx, 1 / cam.fy, cam.h, cam.w)
    assert np.abs(ref - img).max() <= 1e-5
    gm2, gc2, gcol, ga = O.render_rgb_bwd(g["mean2d"], g["cov2d"], sc["color"], sc["alpha"], g["start"], g["end"], g["ids"],
                                          ref, go.astype(np.float32), cam.topleft, 1 / cam.fx, 1 / cam.fy, cam.h, cam.w)
    gm, gq, gs = O.project_bwd(sc["mean"], sc["qvec"], sc["svec"], cam.c2w, gm2, gc2, None, False)
    _assert_entrywise({"mean": gm, "qvec": gq, "svec": gs, "alpha": ga, "color": gcol}, fd, ("oracle rgb", seed))


# ---- the HIP path -----------------------------------------------------------------------------------------------------
def _T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("seed,C,basis", [(0, 4, "exact"), (0, 4, "auto"), (4, 4, "auto"), (1, 2, "auto"), (2, 3, "exact"),
                                          (3, 1, "auto"), (7, 4, "auto")])
def test_hip_frame_sh_gradients_against_finite_differences(seed, C, basis):
    from gsgen_amd import renderer as R, _capi
    cam, sc, g, go = _case(seed, C)
    bg = np.array([0.3, 0.1, 0.2], np.float32)
    P64, img, fd = _fd(sc, cam, C, go, g, KEYS_SH, bg)
    P = {k: _T(sc[k]).requires_grad_(True) for k in KEYS_SH}
    buf = R.FrameBuffers(sc["mean"].shape[0], cam.w, cam.h, torch.device("cuda:0"))
    rgb, _ = R.render_frame(P["mean"], P["qvec"], P["svec"], P["alpha"], P["sh"], R.CameraInfo(*cam.intr), cam.c2w, buf, C=C,
                            bg_rgb=_T(bg), detach_depth=False, sh_basis=basis)
    (rgb * _T(go.astype(np.float32))).sum().backward()
    assert buf.ensure_capacity() and int(buf.total.item()) == g["D"]
    assert np.array_equal(buf.ids.cpu().numpy()[:g["D"]], g["ids"])  # the lists the fp64 model froze are the GPU's
    assert np.abs(rgb.detach().cpu().numpy() - img).max() <= 1e-4
    if C == 4 and basis == "auto":  # the polynomial form really took this frame: most splats are within their bound
        rows = P["sh"].detach()[:, :, 1:].abs().sum(-1).max(1).values.cpu().numpy()
        ok = np.array([_capi.load().sh_poly_applies(float(r_), 1 / cam.fx, 4) for r_ in rows[rows > 0]])
        assert ok.mean() >= 0.75, ok.mean()  # (the few beyond it are evaluated exactly, entry by entry, inside the same kernel)
    _assert_entrywise({k: P[k].grad.cpu().numpy() for k in KEYS_SH}, fd, ("hip frame", seed, C, basis))


@pytest.mark.gpu
def test_hip_batched_sh_gradients_against_finite_differences():
    """two cameras through BatchRenderer (one autograd node, shared gradient accumulators): the gradient of the summed loss
    against the sum of the two cameras' finite differences"""
    from gsgen_amd import renderer as R
    from gsgen_amd.batch import BatchRenderer
    C, n = 4, 8
    cams = [scenes.Camera(32, 32, fx=600.0 + 80 * i, c2w=scenes.orbit(2.5, 15.0 + 20 * i, 30.0 + 70 * i)) for i in range(2)]
    sc = FD.tiny_scene(n, 11, C, cams[0])
    gos = np.random.default_rng(3).normal(size=(2, 32, 32, 3))
    bg = np.array([0.2, 0.4, 0.1], np.float32)
    want = {k: 0.0 for k in KEYS_SH}
    for cam, go in zip(cams, gos):
        g = scenes.oracle_geometry(sc, cam)
        assert g["mask"].all()
        _, _, fd = _fd(sc, cam, C, go, g, KEYS_SH, bg)
        want = {k: want[k] + fd[k] for k in KEYS_SH}
    P = {k: _T(sc[k]).requires_grad_(True) for k in KEYS_SH}
    br = BatchRenderer(n, 32, 32, torch.device("cuda:0"), max_batch=2)
    rgb, _ = br.render(P["mean"], P["qvec"], P["svec"], P["alpha"], P["sh"], [R.CameraInfo(*c.intr) for c in cams],
                       [c.c2w for c in cams], C=C, bg_rgb=_T(bg), detach_depth=False)
    (rgb * _T(gos.astype(np.float32))).sum().backward()
    assert br.ensure_capacity(2)
    _assert_entrywise({k: P[k].grad.cpu().numpy() for k in KEYS_SH}, want, "hip batch")


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [5, 6])
def test_hip_frame_rgb_gradients_against_finite_differences(seed):
    from gsgen_amd import renderer as R
    cam, sc, g, go = _case(seed, 0)
    _, img, fd = _fd(sc, cam, 0, go, g, KEYS_RGB, None)
    P = {k: _T(sc[k]).requires_grad_(True) for k in KEYS_RGB}
    buf = R.FrameBuffers(sc["mean"].shape[0], cam.w, cam.h, torch.device("cuda:0"))
    rgb, _ = R.render_frame(P["mean"], P["qvec"], P["svec"], P["alpha"], P["color"], R.CameraInfo(*cam.intr), cam.c2w, buf, C=0,
                            detach_depth=False)
    (rgb * _T(go.astype(np.float32))).sum().backward()
    assert buf.ensure_capacity()
    assert np.abs(rgb.detach().cpu().numpy() - img).max() <= 1e-5
    _assert_entrywise({k: P[k].grad.cpu().numpy() for k in KEYS_RGB}, fd, ("hip rgb", seed))
