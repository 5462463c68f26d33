"""Generates tests/golden/model/model_batch.npz from the REFERENCE ITSELF, at the level of its public model API
(run in the authoring container only): the reference's GaussianSplattingRenderer (gs/gaussian_splatting.py,
imported from /root/reference through tests/refshim.py, constructed from a config as the trainer does) renders a
two-camera batch with `_backend` bound to the reference's OWN CUDA kernels compiled for the CPU
(oracle/_ref/libgs_ref.so, oracle/ref.py) -- no code of this repo's product or oracle is on the path.

    forward(batch)  -> rgb, depth, opacity, z_var                   (gs/gaussian_splatting.py:1423-1466, :1198-1421)
    loss.backward() -> gradients of the five RAW parameter fields   (through the reference's activations)
    post_backward() -> mean_2d_grad_accum, cnt; max_radii2d         (:464-469, :1240-1245)

The GPU test (tests/test_gpu_golden.py::test_fused_model_path_matches_the_reference_model_golden) feeds the same raw
parameters, cameras and output gradients to this repo's fused path (BatchRenderer.render_heads + DensifyStats) and
compares everything; tests/test_reference_python_on_mirror.py runs the reference's model class on the `_gs` mirror
against the same file.

    python tests/golden/make_golden_model.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import refshim  # noqa: E402
import scenes  # noqa: E402
from oracle import ref as Rf, ref_build  # noqa: E402

BG = [0.1, 0.2, 0.3]


class Cfg(dict):
    """what the reference reads its OmegaConf node through: attribute access, .get, hasattr"""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None


def model_cfg():
    """the renderer node of conf/base.yaml:129-160 with a fixed background"""
    return Cfg(device="cpu", svec_act="exp", alpha_act="sigmoid", color_act="sigmoid", tile_size=16,
               frustum_culling_radius=6.0, tile_culling_type="aabb", tile_culling_thresh=0.01, tile_culling_radius=6.0,
               T_thresh=1e-4, skip_frustum_culling=False, normal_as_rgb=False, debug=False, depth_detach=True,
               background=Cfg(type="fixed", device="cpu", color=BG, random_aug=False, random_aug_prob=0.0),
               densify=Cfg(enabled=True), prune=Cfg(enabled=False))


def case():
    sc = scenes.random_scene(600, seed=11, svec=0.05, spread=1.2, C=1)
    cams = [scenes.Camera(72, 56, fx=66.0, c2w=scenes.orbit(2.4, 20, 40)),
            scenes.Camera(72, 56, fx=80.0, c2w=scenes.orbit(1.6, -10, 200))]
    return sc, cams


def _n(t):
    return np.ascontiguousarray(t.detach().numpy())


class ReferenceKernels(types.ModuleType):
    """`_gs` as the reference's Python sees it, every entry point executed by the reference's own kernels
    (caller-allocated outputs filled in place, gradients accumulated: gs/src/render.cu)"""

    def __init__(self):
        super().__init__("_gs")

    @staticmethod
    def culling_gaussian_bsphere(mean, qvec, svec, normal, pts, mask, thresh):
        mask.copy_(torch.from_numpy(Rf.cull_bsphere(_n(mean), _n(qvec), _n(svec), _n(normal), _n(pts), thresh)))

    @staticmethod
    def tile_culling_aabb_start_end(tl, br, ids, start, end, depth, nth, ntw):
        i_, s_, e_ = Rf.bin_sort(_n(tl), _n(br), _n(depth), nth, ntw, int(ids.shape[0]))
        ids.copy_(torch.from_numpy(i_)); start.copy_(torch.from_numpy(s_)); end.copy_(torch.from_numpy(e_))

    @staticmethod
    def tile_based_vol_rendering_start_end_with_T(mean, cov, color, alpha, start, end, ids, out, topleft, ts, nth, ntw,
                                                  psx, psy, H, W, thresh, T):
        assert ts == 16
        o, t_ = Rf.render_rgb_fwd(_n(mean), _n(cov), _n(color), _n(alpha), _n(start), _n(end), _n(ids), _n(topleft),
                                  psx, psy, H, W, thresh)
        out.copy_(torch.from_numpy(o).reshape(out.shape)); T.copy_(torch.from_numpy(t_).reshape(T.shape))

    @staticmethod
    def tile_based_vol_rendering_backward_start_end(mean, cov, color, alpha, start, end, ids, out, g_mean, g_cov, g_color,
                                                    g_alpha, g_out, topleft, ts, nth, ntw, psx, psy, H, W, thresh):
        g = Rf.render_rgb_bwd(_n(mean), _n(cov), _n(color), _n(alpha), _n(start), _n(end), _n(ids), _n(out), _n(g_out),
                              _n(topleft), psx, psy, H, W, thresh)
        for dst, src in zip((g_mean, g_cov, g_color, g_alpha), g):
            dst.add_(torch.from_numpy(src).reshape(dst.shape))

    @staticmethod
    def tile_based_vol_rendering_scalar(mean, cov, scalar, alpha, start, end, ids, out, topleft, ts, nth, ntw, psx, psy,
                                        H, W, thresh, T):
        o, t_ = Rf.render_scalar_fwd(_n(mean), _n(cov), _n(scalar).reshape(-1), _n(alpha), _n(start), _n(end), _n(ids),
                                     _n(topleft), psx, psy, H, W, thresh)
        out.copy_(torch.from_numpy(o).reshape(out.shape)); T.copy_(torch.from_numpy(t_).reshape(T.shape))

    @staticmethod
    def tile_based_vol_rendering_scalar_backward(mean, cov, scalar, alpha, start, end, ids, out, g_mean, g_cov, g_scalar,
                                                 g_alpha, g_out, topleft, ts, nth, ntw, psx, psy, H, W, thresh):
        n = mean.shape[0]  # the "ones" scalar of the opacity pass is sized N_full (gs/gaussian_splatting.py:1359)
        g = Rf.render_scalar_bwd(_n(mean), _n(cov), _n(scalar).reshape(-1)[:n], _n(alpha), _n(start), _n(end), _n(ids),
                                 _n(out), _n(g_out), _n(topleft), psx, psy, H, W, thresh)
        g_mean.add_(torch.from_numpy(g[0])); g_cov.add_(torch.from_numpy(g[1]).reshape(g_cov.shape))
        g_scalar.view(-1)[:n].add_(torch.from_numpy(g[2])); g_alpha.add_(torch.from_numpy(g[3]))


def generate():
    refshim.install()
    backend = ReferenceKernels()
    sys.modules["_gs"] = backend
    dm = types.ModuleType("kornia.geometry.depth")  # utils/ops.py:5 (unused on this path)
    dm.depth_to_3d = None
    sys.modules["kornia.geometry.depth"] = dm
    sys.modules["kornia"].__path__ = []
    sys.modules["kornia.geometry"].__path__ = []
    import gs.renderer as GR
    import gs.gaussian_splatting as M
    from utils.camera import CameraInfo
    GR._backend = backend
    M._backend = backend
    sc, cams = case()
    t = lambda a: torch.tensor(np.ascontiguousarray(a))  # noqa: E731
    model = M.GaussianSplattingRenderer(model_cfg(), {k: t(sc[k]) for k in ("mean", "qvec", "svec", "color", "alpha")})
    model.train()
    out = model({"c2w": torch.stack([t(c.c2w) for c in cams]), "camera_info": [CameraInfo(*c.intr) for c in cams]})
    rng = np.random.default_rng(5)
    go = {k: rng.normal(size=tuple(v.shape)).astype(np.float32) for k, v in out.items()}
    sum((out[k] * t(go[k])).sum() for k in out).backward()
    masks = np.stack([m_.numpy() for m_ in model.masks])
    model.post_backward()
    raw = {"mean": model.mean, "qvec": model.qvec, "svec": model.svec_before_activation,
           "color": model.color_before_activation, "alpha": model.alpha_before_activation}
    res = {"bg": np.array(BG, np.float32), "masks": masks,
           "cam_intr": np.array([c.intr for c in cams], np.float64), "c2w": np.stack([c.c2w for c in cams])}
    for k, v in raw.items():
        res["raw_" + k] = _n(v)
        res["grad_" + k] = _n(v.grad)
    for k, v in out.items():
        res["out_" + k] = _n(v)
        res["go_" + k] = go[k]
    res["max_radii2d"], res["grad_accum"], res["cnt"] = _n(model.max_radii2d), _n(model.mean_2d_grad_accum), _n(model.cnt)
    os.makedirs(os.path.join(HERE, "model"), exist_ok=True)
    np.savez_compressed(os.path.join(HERE, "model", "model_batch.npz"), **res)
    return res


if __name__ == "__main__":
    if not refshim.available():
        raise SystemExit("needs /root/reference")
    ref_build.build()
    r = generate()
    print("visible", r["masks"].sum(1), "rgb mean", float(r["out_rgb"].mean()), "cnt sum", float(r["cnt"].sum()),
          os.path.getsize(os.path.join(HERE, "model", "model_batch.npz")) // 1024, "KiB")
