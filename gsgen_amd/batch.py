"""Batched-camera caller of the fused frame path.

The reference renders the cameras of a batch strictly one after another in a Python loop and
stacks the per-camera dicts (gs/gaussian_splatting.py:1423-1466).  One 800x800 frame of 100k
Gaussians does not fill an MI355X (2.5k tiles on 256 CUs, each kernel ending on its longest
tiles), so this caller renders the batch with ONE enqueue per stage -- geometry, compositing forward,
compositing backward, projection backward each launch once for all cameras (gridDim = views x tiles) --
and nothing synchronises with the host.  The whole batch is ONE autograd node; every camera adds
atomically into one set of parameter gradients (no per-camera zero-fill of the SH gradient, 19 MB at
100k and C = 4, and no B-way gradient sum afterwards).

Optionally (pipeline=True) the batch runs as TWO half-batches on two streams -- the renderer's own side stream, forked from
and joined to the caller's stream around the forward and around the backward (a loss needs every image of the batch before
any backward starts).  Measured in round 5 (profiles/r05_notes.md): with that join the split gains nothing with one step in
flight (cfg2 SH 4 823 vs 4 821 renders/s, RGB + heads +0.7 %) and LOSES 6-8 % at the trainer's 4 x 512^2 (917 vs 863 it/s) --
the two 4-view launches of a stage are less efficient than one 8-view launch, and what round 4 had measured as +4..9 % came
from letting one half's backward overlap the other's forward, which no real training step can do.  Hence off by default;
images are the same bits either way, gradients the same up to the order of the atomics.

Every camera of the batch owns a FrameBuffers slot because its backward needs the lists and
projected records of its forward (22 B x N + 12 B x D_cap per slot).

Pair-list overflow (a camera needing more (tile, Gaussian) pairs than its slot holds) is never a
finite blank image: see renderer.FrameBuffers and BatchRenderer.check_overflow.
"""
import ctypes

import numpy as np
import torch

from . import _capi
from . import renderer as R
from .renderer import _p, PairListOverflow


# utils/activations.py:36-57 as gsgen_activate_fields numbers them (include/gsgen_hip.h), and their torch forms (the fallback path)
ACTIVATION_CODES = {"nothing": 0, "exp": 1, "sigmoid": 2, "abs": 3, "relu": 4, "softplus": 5, "biased_relu": 6, "biased_abs": 7}
TORCH_ACTIVATIONS = {"nothing": lambda x: x, "exp": torch.exp, "sigmoid": torch.sigmoid, "abs": torch.abs, "relu": torch.relu,
                     "softplus": torch.nn.functional.softplus, "biased_relu": lambda x: torch.relu(x) + 1e-3,
                     "biased_abs": lambda x: torch.abs(x) + 1e-3}

_EXT = [False]  # the compiled `_gsbatch` module (csrc/torch_batch.cpp), None when it is not built; looked up once


def _batch_ext():
    if _EXT[0] is False:
        import glob
        import importlib.util
        import os
        hits = glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ext", "_gsbatch.*.so"))
        mod = None
        if hits:
            _capi.load()  # (libgsgen_hip.so and torch's HIP runtime first: one libamdhip64 per process)
            spec = importlib.util.spec_from_file_location("_gsbatch", hits[0])
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
        _EXT[0] = mod
    return _EXT[0]


def _bg_grad(ctx, grad_rgb, T):
    """d/d bg of rgb = ... + T * bg (gs/renderer.py:1283: nan_to_num(grad * T)), reduced to bg's shape; None unless
    the background takes part in the graph (the reference's ConstBackground / MLPBackground are trainable)"""
    if getattr(ctx, "bg_shape", None) is None or grad_rgb is None:
        return None
    return torch.nan_to_num(grad_rgb * T).sum_to_size(ctx.bg_shape)


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NO_GUARD = _NoGuard()


def _on(dev):
    """`with _on(dev):` = torch.cuda.device(dev), free when dev is already the current device (the context manager costs
    ~5 us per use; a step of the fused path entered it four times)"""
    return _NO_GUARD if torch.cuda.current_device() == dev.index else torch.cuda.device(dev)


def _sub(table, lo, n):
    """rows lo .. lo + n of a ctypes array as an array of its own (the same memory: what a half-batch hands the C ABI)"""
    if lo == 0 and n == len(table):
        return table
    et = table._type_
    return (et * n).from_address(ctypes.addressof(table) + lo * ctypes.sizeof(et))


class _render_batch(torch.autograd.Function):
    """(mean, qvec, svec, alpha, sh|color) -> rgb [B,H,W,3], T [B,H,W,1] for B cameras."""

    @staticmethod
    def forward(ctx, mean, qvec, svec, alpha, col, cams, br, B, C, bg_rgb, thresh, detach_depth, stats):
        """one enqueue of the geometry kernels (gridDim.y = cameras) and one compositing launch per half-batch.  Nothing is
        filled between the stages: the projection launch zeroes the step's gradient accumulators (the renderer's per-view
        blocks and the shared block allocated here, gsgen_frame_geometry_batch_zero), the compositing launch writes every
        pixel of out / T."""
        mean, qvec, svec = mean.contiguous(), qvec.contiguous(), svec.contiguous()
        alpha, col = alpha.contiguous(), col.contiguous()
        lib = _capi.load()
        H, W, N, dev = br.H, br.W, br.N, mean.device
        # (BatchRenderer.render has run _begin_batch: overflow check + generation)
        out = torch.empty(B, H, W, 3, device=dev, dtype=torch.float32)
        T = torch.empty(B, H, W, 1, device=dev, dtype=torch.float32)
        out_p, T_p = out.data_ptr(), T.data_ptr()
        kind = "sh" if C > 0 else "rgb"  # C == 0: post-activation colours
        # d L / d alpha [N] | d L / d col, shared by the views: returned by the backward, hence allocated per batch -- and
        # zeroed by this forward's projection launch
        gsh = torch.empty(br._Np + (col.numel() + 3) // 4 * 4, device=dev, dtype=torch.float32)
        with _on(dev):
            parts = br._fork(B)
            views = br._geometry(kind, B, parts, mean, qvec, svec, gsh, stats)
            bg_p = _p(bg_rgb)
            cis = br._cis
            if C > 0:
                rr, nf = br._route_args
                for i in range(B):
                    ci, v = cis[i], views[i]
                    v.pixel_size_x, v.pixel_size_y = 1.0 / ci.fx, 1.0 / ci.fy
                    v.T, v.bg_rgb, v.out = T_p + 4 * H * W * i, bg_p, out_p + 12 * H * W * i
                    v.route_report, v.no_fallback = rr, nf
            else:
                for i in range(B):
                    ci, v = cis[i], views[i]
                    v.pixel_size_x, v.pixel_size_y = 1.0 / ci.fx, 1.0 / ci.fy
                    v.T, v.out6 = T_p + 4 * H * W * i, out_p + 12 * H * W * i  # read as [H,W,3] by the RGB entry points
            nth, ntw = br.slots[0].nth, br.slots[0].ntw
            for k, (lo, n, s) in enumerate(parts):
                if C > 0:
                    lib.vol_render_sh_batch_routed(n, _sub(views, lo, n), N, _p(col), _p(alpha), 16, nth, ntw, H, W, C,
                                                   thresh, br.segments, _p(br._sh_bound), _p(br._sh_rows), _p(br._bws[k]), s)
                else:
                    lib.vol_render_rgb_batch(n, _sub(views, lo, n), N, _p(col), _p(alpha), 16, nth, ntw, H, W, thresh,
                                             _p(br._bws[k]), s)
            br._join(parts)
        if C == 0 and bg_rgb is not None:
            out = out + T * bg_rgb  # gs/renderer.py:1182; `out` (saved below) is what the backward reads as final
            for i in range(B):
                views[i].out6 = out.data_ptr() + 12 * H * W * i
        ctx.gen = br._generation
        ctx.views, ctx.gsh, ctx.parts = views, gsh, [(lo, n) for lo, n, _ in parts]
        ctx.save_for_backward(mean, qvec, svec, alpha, col, cams, out, T)
        ctx.bg_shape = tuple(bg_rgb.shape) if (bg_rgb is not None and ctx.needs_input_grad[9]) else None
        ctx.br, ctx.B, ctx.C, ctx.thresh, ctx.detach, ctx.stats = br, B, C, thresh, detach_depth, stats
        ctx.sh_bound, ctx.sh_rows = br._sh_bound, br._sh_rows  # forward and backward of a batch route on the same device values
        ctx.mark_non_differentiable(T)
        return out, T

    @staticmethod
    def backward(ctx, grad, _gT):
        ctx.br._check_generation(ctx.gen)
        mean, qvec, svec, alpha, col, cams, out, T = ctx.saved_tensors
        br, B, C, thresh, stats = ctx.br, ctx.B, ctx.C, ctx.thresh, ctx.stats
        lib = _capi.load()
        H, W, N, dev = br.H, br.W, br.N, mean.device
        grad = grad.contiguous()
        Np = br._Np
        gsh, ctx.gsh = ctx.gsh, None
        if gsh is None:  # a second backward through the same graph (retain_graph): the accumulators were handed out
            gsh = torch.zeros(Np + (col.numel() + 3) // 4 * 4, device=dev, dtype=torch.float32)
            br._g2d[:B].zero_()
        g_alpha, g_col = gsh[:N], gsh[Np:Np + col.numel()].view(col.shape)
        g3d = torch.empty(10 * N, device=dev, dtype=torch.float32)   # mean | qvec | svec: overwritten
        g_mean, g_qvec, g_svec = g3d[:3 * N].view(N, 3), g3d[3 * N:7 * N].view(N, 4), g3d[7 * N:].view(N, 3)
        grad_p = grad.data_ptr()
        views = ctx.views
        if C > 0:
            for i in range(B):
                views[i].grad_out = grad_p + 12 * H * W * i
        else:
            for i in range(B):
                views[i].grad_out6 = grad_p + 12 * H * W * i
        nth, ntw = br.slots[0].nth, br.slots[0].ntw
        with _on(dev):
            parts = br._fork(B, ctx.parts)
            for k, (lo, n, s) in enumerate(parts):
                if C > 0:  # the moment form (round 6, include/gsgen_hip.h): expanded by the projection backward below
                    lib.vol_render_backward_sh_batch_routed_moments(n, _sub(views, lo, n), N, _p(col), _p(alpha), _p(g_col),
                                                                    _p(g_alpha), 16, nth, ntw, H, W, C, thresh, br.segments,
                                                                    _p(ctx.sh_bound), _p(ctx.sh_rows), _p(br._bws[k]), s)
                else:
                    lib.vol_render_rgb_backward_batch(n, _sub(views, lo, n), N, _p(col), _p(alpha), _p(g_col), _p(g_alpha), 16,
                                                      nth, ntw, H, W, thresh, _p(br._bws[k]), s)
            br._join(parts)
            s = parts[0][2]
            acc = stats.grad_accum if stats is not None else None
            if C > 0:  # (the backward's statistics -- sum |d L / d mean2d|, visits -- are summed by this launch itself)
                lib.project_gaussians_backward_batch_moments_sh(B, N, _p(mean), _p(qvec), _p(svec), br._ptr_table("cam", B),
                                                                int(ctx.detach), br._mask_table(B), br._ptr_table("g_mean2d", B),
                                                                br._ptr_table("g_cov2d", B), br._ptr_table("cov2d", B), _p(g_mean),
                                                                _p(g_qvec), _p(g_svec), _p(acc), _p(stats.cnt) if acc is not None else None, s)
            else:
                lib.project_gaussians_backward_batch(B, N, _p(mean), _p(qvec), _p(svec), br._ptr_table("cam", B),
                                                     int(ctx.detach), br._mask_table(B), br._ptr_table("g_mean2d", B),
                                                     br._ptr_table("g_cov2d", B), None, _p(g_mean), _p(g_qvec), _p(g_svec), s)
                if acc is not None:
                    lib.densify_update_batch(B, N, None, br._ptr_table("g_mean2d", B), br._mask_table(B), None,
                                             _p(stats.grad_accum), _p(stats.cnt), s)
        return (g_mean, g_qvec, g_svec, g_alpha, g_col, None, None, None, None, _bg_grad(ctx, grad, T), None, None, None)


def _bg_rows(bg, B):
    """-> (tensor to keep alive, [address of view i's background colour]) for bg_rgb of 3 or 3 B elements (any broadcastable shape)"""
    if bg is None:
        return None, [None] * B
    if bg.numel() == 3 and bg.is_contiguous():
        return bg, [bg.data_ptr()] * B
    rows = bg.detach().expand(B, 1, 1, 3).contiguous()
    return rows, [rows.data_ptr() + 12 * i for i in range(B)]


class _render_batch_heads(torch.autograd.Function):
    """(mean, qvec, svec, alpha, color) -> rgb [B,H,W,3], depth, opacity, depth^2 (or z_var), T [B,H,W,1]: the default
    (rgb_only = False) output set of GaussianSplattingRenderer.forward (gs/gaussian_splatting.py:1304-1416,
    :1423-1466), one fused compositing pass per camera (gsgen_vol_render_rgbd) instead of four.  Round 6: four separate contiguous
    images, the background and (z_var) the depth variance formed inside the launches -- see csrc/torch_batch.cpp, HeadsFn."""

    @staticmethod
    def forward(ctx, mean, qvec, svec, alpha, col, cams, br, B, bg_rgb, thresh, detach_depth, stats, z_var):
        mean, qvec, svec = mean.contiguous(), qvec.contiguous(), svec.contiguous()
        alpha, col = alpha.contiguous(), col.contiguous()
        lib = _capi.load()
        H, W, N, dev = br.H, br.W, br.N, mean.device
        # (BatchRenderer.render_heads has run _begin_batch; every pixel of the images / T is written by the batched forward, the
        # gradient accumulators are zeroed by the projection launch: no fill kernels)
        f = dict(device=dev, dtype=torch.float32)
        rgb = torch.empty(B, H, W, 3, **f)
        dep, opa, zz, T = (torch.empty(B, H, W, 1, **f) for _ in range(4))
        bg_grad = bg_rgb is not None and ctx.needs_input_grad[8]
        gsh = torch.empty(br._Np + (256 * B if bg_grad else 0), **f)  # d L / d alpha, shared by the views | d L / d bg partial rows
        bg_keep, bg_ptrs = _bg_rows(bg_rgb, B)
        with _on(dev):
            parts = br._fork(B)
            views = br._geometry("rgbd", B, parts, mean, qvec, svec, gsh, stats)
            cis = br._cis
            for i in range(B):
                ci, v = cis[i], views[i]
                v.pixel_size_x, v.pixel_size_y = 1.0 / ci.fx, 1.0 / ci.fy
                v.out6, v.T = None, T.data_ptr() + 4 * H * W * i
                v.out_rgb = rgb.data_ptr() + 12 * H * W * i
                v.out_depth, v.out_opacity, v.out_depth2 = (x.data_ptr() + 4 * H * W * i for x in (dep, opa, zz))
                v.bg_rgb, v.depth_variance = bg_ptrs[i], 1 if z_var else 0
                v.grad_bg = gsh.data_ptr() + 4 * (br._Np + 256 * i) if bg_grad else None
            nth, ntw = br.slots[0].nth, br.slots[0].ntw
            for k, (lo, n, s) in enumerate(parts):
                lib.vol_render_rgbd_batch(n, _sub(views, lo, n), N, _p(col), _p(alpha), 16, nth, ntw, H, W, thresh,
                                          _p(br._bws[k]), s)
            br._join(parts)
        ctx.gen = br._generation
        ctx.views, ctx.gsh, ctx.parts = views, gsh, [(lo, n) for lo, n, _ in parts]
        ctx.save_for_backward(mean, qvec, svec, alpha, col, cams, rgb, dep, opa, zz, T)
        ctx.bg_keep, ctx.bg_grad = bg_keep, bg_grad
        ctx.bg_shape = tuple(bg_rgb.shape) if bg_grad else None
        ctx.br, ctx.B, ctx.thresh, ctx.detach, ctx.stats = br, B, thresh, detach_depth, stats
        ctx.mark_non_differentiable(T)
        return rgb, dep, opa, zz, T

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_opac, g_z2, _gT):
        """one compositing launch per half-batch that reads the four head gradients in place, forms the depth variance's chain rule and
        the background's gradient itself, one projection launch that expands the moments, forms d L / d depth = g3 + 2 depth g5 and
        sums the colour gradient over the views (gsgen_project_gaussians_backward_batch_heads_moments)"""
        ctx.br._check_generation(ctx.gen)
        mean, qvec, svec, alpha, col, cams = ctx.saved_tensors[:6]
        br, B, thresh, stats = ctx.br, ctx.B, ctx.thresh, ctx.stats
        lib = _capi.load()
        H, W, N, dev = br.H, br.W, br.N, mean.device
        parts_g = [g.contiguous() if g is not None else None for g in (g_rgb, g_depth, g_opac, g_z2)]
        gsh, ctx.gsh = ctx.gsh, None
        if gsh is None:  # a second backward through the same graph (retain_graph): the accumulators were handed out
            gsh = torch.zeros(br._Np + (256 * B if ctx.bg_grad else 0), device=dev, dtype=torch.float32)
            br._g2d[:B].zero_()
            br._chan6()[:B].zero_()
        g_alpha = gsh[:N]
        g3d = torch.empty(13 * N, device=dev, dtype=torch.float32)  # mean | qvec | svec | colour: overwritten
        g_mean, g_qvec = g3d[:3 * N].view(N, 3), g3d[3 * N:7 * N].view(N, 4)
        g_svec, g_col = g3d[7 * N:10 * N].view(N, 3), g3d[10 * N:].view(N, 3)
        views = ctx.views
        pp = [x.data_ptr() if x is not None else None for x in parts_g]
        for i in range(B):
            v = views[i]
            v.grad_out6 = None
            v.grad_rgb = pp[0] + 12 * H * W * i if pp[0] is not None else None
            v.grad_depth = pp[1] + 4 * H * W * i if pp[1] is not None else None
            v.grad_opacity = pp[2] + 4 * H * W * i if pp[2] is not None else None
            v.grad_depth2 = pp[3] + 4 * H * W * i if pp[3] is not None else None
            v.grad_bg = gsh.data_ptr() + 4 * (br._Np + 256 * i) if ctx.bg_grad else None
        nth, ntw = br.slots[0].nth, br.slots[0].ntw
        with _on(dev):
            parts = br._fork(B, ctx.parts)
            for k, (lo, n, s) in enumerate(parts):
                # the moment form (round 6, include/gsgen_hip.h): ten components per (tile, Gaussian) instead of thirteen
                lib.vol_render_rgbd_backward_batch_moments(n, _sub(views, lo, n), N, _p(col), _p(alpha), _p(g_alpha), 16, nth, ntw,
                                                           H, W, thresh, _p(br._bws[k]), s)
            br._join(parts)
            s = parts[0][2]
            # ... expanded per (view, Gaussian); leaves d L / d mean2d in the per-view blocks for the statistics below
            lib.project_gaussians_backward_batch_heads_moments(B, N, _p(mean), _p(qvec), _p(svec), br._ptr_table("cam", B),
                                                               int(ctx.detach), br._mask_table(B), br._ptr_table("g_mean2d", B),
                                                               br._ptr_table("g_cov2d", B), br._ptr_table("g_chan6", B),
                                                               br._ptr_table("depth", B), br._ptr_table("cov2d", B),
                                                               br._ptr_table("chol", B), _p(g_mean), _p(g_qvec), _p(g_svec),
                                                               _p(g_col), _p(stats.grad_accum) if stats is not None else None,
                                                               _p(stats.cnt) if (stats is not None and stats.grad_accum is not None) else None, s)
        g_bg = None
        if ctx.bg_grad and g_rgb is not None:
            g_bg = gsh[br._Np:br._Np + 256 * B].view(B, 64, 4)[..., :3].sum(1).view(B, 1, 1, 3).sum_to_size(ctx.bg_shape)
        return (g_mean, g_qvec, g_svec, g_alpha, g_col, None, None, None, g_bg, None, None, None, None)


class BatchRenderer:
    """Renders [B] cameras of one (W, H) shape for a fixed Gaussian count N."""

    def __init__(self, N, W, H, device, max_batch, D_cap=None, segments=1, strict=False, pipeline=False, device_cameras=False):
        """D_cap: capacity of every slot's (tile, Gaussian) pair list.  None (default): the FIRST batch is rendered
        synchronously (one host sync) and sizes all slots at 1.5 x the largest count it saw; later batches report their
        counts through the geometry launch itself (no sync, renderer.PairCountReport) and the lists are regrown before the
        scene outgrows them.  A camera that still overflows is rendered as NaN, never as a finite blank image, and the next
        render()/check_overflow() raises PairListOverflow with the lists already regrown.
        strict=True: every batch is rendered synchronously (count read back, regrown and binned again if it did not fit):
        lossless, one host sync per batch -- what the reference pays per CAMERA (gs/culling.py:34).
        pipeline: False (default) -- one launch per stage for the whole batch; True -- two half-batches on two streams (the
        caller's and the renderer's own, forked and joined inside the call); "auto" -- True from 4 cameras on.  Measured not to
        pay under the join a loss imposes (module docstring); kept for callers whose backward may start per half.
        segments: backward workgroups per tile (FrameBuffers).
        device_cameras: False (default) -- a batch's camera blocks travel in kernel arguments (gsgen_upload_small) and its pixel sizes
        in the view tables, i.e. in kernel arguments too; True -- the compositing kernels read both from DEVICE memory (filled by one
        enqueue, upload_cameras), and a render enqueued while the stream is being captured uploads nothing: a
        captured hipGraph of a step (gsgen_amd.graph.CapturedStep) then replays for whatever cameras were uploaded before the
        replay -- poses and intrinsics.  RGB + heads and RGB batches (render_heads, render with C = 0); SH batches keep their pixel
        sizes in the view tables."""
        self.N, self.W, self.H, self.device = N, W, H, torch.device(device)
        self.device_cameras = bool(device_cameras)
        self.segments = int(segments)
        self.strict = bool(strict)
        if pipeline not in ("auto", True, False):
            raise ValueError("pipeline: 'auto', True or False")
        self.pipeline = pipeline
        # the slots' pair counters live in one tensor (one copy brings a batch's counts to the host where a sync is wanted)
        self._totals = torch.zeros(max_batch, device=device, dtype=torch.int32)
        self._report = R.PairCountReport(max_batch)
        # SH degree 3, per-tile routing: one more host-visible word -- "a tile crowded with splats beyond the bound was seen" -- written by
        # the polynomial forward (gsgen_sh_view::route_report).  Three clean reports in a row and the persistent exact fallbacks are no
        # longer enqueued (no_fallback: their workgroups, 152 registers each, otherwise have to find room on a full chip to discover
        # that they have nothing to do -- 0.23 + 0.09 ms of stream time per step in flight in round 5's trace); one report brings them back.
        self._route = R.PairCountReport(1)
        self._route_clean = 0
        self._route_args = (None, 0)
        self.strict = self.strict or self._report.unmapped  # (no report channel: size every batch with the read-back)
        self._gen = np.zeros(1, np.int64)  # the generation counter, in memory the C++ autograd node reads too
        self._plans = {}
        self.use_ext = True    # the C++ autograd node (csrc/torch_batch.cpp) where it is built and applicable
        self._sh_bound = self._sh_rows = None
        self._table_cache = {}
        # the slots' depth buffers are the rows of one matrix (the heads' backward reads depths[:B] as one tensor)
        self._depths = torch.empty(max_batch, N, device=device, dtype=torch.float32)
        self.slots = [R.FrameBuffers(N, W, H, device, D_cap=D_cap, segments=self.segments, total=self._totals[i:i + 1],
                                     depth=self._depths[i].view(N, 1), report=(self._report, i), strict=self.strict)
                      for i in range(max_batch)]
        self._ptr_tabs = {}
        self._side = None      # the second stream of a pipelined batch, made on first use
        self._ev = None
        # Buffers a step needs and nobody else sees live as long as the renderer (round 4: nothing is allocated or filled per
        # step except what is handed to the caller): the camera rows of the batch in flight, the kernel-parameter tables of its
        # launches, and the per-view gradient accumulators mean2d (2) | cov2d (4) x Np (+ the six channels of render_heads, made
        # on its first call: 24 B x N per slot) -- zeroed by the projection launch of the step's forward
        # (gsgen_frame_geometry_batch_zero).  Only one batch per renderer is between forward and backward
        # (_check_generation), so one set is enough.  Memory per slot beyond the lists: (22 + 24 [+ 24]) B x N.
        lib = _capi.load()
        self._Np = (N + 3) // 4 * 4
        # camera rows [max_batch, 68] | pixel sizes [max_batch, 2] (device_cameras: gsgen_rgbd_view::pixel_size_dev)
        self._camblock = torch.empty(max_batch * 70, device=device, dtype=torch.float32)
        self._cams = self._camblock[:max_batch * 68].view(max_batch, 68)
        self._pix = self._camblock[max_batch * 68:].view(max_batch, 2)
        self._hostblock = np.zeros(max_batch * 70, np.float32)
        nth_, ntw_ = R.n_tiles(H, W)
        self._nb_sh = lib.sh_batch_workspace_bytes_routed(max_batch, nth_ * ntw_)  # parameter tables + per-tile routing flags
        # one batch workspace per half (routing flags of its views) + the geometry launches' view tables
        self._bws = [torch.empty(self._nb_sh, device=device, dtype=torch.uint8) for _ in range(2)]
        self._gws = torch.empty(lib.frame_batch_workspace_bytes(max_batch), device=device, dtype=torch.uint8)
        self._gv_bytes = lib.frame_batch_workspace_bytes(1)
        self._g2d = torch.empty(max_batch, 6 * self._Np, device=device, dtype=torch.float32)
        self._gch = None
        self._chols = None
        self._rows = torch.zeros(N, device=device, dtype=torch.float32)  # per-splat bounds of the batch in flight (gsgen_sh_l1_bound_rows)
        self._smax = torch.zeros(1, device=device, dtype=torch.float32)  # ... and their maximum (a scene within a view's bound skips the per-entry tests)
        # per camera: cam block (56) | topleft (2) | rotation rows (9) | pad -> 68 floats, packed on the host and sent
        # through kernel arguments (gsgen_upload_small): no pinned ring, no copy event, the host never waits
        self._host = np.zeros((max_batch, 68), np.float32)
        self._poses = np.zeros((max_batch, 12), np.float32)
        self._intr = np.zeros((max_batch, 8), np.float64)
        self._cis = []
        self._last_parts = [(0, 0)]
        self._bound_tick = 0

    @property
    def _generation(self):
        return int(self._gen[0])

    @_generation.setter
    def _generation(self, v):
        self._gen[0] = v

    # ---- buffers and tables ------------------------------------------------------------------------------------------
    def _chan6(self):
        """[max_batch, 6 Np]: the per-view accumulators of d L / d (r, g, b, depth, 1, depth^2) (render_heads only)"""
        if self._gch is None:
            self._gch = torch.empty(len(self.slots), 6 * self._Np, device=self.device, dtype=torch.float32)
            self._ptr_tabs.clear()
            self._table_cache.clear()
            self._plans.clear()
        return self._gch

    def _chol(self, i):
        """slot i's [N,4] prepared evaluation records (gsgen_geometry_view::chol; RGB and RGB + heads batches only)"""
        if self._chols is None:
            self._chols = torch.empty(len(self.slots), self._Np, 4, device=self.device, dtype=torch.float32)
        return self._chols[i]

    def _upload(self, cam_infos, c2ws, frustum_radius, tile_radius):
        """The batch's camera blocks, packed by ONE library call (gsgen_pack_camera_blocks) and sent through kernel
        arguments (gsgen_upload_small).  (Per camera in Python -- numpy slices, one ctypes call each -- this was 15 us a
        camera: a tenth of a 4 x 512^2 step.)"""
        B = len(cam_infos)
        poses, intr, h = self._poses, self._intr, self._host
        if isinstance(c2ws, np.ndarray) and c2ws.ndim == 3:  # poses already stacked: [B, 3 or 4, 4]
            poses[:B] = c2ws[:, :3, :4].reshape(B, 12)
        else:
            for i, c2w in enumerate(c2ws):
                if isinstance(c2w, torch.Tensor):
                    c2w = c2w.detach().cpu().numpy()
                poses[i] = c2w.ravel()[:12] if (type(c2w) is np.ndarray and c2w.dtype == np.float32) else \
                    np.asarray(c2w, np.float32).reshape(-1)[:12]
        intr[:B] = [(ci.fx, ci.fy, ci.cx, ci.cy, ci.w, ci.h, ci.near_plane, ci.far_plane) for ci in cam_infos]  # (one assignment)
        lib = _capi.load()
        if self.device_cameras:
            # a render enqueued into a stream capture uploads nothing: the replay renders what upload_cameras put in place before it
            if torch.cuda.is_current_stream_capturing():
                return self._cams[:B]
            nb = len(self.slots)
            hp = self._hostblock
            lib.pack_camera_blocks(B, poses.ctypes.data, 12, intr.ctypes.data, frustum_radius, tile_radius, hp.ctypes.data)
            px = hp[nb * 68:].reshape(nb, 2)
            px[:B, 0] = 1.0 / intr[:B, 0]  # (rounded to fp32 exactly as the view tables' floats are)
            px[:B, 1] = 1.0 / intr[:B, 1]
            # rows and pixel sizes in one enqueue, through kernel arguments like the default upload (outside a capture that is as good
            # as anywhere; the source is reusable at once -- a DMA copy from pinned memory was measured first: 10 us of stream time each)
            lib.upload_small(_p(self._camblock), hp.ctypes.data, nb * 280, torch.cuda.current_stream(self.device).cuda_stream)
            return self._cams[:B]
        lib.pack_camera_blocks(B, poses.ctypes.data, 12, intr.ctypes.data, frustum_radius, tile_radius, h.ctypes.data)
        # the renderer's own device block: its rows are read by the batch's backward, and no other batch of this renderer
        # may come between a forward and its backward (_check_generation)
        lib.upload_small(_p(self._cams), h.ctypes.data, B * 272, torch.cuda.current_stream(self.device).cuda_stream)
        return self._cams[:B]

    def upload_cameras(self, cam_infos, c2ws, frustum_radius=6.0, tile_radius=6.0):
        """device_cameras renderers: put a batch's cameras (poses and intrinsics) in place on the current stream without rendering --
        what a captured step's replay then renders (gsgen_amd.graph.CapturedStep does this for you)"""
        if not self.device_cameras:
            raise RuntimeError("upload_cameras: this BatchRenderer was built without device_cameras=True")
        self._check_batch(cam_infos)
        self._upload(cam_infos, c2ws, frustum_radius, tile_radius)
        self._cis = list(cam_infos)

    def _mask_table(self, B):
        """void*[B] of the slots' visibility masks (they never move): built once per batch size"""
        t = self._ptr_tabs.get(B)
        if t is None:
            t = self._ptr_tabs[B] = (ctypes.c_void_p * B)(*[_p(self.slots[i].mask) for i in range(B)])
        return t

    def _ptr_table(self, kind, B):
        """void*[B] of per-view addresses that never move: the camera rows, the slots' cov2d / depth buffers, the per-view
        gradient blocks (built once per kind and batch size)"""
        key = (kind, B)
        t = self._ptr_tabs.get(key)
        if t is None:
            g0, st = self._g2d.data_ptr(), 4 * 6 * self._Np
            if kind == "cam":
                a = [self._cams.data_ptr() + 272 * i for i in range(B)]
            elif kind == "cov2d":
                a = [_p(self.slots[i].cov2d) for i in range(B)]
            elif kind == "depth":
                a = [_p(self.slots[i].depth) for i in range(B)]
            elif kind == "chol":
                a = [_p(self._chol(i)) for i in range(B)]
            elif kind == "g_mean2d":
                a = [g0 + st * i for i in range(B)]
            elif kind == "g_cov2d":
                a = [g0 + st * i + 4 * 2 * self._Np for i in range(B)]
            elif kind == "g_chan6":
                c0 = self._chan6().data_ptr()
                a = [c0 + st * i for i in range(B)]
            else:
                raise KeyError(kind)
            t = self._ptr_tabs[key] = (ctypes.c_void_p * B)(*a)
        return t

    def _tables(self, kind):
        """(GeometryView[max_batch], ShView | RgbdView[max_batch]) with every per-slot buffer address filled in; rebuilt
        when the pair lists are regrown.  One set per kind ("sh", "rgb", "rgbd"): a batch's backward reads the
        view table its forward filled, and only one batch per BatchRenderer is between forward and backward
        (_check_generation)."""
        cap = self.slots[0].D_cap
        hit = self._table_cache.get(kind)
        if hit is not None and hit[0] == cap:
            return hit[1], hit[2]
        n = len(self.slots)
        geo = (_capi.GeometryView * n)()
        views = ((_capi.ShView if kind == "sh" else _capi.RgbdView) * n)()
        cams_p, g0, st = self._cams.data_ptr(), self._g2d.data_ptr(), 4 * 6 * self._Np
        c0 = self._chan6().data_ptr() if kind == "rgbd" else None
        for i, buf in enumerate(self.slots):
            g, v = geo[i], views[i]
            cam = cams_p + 272 * i  # row i: cam block | topleft at +56 floats | rotation at +58
            g.cam = cam
            g.mean2d, g.cov2d, g.depth, g.mask = _p(buf.mean2d), _p(buf.cov2d), _p(buf.depth), _p(buf.mask)
            g.gaussian_ids, g.start, g.end, g.total = _p(buf.ids), _p(buf.start), _p(buf.end), _p(buf.total)
            g.workspace, g.workspace_bytes, g.D_cap = _p(buf.ws), buf.ws.numel(), buf.D_cap
            g.pair_report = self._report.ptr(i)
            if kind != "sh":  # the compositing kernels' evaluation records, prepared by the projection launch (include/gsgen_hip.h)
                g.chol = v.chol = _p(self._chol(i))
            # this view's gradient accumulators: zero-filled by the projection launch, read by the projection backward
            g.zero_grad_mean2d = v.grad_mean = g0 + st * i
            g.zero_grad_cov2d = v.grad_cov = g0 + st * i + 4 * 2 * self._Np
            v.mean, v.cov, v.start, v.end, v.gaussian_ids = _p(buf.mean2d), _p(buf.cov2d), _p(buf.start), _p(buf.end), _p(buf.ids)
            v.tile_order = buf.tile_order()
            v.topleft = cam + 224
            if kind == "sh":
                v.c2w = cam + 232
                v.segment_workspace = _p(buf.seg_ws) if self.segments > 1 else None
            else:
                v.depth = _p(buf.depth)
                if kind == "rgbd":
                    g.zero_grad_chan6 = v.grad_chan6 = c0 + st * i
                if self.device_cameras:
                    v.pixel_size_dev = self._pix.data_ptr() + 8 * i
        self._table_cache[kind] = (cap, geo, views)
        return geo, views

    # ---- the C++ autograd node ---------------------------------------------------------------------------------------
    def _plan(self, kind, B):
        """-> (address of the _gsbatch.Plan of this (kind, batch size), the view table) when the batch can take the C++ node:
        the module is built, the lists are sized, no host sync is wanted (strict), one launch per stage (no half-batches);
        else None (the Python Functions: same launches)."""
        ext = _batch_ext() if self.use_ext else None
        if ext is None or self.strict or B == 0 or len(self._split(B)) > 1 or self.slots[0].needs_sync_sizing():
            return None
        cap = self.slots[0].D_cap
        key = (kind, B)
        hit = self._plans.get(key)
        if hit is not None and hit[0] == cap:
            return hit[2], hit[3]
        geo, views = self._tables(kind)
        adr = ctypes.addressof
        heads = kind == "rgbd"
        plan = ext.Plan({"rgbd": 0, "rgb": 1, "sh": 2}[kind], B, self.N, self._Np, self.W, self.H, self.slots[0].nth, self.slots[0].ntw,
                        self.segments, adr(geo), adr(views), adr(self._ptr_table("cam", B)), adr(self._mask_table(B)),
                        adr(self._ptr_table("g_mean2d", B)), adr(self._ptr_table("g_cov2d", B)),
                        adr(self._ptr_table("g_chan6", B)) if heads else 0, adr(self._ptr_table("depth", B)),
                        adr(self._ptr_table("cov2d", B)), adr(self._ptr_table("chol", B)) if kind != "sh" else 0, _p(self._gws),
                        _p(self._bws[0]), self._gen.ctypes.data, self._g2d,
                        self._chan6() if heads else None,
                        # what a pending backward of this plan needs alive even if the renderer is dropped first: the host tables,
                        # the generation cell, the slots' buffers as they are NOW (a regrowth replaces them and the plan), the
                        # workspaces -- not the renderer itself (it caches the plan: a cycle Python could not collect)
                        [geo, views, dict(self._ptr_tabs), self._gen, self._cams, self._gws, self._bws[0], self._rows, self._smax, self._chols,
                         [(b_.mean2d, b_.cov2d, b_.depth, b_.mask, b_.ids, b_.start, b_.end, b_.ws, b_.total, b_.seg_ws)
                          for b_ in self.slots[:B]]])
        va = np.ctypeslib.as_array(ctypes.cast(views, ctypes.POINTER(ctypes.c_uint8)), (ctypes.sizeof(views),)).view(np.dtype(views._type_))
        self._plans[key] = (cap, plan, plan.address(), va)
        return plan.address(), va

    def _stats_args(self, stats):
        return (None, None, None) if stats is None else (stats.max_radii2d, stats.grad_accum, stats.cnt)

    # ---- half-batches --------------------------------------------------------------------------------------------------
    def _fork(self, B, split=None):
        """-> [(first view, views, raw stream)]: the batch as one part on the current stream, or as two halves -- the second
        on the renderer's side stream, which waits for everything enqueued on the current stream so far.  `split`: the
        forward's partition, for its backward."""
        cur = torch.cuda.current_stream(self.device)
        if split is None:
            split = self._split(B)
        self._last_parts = split
        if len(split) == 1:
            return [(0, split[0][1], cur.cuda_stream)]
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
            self._ev = [torch.cuda.Event(), torch.cuda.Event()]
        self._ev[0].record(cur)
        self._side.wait_event(self._ev[0])
        return [(split[0][0], split[0][1], cur.cuda_stream), (split[1][0], split[1][1], self._side.cuda_stream)]

    def _split(self, B):
        two = B >= 2 and (self.pipeline is True or (self.pipeline == "auto" and B >= 4))
        h = (B + 1) // 2
        return [(0, h), (h, B - h)] if two else [(0, B)]

    def _join(self, parts):
        """the current stream waits for the side stream's half (what the call allocated on the current stream and used on
        the side stream is therefore free to be reused in current-stream order: no record_stream needed)"""
        if len(parts) > 1:
            self._ev[1].record(self._side)
            torch.cuda.current_stream(self.device).wait_event(self._ev[1])

    def _geometry(self, kind, B, parts, mean, qvec, svec, gsh, stats=None):
        """the geometry chain of every part (the first one zero-fills the shared gradient block) -> the view table.  While
        the lists are unsized (or strict) the counts are read back -- one host sync -- and, if a camera's pairs did not
        fit, all lists are regrown and the batch binned again: lossless."""
        lib = _capi.load()
        N = self.N
        sync = self.slots[0].needs_sync_sizing() if B else False
        while True:
            geo, views = self._tables(kind)
            mr = _p(stats.max_radii2d) if stats is not None else None  # the forward's statistic, raised by the projection launch
            for i in range(B):
                geo[i].max_radii2d = mr
            for k, (lo, n, s) in enumerate(parts):
                lib.frame_geometry_batch_zero(n, _sub(geo, lo, n), N, _p(mean), _p(qvec), _p(svec), self.W, self.H,
                                              _p(gsh) if k == 0 else None, gsh.numel() if k == 0 else 0,
                                              _p(self._gws) + lo * self._gv_bytes, s)
            if not sync:
                return views
            if len(parts) > 1:
                self._side.synchronize()
            need = max(R.pair_count(c) for c in self._totals[:B].tolist())  # (the sync)
            self._report.clear()
            if need <= self.slots[0].D_cap:
                for s_ in self.slots:
                    s_.sized = True
                return views
            self._regrow(need)

    def _regrow(self, need):
        for s_ in self.slots:
            s_._alloc_pairs(R._cap_for(need), need)  # (raises beyond 2^31 - 1 pairs: no list can hold that frame)
            s_.sized = True
        self._generation += 1  # a pending backward would read freed lists

    # ---- one-forward-one-backward contract and overflow detection ------------------------------------------------
    def _begin_batch(self, B):
        """Every forward starts here.  The batch's backward reads the lists its forward left in the slots, so a
        later render (or a regrown slot) invalidates it: the generation the forward stores is checked in backward."""
        self.check_overflow()
        self._generation += 1

    def _check_generation(self, gen):
        if gen != self._generation:
            raise RuntimeError("gsgen_amd.BatchRenderer: another render() / render_heads() (or a regrown slot) came "
                               "between this batch's forward and its backward -- the lists the backward needs are gone. "
                               "Call backward before the next render, or use one BatchRenderer per batch in flight "
                               "(e.g. for gradient accumulation or an evaluation render in between).")

    def check_overflow(self):
        """No sync.  Reads what the geometry launches of earlier batches reported (every batch, every view -- nothing is
        sampled): a camera whose pairs did not fit -> all lists regrown, PairListOverflow raised (that camera's image was
        NaN, it contributed no gradients: repeat the step); the largest count within 25 % of the capacity -> regrown quietly,
        before anything overflows.  Called by every render; call it yourself after replaying a captured step."""
        rep = self._report
        if rep.any_overflow():
            bad = {i: rep.overflow(i) for i in range(rep.n) if rep.overflow(i)}
            rep.clear()
            old = self.slots[0].D_cap
            need = max(bad.values())
            if need > old:
                self._regrow(need)
            raise PairListOverflow(
                f"gsgen_amd: camera(s) {sorted(bad)} of an earlier batch needed up to {need} (tile, Gaussian) pairs, capacity was "
                f"{old}: their images and T are NaN and they contributed no gradients.  The lists have been regrown to "
                f"{self.slots[0].D_cap}: repeat that step (or use strict=True / ensure_capacity() to rule this out).")
        if self.slots[0].sized:
            last = max(rep.last(i) for i in range(rep.n))
            cap = self.slots[0].D_cap
            if last <= cap and last * 1.25 > cap:
                self._regrow(last)
        return True

    def ensure_capacity(self, B=None):
        """One host sync: did every camera of the last batch fit?  Regrows the lists if not (False: those cameras' images
        are NaN -- render again)."""
        counts = [R.pair_count(c) for c in self._totals[:B].tolist()]
        self._report.clear()
        for s_ in self.slots:
            s_.sized = True
        need = max(counts, default=0)
        if need > self.slots[0].D_cap:
            self._regrow(need)
            return False
        return True

    # ---- the public calls ----------------------------------------------------------------------------------------------
    def _check_params(self, mean, qvec, svec, alpha, col, ncol, stats):
        """shapes against the renderer's N (a renderer reused after densify / prune must be an error, not an out-of-bounds access
        on the device: the launches take raw pointers and self.N -- ADVICE r5)"""
        N = self.N
        for name, t, per in (("mean", mean, 3), ("qvec", qvec, 4), ("svec", svec, 3), ("alpha", alpha, 1), ("colour", col, ncol)):
            if not (isinstance(t, torch.Tensor) and t.dtype == torch.float32 and t.device == mean.device and t.numel() == N * per
                    and (per == 1 or t.shape[0] == N)):
                raise ValueError(f"gsgen_amd.BatchRenderer: {name} must be a float32 tensor of {N} x {per} elements on {mean.device} "
                                 f"(got {tuple(getattr(t, 'shape', ()))}); build a new BatchRenderer after densify / prune")
        if stats is not None:
            for name in ("max_radii2d", "grad_accum", "cnt"):
                t = getattr(stats, name, None)
                if t is not None and (t.numel() < N or t.dtype != torch.float32 or t.device != mean.device or not t.is_contiguous()):
                    raise ValueError(f"gsgen_amd.BatchRenderer: stats.{name} must be a contiguous float32 tensor of >= {N} elements on {mean.device}")

    def _check_batch(self, cam_infos):
        B = len(cam_infos)
        if B > len(self.slots):
            raise ValueError(f"batch of {B} cameras, renderer was sized for {len(self.slots)}")
        for ci in cam_infos:
            if (ci.w, ci.h) != (self.W, self.H):
                raise ValueError("every camera of a batch must have the renderer's (W, H)")
        return B

    def render(self, mean, qvec, svec, alpha, col, cam_infos, c2ws, C=0, bg_rgb=None, thresh=1e-4,
               frustum_radius=6.0, tile_radius=6.0, detach_depth=True, stats=None, sh_basis="auto", sh_l1_bound=None,
               verify_bound=False):
        """-> (rgb [B,H,W,3], T [B,H,W,1]); differentiable wrt mean, qvec, svec, alpha, col.

        cam_infos: B CameraInfo of this renderer's (W, H); c2ws: B poses [3,4] (arrays or tensors).
        col is sh_coeffs [N,3,C*C] for C in 1..4, post-activation rgb [N,3] for C == 0.
        sh_basis (C == 4): "auto" (default) -- the per-splat coefficient bounds S_i = max_c sum_{k>=1} |sh[i][c][k]| are measured
        on the device in front of the launch (one 10-us pass on the render's stream, no host sync) and the kernels route on them
        per list entry and per tile: the tile-local polynomial form of the per-pixel SH basis where 0.25 S_i delta^3 <= 1e-5 - 8.7e-7
        (colours within 1e-5 of the exact kernels'), the exact evaluation elsewhere (include/gsgen_hip.h "the coefficient
        bound").  "exact": the exact kernels only.  sh_l1_bound: a 1-float DEVICE tensor that already holds max_i S_i for THESE
        coefficients (e.g. renderer.sh_l1_bound_device(sh) evaluated once for several batches of one optimiser step) -- skips
        the pass; verify_bound=True checks such a tensor on the device first (debug: one host sync, raises if any row exceeds it).
        """
        if sh_basis not in ("auto", "exact"):
            raise ValueError("sh_basis: 'auto' or 'exact'")
        B = self._check_batch(cam_infos)
        if B == 0:
            z = torch.zeros(0, self.H, self.W, 3, device=self.device)
            return z, z[..., :1]
        self._check_params(mean, qvec, svec, alpha, col, 3 * int(C) * int(C) if int(C) > 0 else 3, stats)
        cams = self._upload(cam_infos, c2ws, frustum_radius, tile_radius)
        self._cis = list(cam_infos)
        self._sh_bound = self._sh_rows = None
        if int(C) == 4 and sh_basis == "auto":
            if sh_l1_bound is not None:
                if not (isinstance(sh_l1_bound, torch.Tensor) and sh_l1_bound.is_cuda and sh_l1_bound.dtype == torch.float32
                        and sh_l1_bound.numel() == 1):
                    raise ValueError("sh_l1_bound: a 1-float CUDA tensor (the bound lives on the device; "
                                     "renderer.sh_l1_bound_device computes it)")
                if verify_bound:
                    R.verify_sh_l1_bound(col, sh_l1_bound)
                self._sh_bound = sh_l1_bound
            else:  # per-splat bounds + their maximum: the launches take the view's bound first, then route per entry / tile
                self._sh_rows = self._measure_bound(col)
                self._sh_bound = self._smax if self._sh_rows is self._rows else None
        # every forward starts with the overflow check -- BEFORE the plan is fetched: a quiet regrow replaces the slots' lists and
        # the plan that points at them (the step it was triggered for used to run on the stale plan)
        self._begin_batch(B)
        self._route_args = self._route_mode() if (int(C) == 4 and self._sh_rows is not None) else (None, 0)
        fast = self._plan("sh" if int(C) > 0 else "rgb", B)
        if fast is not None:  # one C++ autograd node (csrc/torch_batch.cpp): the same launches, a third of the host time
            self._last_parts = [(0, B)]
            va = fast[1]
            va["pixel_size_x"][:B] = 1.0 / self._intr[:B, 0]
            va["pixel_size_y"][:B] = 1.0 / self._intr[:B, 1]
            if int(C) > 0:
                va["route_report"][:B] = self._route_args[0] or 0
                va["no_fallback"][:B] = self._route_args[1]
            out, T = _batch_ext().render(fast[0], mean, qvec, svec, alpha, col, bg_rgb, int(C), float(thresh), bool(detach_depth),
                                         self._sh_bound, self._sh_rows, *self._stats_args(stats))
            return out, T
        return _render_batch.apply(mean, qvec, svec, alpha, col, cams, self, B, int(C), bg_rgb, float(thresh),
                                   bool(detach_depth), stats)

    def _route_mode(self):
        """-> (device address of the crowded-tile report word or None, no_fallback 0 | 1) for the batch about to be enqueued: reads and
        clears what earlier batches reported (plain host memory: no sync; an answer may be a batch or two late -- it is a hint, either
        mode renders every tile correctly)"""
        rep = self._route
        if rep.ptr(0) is None:
            return None, 0
        seen = int(rep._np[0, 0])
        rep._np[0, 0] = 0
        self._route_clean = 0 if seen else self._route_clean + 1
        return rep.ptr(0), 1 if self._route_clean >= 3 else 0

    def _measure_bound(self, col):
        """renderer.sh_row_bounds_device into the renderer's own [N] floats (read by this batch's forward and backward only)"""
        if col.dim() != 3 or col.shape[1] != 3 or col.shape[2] != 16 or col.dtype != torch.float32 or not col.is_contiguous() \
                or col.shape[0] != self.N:
            return R.sh_row_bounds_device(col)  # (other layouts: the checked path)
        # the maximum is a RUNNING one (no 4-byte fill kernel in front of every step's pass: it was a launch of its own in the
        # step's chain): always an upper bound of the current coefficients' -- what the per-view shortcut needs -- and made
        # tight again every 64th batch
        self._bound_tick += 1
        with _on(self.device):
            if self._bound_tick % 64 == 1 and not torch.cuda.is_current_stream_capturing():
                self._smax.zero_()
            _capi.load().sh_l1_bound_rows_running(self.N, col.data_ptr(), 4, self._smax.data_ptr(), self._rows.data_ptr(),
                                                  torch.cuda.current_stream(self.device).cuda_stream)
        return self._rows

    def render_heads(self, mean, qvec, svec, alpha, color, cam_infos, c2ws, bg_rgb=None, thresh=1e-4,
                     frustum_radius=6.0, tile_radius=6.0, detach_depth=True, stats=None, z_var=False, activations=None):
        """-> (rgb [B,H,W,3], depth, opacity, depth2, T [B,H,W,1]) from post-activation colours [N,3]: what
        GaussianSplattingRenderer.forward returns with rgb_only = False, one compositing pass per camera.  Five separate contiguous
        tensors.  bg_rgb (3 or 3 B elements, any broadcastable shape; differentiable): composited inside the forward launch.
        z_var=True: the fourth output is the depth variance depth2 - depth^2 the reference's model returns
        (gs/gaussian_splatting.py:1397), formed -- and differentiated -- inside the launches.
        activations=(svec_act, alpha_act, color_act) -- names of utils/activations.py:36-57: svec / alpha / color are then the model's
        RAW parameters (svec_before_activation ...), activated by one launch inside the batch's autograd node (and differentiated by
        one) instead of three torch kernels and three autograd nodes: host time of a small training step."""
        if activations is not None:
            codes = [ACTIVATION_CODES.get(a) for a in activations]
            if None in codes:
                raise ValueError(f"activations {activations}: known are {sorted(ACTIVATION_CODES)}")
        else:
            codes = [-1, -1, -1]
        B = self._check_batch(cam_infos)
        if B == 0:
            z = torch.zeros(0, self.H, self.W, 3, device=self.device)
            return z, z[..., :1], z[..., :1], z[..., :1], z[..., :1]
        self._check_params(mean, qvec, svec, alpha, color, 3, stats)
        cams = self._upload(cam_infos, c2ws, frustum_radius, tile_radius)
        self._cis = list(cam_infos)
        self._begin_batch(B)  # (before the plan: see render)
        fast = self._plan("rgbd", B)
        if fast is not None:  # one C++ autograd node (csrc/torch_batch.cpp)
            self._last_parts = [(0, B)]
            va = fast[1]
            va["pixel_size_x"][:B] = 1.0 / self._intr[:B, 0]
            va["pixel_size_y"][:B] = 1.0 / self._intr[:B, 1]
            return tuple(_batch_ext().render_heads(fast[0], mean, qvec, svec, alpha, color, bg_rgb, float(thresh),
                                                   bool(detach_depth), *self._stats_args(stats), bool(z_var), *codes))
        if activations is not None:  # (the Python Functions take activated fields: torch's own kernels, the same values)
            svec, alpha, color = (TORCH_ACTIVATIONS[a](x) for a, x in zip(activations, (svec, alpha, color)))
        return _render_batch_heads.apply(mean, qvec, svec, alpha, color, cams, self, B, bg_rgb, float(thresh),
                                         bool(detach_depth), stats, bool(z_var))

    def routing_flags(self, B):
        """uint8 [B, n_tiles]: what the last SH degree-3 batch's polynomial forward decided per tile -- 1 = a quarter of a staged
        batch of the tile's list exceeds the bound for the view's pixel size, the exact kernel rendered it; 0 = the
        polynomial form (empty tiles: 0).  Reports and tests only; reading it synchronises like any tensor read."""
        lib = _capi.load()
        T = self.slots[0].nth * self.slots[0].ntw
        rows = []
        for k, (lo, n) in enumerate(self._last_parts):
            o = lib.sh_batch_workspace_bytes(n)  # (behind the two parameter tables of an n-view batch)
            rows.append(self._bws[k][o:o + n * T].view(n, T))
        return torch.cat(rows)[:B] if len(rows) > 1 else rows[0][:B]
