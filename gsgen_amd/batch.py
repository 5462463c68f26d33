"""Batched-camera caller of the fused frame path.

The reference renders the cameras of a batch strictly one after another in a Python loop and
stacks the per-camera dicts (gs/gaussian_splatting.py:1423-1466).  One 800x800 frame of 100k
Gaussians does not fill an MI355X (2.5k tiles on 256 CUs, each kernel ending on its longest
tiles), so this caller keeps several cameras in flight instead: camera i is enqueued on HIP
stream i % n_streams with its own FrameBuffers, all per-camera constants go up in ONE host to
device copy, and nothing synchronises with the host.  The whole batch is ONE autograd node: its
backward fans out over the same streams, and every camera adds atomically into one set of
parameter gradients (compositing already accumulates; the projection backward runs in its
accumulate form), so there is no per-camera zero-fill of the SH gradient (19 MB at 100k, C=4)
and no B-way gradient sum afterwards.

Every camera of the batch owns a FrameBuffers slot because its backward needs the lists and
projected records of its forward (22 + 12 B x D_cap per slot; 64 cameras at cfg2 ~ 2 GB of the
288 GB).
"""
import ctypes

import numpy as np
import torch

from . import _capi
from . import renderer as R
from .renderer import _p


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


def _tab(addresses):
    """host array of device pointers for the *_batch entry points"""
    return (ctypes.c_void_p * len(addresses))(*addresses)


class _render_batch(torch.autograd.Function):
    """(mean, qvec, svec, alpha, sh|color) -> rgb [B,H,W,3], T [B,H,W,1] for B cameras."""

    @staticmethod
    def forward(ctx, mean, qvec, svec, alpha, col, cams, br, B, C, bg_rgb, thresh, detach_depth, stats):
        mean, qvec, svec = mean.contiguous(), qvec.contiguous(), svec.contiguous()
        alpha, col = alpha.contiguous(), col.contiguous()
        lib = _capi.load()
        H, W, N, dev = br.H, br.W, br.N, mean.device
        fused = br.fused_launch and B > 0
        ctx.gen = br._begin_batch(B)
        if fused:  # the batched forward writes every pixel of out / T (empty tiles included): no fill kernels
            out = torch.empty(B, H, W, 3, device=dev, dtype=torch.float32)
            T = torch.empty(B, H, W, 1, device=dev, dtype=torch.float32)
            return _render_batch._forward_fused(ctx, mean, qvec, svec, alpha, col, cams, br, B, C, bg_rgb, thresh,
                                                detach_depth, stats, out, T)
        out = torch.zeros(B, H, W, 3, device=dev, dtype=torch.float32)
        T = torch.ones(B, H, W, 1, device=dev, dtype=torch.float32)
        cams_p, out_p, T_p = cams.data_ptr(), out.data_ptr(), T.data_ptr()
        cur = br._fork(B, (cams, out, T) + tuple(x for x in (br._sh_bound, br._sh_rows) if x is not None))
        with torch.cuda.device(dev):
            for i in range(B):
                buf, s, ci = br.slots[i], br.streams[i % len(br.streams)].cuda_stream, br._cis[i]
                cam = cams_p + 272 * i  # row i: cam block | topleft at +56 floats | rotation at +58
                psx, psy = 1.0 / ci.fx, 1.0 / ci.fy
                lib.frame_geometry(N, _p(mean), _p(qvec), _p(svec), cam, W, H, buf.D_cap, _p(buf.mean2d),
                                   _p(buf.cov2d), _p(buf.depth), _p(buf.mask), _p(buf.ids), _p(buf.start), _p(buf.end),
                                   _p(buf.total), _p(buf.ws), buf.ws.numel(), s)
                if stats is not None:
                    lib.densify_update(N, _p(buf.cov2d), None, _p(buf.mask), _p(stats.max_radii2d), None, None, s)
                if C > 0:
                    lib.vol_render_sh_routed(N, buf.D_cap, _p(buf.mean2d), _p(buf.cov2d), _p(col), _p(alpha),
                                             _p(buf.start), _p(buf.end), _p(buf.ids), out_p + 12 * H * W * i, cam + 224,
                                             cam + 232, 16, buf.nth, buf.ntw, psx, psy, H, W, C, thresh,
                                             _p(bg_rgb), T_p + 4 * H * W * i, buf.tile_order(), None, 0,
                                             _p(br._sh_bound), _p(br._sh_rows), s)
                else:
                    lib.vol_render_start_end_with_T(N, buf.D_cap, _p(buf.mean2d), _p(buf.cov2d), _p(col), _p(alpha),
                                                    _p(buf.start), _p(buf.end), _p(buf.ids), out_p + 12 * H * W * i,
                                                    cam + 224, 16, buf.nth, buf.ntw, psx, psy, H, W, thresh,
                                                    T_p + 4 * H * W * i, s)
        br._join(B, cur)
        br._end_batch(B)
        ctx.views = ctx.bws = None
        if C == 0 and bg_rgb is not None:
            out = out + T * bg_rgb  # gs/renderer.py:1182
        ctx.save_for_backward(mean, qvec, svec, alpha, col, cams, out, T)
        ctx.bg_shape = tuple(bg_rgb.shape) if (bg_rgb is not None and ctx.needs_input_grad[9]) else None
        ctx.br, ctx.B, ctx.C, ctx.thresh, ctx.detach, ctx.stats = br, B, C, thresh, detach_depth, stats
        ctx.sh_bound, ctx.sh_rows = br._sh_bound, br._sh_rows  # forward and backward of a batch route on the same device values
        ctx.cis = list(br._cis[:B])
        ctx.mark_non_differentiable(T)
        return out, T

    @staticmethod
    def _forward_fused(ctx, mean, qvec, svec, alpha, col, cams, br, B, C, bg_rgb, thresh, detach_depth, stats, out, T):
        """the whole batch on the current stream: one enqueue of the geometry kernels (gridDim.y = cameras), one
        compositing launch.  Nothing is filled between the stages: the projection launch zeroes the step's gradient
        accumulators (the renderer's per-view blocks and the shared block allocated here, gsgen_frame_geometry_batch_zero),
        the compositing launch writes every pixel of out / T."""
        lib = _capi.load()
        H, W, N, dev = br.H, br.W, br.N, mean.device
        out_p, T_p = out.data_ptr(), T.data_ptr()
        s = torch.cuda.current_stream(dev).cuda_stream
        # the slots' buffer addresses, the camera rows and the gradient blocks sit in cached tables (BatchRenderer._tables);
        # only what changes per call is set
        geo, views = br._tables("sh" if C > 0 else "rgb")  # C == 0: post-activation colours
        bg_p = _p(bg_rgb)
        cis = br._cis
        if C > 0:
            for i in range(B):
                ci, v = cis[i], views[i]
                v.pixel_size_x, v.pixel_size_y = 1.0 / ci.fx, 1.0 / ci.fy
                v.T, v.bg_rgb, v.out = T_p + 4 * H * W * i, bg_p, out_p + 12 * H * W * i
        else:
            for i in range(B):
                ci, v = cis[i], views[i]
                v.pixel_size_x, v.pixel_size_y = 1.0 / ci.fx, 1.0 / ci.fy
                v.T, v.out6 = T_p + 4 * H * W * i, out_p + 12 * H * W * i  # read as [H,W,3] by the RGB entry points
        # d L / d alpha [N] | d L / d col, shared by the views: returned by the backward, hence allocated per batch -- and
        # zeroed by this forward's projection launch
        gsh = torch.empty(br._Np + (col.numel() + 3) // 4 * 4, device=dev, dtype=torch.float32)
        nb_sh = br._nb_sh
        with _on(dev):
            lib.frame_geometry_batch_zero(B, geo, N, _p(mean), _p(qvec), _p(svec), W, H, _p(gsh), gsh.numel(),
                                          _p(br._bws) + nb_sh, s)
            br._end_batch(B)
            if stats is not None:
                lib.densify_update_batch(B, N, br._ptr_table("cov2d", B), None, br._mask_table(B), _p(stats.max_radii2d), None,
                                         None, s)
            if C > 0:
                lib.vol_render_sh_batch_routed(B, views, N, _p(col), _p(alpha), 16, br.slots[0].nth, br.slots[0].ntw, H, W, C,
                                               thresh, br.segments, _p(br._sh_bound), _p(br._sh_rows), _p(br._bws), s)
            else:
                lib.vol_render_rgb_batch(B, views, N, _p(col), _p(alpha), 16, br.slots[0].nth, br.slots[0].ntw, H, W,
                                         thresh, _p(br._bws), s)
        if C == 0 and bg_rgb is not None:
            out = out + T * bg_rgb  # gs/renderer.py:1182; `out` (saved below) is what the backward reads as final
            for i in range(B):
                views[i].out6 = out.data_ptr() + 12 * H * W * i
        ctx.views, ctx.gsh = views, gsh
        ctx.save_for_backward(mean, qvec, svec, alpha, col, cams, out, T)
        ctx.bg_shape = tuple(bg_rgb.shape) if (bg_rgb is not None and ctx.needs_input_grad[9]) else None
        ctx.br, ctx.B, ctx.C, ctx.thresh, ctx.detach, ctx.stats = br, B, C, thresh, detach_depth, stats
        ctx.sh_bound, ctx.sh_rows = br._sh_bound, br._sh_rows  # forward and backward of a batch route on the same device values
        ctx.cis = list(br._cis[:B])
        ctx.mark_non_differentiable(T)
        return out, T

    @staticmethod
    def _backward_fused(ctx, grad):
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
        s = torch.cuda.current_stream(dev).cuda_stream
        views = ctx.views
        if C > 0:
            for i in range(B):
                views[i].grad_out = grad_p + 12 * H * W * i
        else:
            for i in range(B):
                views[i].grad_out6 = grad_p + 12 * H * W * i
        with _on(dev):
            if C > 0:
                lib.vol_render_backward_sh_batch_routed(B, views, N, _p(col), _p(alpha), _p(g_col), _p(g_alpha), 16,
                                                        br.slots[0].nth, br.slots[0].ntw, H, W, C, thresh, br.segments,
                                                        _p(ctx.sh_bound), _p(ctx.sh_rows), _p(br._bws), s)
            else:
                lib.vol_render_rgb_backward_batch(B, views, N, _p(col), _p(alpha), _p(g_col), _p(g_alpha), 16,
                                                  br.slots[0].nth, br.slots[0].ntw, H, W, thresh, _p(br._bws), s)
            lib.project_gaussians_backward_batch(B, N, _p(mean), _p(qvec), _p(svec), br._ptr_table("cam", B),
                                                 int(ctx.detach), br._mask_table(B), br._ptr_table("g_mean2d", B),
                                                 br._ptr_table("g_cov2d", B), None, _p(g_mean), _p(g_qvec), _p(g_svec), s)
            if stats is not None:
                lib.densify_update_batch(B, N, None, br._ptr_table("g_mean2d", B), br._mask_table(B), None,
                                         _p(stats.grad_accum), _p(stats.cnt), s)
        return (g_mean, g_qvec, g_svec, g_alpha, g_col, None, None, None, None, _bg_grad(ctx, grad, T), None, None, None)

    @staticmethod
    def backward(ctx, grad, _gT):
        ctx.br._check_generation(ctx.gen)
        if ctx.views is not None:
            return _render_batch._backward_fused(ctx, grad)
        mean, qvec, svec, alpha, col, cams, out, T = ctx.saved_tensors
        br, B, C, thresh, stats = ctx.br, ctx.B, ctx.C, ctx.thresh, ctx.stats
        lib = _capi.load()
        H, W, N, dev = br.H, br.W, br.N, mean.device
        grad = grad.contiguous()
        g2d = torch.zeros(B, 6 * N, device=dev, dtype=torch.float32)  # per camera: mean2d | cov2d
        g3d = torch.zeros(11 * N, device=dev, dtype=torch.float32)   # shared: mean | qvec | svec | alpha
        g_mean, g_qvec = g3d[:3 * N].view(N, 3), g3d[3 * N:7 * N].view(N, 4)
        g_svec, g_alpha = g3d[7 * N:10 * N].view(N, 3), g3d[10 * N:]
        g_col = torch.zeros_like(col)
        cams_p, out_p, grad_p, g2d_p = cams.data_ptr(), out.data_ptr(), grad.data_ptr(), g2d.data_ptr()
        cur = br._fork(B, (grad, g2d, g3d, g_col) + tuple(x for x in (ctx.sh_bound, ctx.sh_rows) if x is not None))
        with torch.cuda.device(dev):
            for i in range(B):
                buf, s, ci = br.slots[i], br.streams[i % len(br.streams)].cuda_stream, ctx.cis[i]
                cam = cams_p + 272 * i
                g_mean2d = g2d_p + 24 * N * i
                g_cov2d = g_mean2d + 8 * N
                psx, psy = 1.0 / ci.fx, 1.0 / ci.fy
                if C > 0:
                    lib.vol_render_backward_sh_routed(N, buf.D_cap, _p(buf.mean2d), _p(buf.cov2d), _p(col), _p(alpha),
                                                      _p(buf.start), _p(buf.end), _p(buf.ids), out_p + 12 * H * W * i,
                                                      g_mean2d, g_cov2d, _p(g_col), _p(g_alpha),
                                                      grad_p + 12 * H * W * i, cam + 224, cam + 232, 16, buf.nth,
                                                      buf.ntw, psx, psy, H, W, C, thresh, None, buf.tile_order(), None, 0,
                                                      _p(ctx.sh_bound), _p(ctx.sh_rows), s)
                else:
                    lib.vol_render_backward_start_end(N, buf.D_cap, _p(buf.mean2d), _p(buf.cov2d), _p(col), _p(alpha),
                                                      _p(buf.start), _p(buf.end), _p(buf.ids), out_p + 12 * H * W * i,
                                                      g_mean2d, g_cov2d, _p(g_col), _p(g_alpha),
                                                      grad_p + 12 * H * W * i, cam + 224, 16, buf.nth, buf.ntw, psx,
                                                      psy, H, W, thresh, s)
                lib.project_gaussians_backward_accum(N, _p(mean), _p(qvec), _p(svec), cam, int(ctx.detach),
                                                     _p(buf.mask), g_mean2d, g_cov2d, None, _p(g_mean),
                                                     _p(g_qvec), _p(g_svec), s)
                if stats is not None:
                    lib.densify_update(N, None, g_mean2d, _p(buf.mask), None, _p(stats.grad_accum),
                                       _p(stats.cnt), s)
        br._join(B, cur)
        return (g_mean, g_qvec, g_svec, g_alpha, g_col, None, None, None, None, _bg_grad(ctx, grad, T), None, None, None)


class _render_batch_heads(torch.autograd.Function):
    """(mean, qvec, svec, alpha, color) -> rgb [B,H,W,3], depth, opacity, depth^2, T [B,H,W,1]: the default
    (rgb_only = False) output set of GaussianSplattingRenderer.forward (gs/gaussian_splatting.py:1304-1416,
    :1423-1466), one fused compositing pass per camera (gsgen_vol_render_rgbd) instead of four."""

    @staticmethod
    def forward(ctx, mean, qvec, svec, alpha, col, cams, br, B, bg_rgb, thresh, detach_depth, stats):
        mean, qvec, svec = mean.contiguous(), qvec.contiguous(), svec.contiguous()
        alpha, col = alpha.contiguous(), col.contiguous()
        lib = _capi.load()
        H, W, N, dev = br.H, br.W, br.N, mean.device
        ctx.views = ctx.gsh = None
        ctx.gen = br._begin_batch(B)
        if br.fused_launch and B > 0:  # one enqueue per stage for the whole batch, on the current stream
            # (every pixel of out6 / T is written by the batched forward, the gradient accumulators are zeroed by the
            # projection launch: no fill kernels)
            out6 = torch.empty(B, H, W, 6, device=dev, dtype=torch.float32)
            T = torch.empty(B, H, W, 1, device=dev, dtype=torch.float32)
            out_p, T_p = out6.data_ptr(), T.data_ptr()
            s = torch.cuda.current_stream(dev).cuda_stream
            geo, views = br._tables("rgbd")
            cis = br._cis
            for i in range(B):
                ci, v = cis[i], views[i]
                v.pixel_size_x, v.pixel_size_y = 1.0 / ci.fx, 1.0 / ci.fy
                v.out6, v.T = out_p + 24 * H * W * i, T_p + 4 * H * W * i
            gsh = torch.empty(br._Np, device=dev, dtype=torch.float32)  # d L / d alpha, shared by the views
            with _on(dev):
                lib.frame_geometry_batch_zero(B, geo, N, _p(mean), _p(qvec), _p(svec), W, H, _p(gsh), gsh.numel(),
                                              _p(br._bws) + br._nb_sh, s)
                br._end_batch(B)
                if stats is not None:
                    lib.densify_update_batch(B, N, br._ptr_table("cov2d", B), None, br._mask_table(B),
                                             _p(stats.max_radii2d), None, None, s)
                lib.vol_render_rgbd_batch(B, views, N, _p(col), _p(alpha), 16, br.slots[0].nth, br.slots[0].ntw, H, W,
                                          thresh, _p(br._bws), s)
            ctx.views, ctx.gsh = views, gsh
        else:
            out6 = torch.zeros(B, H, W, 6, device=dev, dtype=torch.float32)
            T = torch.ones(B, H, W, 1, device=dev, dtype=torch.float32)
            _render_batch_heads._forward_streams(br, B, lib, mean, qvec, svec, alpha, col, cams, out6, T, thresh, stats)
        if bg_rgb is not None:
            out6[..., :3] += T * bg_rgb  # gs/renderer.py:1182
        ctx.save_for_backward(mean, qvec, svec, alpha, col, cams, out6, T)
        ctx.bg_shape = tuple(bg_rgb.shape) if (bg_rgb is not None and ctx.needs_input_grad[8]) else None
        ctx.br, ctx.B, ctx.thresh, ctx.detach, ctx.stats = br, B, thresh, detach_depth, stats
        ctx.cis = list(br._cis[:B])
        ctx.mark_non_differentiable(T)
        return out6[..., :3], out6[..., 3:4], out6[..., 4:5], out6[..., 5:6], T

    @staticmethod
    def _forward_streams(br, B, lib, mean, qvec, svec, alpha, col, cams, out6, T, thresh, stats):
        """one chain per camera on the side streams"""
        H, W, N, dev = br.H, br.W, br.N, mean.device
        cur = br._fork(B, (cams, out6, T))
        cams_p, out_p, T_p = cams.data_ptr(), out6.data_ptr(), T.data_ptr()
        with torch.cuda.device(dev):
            for i in range(B):
                buf, s, ci = br.slots[i], br.streams[i % len(br.streams)].cuda_stream, br._cis[i]
                cam = cams_p + 272 * i
                lib.frame_geometry(N, _p(mean), _p(qvec), _p(svec), cam, W, H, buf.D_cap, _p(buf.mean2d),
                                   _p(buf.cov2d), _p(buf.depth), _p(buf.mask), _p(buf.ids), _p(buf.start), _p(buf.end),
                                   _p(buf.total), _p(buf.ws), buf.ws.numel(), s)
                if stats is not None:
                    lib.densify_update(N, _p(buf.cov2d), None, _p(buf.mask), _p(stats.max_radii2d), None, None, s)
                lib.vol_render_rgbd(N, buf.D_cap, _p(buf.mean2d), _p(buf.cov2d), _p(col), _p(buf.depth), _p(alpha),
                                    _p(buf.start), _p(buf.end), _p(buf.ids), out_p + 24 * H * W * i, cam + 224, 16,
                                    buf.nth, buf.ntw, 1.0 / ci.fx, 1.0 / ci.fy, H, W, thresh, T_p + 4 * H * W * i,
                                    buf.tile_order(), s)
        br._join(B, cur)
        br._end_batch(B)

    @staticmethod
    def _backward_fused(ctx, g_rgb, g_depth, g_opac, g_z2):
        """one compositing launch that reads the four head gradients in place (no [B,H,W,6] image is assembled: that
        concatenation was 8 % of a step at 8 x 800^2), one projection launch that forms d L / d depth = g3 + 2 depth g5 and
        sums the colour gradient over the views itself (gsgen_project_gaussians_backward_batch_heads)"""
        mean, qvec, svec, alpha, col, cams, out6, T = ctx.saved_tensors
        br, B, thresh, stats = ctx.br, ctx.B, ctx.thresh, ctx.stats
        lib = _capi.load()
        H, W, N, dev = br.H, br.W, br.N, mean.device
        parts = [g.contiguous() if g is not None else None for g in (g_rgb, g_depth, g_opac, g_z2)]
        gsh, ctx.gsh = ctx.gsh, None
        if gsh is None:  # a second backward through the same graph (retain_graph): the accumulators were handed out
            gsh = torch.zeros(br._Np, device=dev, dtype=torch.float32)
            br._g2d[:B].zero_()
        g_alpha = gsh[:N]
        g3d = torch.empty(13 * N, device=dev, dtype=torch.float32)  # mean | qvec | svec | colour: overwritten
        g_mean, g_qvec = g3d[:3 * N].view(N, 3), g3d[3 * N:7 * N].view(N, 4)
        g_svec, g_col = g3d[7 * N:10 * N].view(N, 3), g3d[10 * N:].view(N, 3)
        s = torch.cuda.current_stream(dev).cuda_stream
        views = ctx.views
        pp = [x.data_ptr() if x is not None else None for x in parts]
        for i in range(B):
            v = views[i]
            v.grad_out6 = None
            v.grad_rgb = pp[0] + 12 * H * W * i if pp[0] is not None else None
            v.grad_depth = pp[1] + 4 * H * W * i if pp[1] is not None else None
            v.grad_opacity = pp[2] + 4 * H * W * i if pp[2] is not None else None
            v.grad_depth2 = pp[3] + 4 * H * W * i if pp[3] is not None else None
        with _on(dev):
            lib.vol_render_rgbd_backward_batch(B, views, N, _p(col), _p(alpha), _p(g_alpha), 16, br.slots[0].nth,
                                               br.slots[0].ntw, H, W, thresh, _p(br._bws), s)
            lib.project_gaussians_backward_batch_heads(B, N, _p(mean), _p(qvec), _p(svec), br._ptr_table("cam", B),
                                                       int(ctx.detach), br._mask_table(B), br._ptr_table("g_mean2d", B),
                                                       br._ptr_table("g_cov2d", B), br._ptr_table("g_chan6", B),
                                                       br._ptr_table("depth", B), _p(g_mean), _p(g_qvec), _p(g_svec),
                                                       _p(g_col), s)
            if stats is not None:
                lib.densify_update_batch(B, N, None, br._ptr_table("g_mean2d", B), br._mask_table(B), None,
                                         _p(stats.grad_accum), _p(stats.cnt), s)
        return (g_mean, g_qvec, g_svec, g_alpha, g_col, None, None, None, _bg_grad(ctx, g_rgb, T), None, None, None)

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_opac, g_z2, _gT):
        ctx.br._check_generation(ctx.gen)
        if ctx.views is not None:
            return _render_batch_heads._backward_fused(ctx, g_rgb, g_depth, g_opac, g_z2)
        mean, qvec, svec, alpha, col, cams, out6, T = ctx.saved_tensors
        br, B, thresh, stats = ctx.br, ctx.B, ctx.thresh, ctx.stats
        lib = _capi.load()
        H, W, N, dev = br.H, br.W, br.N, mean.device
        parts = [g.contiguous() if g is not None else None for g in (g_rgb, g_depth, g_opac, g_z2)]
        z = lambda g, c: g if g is not None else torch.zeros(B, H, W, c, device=dev)  # noqa: E731
        go6 = torch.cat([z(parts[0], 3), z(parts[1], 1), z(parts[2], 1), z(parts[3], 1)], dim=-1).contiguous()
        gbuf = torch.zeros(2 * B * 6 * N, device=dev, dtype=torch.float32)  # one fill for both
        g2d = gbuf[:B * 6 * N].view(B, 6 * N)      # per camera: mean2d | cov2d
        gch = gbuf[B * 6 * N:].view(B, N, 6)       # per camera: rgb | depth | opacity | depth^2
        gdp = torch.empty(B, N, device=dev, dtype=torch.float32)       # per camera: d L / d (view-space depth)
        g3d = torch.zeros(11 * N, device=dev, dtype=torch.float32)     # shared: mean | qvec | svec | alpha
        g_mean, g_qvec = g3d[:3 * N].view(N, 3), g3d[3 * N:7 * N].view(N, 4)
        g_svec, g_alpha = g3d[7 * N:10 * N].view(N, 3), g3d[10 * N:]
        cams_p, out_p, g2d_p = cams.data_ptr(), out6.data_ptr(), g2d.data_ptr()
        go_p = go6.data_ptr()
        cur = br._fork(B, (go6, g2d, gch, gdp, g3d))
        with torch.cuda.device(dev):
            for i in range(B):
                buf, stream, ci = br.slots[i], br.streams[i % len(br.streams)], ctx.cis[i]
                s = stream.cuda_stream
                cam = cams_p + 272 * i
                g_mean2d = g2d_p + 24 * N * i
                g_cov2d = g_mean2d + 8 * N
                lib.vol_render_rgbd_backward(N, buf.D_cap, _p(buf.mean2d), _p(buf.cov2d), _p(col), _p(buf.depth), _p(alpha),
                                             _p(buf.start), _p(buf.end), _p(buf.ids), out_p + 24 * H * W * i, g_mean2d,
                                             g_cov2d, _p(gch[i]), _p(g_alpha), go_p + 24 * H * W * i, cam + 224, 16,
                                             buf.nth, buf.ntw, 1.0 / ci.fx, 1.0 / ci.fy, H, W, thresh, buf.tile_order(), s)
                with torch.cuda.stream(stream):  # the depth head and the depth^2 head both feed the depth
                    torch.addcmul(gch[i, :, 3], buf.depth.view(-1), gch[i, :, 5], value=2.0, out=gdp[i])
                lib.project_gaussians_backward_accum(N, _p(mean), _p(qvec), _p(svec), cam, int(ctx.detach),
                                                     _p(buf.mask), g_mean2d, g_cov2d, _p(gdp[i]), _p(g_mean),
                                                     _p(g_qvec), _p(g_svec), s)
                if stats is not None:
                    lib.densify_update(N, None, g_mean2d, _p(buf.mask), None, _p(stats.grad_accum),
                                       _p(stats.cnt), s)
        br._join(B, cur)
        g_col = gch[:, :, :3].sum(0)
        return (g_mean, g_qvec, g_svec, g_alpha, g_col, None, None, None, _bg_grad(ctx, g_rgb, T), None, None, None)


class BatchRenderer:
    """Renders [B] cameras of one (W, H) shape for a fixed Gaussian count N."""

    def __init__(self, N, W, H, device, max_batch, n_streams=3, D_cap=None, fused_launch=True, segments=1):
        """fused_launch: an SH batch is ONE enqueue per stage on the current stream -- geometry, compositing
        forward, compositing backward and projection backward each launch once for all cameras
        (gsgen_frame_geometry_batch, gsgen_vol_render_sh_batch, ..._backward_sh_batch,
        gsgen_project_gaussians_backward_batch) -- instead of one chain per camera spread over
        `n_streams` side streams; post-activation colours (C = 0: gsgen_vol_render_rgb_batch) and render_heads
        (gsgen_vol_render_rgbd_batch / _backward_batch) likewise.
        cfg2: 2850 vs 2710 renders/s (profiles/r01_notes.md).  segments: backward workgroups per tile
        (FrameBuffers), fused launches only."""
        self.N, self.W, self.H, self.device = N, W, H, torch.device(device)
        self.fused_launch, self.segments = bool(fused_launch), int(segments) if fused_launch else 1
        # the slots' pair counters live in one tensor: one copy brings a batch's counts to the host (overflow detection
        # without a sync, see FrameBuffers.check_overflow)
        self._totals = torch.zeros(max_batch, device=device, dtype=torch.int32)
        self._monitor = R.PairCountMonitor(max_batch)
        # pair counts follow every `monitor_every`-th batch to the host (an async copy + an event each: ~25 us of host time;
        # an overflowing scene overflows in the following batches too, so sampling delays the report by a few batches)
        self.monitor_every, self._tick = 4, 0
        self._generation = 0
        self._sh_bound = self._sh_rows = None
        self._table_cache = {}
        # the slots' depth buffers are the rows of one matrix (the heads' backward reads depths[:B] as one tensor)
        self._depths = torch.empty(max_batch, N, device=device, dtype=torch.float32)
        self.slots = [R.FrameBuffers(N, W, H, device, D_cap=D_cap, segments=self.segments, total=self._totals[i:i + 1],
                                     depth=self._depths[i].view(N, 1))
                      for i in range(max_batch)]
        self._ptr_tabs = {}
        self.streams = [torch.cuda.Stream(device=device) for _ in range(max(1, n_streams))]
        # Buffers a step needs and nobody else sees live as long as the renderer (round 4: nothing is allocated or filled per
        # step except what is handed to the caller): the camera rows of the batch in flight, the kernel-parameter tables of its
        # launches, and the per-view gradient accumulators mean2d (2) | cov2d (4) | channels (6) x Np -- zeroed by the
        # projection launch of the step's forward (gsgen_frame_geometry_batch_zero).  Only one batch per renderer is between
        # forward and backward (_check_generation), so one set is enough.
        lib = _capi.load()
        self._Np = (N + 3) // 4 * 4
        self._cams = torch.empty(max_batch, 68, device=device, dtype=torch.float32)
        nth_, ntw_ = R.n_tiles(H, W)
        self._nb_sh = lib.sh_batch_workspace_bytes_routed(max_batch, nth_ * ntw_)  # parameter tables + per-tile routing flags
        self._bws = torch.empty(self._nb_sh + lib.frame_batch_workspace_bytes(max_batch), device=device, dtype=torch.uint8)
        self._g2d = torch.empty(max_batch, 12 * self._Np, device=device, dtype=torch.float32)
        self._rows = torch.zeros(N, device=device, dtype=torch.float32)  # per-splat bounds of the batch in flight (gsgen_sh_l1_bound_rows)
        self._smax = torch.zeros(1, device=device, dtype=torch.float32)  # ... and their maximum (a scene within a view's bound skips the per-entry tests)
        # per camera: cam block (56) | topleft (2) | rotation rows (9) | pad -> 68 floats, packed on the host and sent
        # through kernel arguments (gsgen_upload_small): no pinned ring, no copy event, the host never waits
        self._host = np.zeros((max_batch, 68), np.float32)
        self._poses = np.zeros((max_batch, 12), np.float32)
        self._intr = np.zeros((max_batch, 8), np.float64)
        self._cis = []

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
        for i, ci in enumerate(cam_infos):
            intr[i] = (ci.fx, ci.fy, ci.cx, ci.cy, ci.w, ci.h, ci.near_plane, ci.far_plane)
        lib = _capi.load()
        lib.pack_camera_blocks(B, poses.ctypes.data, 12, intr.ctypes.data, frustum_radius, tile_radius, h.ctypes.data)
        # the renderer's own device block: its rows are read by the batch's backward, and no other batch of this renderer
        # may come between a forward and its backward (_check_generation)
        lib.upload_small(_p(self._cams), h.ctypes.data, B * 272, torch.cuda.current_stream(self.device).cuda_stream)
        return self._cams[:B]

    def _mask_table(self, B):
        """void*[B] of the slots' visibility masks (they never move): built once per batch size"""
        t = self._ptr_tabs.get(B)
        if t is None:
            import ctypes
            t = self._ptr_tabs[B] = (ctypes.c_void_p * B)(*[_p(self.slots[i].mask) for i in range(B)])
        return t

    def _ptr_table(self, kind, B):
        """void*[B] of per-view addresses that never move: the camera rows, the slots' cov2d / depth buffers, the per-view
        gradient blocks (built once per kind and batch size)"""
        key = (kind, B)
        t = self._ptr_tabs.get(key)
        if t is None:
            import ctypes
            g0, st = self._g2d.data_ptr(), 4 * 12 * self._Np
            if kind == "cam":
                a = [self._cams.data_ptr() + 272 * i for i in range(B)]
            elif kind == "cov2d":
                a = [_p(self.slots[i].cov2d) for i in range(B)]
            elif kind == "depth":
                a = [_p(self.slots[i].depth) for i in range(B)]
            elif kind == "g_mean2d":
                a = [g0 + st * i for i in range(B)]
            elif kind == "g_cov2d":
                a = [g0 + st * i + 4 * 2 * self._Np for i in range(B)]
            elif kind == "g_chan6":
                a = [g0 + st * i + 4 * 6 * self._Np for i in range(B)]
            else:
                raise KeyError(kind)
            t = self._ptr_tabs[key] = (ctypes.c_void_p * B)(*a)
        return t

    def _tables(self, kind):
        """(GeometryView[max_batch], ShView | RgbdView[max_batch]) with every per-slot buffer address filled in; rebuilt
        when a slot's pair list is regrown.  One set per kind ("sh", "rgb", "rgbd"): a batch's backward reads the
        view table its forward filled, and only one batch per BatchRenderer is between forward and backward
        (_check_generation)."""
        caps = tuple(s.D_cap for s in self.slots)
        hit = self._table_cache.get(kind)
        if hit is not None and hit[0] == caps:
            return hit[1], hit[2]
        n = len(self.slots)
        geo = (_capi.GeometryView * n)()
        views = ((_capi.ShView if kind == "sh" else _capi.RgbdView) * n)()
        cams_p, g0, st = self._cams.data_ptr(), self._g2d.data_ptr(), 4 * 12 * self._Np
        for i, buf in enumerate(self.slots):
            g, v = geo[i], views[i]
            cam = cams_p + 272 * i  # row i: cam block | topleft at +56 floats | rotation at +58
            g.cam = cam
            g.mean2d, g.cov2d, g.depth, g.mask = _p(buf.mean2d), _p(buf.cov2d), _p(buf.depth), _p(buf.mask)
            g.gaussian_ids, g.start, g.end, g.total = _p(buf.ids), _p(buf.start), _p(buf.end), _p(buf.total)
            g.workspace, g.workspace_bytes, g.D_cap = _p(buf.ws), buf.ws.numel(), buf.D_cap
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
                    g.zero_grad_chan6 = v.grad_chan6 = g0 + st * i + 4 * 6 * self._Np
        self._table_cache[kind] = (caps, geo, views)
        return geo, views

    # ---- one-forward-one-backward contract and overflow detection ------------------------------------------------
    def _begin_batch(self, B):
        """Every forward starts here.  The batch's backward reads the lists its forward left in the slots, so a
        later render (or a regrown slot) invalidates it: the generation it returns is checked in backward."""
        self.check_overflow()
        self._generation += 1
        return self._generation

    def _end_batch(self, B):
        """behind the geometry enqueue: the batch's pair counts follow it to the host (one async copy, one event, into
        the monitor's ring: no batch's counts are ever dropped unread)"""
        self._tick += 1
        if self._tick % self.monitor_every == 1 or self.monitor_every <= 1:
            self._monitor.record(self._totals, B, torch.cuda.current_stream(self.device))

    def _check_generation(self, gen):
        if gen != self._generation:
            raise RuntimeError("gsgen_amd.BatchRenderer: another render() / render_heads() (or a regrown slot) came "
                               "between this batch's forward and its backward -- the lists the backward needs are gone. "
                               "Call backward before the next render, or use one BatchRenderer per batch in flight "
                               "(e.g. for gradient accumulation or an evaluation render in between).")

    def check_overflow(self):
        """No sync (unless the host is more than PairCountMonitor.depth batches ahead): if pair counts of earlier batches
        have reached the host and one exceeded its slot's capacity, grow that slot and warn (that camera was rendered as
        background only, with zero gradients)."""
        ok = True
        worst = {}
        if not self._monitor.has_news():
            return ok
        for counts in self._monitor.drain():
            for i, need in enumerate(counts):
                worst[i] = max(worst.get(i, 0), need)
        for i, need in worst.items():
            s = self.slots[i]
            if need > s.D_cap:
                import warnings
                old = s.D_cap
                s._alloc_pairs(int(need * 1.25) + 1024)
                self._generation += 1
                warnings.warn(f"gsgen_amd: camera {i} of an earlier batch needed {need} (tile, Gaussian) pairs, "
                              f"capacity was {old}: it was rendered as BACKGROUND ONLY with zero gradients.  The slot has "
                              f"been regrown to {s.D_cap}; call BatchRenderer.ensure_capacity() after a render to catch "
                              f"this synchronously.", RuntimeWarning, stacklevel=3)
                ok = False
        return ok

    def _fork(self, B, tensors):
        """side streams wait for the current stream; `tensors` (allocated on the current stream)
        are about to be used on them"""
        cur = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(cur)
        for st in self.streams[:min(B, len(self.streams))]:
            st.wait_event(ready)
            for t in tensors:
                t.record_stream(st)
        return cur

    def _join(self, B, cur):
        for st in self.streams[:min(B, len(self.streams))]:
            cur.wait_stream(st)

    def render(self, mean, qvec, svec, alpha, col, cam_infos, c2ws, C=0, bg_rgb=None, thresh=1e-4,
               frustum_radius=6.0, tile_radius=6.0, detach_depth=True, stats=None, sh_basis="auto", sh_l1_bound=None,
               verify_bound=False):
        """-> (rgb [B,H,W,3], T [B,H,W,1]); differentiable wrt mean, qvec, svec, alpha, col.

        cam_infos: B CameraInfo of this renderer's (W, H); c2ws: B poses [3,4] (arrays or tensors).
        col is sh_coeffs [N,3,C*C] for C in 1..4, post-activation rgb [N,3] for C == 0.
        sh_basis (C == 4): "auto" (default) -- the coefficient bound S = max_i max_c sum_{k>=1} |sh[i][c][k]| is measured on the
        device in front of the launch (one 5-us pass on the render's stream, no host sync) and the kernels route on it per view:
        the tile-local polynomial form of the per-pixel SH basis where 0.25 S 0.7 delta^3 <= 1e-5, the exact kernel elsewhere
        (include/gsgen_hip.h "the coefficient bound"; images within 1e-5 of the exact kernels, +20 % renders/s at 8 x 800^2).
        "exact": the exact kernels only.  sh_l1_bound: a 1-float DEVICE tensor that already holds S for THESE coefficients
        (e.g. renderer.sh_l1_bound_device(sh) evaluated once for several batches of one optimiser step) -- skips the pass; verify_bound=True
        checks such a tensor on the device first (debug: one host sync, raises if any row exceeds it).
        """
        if sh_basis not in ("auto", "exact"):
            raise ValueError("sh_basis: 'auto' or 'exact'")
        B = len(cam_infos)
        if B > len(self.slots):
            raise ValueError(f"batch of {B} cameras, renderer was sized for {len(self.slots)}")
        for ci in cam_infos:
            if (ci.w, ci.h) != (self.W, self.H):
                raise ValueError("every camera of a batch must have the renderer's (W, H)")
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
        return _render_batch.apply(mean, qvec, svec, alpha, col, cams, self, B, int(C), bg_rgb, float(thresh),
                                   bool(detach_depth), stats)

    def _measure_bound(self, col):
        """renderer.sh_row_bounds_device into the renderer's own [N] floats (read by this batch's forward and backward only)"""
        if col.dim() != 3 or col.shape[1] != 3 or col.shape[2] != 16 or col.dtype != torch.float32 or not col.is_contiguous() \
                or col.shape[0] != self.N:
            return R.sh_row_bounds_device(col)  # (other layouts: the checked path)
        with _on(self.device):
            _capi.load().sh_l1_bound_rows(self.N, col.data_ptr(), 4, self._smax.data_ptr(), self._rows.data_ptr(),
                                          torch.cuda.current_stream(self.device).cuda_stream)
        return self._rows

    def render_heads(self, mean, qvec, svec, alpha, color, cam_infos, c2ws, bg_rgb=None, thresh=1e-4,
                     frustum_radius=6.0, tile_radius=6.0, detach_depth=True, stats=None):
        """-> (rgb [B,H,W,3], depth, opacity, depth2, T [B,H,W,1]) from post-activation colours [N,3]: what
        GaussianSplattingRenderer.forward returns with rgb_only = False, one compositing pass per camera."""
        B = len(cam_infos)
        if B > len(self.slots):
            raise ValueError(f"batch of {B} cameras, renderer was sized for {len(self.slots)}")
        for ci in cam_infos:
            if (ci.w, ci.h) != (self.W, self.H):
                raise ValueError("every camera of a batch must have the renderer's (W, H)")
        cams = self._upload(cam_infos, c2ws, frustum_radius, tile_radius)
        self._cis = list(cam_infos)
        return _render_batch_heads.apply(mean, qvec, svec, alpha, color, cams, self, B, bg_rgb, float(thresh),
                                         bool(detach_depth), stats)

    def routing_flags(self, B):
        """uint8 [B, n_tiles] (a view of the batch workspace): what the last SH degree-3 batch's polynomial forward decided per
        tile -- 1 = a splat the tile staged exceeds the bound for the view's pixel size, the exact kernel rendered it; 0 = the
        polynomial form (empty tiles: 0).  Reports and tests only; reading it synchronises like any tensor read."""
        lib = _capi.load()
        T = self.slots[0].nth * self.slots[0].ntw
        o = lib.sh_batch_workspace_bytes(B)  # (behind the two parameter tables of a B-view batch)
        return self._bws[o:o + B * T].view(B, T)

    def ensure_capacity(self, B=None):
        """One host sync: grows any slot whose pair list overflowed in the last batch.  Returns
        False if a slot had to grow (that camera's image was rendered empty: render again)."""
        self._monitor.clear()
        ok = True
        for s in self.slots[:B]:
            ok = s.ensure_capacity() and ok
        if not ok:
            self._generation += 1  # a pending backward would read freed lists
        return ok
