"""Stand-ins for two of the reference's compositing autograd.Functions (gs/renderer.py:999-1283), used ONLY by
tests/test_gpu_api.py::test_reference_call_sequence_render_one to replay render_one's call sequence on the GPU box,
where /root/reference (and with it the real classes) does not exist.  Same argument lists, return shapes, saved
tensors and backward outputs; the real classes are exercised on this package's mirror, imported from the reference,
by tests/test_reference_python_on_mirror.py.  Test infrastructure -- not part of the product."""
import torch

from gsgen_amd import _gs as _backend


class _render_with_T(torch.autograd.Function):
    """gs/renderer.py:1135-1283"""

    @staticmethod
    def forward(ctx, mean, cov, scalar, alpha, start, end, gaussian_ids, topleft, tile_size, n_tiles_h,
                n_tiles_w, pixel_size_x, pixel_size_y, H, W, thresh, bg):
        out = torch.zeros([H, W, 3], dtype=torch.float32, device=mean.device)
        T = torch.ones_like(out[..., :1])
        _backend.tile_based_vol_rendering_start_end_with_T(
            mean, cov, scalar, alpha, start, end, gaussian_ids, out, topleft, tile_size, n_tiles_h,
            n_tiles_w, pixel_size_x, pixel_size_y, H, W, thresh, T)
        out = out + T * bg
        ctx.save_for_backward(mean, cov, scalar, alpha, start, end, gaussian_ids, out, topleft, T)
        ctx.const = [tile_size, n_tiles_h, n_tiles_w, pixel_size_x, pixel_size_y, H, W, thresh]
        return out

    @staticmethod
    def backward(ctx, grad):
        mean, cov, color, alpha, start, end, gaussian_ids, out, topleft, T = ctx.saved_tensors
        grad = grad.contiguous()
        grad_mean = torch.zeros_like(mean)
        grad_cov = torch.zeros_like(cov)
        grad_color = torch.zeros_like(color)
        grad_alpha = torch.zeros_like(alpha)
        tile_size, n_tiles_h, n_tiles_w, pixel_size_x, pixel_size_y, H, W, thresh = ctx.const
        _backend.tile_based_vol_rendering_backward_start_end(
            mean, cov, color, alpha, start, end, gaussian_ids, out, grad_mean, grad_cov, grad_color,
            grad_alpha, grad, topleft, tile_size, n_tiles_h, n_tiles_w, pixel_size_x, pixel_size_y, H,
            W, thresh)
        return (grad_mean, grad_cov, grad_color, grad_alpha) + (None,) * 12 + (
            torch.nan_to_num(grad * T),)


class _render_scalar(torch.autograd.Function):
    """gs/renderer.py:999-1132.  T is the caller's [H,W,1] tensor, overwritten in place."""

    @staticmethod
    def forward(ctx, mean, cov, scalar, alpha, start, end, gaussian_ids, topleft, tile_size, n_tiles_h,
                n_tiles_w, pixel_size_x, pixel_size_y, H, W, thresh, T):
        out = torch.zeros([H * W], dtype=torch.float32, device=mean.device)
        scalar = scalar.contiguous()
        _backend.tile_based_vol_rendering_scalar(
            mean, cov, scalar, alpha, start, end, gaussian_ids, out, topleft, tile_size, n_tiles_h,
            n_tiles_w, pixel_size_x, pixel_size_y, H, W, thresh, T)
        ctx.save_for_backward(mean, cov, scalar, alpha, start, end, gaussian_ids, out, topleft)
        ctx.const = [tile_size, n_tiles_h, n_tiles_w, pixel_size_x, pixel_size_y, H, W, thresh]
        return out

    @staticmethod
    def backward(ctx, grad):
        mean, cov, scalar, alpha, start, end, gaussian_ids, out, topleft = ctx.saved_tensors
        grad = grad.contiguous()
        grad_mean = torch.zeros_like(mean)
        grad_cov = torch.zeros_like(cov)
        grad_scalar = torch.zeros_like(scalar)
        grad_alpha = torch.zeros_like(alpha)
        tile_size, n_tiles_h, n_tiles_w, pixel_size_x, pixel_size_y, H, W, thresh = ctx.const
        _backend.tile_based_vol_rendering_scalar_backward(
            mean, cov, scalar, alpha, start, end, gaussian_ids, out, grad_mean, grad_cov, grad_scalar,
            grad_alpha, grad, topleft, tile_size, n_tiles_h, n_tiles_w, pixel_size_x, pixel_size_y, H,
            W, thresh)
        return (grad_mean, grad_cov, grad_scalar, grad_alpha) + (None,) * 13


render_scalar = _render_scalar.apply
render_with_T = _render_with_T.apply
