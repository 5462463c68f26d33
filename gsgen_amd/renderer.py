"""Host side of the rasterizer above the `_gs` mirror.

`project_gaussians` and `tile_culling_aabb_count` replace the reference's PyTorch implementations
(gs/renderer.py:391-421, gs/culling.py:8-37) with the HIP kernels and keep their signatures.
`render_rgb_heads` fuses render_one's four compositing passes; `render_frame` is the additive fused path
(cull -> project -> bin/sort -> composite, no host sync) used by bench.py, BatchRenderer and the tests.
The reference's compositing autograd.Functions (gs/renderer.py:424-1283) are deliberately not re-typed here: they
are the reference's Python and run unmodified on gsgen_amd._gs (tests/test_reference_python_on_mirror.py).
"""

import numpy as np
import torch

from . import _capi


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _p(t):
    return t.data_ptr() if t is not None else None


# ---------------------------------------------------------------------------------------------
# projection (gs/renderer.py:391-421)
# ---------------------------------------------------------------------------------------------
class _project_gaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mean, qvec, svec, c2w, detach_depth):
        mean, qvec, svec = mean.contiguous(), qvec.contiguous(), svec.contiguous()
        c2w = c2w.contiguous().float()
        N = mean.size(0)
        dev = mean.device
        mean2d = torch.empty(N, 2, device=dev, dtype=torch.float32)
        cov2d = torch.empty(N, 2, 2, device=dev, dtype=torch.float32)
        JW = torch.empty(N, 3, 3, device=dev, dtype=torch.float32)
        depth = torch.empty(N, 1, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _capi.load().project_gaussians(N, _p(mean), _p(qvec), _p(svec), _p(c2w), _p(mean2d),
                                           _p(cov2d), _p(JW), _p(depth), _stream(mean))
        ctx.save_for_backward(mean, qvec, svec, c2w)
        ctx.detach_depth = bool(detach_depth)
        ctx.mark_non_differentiable(JW)  # "JW should not be updated" (gs/renderer.py:405)
        return mean2d, cov2d, JW, depth

    @staticmethod
    def backward(ctx, g_mean2d, g_cov2d, g_JW, g_depth):
        mean, qvec, svec, c2w = ctx.saved_tensors
        N = mean.size(0)
        dev = mean.device
        g_mean2d = (g_mean2d if g_mean2d is not None else torch.zeros(N, 2, device=dev)).contiguous()
        g_cov2d = (g_cov2d if g_cov2d is not None else torch.zeros(N, 2, 2, device=dev)).contiguous()
        g_depth = g_depth.contiguous() if g_depth is not None else None
        g_mean = torch.empty_like(mean)
        g_qvec = torch.empty_like(qvec)
        g_svec = torch.empty_like(svec)
        with torch.cuda.device(dev):
            _capi.load().project_gaussians_backward(
                N, _p(mean), _p(qvec), _p(svec), _p(c2w), int(ctx.detach_depth), _p(g_mean2d),
                _p(g_cov2d), _p(g_depth), _p(g_mean), _p(g_qvec), _p(g_svec), _stream(mean))
        return g_mean, g_qvec, g_svec, None, None


def project_gaussians(mean, qvec, svec, c2w, detach_depth=False):
    """Same contract as gs/renderer.py:391-421: -> (mean2d [N,2], cov2d [N,2,2], JW [N,3,3],
    depth [N,1]); gradients flow to mean, qvec, svec (J is a constant; the depth of the
    perspective divide is detached iff detach_depth)."""
    return _project_gaussians.apply(mean, qvec, svec, c2w, detach_depth)


@torch.no_grad()
def tile_culling_aabb_count(mean, cov, tile_size, camera_info, D, sync=True):
    """gs/culling.py:8-37 on the device.  Returns (N_with_dub, aabb_topleft, aabb_bottomright);
    with sync=False N_with_dub is a device uint32 tensor instead of a python int (the
    reference's `.item()` host sync is the only reason for a sync here)."""
    N = mean.size(0)
    dev = mean.device
    tl = torch.empty(N, 2, device=dev, dtype=torch.int32)
    br = torch.empty(N, 2, device=dev, dtype=torch.int32)
    total = torch.empty(1, device=dev, dtype=torch.int32)
    mean, cov = mean.contiguous(), cov.contiguous()
    with torch.cuda.device(dev):
        _capi.load().tile_culling_aabb_count(
            N, _p(mean), _p(cov), int(tile_size), float(camera_info.fx), float(camera_info.fy),
            float(camera_info.cx), float(camera_info.cy), int(camera_info.w), int(camera_info.h),
            float(D), _p(tl), _p(br), _p(total), _stream(mean))
    return (int(total.item()) if sync else total), tl, br


# ---------------------------------------------------------------------------------------------
# fused RGB + auxiliary heads (additive; the reference's own six autograd.Functions of gs/renderer.py:424-1283 are
# NOT restated here: a user of the reference keeps them and only swaps `_gs` -- tests/test_reference_python_on_mirror.py
# runs them, imported from the reference, on this package's mirror)
# ---------------------------------------------------------------------------------------------
class _render_rgb_heads(torch.autograd.Function):
    """RGB + depth + opacity + depth^2 in ONE compositing pass: replaces the
    render_with_T + 3x render_scalar sequence of gs/gaussian_splatting.py:1304-1403.  Returns
    (rgb [H,W,3] incl. T*bg, depth [H,W,1], opacity [H,W,1], z2 [H,W,1], T [H,W,1]); gradients flow
    to mean2d, cov2d, color, depth, alpha and bg exactly as the four reference Functions' sum."""

    @staticmethod
    def forward(ctx, mean, cov, color, depth, alpha, start, end, gaussian_ids, topleft, n_tiles_h, n_tiles_w,
                pixel_size_x, pixel_size_y, H, W, thresh, bg, tile_order):
        dev = mean.device
        depth_c = depth.contiguous()
        out6 = torch.zeros(H, W, 6, dtype=torch.float32, device=dev)
        T = torch.ones(H, W, 1, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _capi.load().vol_render_rgbd(mean.size(0), gaussian_ids.size(0), _p(mean), _p(cov), _p(color), _p(depth_c),
                                         _p(alpha), _p(start), _p(end), _p(gaussian_ids), _p(out6), _p(topleft), 16,
                                         n_tiles_h, n_tiles_w, pixel_size_x, pixel_size_y, H, W, thresh, _p(T),
                                         tile_order, _stream(mean))
        if bg is not None:
            out6[..., :3] += T * bg
        ctx.save_for_backward(mean, cov, color, depth_c, alpha, start, end, gaussian_ids, topleft, out6, T)
        ctx.const = [n_tiles_h, n_tiles_w, pixel_size_x, pixel_size_y, H, W, thresh, tile_order, bg is not None]
        return out6[..., :3], out6[..., 3:4], out6[..., 4:5], out6[..., 5:6], T

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_opac, g_z2, g_T):
        mean, cov, color, depth, alpha, start, end, gaussian_ids, topleft, out6, T = ctx.saved_tensors
        n_tiles_h, n_tiles_w, psx, psy, H, W, thresh, tile_order, has_bg = ctx.const
        dev = mean.device
        N = mean.size(0)
        z = lambda g, c: g if g is not None else torch.zeros(H, W, c, device=dev)  # noqa: E731
        go6 = torch.cat([z(g_rgb, 3), z(g_depth, 1), z(g_opac, 1), z(g_z2, 1)], dim=-1).contiguous()
        g_mean, g_cov = torch.zeros_like(mean), torch.zeros_like(cov)
        g_chan, g_alpha = torch.zeros(N, 6, device=dev), torch.zeros_like(alpha)
        with torch.cuda.device(dev):
            _capi.load().vol_render_rgbd_backward(N, gaussian_ids.size(0), _p(mean), _p(cov), _p(color), _p(depth),
                                                  _p(alpha), _p(start), _p(end), _p(gaussian_ids), _p(out6), _p(g_mean),
                                                  _p(g_cov), _p(g_chan), _p(g_alpha), _p(go6), _p(topleft), 16,
                                                  n_tiles_h, n_tiles_w, psx, psy, H, W, thresh, tile_order, _stream(mean))
        d = depth.reshape(N)
        g_d = (g_chan[:, 3] + 2.0 * d * g_chan[:, 5]).reshape(depth.shape)
        g_bg = torch.nan_to_num(g_rgb * T) if (has_bg and g_rgb is not None) else None
        return (g_mean, g_cov, g_chan[:, :3].contiguous(), g_d, g_alpha) + (None,) * 11 + (g_bg, None)


def render_rgb_heads(mean, cov, color, depth, alpha, start, end, gaussian_ids, topleft, n_tiles_h, n_tiles_w,
                     pixel_size_x, pixel_size_y, H, W, thresh, bg=None, tile_order=None):
    return _render_rgb_heads.apply(mean, cov, color, depth, alpha, start, end, gaussian_ids, topleft, n_tiles_h,
                                   n_tiles_w, pixel_size_x, pixel_size_y, H, W, thresh, bg, tile_order)



# ---------------------------------------------------------------------------------------------
# camera packing for the fused path
# ---------------------------------------------------------------------------------------------
class CameraInfo:
    """Field-compatible with utils/camera.py:219-259 (fx, fy, cx, cy, w, h, near/far planes,
    yfov, aspect)."""

    def __init__(self, fx, fy, cx, cy, w, h, near_plane=0.01, far_plane=100.0):
        self.fx, self.fy, self.cx, self.cy = float(fx), float(fy), float(cx), float(cy)
        self.w, self.h = int(w), int(h)
        self.yfov = 2 * np.arctan(self.h / (2 * self.fy))
        self.aspect = self.w / self.h
        self.near_plane, self.far_plane = float(near_plane), float(far_plane)

    def get_frustum(self, c2w):
        """utils/camera.py:260-294 in fp32 numpy, operation for operation (python-double scalars
        are rounded to fp32 before they meet an fp32 array, as torch does)."""
        c2w = np.asarray(c2w, np.float32)
        f32 = np.float32
        up, right, lookat, t = -c2w[:, 1], c2w[:, 0], c2w[:, 2], c2w[:, 3]
        half_v = self.far_plane * np.tan(self.yfov * 0.5)
        half_h = half_v * self.aspect
        near_point = f32(self.near_plane) * lookat
        far_point = f32(self.far_plane) * lookat
        cr = lambda a, b: np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2],  # noqa: E731
                                    a[0] * b[1] - a[1] * b[0]], np.float32)
        normals = np.stack([
            lookat, -lookat,
            cr(far_point - f32(half_h) * right, up), cr(up, far_point + f32(half_h) * right),
            cr(far_point + f32(half_v) * up, right), cr(right, far_point - f32(half_v) * up)])
        nrm = np.sqrt((normals * normals)[:, 0] + (normals * normals)[:, 1] + (normals * normals)[:, 2])
        normals = (normals / np.maximum(nrm, f32(1e-12))[:, None]).astype(np.float32)
        pts = np.stack([near_point + t, far_point + t, t, t, t, t]).astype(np.float32)
        return normals, pts

    def pack(self, c2w, frustum_radius=6.0, tile_radius=6.0):
        """-> float32[56], the `cam` block of gsgen_frame_geometry (include/gsgen_hip.h)."""
        cam = np.empty(56, np.float32)
        self.pack_into(cam, c2w, frustum_radius, tile_radius)
        return cam

    def pack_into(self, cam, c2w, frustum_radius=6.0, tile_radius=6.0):
        """Fills a contiguous float32[>=56] host array in place (gsgen_pack_camera: the frustum of
        get_frustum() computed by the library's host code, a few microseconds per camera)."""
        c2w = np.ascontiguousarray(np.asarray(c2w, np.float32).reshape(-1)[:12])
        _capi.load().pack_camera(c2w.ctypes.data, self.fx, self.fy, self.cx, self.cy, self.w, self.h,
                                 self.near_plane, self.far_plane, frustum_radius, tile_radius, cam.ctypes.data)


def n_tiles(H, W, tile_size=16):
    return (H + tile_size - 1) // tile_size, (W + tile_size - 1) // tile_size


def sh_l1_bound_device(sh, out=None):
    """S = max over splats and channels of sum_{k >= 1} |sh[i][c][k]| as a 1-float DEVICE tensor: one coalesced pass over the
    coefficients on the current stream (~5 us for 100 k splats), no host sync.  This is what the SH launches route on
    (include/gsgen_hip.h, "the coefficient bound"): render_frame / BatchRenderer.render call it in their forward, so the
    value always belongs to the coefficients being rendered."""
    if sh.dtype != torch.float32:
        raise ValueError("sh_l1_bound wants fp32 SH coefficients [N, 3, C*C] (or the same memory as [N, 3*C*C])")
    sh = sh.detach().contiguous()
    if sh.dim() == 2 and sh.shape[1] % 3 == 0:  # the flat layout some callers keep: [N, 3 C*C]
        sh = sh.view(sh.shape[0], 3, sh.shape[1] // 3)
    C = int(round(sh.shape[-1] ** 0.5)) if sh.dim() == 3 else 0
    if sh.dim() != 3 or sh.shape[1] != 3 or C * C != sh.shape[-1] or not 1 <= C <= 4:
        raise ValueError("sh_l1_bound wants fp32 SH coefficients [N, 3, C*C] (or the same memory as [N, 3*C*C]), C in 1..4")
    if out is None:
        out = torch.empty(1, device=sh.device, dtype=torch.float32)
    with torch.cuda.device(sh.device):
        _capi.load().sh_l1_bound(sh.shape[0], _p(sh), C, _p(out), _stream(sh))
    return out


def sh_row_bounds_device(sh, out=None, out_max=None):
    """Per-SPLAT coefficient bounds as a DEVICE tensor [N]: rows[i] = max over the three channels of sum_{k >= 1} |sh[i][c][k]|
    (gsgen_sh_l1_bound_rows: one coalesced pass on the current stream, no host sync).  The SH launches of the fused paths
    route per TILE on it (include/gsgen_hip.h, "per-TILE routing"): a splat with large higher-band coefficients, or a wide
    camera, costs the tiles it touches, not the view.  out_max (optional, 1 float): receives the global maximum."""
    if sh.dtype != torch.float32:
        raise ValueError("sh_row_bounds wants fp32 SH coefficients [N, 3, C*C]")
    sh = sh.detach().contiguous()
    if sh.dim() == 2 and sh.shape[1] % 3 == 0:
        sh = sh.view(sh.shape[0], 3, sh.shape[1] // 3)
    C = int(round(sh.shape[-1] ** 0.5)) if sh.dim() == 3 else 0
    if sh.dim() != 3 or sh.shape[1] != 3 or C * C != sh.shape[-1] or not 1 <= C <= 4:
        raise ValueError("sh_row_bounds wants fp32 SH coefficients [N, 3, C*C], C in 1..4")
    if out is None:
        out = torch.empty(sh.shape[0], device=sh.device, dtype=torch.float32)
    for name, t_, need in (("out", out, sh.shape[0]), ("out_max", out_max, 1)):
        if t_ is not None and not (isinstance(t_, torch.Tensor) and t_.dtype == torch.float32 and t_.device == sh.device
                                   and t_.is_contiguous() and t_.numel() >= need):  # (the kernel writes `need` floats through a raw pointer)
            raise ValueError(f"sh_row_bounds: {name} must be a contiguous fp32 tensor of at least {need} element(s) on {sh.device}")
    with torch.cuda.device(sh.device):
        _capi.load().sh_l1_bound_rows(sh.shape[0], _p(sh), C, _p(out_max), _p(out), _stream(sh))
    return out


def sh_l1_bound(sh):
    """The same value as a Python float (ONE host sync): for reports and tests, never needed by the render path."""
    return float(sh_l1_bound_device(sh).item())


def verify_sh_l1_bound(sh, bound):
    """Debug check of a bound produced elsewhere (a 1-float device tensor, e.g. a by-product of an optimiser pass): counts, on
    the device, the (splat, channel) rows whose sum_{k >= 1} |sh| exceeds it and raises if there are any (one host sync)."""
    sh = sh.detach().contiguous()
    n_bad = torch.empty(1, device=sh.device, dtype=torch.int32)
    C = int(round(sh.shape[-1] ** 0.5))
    with torch.cuda.device(sh.device):
        _capi.load().sh_l1_bound_check(sh.shape[0], _p(sh), C, _p(bound), _p(n_bad), _stream(sh))
    n = int(n_bad.item())
    if n:
        raise RuntimeError(f"gsgen_amd: the SH coefficient bound {float(bound.item()):.6g} is exceeded by {n} (splat, channel) rows: "
                           "it is stale or wrong, and the polynomial SH basis would silently break the 1e-4 image contract")


def pair_count(v):
    """The device's uint32 pair count as a Python int (it lives in an int32 tensor).  Beyond int32 -- the kernels saturate at
    2^32 - 1 -- no pair buffer can hold the frame (list positions are int32 in the reference's layout): that is a diverged
    scene (scales far larger than the view), reported as such instead of being 'regrown' for."""
    n = int(v) & 0xFFFFFFFF
    if n > 0x7FFFFFFF:
        raise RuntimeError(f"gsgen_amd: the frame needs {'at least 2^32 - 1' if n == 0xFFFFFFFF else n} (tile, Gaussian) pairs, beyond "
                           "the int32 list positions of the reference's layout -- the Gaussians' scales have diverged")
    return n


class PairListOverflow(RuntimeError):
    """A frame's (tile, Gaussian) pairs did not fit its list: that frame's image and T are NaN (never a finite blank image)
    and it contributed no gradients.  Raised by the first render / check_overflow() after the device reported it; by then the
    buffers have been regrown, so repeating the step succeeds.  `strict=True` (one host sync per frame or batch, what the
    reference pays per camera, gs/culling.py:34) makes overflow impossible instead."""


class PairCountReport:
    """The geometry launches' own report to the host (include/gsgen_hip.h, "pair_report"): per view two uint32 in pinned,
    device-mapped host memory -- [0] the last frame's pair count (every frame), [1] the largest count of a frame that did NOT
    fit (kept until cleared here).  The kernel stores them over the fabric; nothing is copied, no event is recorded, the
    host never waits, and a hipGraph replay reports exactly like an eager launch.  (Rounds 2-4: an asynchronous copy + event
    per frame into a ring of pinned blocks, sampled every 4th batch -- a camera overflowing in an unsampled batch went
    unreported)"""

    def __init__(self, n):
        self.n = int(n)
        self.host = torch.zeros(self.n, 2, dtype=torch.int32)
        self._dev = None
        # unmapped: a GPU process whose pinned block has no device-side address -- nothing would ever be reported, an outgrown
        # list would render NaN frames for ever with nobody told.  FrameBuffers / BatchRenderer then size every
        # frame with the synchronous read-back (as strict=True does): lossless, one host sync per frame or batch.
        self.unmapped = False
        if torch.cuda.is_available():  # (host-only processes -- bench.py's dry run, the CPU tests -- keep a plain block)
            self.host = self.host.pin_memory()
            self._dev = _capi.load().host_device_pointer(self.host.data_ptr())
            if not self._dev:
                import warnings
                self._dev, self.unmapped = None, True
                warnings.warn("gsgen_amd: the pinned pair-count report block is not mapped into the device's address space; every "
                              "frame will be sized with a host read-back (as strict=True) instead of running unmonitored",
                              RuntimeWarning, stacklevel=3)
        self._np = self.host.numpy().view(np.uint32)  # the same memory

    def ptr(self, i):
        """device-side address of view i's two words (None: no report, host-only process)"""
        return None if self._dev is None else self._dev + 8 * i

    def last(self, i):
        return pair_count(self._np[i, 0])

    def overflow(self, i):
        return pair_count(self._np[i, 1])

    def any_overflow(self):
        return bool(self._np[:, 1].any())

    def clear(self, i=None):
        if i is None:
            self._np[:, 1] = 0
        else:
            self._np[i, 1] = 0


def _cap_for(need):
    """list capacity for a frame that needs `need` pairs: x 1.5 (sized from what was seen, not 16 N)"""
    return int(need * 1.5) + 4096


class FrameBuffers:
    """Device buffers of the fused path for one (N, W, H) shape.  D_cap is the capacity of the (tile, Gaussian) pair list.

    A frame whose pairs do not fit is NEVER a finite blank image: its image and T are NaN (the compositing forwards poison
    every tile, GSGEN_LIST_OVERFLOW), it contributes no gradients, and the next render through these buffers -- or
    check_overflow() -- regrows them and raises PairListOverflow.  Three ways not to get there:
      * D_cap=None (default): the FIRST frame is rendered synchronously (one host sync: count, regrow, bin again if it did
        not fit) and sizes the list at 1.5 x what it needed; afterwards every frame's count reaches the host through the
        geometry launch's own report (PairCountReport, no sync) and the list is regrown BEFORE the scene outgrows it
        (count x 1.25 > capacity);
      * strict=True: every frame is rendered that way (the reference's per-camera `.item()`, gs/culling.py:34): lossless;
      * ensure_capacity() after a render (one sync): False if that frame has to be rendered again."""

    def __init__(self, N, W, H, device, D_cap=None, segments=1, total=None, depth=None, strict=False, report=None):
        """total: optional int32 [1] device tensor to use as this buffer set's pair counter (BatchRenderer keeps the
        counters of its slots in one tensor so that one copy brings a whole batch's counts to the host).
        depth: optional float32 [N, 1] device tensor to use as the depth buffer (BatchRenderer: the rows of one matrix).
        report: (PairCountReport, index) shared with a BatchRenderer, which then does the overflow handling itself.
        segments > 1: the SH backward runs one workgroup per (tile, 32-entry list segment) from
        checkpoints the forward leaves in `seg_ws` (gsgen_vol_render_sh_segmented) -- shorter tail for
        a lone render, slightly more total work; 1 (one workgroup per tile) is best when several
        renders are in flight."""
        self.N, self.W, self.H, self.device = N, W, H, device
        self.nth, self.ntw = n_tiles(H, W)
        self.segments = int(segments)
        self.seg_ws = None
        if self.segments > 1:
            self.seg_ws = torch.empty(_capi.load().segment_workspace_bytes(self.nth * self.ntw, self.segments),
                                      device=device, dtype=torch.uint8)
        f = dict(device=device, dtype=torch.float32)
        self.mean2d = torch.empty(N, 2, **f)
        self.cov2d = torch.empty(N, 2, 2, **f)
        self.depth = depth if depth is not None else torch.empty(N, 1, **f)
        self.mask = torch.empty(N, device=device, dtype=torch.bool)
        self.start = torch.empty(self.nth * self.ntw, device=device, dtype=torch.int32)
        self.end = torch.empty_like(self.start)
        self.total = total if total is not None else torch.zeros(1, device=device, dtype=torch.int32)
        self.D_cap = 0
        # `generation` counts the forwards that rewrote these buffers (and regrowths): a backward whose forward is no
        # longer the latest user raises instead of reading another frame's lists.
        self.generation = 0
        self.strict = bool(strict)
        self._owns_report = report is None
        self._report, self._ri = (PairCountReport(1), 0) if report is None else report
        self.strict = self.strict or self._report.unmapped
        # sized: the capacity comes from a measured frame (or from the caller, who then answers for it)
        self.sized = D_cap is not None
        self._alloc_pairs(D_cap if D_cap else max(4 * N, 1 << 16))

    def _alloc_pairs(self, D_cap, need=0):
        """need: the pair count this (re)allocation answers to.  List positions are int32 (the reference's start / end layout): a
        frame beyond 2^31 - 1 pairs cannot be held by ANY list -- an error, not another round of the callers' retry loops
        (a clamped capacity that no longer grows made frame_geometry / BatchRenderer._geometry spin)."""
        if int(need) > 0x7FFFFFFF:
            raise RuntimeError(f"gsgen_amd: a frame needs {int(need)} (tile, Gaussian) pairs, beyond the int32 list positions of the "
                               "reference's layout -- the Gaussians' scales have diverged")
        self.generation += 1  # pending backwards hold pointers into the old lists
        self.D_cap = int(min(D_cap, 0x7FFFFFFF))
        self.ids = torch.empty(self.D_cap, device=self.device, dtype=torch.int32)
        nbytes = _capi.load().frame_workspace_bytes(self.N, self.D_cap, self.nth * self.ntw)
        self.ws = torch.empty(nbytes, device=self.device, dtype=torch.uint8)

    def report_ptr(self):
        return self._report.ptr(self._ri)

    def row_bounds(self):
        """float32 [N + 1]: the per-splat coefficient bounds of the frame in flight (sh_row_bounds_device) and their maximum
        behind them, read by its backward"""
        if getattr(self, "_rows", None) is None:
            self._rows = torch.empty(self.N + 1, device=self.device, dtype=torch.float32)
        return self._rows

    def tile_order(self):
        """device address of the longest-list-first launch order written by frame_geometry"""
        return _capi.load().frame_tile_order(self.ws.data_ptr(), self.N, self.D_cap, self.nth * self.ntw)

    def ensure_capacity(self):
        """Host-side check (one sync): did the last frame fit?  Grows the pair buffers if not (False: that frame's image is
        NaN -- render it again)."""
        need = pair_count(self.total.item())
        self.sized = True
        if self._owns_report:
            self._report.clear()
        if need > self.D_cap:
            self._alloc_pairs(_cap_for(need), need)
            return False
        return True

    def needs_sync_sizing(self):
        """does the next frame have to be rendered synchronously (strict, or a default-sized list nobody has measured yet)?
        Under stream capture nothing can be measured: capturing with an unmeasured list is an error, not a silent risk."""
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            if not self.sized or self.strict:
                raise RuntimeError("gsgen_amd: rendering under stream capture with " +
                                   ("strict=True (a host sync per frame cannot be captured)" if self.sized else
                                    "a pair list nobody has sized: render once eagerly (the first frame sizes the list) or "
                                    "call ensure_capacity() before capturing") +
                                   ".  (After a replay, check_overflow() tells whether the replayed frame still fit.)")
            return False
        return self.strict or not self.sized

    def begin_frame(self):
        """Every forward through these buffers starts here: -> the generation its backward must still find."""
        if self._owns_report:
            self.check_overflow()
        self.generation += 1
        return self.generation

    def check_overflow(self):
        """No sync.  Looks at what the geometry launches of earlier frames reported: a frame that did not fit -> regrow and
        raise PairListOverflow (its image was NaN); a count within 25 % of the capacity -> regrow quietly, before it
        overflows.  Also the way to learn of an overflow inside a hipGraph replay."""
        need = self._report.overflow(self._ri)
        if need:
            old = self.D_cap
            self._report.clear(self._ri)
            if need > self.D_cap:
                self._alloc_pairs(_cap_for(need), need)
            raise PairListOverflow(
                f"gsgen_amd: an earlier frame through these buffers needed {need} (tile, Gaussian) pairs, capacity was {old}: "
                f"its image and T are NaN and it contributed no gradients.  The buffers have been regrown to {self.D_cap}: "
                f"render that frame again (or use strict=True / ensure_capacity() to rule this out).")
        last = self._report.last(self._ri)
        if self.sized and last <= self.D_cap and last * 1.25 > self.D_cap:
            self._alloc_pairs(_cap_for(last))
        return True


def frame_geometry(mean, qvec, svec, cam_dev, buf):
    """cull + project + AABB + bin + per-tile sort in one enqueue.  No host sync -- except while the list is unsized or
    `buf.strict`: then the count is read back and, if the pairs did not fit, the list regrown and the frame binned again
    (lossless; FrameBuffers.needs_sync_sizing)."""
    lib = _capi.load()
    sync = buf.needs_sync_sizing()
    with torch.cuda.device(mean.device):
        while True:
            lib.frame_geometry_report(
                buf.N, _p(mean), _p(qvec), _p(svec), _p(cam_dev), buf.W, buf.H, buf.D_cap, _p(buf.mean2d),
                _p(buf.cov2d), _p(buf.depth), _p(buf.mask), _p(buf.ids), _p(buf.start), _p(buf.end),
                _p(buf.total), buf.report_ptr(), _p(buf.ws), buf.ws.numel(), _stream(mean))
            if not sync or buf.ensure_capacity():  # (ensure_capacity: one sync; regrows and says False if the pairs did not fit)
                return


class DensifyStats:
    """The three per-Gaussian running statistics the trainer's densify/prune step reads
    (gs/gaussian_splatting.py:477-479): max_radii2d, mean_2d_grad_accum, cnt (all f32 [N]).

    `render_frame(..., stats=s)` updates max_radii2d in the forward and grad_accum/cnt in the
    backward (the reference does the latter in update_densify_info(), :464-469), each as one
    masked pass on the render's stream."""

    def __init__(self, N, device):
        self.max_radii2d = torch.zeros(N, device=device, dtype=torch.float32)
        self.grad_accum = torch.zeros(N, device=device, dtype=torch.float32)
        self.cnt = torch.zeros(N, device=device, dtype=torch.float32)

    def update_radii(self, cov2d, mask):
        with torch.cuda.device(cov2d.device):
            _capi.load().densify_update(cov2d.shape[0], _p(cov2d), None, _p(mask), _p(self.max_radii2d), None,
                                        None, _stream(cov2d))

    def update_grad(self, grad_mean2d, mask):
        with torch.cuda.device(grad_mean2d.device):
            _capi.load().densify_update(grad_mean2d.shape[0], None, _p(grad_mean2d), _p(mask), None,
                                        _p(self.grad_accum), _p(self.cnt), _stream(grad_mean2d))


class _render_frame(torch.autograd.Function):
    """Fused differentiable frame: (mean, qvec, svec, alpha, sh|color) -> rgb [H,W,3] (+T)."""

    @staticmethod
    def forward(ctx, mean, qvec, svec, alpha, col, cam_dev, topleft, rot, bg_rgb, buf, cam_info, C,
                thresh, detach_depth, stats, sh_basis):
        mean, qvec, svec = mean.contiguous(), qvec.contiguous(), svec.contiguous()
        alpha, col = alpha.contiguous(), col.contiguous()
        lib = _capi.load()
        H, W = buf.H, buf.W
        dev = mean.device
        buf.begin_frame()
        frame_geometry(mean, qvec, svec, cam_dev, buf)
        ctx.gen = buf.generation  # (read behind the geometry: sizing an unmeasured list regrows it inside this forward)
        if stats is not None:
            stats.update_radii(buf.cov2d, buf.mask)
        out = torch.zeros(H, W, 3, device=dev, dtype=torch.float32)
        T = torch.ones(H, W, 1, device=dev, dtype=torch.float32)
        psx, psy = 1.0 / cam_info.fx, 1.0 / cam_info.fy
        s = _stream(mean)
        # SH degree 3, "auto": per-splat bounds measured on the device, the kernels route per tile on them
        ctx.sh_bound = None
        if C == 4 and sh_basis == "auto":
            if col.shape[0] != buf.N:
                raise ValueError(f"render_frame: {col.shape[0]} coefficient rows for buffers of {buf.N} Gaussians")
            ctx.sh_bound = buf.row_bounds()
            sh_row_bounds_device(col, out=ctx.sh_bound[:buf.N], out_max=ctx.sh_bound[buf.N:])
        smax = None if ctx.sh_bound is None else ctx.sh_bound.data_ptr() + 4 * buf.N  # (the view's bound: decides first)
        with torch.cuda.device(dev):
            if C > 0:
                lib.vol_render_sh_routed(buf.N, buf.D_cap, _p(buf.mean2d), _p(buf.cov2d), _p(col), _p(alpha),
                                          _p(buf.start), _p(buf.end), _p(buf.ids), _p(out), _p(topleft), _p(rot),
                                          16, buf.nth, buf.ntw, psx, psy, H, W, C, thresh, _p(bg_rgb), _p(T),
                                          buf.tile_order(), _p(buf.seg_ws), buf.segments, smax, _p(ctx.sh_bound), s)
            else:
                lib.vol_render_start_end_with_T(buf.N, buf.D_cap, _p(buf.mean2d), _p(buf.cov2d), _p(col),
                                                _p(alpha), _p(buf.start), _p(buf.end), _p(buf.ids),
                                                _p(out), _p(topleft), 16, buf.nth, buf.ntw, psx, psy, H,
                                                W, thresh, _p(T), s)
                if bg_rgb is not None:
                    out = out + T * bg_rgb
        ctx.save_for_backward(mean, qvec, svec, alpha, col, cam_dev, topleft, rot, out, T)
        ctx.buf, ctx.cam_info, ctx.C, ctx.thresh, ctx.detach = buf, cam_info, C, thresh, detach_depth
        ctx.bg_shape = tuple(bg_rgb.shape) if (bg_rgb is not None and ctx.needs_input_grad[8]) else None
        ctx.stats = stats
        ctx.mark_non_differentiable(T)
        return out, T

    @staticmethod
    def backward(ctx, grad, _gT):
        mean, qvec, svec, alpha, col, cam_dev, topleft, rot, out, T = ctx.saved_tensors
        buf, ci, C, thresh = ctx.buf, ctx.cam_info, ctx.C, ctx.thresh
        if ctx.gen != buf.generation:
            raise RuntimeError("gsgen_amd.render_frame: these FrameBuffers were used by a later render (or regrown) "
                               "before this frame's backward ran -- its lists are gone.  Run backward before the next "
                               "render with the same buffers, or give every frame in flight its own FrameBuffers.")
        lib = _capi.load()
        dev = mean.device
        H, W, N = buf.H, buf.W, buf.N
        grad = grad.contiguous()
        g2 = torch.zeros(7 * N, device=dev, dtype=torch.float32)  # one memset: mean2d | cov2d | alpha
        g_mean2d, g_cov2d, g_alpha = g2[:2 * N].view(N, 2), g2[2 * N:6 * N].view(N, 4), g2[6 * N:]
        g_col = torch.zeros_like(col)
        psx, psy = 1.0 / ci.fx, 1.0 / ci.fy
        s = _stream(mean)
        with torch.cuda.device(dev):
            if C > 0:
                lib.vol_render_backward_sh_routed(N, buf.D_cap, _p(buf.mean2d), _p(buf.cov2d), _p(col), _p(alpha),
                                                   _p(buf.start), _p(buf.end), _p(buf.ids), _p(out), _p(g_mean2d),
                                                   _p(g_cov2d), _p(g_col), _p(g_alpha), _p(grad), _p(topleft),
                                                   _p(rot), 16, buf.nth, buf.ntw, psx, psy, H, W, C, thresh, None,
                                                   buf.tile_order(), _p(buf.seg_ws), buf.segments,
                                                   None if ctx.sh_bound is None else ctx.sh_bound.data_ptr() + 4 * N,
                                                   _p(ctx.sh_bound), s)
            else:
                lib.vol_render_backward_start_end(N, buf.D_cap, _p(buf.mean2d), _p(buf.cov2d), _p(col),
                                                  _p(alpha), _p(buf.start), _p(buf.end), _p(buf.ids),
                                                  _p(out), _p(g_mean2d), _p(g_cov2d), _p(g_col),
                                                  _p(g_alpha), _p(grad), _p(topleft), 16, buf.nth,
                                                  buf.ntw, psx, psy, H, W, thresh, s)
            g_mean = torch.empty_like(mean); g_qvec = torch.empty_like(qvec); g_svec = torch.empty_like(svec)
            lib.project_gaussians_backward_masked(N, _p(mean), _p(qvec), _p(svec), _p(cam_dev),
                                                  int(ctx.detach), _p(buf.mask), _p(g_mean2d), _p(g_cov2d),
                                                  None, _p(g_mean), _p(g_qvec), _p(g_svec), s)
        if ctx.stats is not None:
            ctx.stats.update_grad(g_mean2d, buf.mask)
        # d/d bg of out = ... + T * bg (gs/renderer.py:1283: nan_to_num(grad * T)), reduced to bg's shape
        g_bg = torch.nan_to_num(grad * T).sum_to_size(ctx.bg_shape) if ctx.bg_shape is not None else None
        return (g_mean, g_qvec, g_svec, g_alpha, g_col, None, None, None, g_bg) + (None,) * 7


def render_frame(mean, qvec, svec, alpha, col, cam_info, c2w, buf, C=0, bg_rgb=None, thresh=1e-4,
                 frustum_radius=6.0, tile_radius=6.0, detach_depth=True, stats=None, sh_basis="auto"):
    """One differentiable render of `cam_info` at pose `c2w` ([3,4], host array or tensor).

    col is sh_coeffs [N,3,C*C] when C in 1..4, or post-activation rgb [N,3] when C == 0.
    Culled Gaussians keep their index (no mask gathers); gradients come back for every input.
    sh_basis (SH degree 3): "auto" -- the tile-local polynomial form of the per-pixel basis where its error bound holds (decided
    on the device from the coefficients of THIS call, images within 1e-5 of the exact kernels), "exact" -- the exact kernels.
    Returns (rgb [H,W,3], T [H,W,1])."""
    if sh_basis not in ("auto", "exact"):
        raise ValueError("sh_basis: 'auto' or 'exact'")
    c2w_np = c2w.detach().cpu().numpy() if isinstance(c2w, torch.Tensor) else np.asarray(c2w)
    dev = mean.device
    # cam block (56) | pixel origin (2) | rotation rows (9) | pad: packed on the host, sent through kernel arguments
    # (gsgen_upload_small) -- three pageable `.to(device)` copies here each waited for the stream
    h = np.zeros(68, np.float32)
    c2w_f = np.ascontiguousarray(np.asarray(c2w_np, np.float32).reshape(-1)[:12])
    cam_info.pack_into(h, c2w_f, frustum_radius, tile_radius)
    h[56:58] = (-cam_info.cx / cam_info.fx, -cam_info.cy / cam_info.fy)
    h[58:67] = c2w_f.reshape(3, 4)[:, :3].reshape(-1)
    block = torch.empty(68, device=dev, dtype=torch.float32)
    _capi.load().upload_small(_p(block), h.ctypes.data, 272, torch.cuda.current_stream(dev).cuda_stream)
    cam_dev, topleft, rot = block[:56], block[56:58], block[58:67]
    return _render_frame.apply(mean, qvec, svec, alpha, col, cam_dev, topleft, rot, bg_rgb, buf, cam_info,
                               int(C), float(thresh), bool(detach_depth), stats, sh_basis)
