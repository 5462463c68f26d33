"""ctypes binding of the C ABI declared in include/gsgen_hip.h.

`load()` opens gsgen_amd/lib/libgsgen_hip.so (built by gsgen_amd.build for gfx950) and fails
loudly when it is missing -- there is no CPU fallback in this package.  Every wrapper takes
raw device addresses (ints) so it can be driven from torch tensors (`.data_ptr()`), and
returns nothing: a non-zero status raises RuntimeError with gsgen_error_string().
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GSGEN_HIP_LIB: another build of the same library (A/B experiments); default = the in-tree build
DEFAULT_LIB = os.environ.get("GSGEN_HIP_LIB") or os.path.join(_HERE, "lib", "libgsgen_hip.so")

u32, f32, vp, i32, sz = C.c_uint32, C.c_float, C.c_void_p, C.c_int, C.c_size_t



class ShView(C.Structure):
    """gsgen_sh_view (include/gsgen_hip.h): one camera of a batched SH launch; addresses as ints."""
    _fields_ = [("mean", vp), ("cov", vp), ("start", vp), ("end", vp), ("gaussian_ids", vp),
                ("tile_order", vp), ("topleft", vp), ("c2w", vp), ("bg_rgb", vp),
                ("pixel_size_x", f32), ("pixel_size_y", f32), ("out", vp), ("T", vp),
                ("segment_workspace", vp), ("grad_out", vp), ("grad_mean", vp), ("grad_cov", vp), ("route_report", vp),
                ("no_fallback", u32)]


class RgbdView(C.Structure):
    """gsgen_rgbd_view (include/gsgen_hip.h): one camera of a batched RGB + heads launch."""
    _fields_ = [("mean", vp), ("cov", vp), ("depth", vp), ("start", vp), ("end", vp), ("gaussian_ids", vp),
                ("tile_order", vp), ("topleft", vp), ("pixel_size_x", f32), ("pixel_size_y", f32), ("out6", vp),
                ("T", vp), ("grad_out6", vp), ("grad_mean", vp), ("grad_cov", vp), ("grad_chan6", vp),
                ("grad_rgb", vp), ("grad_depth", vp), ("grad_opacity", vp), ("grad_depth2", vp),
                ("out_rgb", vp), ("out_depth", vp), ("out_opacity", vp), ("out_depth2", vp), ("bg_rgb", vp), ("grad_bg", vp),
                ("depth_variance", u32), ("chol", vp), ("pixel_size_dev", vp)]


class GeometryView(C.Structure):
    """gsgen_geometry_view (include/gsgen_hip.h): one camera of a batched geometry enqueue."""
    _fields_ = [("cam", vp), ("mean2d", vp), ("cov2d", vp), ("depth", vp), ("mask", vp), ("gaussian_ids", vp),
                ("start", vp), ("end", vp), ("total", vp), ("workspace", vp), ("workspace_bytes", sz), ("D_cap", u32),
                ("zero_grad_mean2d", vp), ("zero_grad_cov2d", vp), ("zero_grad_chan6", vp), ("pair_report", vp), ("chol", vp),
                ("max_radii2d", vp)]


# name -> argtypes, in the order of include/gsgen_hip.h
SIGNATURES = {
    "gsgen_culling_gaussian_bsphere": [u32, vp, vp, vp, vp, vp, vp, f32, vp],
    "gsgen_tile_culling_aabb_start_end": [u32, u32, u32, u32, vp, vp, vp, vp, vp, vp, vp, sz, vp],
    "gsgen_vol_render_start_end_with_T": [u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, u32, u32, u32,
                                          f32, f32, u32, u32, f32, vp, vp],
    "gsgen_vol_render_backward_start_end": [u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                            vp, vp, u32, u32, u32, f32, f32, u32, u32, f32, vp],
    "gsgen_vol_render_scalar": [u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, u32, u32, u32, f32, f32,
                                u32, u32, f32, vp, vp],
    "gsgen_vol_render_scalar_backward": [u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                         vp, u32, u32, u32, f32, f32, u32, u32, f32, vp],
    "gsgen_vol_render_sh": [u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, u32, u32, u32, f32, f32,
                            u32, u32, u32, f32, vp, vp, vp],
    "gsgen_vol_render_backward_sh": [u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                     vp, u32, u32, u32, f32, f32, u32, u32, u32, f32, vp, vp],
    "gsgen_project_gaussians": [u32, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "gsgen_project_gaussians_backward": [u32, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp],
    "gsgen_project_gaussians_backward_masked": [u32, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp],
    "gsgen_project_gaussians_backward_accum": [u32, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp],
    "gsgen_project_gaussians_backward_batch": [u32, u32, vp, vp, vp, C.POINTER(vp), i32, C.POINTER(vp), C.POINTER(vp),
                                               C.POINTER(vp), C.POINTER(vp), vp, vp, vp, vp],
    "gsgen_project_gaussians_backward_batch_heads": [u32, u32, vp, vp, vp, C.POINTER(vp), i32, C.POINTER(vp), C.POINTER(vp),
                                                     C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), vp, vp, vp, vp, vp],
    "gsgen_project_gaussians_backward_batch_moments_sh": [u32, u32, vp, vp, vp, C.POINTER(vp), i32, C.POINTER(vp), C.POINTER(vp),
                                                          C.POINTER(vp), C.POINTER(vp), vp, vp, vp, vp, vp, vp],
    "gsgen_project_gaussians_backward_batch_heads_moments": [u32, u32, vp, vp, vp, C.POINTER(vp), i32, C.POINTER(vp), C.POINTER(vp),
                                                             C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                                             vp, vp, vp, vp, vp, vp, vp],
    "gsgen_activate_fields": [u32, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp],
    "gsgen_activate_fields_backward": [u32, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp],
    "gsgen_sh_l1_bound_rows": [u32, vp, u32, vp, vp, vp],
    "gsgen_sh_l1_bound_rows_running": [u32, vp, u32, vp, vp, vp],
    "gsgen_vol_render_sh_batch_routed": [u32, C.POINTER(ShView), u32, vp, vp, u32, u32, u32, u32, u32, u32, f32, u32, vp, vp, vp, vp],
    "gsgen_vol_render_backward_sh_batch_routed": [u32, C.POINTER(ShView), u32, vp, vp, vp, vp, u32, u32, u32, u32, u32, u32, f32, u32,
                                                  vp, vp, vp, vp],
    "gsgen_vol_render_backward_sh_batch_routed_moments": [u32, C.POINTER(ShView), u32, vp, vp, vp, vp, u32, u32, u32, u32, u32, u32, f32,
                                                          u32, vp, vp, vp, vp],
    "gsgen_vol_render_sh_routed": [u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, u32, u32, u32, f32, f32, u32, u32, u32, f32,
                                   vp, vp, vp, vp, u32, vp, vp, vp],
    "gsgen_vol_render_backward_sh_routed": [u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, u32, u32, u32,
                                            f32, f32, u32, u32, u32, f32, vp, vp, vp, u32, vp, vp, vp],
    "gsgen_pack_camera": [vp, f32, f32, f32, f32, u32, u32, C.c_double, C.c_double, f32, f32, vp],
    "gsgen_upload_small": [vp, vp, sz, vp],
    "gsgen_pack_camera_blocks": [u32, vp, u32, vp, f32, f32, vp],
    "gsgen_adam_step": [C.c_uint64, vp, vp, vp, vp, u32, vp, vp, f32, f32, f32, u32, vp],
    "gsgen_adam_step_scalars": [u32, vp, f32, f32, u32, vp],
    "gsgen_adam_step_device_scalars": [C.c_uint64, vp, vp, vp, vp, u32, vp, f32, f32, f32, vp, vp],
    "gsgen_densify_update_batch": [u32, u32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), vp, vp, vp, vp],
    "gsgen_densify_update": [u32, vp, vp, vp, vp, vp, vp, vp],
    "gsgen_tile_culling_aabb_count": [u32, vp, vp, u32, f32, f32, f32, f32, u32, u32, f32, vp, vp, vp, vp],
    "gsgen_selftest_reduce_scatter": [u32, vp, vp, vp],
    "gsgen_vol_render_rgbd": [u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, u32, u32, u32, f32, f32, u32, u32,
                              f32, vp, vp, vp],
    "gsgen_vol_render_rgbd_backward": [u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, u32,
                                       u32, u32, f32, f32, u32, u32, f32, vp, vp],
    "gsgen_vol_render_sh_ordered": [u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, u32, u32, u32, f32, f32,
                                    u32, u32, u32, f32, vp, vp, vp, vp],
    "gsgen_vol_render_backward_sh_ordered": [u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                             vp, u32, u32, u32, f32, f32, u32, u32, u32, f32, vp, vp, vp],
    "gsgen_vol_render_sh_segmented": [u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, u32, u32, u32, f32, f32,
                                      u32, u32, u32, f32, vp, vp, vp, vp, u32, vp],
    "gsgen_vol_render_backward_sh_segmented": [u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                               vp, u32, u32, u32, f32, f32, u32, u32, u32, f32, vp, vp, vp, u32, vp],
    "gsgen_vol_render_sh_batch": [u32, C.POINTER(ShView), u32, vp, vp, u32, u32, u32, u32, u32, u32, f32, u32, vp, vp],
    "gsgen_vol_render_backward_sh_batch": [u32, C.POINTER(ShView), u32, vp, vp, vp, vp, u32, u32, u32, u32, u32, u32,
                                           f32, u32, vp, vp],
    "gsgen_vol_render_sh_batch_bounded": [u32, C.POINTER(ShView), u32, vp, vp, u32, u32, u32, u32, u32, u32, f32, u32, vp, vp, vp],
    "gsgen_vol_render_backward_sh_batch_bounded": [u32, C.POINTER(ShView), u32, vp, vp, vp, vp, u32, u32, u32, u32, u32, u32,
                                                   f32, u32, vp, vp, vp],
    "gsgen_vol_render_sh_bounded": [u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, u32, u32, u32, f32, f32,
                                    u32, u32, u32, f32, vp, vp, vp, vp, u32, vp, vp],
    "gsgen_vol_render_backward_sh_bounded": [u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                             vp, u32, u32, u32, f32, f32, u32, u32, u32, f32, vp, vp, vp, u32, vp, vp],
    "gsgen_sh_l1_bound": [u32, vp, u32, vp, vp],
    "gsgen_sh_l1_bound_check": [u32, vp, u32, vp, vp, vp],
    "gsgen_vol_render_rgbd_batch": [u32, C.POINTER(RgbdView), u32, vp, vp, u32, u32, u32, u32, u32, f32, vp, vp],
    "gsgen_vol_render_rgbd_backward_batch": [u32, C.POINTER(RgbdView), u32, vp, vp, vp, u32, u32, u32, u32, u32, f32, vp,
                                             vp],
    "gsgen_vol_render_rgbd_backward_batch_moments": [u32, C.POINTER(RgbdView), u32, vp, vp, vp, u32, u32, u32, u32, u32, f32, vp,
                                                     vp],
    "gsgen_vol_render_rgb_batch": [u32, C.POINTER(RgbdView), u32, vp, vp, u32, u32, u32, u32, u32, f32, vp, vp],
    "gsgen_vol_render_rgb_backward_batch": [u32, C.POINTER(RgbdView), u32, vp, vp, vp, vp, u32, u32, u32, u32, u32, f32,
                                            vp, vp],
    "gsgen_legacy_count_tiles": [u32, u32, vp, vp, vp, u32, u32, u32, f32, f32, f32, vp, vp],
    "gsgen_legacy_image_sort": [u32, u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, u32, u32, u32, f32, f32, f32, vp, sz, vp],
    "gsgen_frame_geometry_batch": [u32, C.POINTER(GeometryView), u32, vp, vp, vp, u32, u32, vp, vp],
    "gsgen_frame_geometry_batch_zero": [u32, C.POINTER(GeometryView), u32, vp, vp, vp, u32, u32, vp, sz, vp, vp],
    "gsgen_frame_geometry": [u32, vp, vp, vp, vp, u32, u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp],
    "gsgen_frame_geometry_report": [u32, vp, vp, vp, vp, u32, u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp],
}
PTR_FUNCS = {
    "gsgen_frame_tile_order": [vp, u32, u32, u32],
    "gsgen_host_device_pointer": [vp],
}
SIZE_FUNCS = {
    "gsgen_tile_culling_workspace_bytes": [u32, u32, u32],
    "gsgen_frame_workspace_bytes": [u32, u32, u32],
    "gsgen_segment_workspace_bytes": [u32, u32],
    "gsgen_sh_batch_workspace_bytes": [u32],
    "gsgen_sh_batch_workspace_bytes_routed": [u32, u32],
    "gsgen_frame_batch_workspace_bytes": [u32],
    "gsgen_legacy_sort_workspace_bytes": [u32, u32],
}
EXPORTS = sorted(list(SIGNATURES) + list(SIZE_FUNCS) + list(PTR_FUNCS)
                 + ["gsgen_version", "gsgen_error_string", "gsgen_kernel_variant", "gsgen_sh_poly_applies"])


class GsgenError(RuntimeError):
    pass


class Lib:
    def __init__(self, path=None):
        path = path or DEFAULT_LIB
        if not os.path.exists(path):
            raise ImportError(
                f"{path} not found: build the HIP extension first (python -m gsgen_amd.build). "
                "gsgen_amd has no CPU fallback.")
        self.path = path
        if path == DEFAULT_LIB or "libgsgen_hip" in os.path.basename(path):
            # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 and asks for it by FILE name
            # (libamdhip64.so), this library asks for the SONAME (libamdhip64.so.7).  If torch's copy is loaded
            # first ours resolves to it; the other way round the process ends up with two runtimes and the
            # kernels of this library are launched on streams the other runtime made ("no ROCm-capable
            # device").  The buffers and streams this binding is driven with are torch's, so torch goes first.
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        # an experiment build (GSGEN_HIP_LIB) is loaded with global symbols: the compiled extensions (_gs, _gsbatch) are linked to
        # libgsgen_hip.so by name and would otherwise bind to the in-tree build they find through their rpath -- an A/B through
        # BatchRenderer's C++ node or the model class then measured the in-tree kernels twice (round 6, sessions 14-20)
        self.cdll = C.CDLL(path, mode=C.RTLD_GLOBAL) if os.environ.get("GSGEN_HIP_LIB") and path == DEFAULT_LIB else C.CDLL(path)
        self.cdll.gsgen_version.restype = C.c_char_p
        self.cdll.gsgen_error_string.restype = C.c_char_p
        self.cdll.gsgen_error_string.argtypes = [i32]
        for name, argt in SIGNATURES.items():
            fn = getattr(self.cdll, name)
            fn.argtypes = argt
            fn.restype = i32
            setattr(self, name[len("gsgen_"):], self._checked(name, fn))
        for name, argt in PTR_FUNCS.items():
            fn = getattr(self.cdll, name)
            fn.argtypes = argt
            fn.restype = vp
            setattr(self, name[len("gsgen_"):], fn)
        for name, argt in SIZE_FUNCS.items():
            fn = getattr(self.cdll, name)
            fn.argtypes = argt
            fn.restype = sz
            setattr(self, name[len("gsgen_"):], fn)

    def _checked(self, name, fn):
        err = self.cdll.gsgen_error_string

        def call(*args):
            rc = fn(*args)
            if rc != 0:
                raise GsgenError(f"{name} failed: {err(rc).decode()} (code {rc})")
        call.__name__ = name
        return call

    def version(self):
        return self.cdll.gsgen_version().decode()

    def sh_poly_applies(self, sh_l1_bound, max_pixel_size, bands=4):
        """the device's routing rule, on the host (for reports): does a view of this pixel size take the polynomial form of
        the SH basis under the coefficient bound value `sh_l1_bound`?"""
        fn = self.cdll.gsgen_sh_poly_applies
        fn.argtypes, fn.restype = [f32, f32, u32], i32
        return bool(fn(float(sh_l1_bound), float(max_pixel_size), int(bands)))

    def kernel_variant(self, stage, bands=4, n_segments=1):
        """name of the compiled compositing kernel a launch of `stage` runs in this process (bands = C)"""
        fn = self.cdll.gsgen_kernel_variant
        fn.argtypes, fn.restype = [C.c_char_p, u32, u32, C.c_char_p, sz], i32
        buf = C.create_string_buffer(192)
        n = fn(stage.encode(), bands, n_segments, buf, len(buf))
        if n <= 0:
            raise ValueError(f"unknown compositing stage {stage!r}")
        return buf.value.decode()


_lib = None


def load(path=None):
    """The process-wide library handle (HIP build).  Raises ImportError when not built."""
    global _lib
    if path is not None:
        return Lib(path)
    if _lib is None:
        _lib = Lib()
    return _lib
