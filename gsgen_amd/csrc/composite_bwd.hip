// composite_bwd.hip -- C entry points of the compositing backward, for gfx950.
//
// Replaces (reference paths relative to /root/reference/gs/src/include):
//   RGB     vol_render.h:866-973 (body :318-418)
//   scalar  vol_render_scalar.h:104-234
//   SH      vol_render_sh.h:268-455 / vol_render_bg.h:131-242
// and the Gaussian backward kernels.h:394-418.
//
// The kernels live in composite.hip next to the forward they share their evaluation code with
// (k_composite_bwd_pixel: unpacked vector form, every mode; k_composite_bwd_sh_vec / k_composite_bwd_chan_vec: packed
// per-pixel arithmetic, optionally one workgroup per list segment).  A splat-parallel form
// existed during round 1 -- splat-parallel: a lane owns a Gaussian, the front-to-back recurrences
// become DPP prefix scans over the lanes, nothing is reduced across pixels -- and lost to both on
// every workload and routing tried (profiles/r01_notes.md); it was removed.
//
// Semantics are the reference's: the prefix at a Gaussian only contains the Gaussians in
// front of it; "stop when T < thresh, tested before each splat"; skip when a*G < 1/255; alpha
// clamped at 0.99 but differentiated as if it were not (vol_render.h:365,409); `final` includes
// bg*T.
#include "composite_common.hpp"

namespace gs {

// composite.hip: picks the kernel for the mode (and the variant table)
int launch_bwd_pixel_dispatch(int mode, int C, const CompParams &p, hipStream_t s);

static int launch_bwd(int mode, int C, const CompParams &p_, hipStream_t s) {
  CompParams p = p_;
  p.n_lo = 0; p.n_hi = 0x7fffffff;
  return launch_bwd_pixel_dispatch(mode, C, p, s);
}

}  // namespace gs

using namespace gs;

extern "C" {

int gsgen_vol_render_backward_start_end(uint32_t N, uint32_t D, const float *mean, const float *cov,
                                        const float *color, const float *alpha, const int *start,
                                        const int *end, const int *gaussian_ids, const float *out,
                                        float *grad_mean, float *grad_cov, float *grad_color,
                                        float *grad_alpha, const float *grad_out,
                                        const float *topleft, uint32_t tile_size,
                                        uint32_t n_tiles_h, uint32_t n_tiles_w, float pixel_size_x,
                                        float pixel_size_y, uint32_t H, uint32_t W, float thresh,
                                        gsgen_stream_t stream) {
  if (int e = check_common(tile_size, start, end, out)) return e;
  if (N == 0 || D == 0) return 0;
  CompParams p{};
  p.mean = mean; p.cov = cov; p.col = color; p.alpha = alpha;
  p.start = start; p.end = end; p.ids = gaussian_ids; p.topleft = topleft;
  p.final_img = out; p.grad_out = grad_out;
  p.g_mean = grad_mean; p.g_cov = grad_cov; p.g_col = grad_color; p.g_alpha = grad_alpha;
  p.ntw = (int)n_tiles_w; p.nth = (int)n_tiles_h; p.H = (int)H; p.W = (int)W;
  p.psx = pixel_size_x; p.psy = pixel_size_y; p.thresh = thresh; p.tile_side = (int)tile_size;
  return launch_bwd(MODE_RGB, 1, p, (hipStream_t)stream);
}

int gsgen_vol_render_rgbd_backward(uint32_t N, uint32_t D, const float *mean, const float *cov, const float *color,
                                   const float *depth, const float *alpha, const int *start, const int *end,
                                   const int *gaussian_ids, const float *out6, float *grad_mean, float *grad_cov,
                                   float *grad_chan6, float *grad_alpha, const float *grad_out6,
                                   const float *topleft, uint32_t tile_size, uint32_t n_tiles_h,
                                   uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y, uint32_t H,
                                   uint32_t W, float thresh, const uint32_t *tile_order, gsgen_stream_t stream) {
  if (int e = check_common(tile_size, start, end, out6)) return e;
  if (N == 0 || D == 0) return 0;
  if (!depth) return GSGEN_EINVAL;
  CompParams p{};
  p.mean = mean; p.cov = cov; p.col = color; p.depth = depth; p.alpha = alpha;
  p.start = start; p.end = end; p.ids = gaussian_ids; p.topleft = topleft;
  p.final_img = out6; p.grad_out = grad_out6;
  p.g_mean = grad_mean; p.g_cov = grad_cov; p.g_col = grad_chan6; p.g_alpha = grad_alpha;
  p.ntw = (int)n_tiles_w; p.nth = (int)n_tiles_h; p.H = (int)H; p.W = (int)W;
  p.psx = pixel_size_x; p.psy = pixel_size_y; p.thresh = thresh; p.tile_side = (int)tile_size;
  p.tile_order = tile_order;
  return launch_bwd(MODE_RGBD, 1, p, (hipStream_t)stream);
}

int gsgen_vol_render_scalar_backward(uint32_t N, uint32_t D, const float *mean, const float *cov,
                                     const float *scalar, const float *alpha, const int *start,
                                     const int *end, const int *gaussian_ids, const float *out,
                                     float *grad_mean, float *grad_cov, float *grad_scalar,
                                     float *grad_alpha, const float *grad_out, const float *topleft,
                                     uint32_t tile_size, uint32_t n_tiles_h, uint32_t n_tiles_w,
                                     float pixel_size_x, float pixel_size_y, uint32_t H, uint32_t W,
                                     float thresh, gsgen_stream_t stream) {
  if (int e = check_common(tile_size, start, end, out)) return e;
  if (N == 0 || D == 0) return 0;
  CompParams p{};
  p.mean = mean; p.cov = cov; p.col = scalar; p.alpha = alpha;
  p.start = start; p.end = end; p.ids = gaussian_ids; p.topleft = topleft;
  p.final_img = out; p.grad_out = grad_out;
  p.g_mean = grad_mean; p.g_cov = grad_cov; p.g_col = grad_scalar; p.g_alpha = grad_alpha;
  p.ntw = (int)n_tiles_w; p.nth = (int)n_tiles_h; p.H = (int)H; p.W = (int)W;
  p.psx = pixel_size_x; p.psy = pixel_size_y; p.thresh = thresh; p.tile_side = (int)tile_size;
  return launch_bwd(MODE_SCALAR, 1, p, (hipStream_t)stream);
}

int gsgen_vol_render_backward_sh_ordered(uint32_t N, uint32_t D, const float *mean, const float *cov,
                                         const float *sh_coeffs, const float *alpha, const int *start,
                                         const int *end, const int *gaussian_ids, const float *out,
                                         float *grad_mean, float *grad_cov, float *grad_sh_coeffs,
                                         float *grad_alpha, const float *grad_out, const float *topleft,
                                         const float *c2w, uint32_t tile_size, uint32_t n_tiles_h,
                                         uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y,
                                         uint32_t H, uint32_t W, uint32_t C, float thresh,
                                         const float *bg_rgb, const uint32_t *tile_order,
                                         gsgen_stream_t stream) {
  return gsgen_vol_render_backward_sh_segmented(N, D, mean, cov, sh_coeffs, alpha, start, end, gaussian_ids, out,
                                                grad_mean, grad_cov, grad_sh_coeffs, grad_alpha, grad_out, topleft,
                                                c2w, tile_size, n_tiles_h, n_tiles_w, pixel_size_x, pixel_size_y, H,
                                                W, C, thresh, bg_rgb, tile_order, nullptr, 0, stream);
}

int gsgen_vol_render_backward_sh_segmented(uint32_t N, uint32_t D, const float *mean, const float *cov,
                                           const float *sh_coeffs, const float *alpha, const int *start,
                                           const int *end, const int *gaussian_ids, const float *out,
                                           float *grad_mean, float *grad_cov, float *grad_sh_coeffs,
                                           float *grad_alpha, const float *grad_out, const float *topleft,
                                           const float *c2w, uint32_t tile_size, uint32_t n_tiles_h,
                                           uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y,
                                           uint32_t H, uint32_t W, uint32_t C, float thresh,
                                           const float *bg_rgb, const uint32_t *tile_order,
                                           const void *segment_workspace, uint32_t n_segments,
                                           gsgen_stream_t stream) {
  return gsgen_vol_render_backward_sh_bounded(N, D, mean, cov, sh_coeffs, alpha, start, end, gaussian_ids, out,
                                              grad_mean, grad_cov, grad_sh_coeffs, grad_alpha, grad_out, topleft,
                                              c2w, tile_size, n_tiles_h, n_tiles_w, pixel_size_x, pixel_size_y, H,
                                              W, C, thresh, bg_rgb, tile_order, segment_workspace, n_segments, nullptr,
                                              stream);
}

int gsgen_vol_render_backward_sh_bounded(uint32_t N, uint32_t D, const float *mean, const float *cov,
                                         const float *sh_coeffs, const float *alpha, const int *start,
                                         const int *end, const int *gaussian_ids, const float *out,
                                         float *grad_mean, float *grad_cov, float *grad_sh_coeffs,
                                         float *grad_alpha, const float *grad_out, const float *topleft,
                                         const float *c2w, uint32_t tile_size, uint32_t n_tiles_h,
                                         uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y,
                                         uint32_t H, uint32_t W, uint32_t C, float thresh,
                                         const float *bg_rgb, const uint32_t *tile_order,
                                         const void *segment_workspace, uint32_t n_segments,
                                         const float *sh_l1_bound, gsgen_stream_t stream) {
  return gsgen_vol_render_backward_sh_routed(N, D, mean, cov, sh_coeffs, alpha, start, end, gaussian_ids, out, grad_mean, grad_cov,
                                             grad_sh_coeffs, grad_alpha, grad_out, topleft, c2w, tile_size, n_tiles_h, n_tiles_w,
                                             pixel_size_x, pixel_size_y, H, W, C, thresh, bg_rgb, tile_order, segment_workspace,
                                             n_segments, sh_l1_bound, nullptr, stream);
}

int gsgen_vol_render_backward_sh_routed(uint32_t N, uint32_t D, const float *mean, const float *cov,
                                        const float *sh_coeffs, const float *alpha, const int *start,
                                        const int *end, const int *gaussian_ids, const float *out,
                                        float *grad_mean, float *grad_cov, float *grad_sh_coeffs,
                                        float *grad_alpha, const float *grad_out, const float *topleft,
                                        const float *c2w, uint32_t tile_size, uint32_t n_tiles_h,
                                        uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y,
                                        uint32_t H, uint32_t W, uint32_t C, float thresh,
                                        const float *bg_rgb, const uint32_t *tile_order,
                                        const void *segment_workspace, uint32_t n_segments,
                                        const float *sh_l1_bound, const float *sh_row_bounds, gsgen_stream_t stream) {
  (void)bg_rgb;  // the background only enters through `out` (= final incl. bg*T)
  if (int e = check_common(tile_size, start, end, out)) return e;
  if (n_segments > 1 && segment_workspace == nullptr) return GSGEN_EINVAL;
  if (C < 1 || C > 4) return GSGEN_EUNSUPPORTED;
  if (!c2w) return GSGEN_EINVAL;
  if (N == 0 || D == 0) return 0;
  CompParams p{};
  p.mean = mean; p.cov = cov; p.col = sh_coeffs; p.alpha = alpha;
  p.start = start; p.end = end; p.ids = gaussian_ids; p.topleft = topleft; p.rot = c2w;
  p.final_img = out; p.grad_out = grad_out;
  p.g_mean = grad_mean; p.g_cov = grad_cov; p.g_col = grad_sh_coeffs; p.g_alpha = grad_alpha;
  p.ntw = (int)n_tiles_w; p.nth = (int)n_tiles_h; p.H = (int)H; p.W = (int)W;
  p.psx = pixel_size_x; p.psy = pixel_size_y; p.thresh = thresh; p.tile_side = (int)tile_size;
  p.tile_order = tile_order;
  p.sh_bound = (C == 4) ? sh_l1_bound : nullptr;  // (both given: the view's bound first, then the tile's list -- as the forward)
  p.sh_rows = (C == 4) ? sh_row_bounds : nullptr;
  if (n_segments > 1) {
    p.nseg = (int)n_segments;
    p.ckpt = reinterpret_cast<float4 *>(const_cast<void *>(segment_workspace));
    p.stop = reinterpret_cast<int *>(p.ckpt + (size_t)n_tiles_h * n_tiles_w * 256 * n_segments);
  }
  return launch_bwd(MODE_SH, (int)C, p, (hipStream_t)stream);
}

int gsgen_vol_render_backward_sh(uint32_t N, uint32_t D, const float *mean, const float *cov,
                                 const float *sh_coeffs, const float *alpha, const int *start,
                                 const int *end, const int *gaussian_ids, const float *out,
                                 float *grad_mean, float *grad_cov, float *grad_sh_coeffs,
                                 float *grad_alpha, const float *grad_out, const float *topleft,
                                 const float *c2w, uint32_t tile_size, uint32_t n_tiles_h,
                                 uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y,
                                 uint32_t H, uint32_t W, uint32_t C, float thresh,
                                 const float *bg_rgb, gsgen_stream_t stream) {
  return gsgen_vol_render_backward_sh_ordered(N, D, mean, cov, sh_coeffs, alpha, start, end, gaussian_ids, out,
                                              grad_mean, grad_cov, grad_sh_coeffs, grad_alpha, grad_out, topleft,
                                              c2w, tile_size, n_tiles_h, n_tiles_w, pixel_size_x, pixel_size_y,
                                              H, W, C, thresh, bg_rgb, nullptr, stream);
}

}  // extern "C"
