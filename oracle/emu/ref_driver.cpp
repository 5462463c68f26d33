// ref_driver.cpp -- C entry points around the REFERENCE's own launchers (gs/src/include/*.h of
// /root/reference, compiled for the CPU by oracle/ref_build.py).  Pointers are host pointers.
// Only the reference's `*_cuda(...)` host launchers are called, exactly as gs/src/render.cu
// calls them; nothing of the reference is restated here.
#include "common.h"
#include "kernels.h"
#include "culling.h"
#include "aabb_culling.h"
#include "vol_render.h"
#include "vol_render_scalar.h"
#include "vol_render_sh.h"
#include "vol_render_bg.h"
#include "tile_ops.h"

extern "C" {

void ref_culling_gaussian_bsphere(uint32_t N, float *mean, float *qvec, float *svec, float *normal, float *pts,
                                  bool *mask, float thresh) {
  culling_gaussian_bsphere_cuda(N, mean, qvec, svec, normal, pts, mask, thresh);
}

void ref_tile_culling_aabb_start_end(uint32_t N, uint32_t D, uint32_t nth, uint32_t ntw, int *gaussian_ids, int *start,
                                     int *end, int *aabb_topleft, int *aabb_bottomright, float *depth) {
  tile_culling_aabb_start_end_cuda(N, D, nth, ntw, gaussian_ids, start, end, aabb_topleft, aabb_bottomright, depth);
}

void ref_vol_render_start_end_with_T(uint32_t N, uint32_t D, float *mean, float *cov, float *color, float *alpha,
                                     int *start, int *end, int *ids, float *out, float *topleft, uint32_t tile_size,
                                     uint32_t nth, uint32_t ntw, float psx, float psy, uint32_t H, uint32_t W,
                                     float thresh, float *T) {
  tile_based_vol_rendering_start_end_cuda_with_T(N, D, mean, cov, color, alpha, start, end, ids, out, topleft,
                                                 tile_size, nth, ntw, psx, psy, H, W, thresh, T);
}

void ref_vol_render_backward_start_end(uint32_t N, uint32_t D, float *mean, float *cov, float *color, float *alpha,
                                       int *start, int *end, int *ids, float *out, float *g_mean, float *g_cov,
                                       float *g_color, float *g_alpha, float *grad_out, float *topleft,
                                       uint32_t tile_size, uint32_t nth, uint32_t ntw, float psx, float psy,
                                       uint32_t H, uint32_t W, float thresh) {
  tile_based_vol_rendering_backward_start_end_cuda(N, D, mean, cov, color, alpha, start, end, ids, out, g_mean, g_cov,
                                                   g_color, g_alpha, grad_out, topleft, tile_size, nth, ntw, psx, psy,
                                                   H, W, thresh);
}

void ref_vol_render_scalar(uint32_t N, uint32_t D, float *mean, float *cov, float *scalar, float *alpha, int *start,
                           int *end, int *ids, float *out, float *topleft, uint32_t tile_size, uint32_t nth,
                           uint32_t ntw, float psx, float psy, uint32_t H, uint32_t W, float thresh, float *T) {
  vol_rendering_scalar_cuda(N, D, mean, cov, scalar, alpha, start, end, ids, out, topleft, tile_size, nth,
                                       ntw, psx, psy, H, W, thresh, T);
}

void ref_vol_render_scalar_backward(uint32_t N, uint32_t D, float *mean, float *cov, float *scalar, float *alpha,
                                    int *start, int *end, int *ids, float *out, float *g_mean, float *g_cov,
                                    float *g_scalar, float *g_alpha, float *grad_out, float *topleft,
                                    uint32_t tile_size, uint32_t nth, uint32_t ntw, float psx, float psy, uint32_t H,
                                    uint32_t W, float thresh) {
  vol_rendering_scalar_backward_cuda(N, D, mean, cov, scalar, alpha, start, end, ids, out, g_mean, g_cov,
                                                g_scalar, g_alpha, grad_out, topleft, tile_size, nth, ntw, psx, psy, H,
                                                W, thresh);
}

#define SH_SWITCH(CALL)          \
  switch (C) {                   \
    case 1: CALL(1); break;      \
    case 2: CALL(2); break;      \
    case 3: CALL(3); break;      \
    case 4: CALL(4); break;      \
    default: break;              \
  }

void ref_vol_render_sh(uint32_t N, uint32_t D, float *mean, float *cov, float *sh, float *alpha, int *start, int *end,
                       int *ids, float *out, float *topleft, float *c2w, uint32_t tile_size, uint32_t nth,
                       uint32_t ntw, float psx, float psy, uint32_t H, uint32_t W, uint32_t C, float thresh,
                       float *bg_rgb) {
#define FWD(CC)                                                                                                    \
  if (bg_rgb)                                                                                                      \
    tile_based_vol_rendering_sh_cuda_with_bg<CC>(N, D, mean, cov, sh, alpha, start, end, ids, out, topleft, c2w,   \
                                                 tile_size, nth, ntw, psx, psy, H, W, thresh, bg_rgb, nullptr);    \
  else                                                                                                             \
    tile_based_vol_rendering_sh_cuda<CC>(N, D, mean, cov, sh, alpha, start, end, ids, out, topleft, c2w, tile_size, \
                                         nth, ntw, psx, psy, H, W, thresh, nullptr)
  SH_SWITCH(FWD)
}

void ref_vol_render_backward_sh(uint32_t N, uint32_t D, float *mean, float *cov, float *sh, float *alpha, int *start,
                                int *end, int *ids, float *out, float *g_mean, float *g_cov, float *g_sh,
                                float *g_alpha, float *grad_out, float *topleft, float *c2w, uint32_t tile_size,
                                uint32_t nth, uint32_t ntw, float psx, float psy, uint32_t H, uint32_t W, uint32_t C,
                                float thresh, float *bg_rgb) {
#define BWD(CC)                                                                                                       \
  if (bg_rgb)                                                                                                         \
    tile_based_vol_rendering_backward_sh_cuda_with_bg<CC>(N, D, mean, cov, sh, alpha, start, end, ids, out, g_mean,   \
                                                          g_cov, g_sh, g_alpha, grad_out, topleft, c2w, tile_size,    \
                                                          nth, ntw, psx, psy, H, W, thresh, bg_rgb, nullptr);         \
  else                                                                                                                \
    tile_based_vol_rendering_backward_sh_cuda<CC>(N, D, mean, cov, sh, alpha, start, end, ids, out, g_mean, g_cov,    \
                                                  g_sh, g_alpha, grad_out, topleft, c2w, tile_size, nth, ntw, psx,    \
                                                  psy, H, W, thresh, nullptr)
  SH_SWITCH(BWD)
}


// legacy binning (render.cu:46-176)
void ref_count_num_gaussians_each_tile(uint32_t N, float *mean, float *cov, float *topleft, uint32_t tile_size,
                                       uint32_t nth, uint32_t ntw, float psx, float psy, int *num_gaussians,
                                       float thresh) {
  count_tiled_gaussians_cuda_sm(N, mean, cov, topleft, tile_size, nth, ntw, psx, psy, num_gaussians, thresh);
}
void ref_count_num_gaussians_each_tile_bcircle(uint32_t N, float *mean, float *radius, float *topleft,
                                               uint32_t tile_size, uint32_t nth, uint32_t ntw, float psx, float psy,
                                               int *num_gaussians) {
  count_tiled_gaussians_bcircle_cuda_sm(N, mean, radius, topleft, tile_size, nth, ntw, psx, psy, num_gaussians);
}
void ref_prepare_image_sort(uint32_t N, uint32_t N_with_dub, int *gaussian_ids, double *tiledepth, float *depth,
                            int *tile_n_gaussians, int *offset, float *mean, float *radius, float *topleft,
                            uint32_t tile_size, uint32_t nth, uint32_t ntw, float psx, float psy) {
  prepare_image_sort_cuda(N, N_with_dub, gaussian_ids, tiledepth, depth, tile_n_gaussians, offset, mean, radius,
                          topleft, tile_size, nth, ntw, psx, psy);
}
void ref_image_sort(uint32_t N, uint32_t N_with_dub, int *gaussian_ids, double *tiledepth, float *depth,
                    int *tile_n_gaussians, int *offset, float *mean, float *cov, float *topleft, uint32_t tile_size,
                    uint32_t nth, uint32_t ntw, float psx, float psy, float thresh) {
  image_sort_cuda(N, N_with_dub, gaussian_ids, tiledepth, depth, tile_n_gaussians, offset, mean, cov, topleft,
                  tile_size, nth, ntw, psx, psy, thresh);
}
}  // extern "C"
