/*
 * gsgen_hip.h -- C ABI of libgsgen_hip.so, the MI355X (gfx950) rasterizer behind GSGEN's
 * `_gs` extension.
 *
 * The reference binds this path through a pybind11/torch module (`gs/src/bindings.cpp:5-82`,
 * prototypes `gs/src/render.h:3-155`) whose wrappers (`gs/src/render.cu`) unpack tensors to
 * raw pointers and call `*_cuda(...)` launchers.  The entry points below are those launchers'
 * replacement: plain pointers + sizes + a hipStream_t, no torch types.  All pointers are DEVICE
 * pointers unless stated.  Every function returns 0 on success, a positive hipError_t value
 * when a HIP call failed, or a negative GSGEN_E* code for a violated precondition; nothing
 * calls exit() (the reference's cudaCheck does: gs/src/include/common.h:56-72).
 *
 * Ownership follows the reference: the caller allocates every array, outputs
 * included, and pre-initialises them as the reference's Python does (out = 0, T = 1,
 * start/end = -1, grad_* = 0 -- the backward entry points ACCUMULATE into grad_*).
 * All work is enqueued on `stream`; no entry point synchronises the device or allocates.
 *
 * Layouts (row-major, fp32 unless noted): mean2d [N,2]; cov2d [N,2,2]; color [N,3];
 * alpha [N]; scalar [N]; sh_coeffs [N,3,C*C]; start/end int32 [n_tiles_h*n_tiles_w];
 * gaussian_ids int32 [D]; out [H,W,3] (or [H,W] scalar); T [H,W]; topleft [2]; bg_rgb [3].
 * tile_size: 16 is the reference's only configured value (conf/base.yaml:132) and the size every tuned kernel variant,
 * the batched, segmented and fused entry points are built for.  The per-camera entry points of the `_gs` surface
 * (gsgen_tile_culling_aabb_count, gsgen_vol_render_start_end_with_T / _backward_start_end, _scalar / _scalar_backward,
 * _sh / _backward_sh) take ANY tile_size from 1 to 32, as the reference's launch of tile_size x tile_size <= 1024 threads does
 * (vol_render.h:1001-1004): 8 runs one wavefront per tile, every other side a four-wavefront workgroup over a 32 x 32 patch of
 * which the caller's tile is the top-left part (untuned: a compatibility path); binning works on tile-index rectangles and is
 * size-agnostic.  tile_size 0 or > 32 returns GSGEN_EUNSUPPORTED.  Per-pixel results do not depend on the tile size.
 */
#ifndef GSGEN_HIP_H
#define GSGEN_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the entry points declared in this header are exported (the declarations
 * carry default visibility; the cross-file helpers of the implementation stay out of the dynamic symbol table). */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

typedef void *gsgen_stream_t; /* hipStream_t; NULL = the legacy default stream */

#define GSGEN_EUNSUPPORTED (-2) /* tile_size not in 1 .. 32 (16 only: batched / segmented / fused), C not in 1..4 */
#define GSGEN_EINVAL (-3)       /* null pointer / inconsistent sizes */
#define GSGEN_EWORKSPACE (-4)   /* workspace too small */

const char *gsgen_version(void);
/* HIP error string for positive return codes, own text for negative ones. */
const char *gsgen_error_string(int code);
/* Which compiled kernel variant a compositing launch runs in this process (gsgen_amd/csrc/composite.hip "launch
 * helpers"; the library reads nothing from the environment).  stage: "sh_fwd", "sh_bwd", "sh_fwd_batch", "sh_bwd_batch"
 * (+ "_poly": the polynomial-basis kernel of a bounded enqueue, see gsgen_vol_render_sh_batch_bounded), "rgb_fwd",
 * "rgb_bwd", "rgbd_fwd_batch", "rgbd_bwd_batch".  Writes a NUL-terminated name into out[out_bytes] and returns its
 * length, 0 for an unknown stage.  Host-only; no counterpart in the reference (bench.py reports it next to the measured
 * kernel time). */
int gsgen_kernel_variant(const char *stage, uint32_t C, uint32_t n_segments, char *out, size_t out_bytes);

/* ---- frustum cull ------------------------------------------------------------------
 * replaces culling_gaussian_bsphere, gs/src/render.h:3 -> render.cu:16-44 ->
 * culling_gaussian_bsphere_cuda culling.h:21-33.  mask is bool[N] (1 byte each). */
int gsgen_culling_gaussian_bsphere(uint32_t N, const float *mean, const float *qvec,
                                   const float *svec, const float *normal /*[6,3]*/,
                                   const float *pts /*[6,3]*/, uint8_t *mask, float thresh,
                                   gsgen_stream_t stream);

/* ---- tile binning + per-tile depth sort ---------------------------------------------
 * replaces tile_culling_aabb_start_end, render.h:61 -> render.cu:381-398 ->
 * tile_culling_aabb_start_end_cuda aabb_culling.h:192-260.  D = gaussian_ids length =
 * sum over Gaussians of (br-tl+1) products (the caller's N_with_dub).  Unlike the reference
 * there is no cudaMalloc/cudaFree/blocking memcpy: temporaries live in `workspace`.
 * D (and D_cap of the fused entry points) must not exceed INT32_MAX -- start / end and the list positions are int32 as in
 * the reference -- else GSGEN_EINVAL.  The fused entry points report the frame's pair count in *total; counts that do not
 * fit 32 bits SATURATE at 2^32 - 1 (and read as "does not fit": nothing is binned), they never wrap. */
size_t gsgen_tile_culling_workspace_bytes(uint32_t N, uint32_t D, uint32_t n_tiles);
int gsgen_tile_culling_aabb_start_end(uint32_t N, uint32_t D, uint32_t n_tiles_h,
                                      uint32_t n_tiles_w, const int *aabb_topleft,
                                      const int *aabb_bottomright, const float *depth,
                                      int *gaussian_ids, int *start, int *end, void *workspace,
                                      size_t workspace_bytes, gsgen_stream_t stream);

/* ---- RGB compositing -----------------------------------------------------------------
 * forward with T: tile_based_vol_rendering_start_end_with_T render.h:149 -> render.cu:989-1012
 *   -> ..._cuda_with_T vol_render.h:1064-1078.  T may be NULL (= tile_based_vol_rendering_start_end,
 *   render.h:65 -> vol_render.h:849-864).
 * backward: tile_based_vol_rendering_backward_start_end render.h:73 -> render.cu:426-482 ->
 *   vol_render.h:975-992.  `out` is the saved forward image INCLUDING T*bg. */
int gsgen_vol_render_start_end_with_T(uint32_t N, uint32_t D, const float *mean, const float *cov,
                                      const float *color, const float *alpha, const int *start,
                                      const int *end, const int *gaussian_ids, float *out,
                                      const float *topleft, uint32_t tile_size, uint32_t n_tiles_h,
                                      uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y,
                                      uint32_t H, uint32_t W, float thresh, float *T,
                                      gsgen_stream_t stream);
int gsgen_vol_render_backward_start_end(uint32_t N, uint32_t D, const float *mean, const float *cov,
                                        const float *color, const float *alpha, const int *start,
                                        const int *end, const int *gaussian_ids, const float *out,
                                        float *grad_mean, float *grad_cov, float *grad_color,
                                        float *grad_alpha, const float *grad_out,
                                        const float *topleft, uint32_t tile_size,
                                        uint32_t n_tiles_h, uint32_t n_tiles_w, float pixel_size_x,
                                        float pixel_size_y, uint32_t H, uint32_t W, float thresh,
                                        gsgen_stream_t stream);

/* ---- scalar compositing (depth / opacity / z^2 heads) ---------------------------------
 * tile_based_vol_rendering_scalar render.h:131 -> vol_render_scalar.h:47-102;
 * tile_based_vol_rendering_scalar_backward render.h:139 -> vol_render_scalar.h:148-234. */
int gsgen_vol_render_scalar(uint32_t N, uint32_t D, const float *mean, const float *cov,
                            const float *scalar, const float *alpha, const int *start,
                            const int *end, const int *gaussian_ids, float *out,
                            const float *topleft, uint32_t tile_size, uint32_t n_tiles_h,
                            uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y, uint32_t H,
                            uint32_t W, float thresh, float *T, gsgen_stream_t stream);
int gsgen_vol_render_scalar_backward(uint32_t N, uint32_t D, const float *mean, const float *cov,
                                     const float *scalar, const float *alpha, const int *start,
                                     const int *end, const int *gaussian_ids, const float *out,
                                     float *grad_mean, float *grad_cov, float *grad_scalar,
                                     float *grad_alpha, const float *grad_out, const float *topleft,
                                     uint32_t tile_size, uint32_t n_tiles_h, uint32_t n_tiles_w,
                                     float pixel_size_x, float pixel_size_y, uint32_t H, uint32_t W,
                                     float thresh, gsgen_stream_t stream);

/* ---- spherical-harmonic compositing ---------------------------------------------------
 * tile_based_vol_rendering_sh render.h:83 / _sh_with_bg render.h:113 -> vol_render_sh.h:171-265,
 * vol_render_bg.h:12-127;  backward render.h:91 / :121 -> vol_render_sh.h:353-480,
 * vol_render_bg.h:131-265.  `c2w` is read as the reference reads it: 9 consecutive floats =
 * three packed rows (vol_render_sh.h:48-55).  bg_rgb == NULL selects the no-background
 * variant; with bg_rgb the forward adds bg*T and writes bg to empty tiles, and `out` handed to
 * the backward includes it.  T (optional, may be NULL) additionally receives the final
 * transmittance -- an output the reference does not have. */
int gsgen_vol_render_sh(uint32_t N, uint32_t D, const float *mean, const float *cov,
                        const float *sh_coeffs, const float *alpha, const int *start,
                        const int *end, const int *gaussian_ids, float *out, const float *topleft,
                        const float *c2w, uint32_t tile_size, uint32_t n_tiles_h,
                        uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y, uint32_t H,
                        uint32_t W, uint32_t C, float thresh, const float *bg_rgb, float *T,
                        gsgen_stream_t stream);
int gsgen_vol_render_backward_sh(uint32_t N, uint32_t D, const float *mean, const float *cov,
                                 const float *sh_coeffs, const float *alpha, const int *start,
                                 const int *end, const int *gaussian_ids, const float *out,
                                 float *grad_mean, float *grad_cov, float *grad_sh_coeffs,
                                 float *grad_alpha, const float *grad_out, const float *topleft,
                                 const float *c2w, uint32_t tile_size, uint32_t n_tiles_h,
                                 uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y,
                                 uint32_t H, uint32_t W, uint32_t C, float thresh,
                                 const float *bg_rgb, gsgen_stream_t stream);

/* ---- additive entry points (no native counterpart in the reference) -------------------
 * EWA projection: the reference does this in PyTorch (gs/renderer.py:366-421); same outputs.
 * c2w is a device pointer to the row-major [3,4] pose.  JW may be NULL.  The backward
 * OVERWRITES g_mean [N,3], g_qvec [N,4], g_svec [N,3]; g_depth may be NULL. */
int gsgen_project_gaussians(uint32_t N, const float *mean, const float *qvec, const float *svec,
                            const float *c2w, float *mean2d, float *cov2d, float *JW, float *depth,
                            gsgen_stream_t stream);
int gsgen_project_gaussians_backward(uint32_t N, const float *mean, const float *qvec,
                                     const float *svec, const float *c2w, int detach_depth,
                                     const float *g_mean2d, const float *g_cov2d,
                                     const float *g_depth, float *g_mean, float *g_qvec,
                                     float *g_svec, gsgen_stream_t stream);
/* Same, rows with mask[i]==0 get zero gradients (mask may be NULL): the backward leg of
 * gsgen_frame_geometry, which keeps every array full-N instead of compacting by the mask. */
int gsgen_project_gaussians_backward_masked(uint32_t N, const float *mean, const float *qvec,
                                            const float *svec, const float *c2w, int detach_depth,
                                            const uint8_t *mask, const float *g_mean2d,
                                            const float *g_cov2d, const float *g_depth,
                                            float *g_mean, float *g_qvec, float *g_svec,
                                            gsgen_stream_t stream);
/* Same, ADDED atomically into caller-zeroed g_mean/g_qvec/g_svec (masked-out rows untouched):
 * the cameras of a batch accumulate into one set of parameter gradients, possibly from
 * concurrent streams. */
int gsgen_project_gaussians_backward_accum(uint32_t N, const float *mean, const float *qvec,
                                           const float *svec, const float *c2w, int detach_depth,
                                           const uint8_t *mask, const float *g_mean2d,
                                           const float *g_cov2d, const float *g_depth,
                                           float *g_mean, float *g_qvec, float *g_svec,
                                           gsgen_stream_t stream);
/* The cameras of a batch in ONE launch: g_* (OVERWRITTEN) = sum over the views of the masked backward
 * above, each Gaussian summed in registers in view order (no atomics, no zero-fill, one pass over the
 * parameters).  c2w / mask / g_mean2d / g_cov2d / g_depth are HOST arrays of n_views DEVICE pointers
 * (read before the call returns); mask and g_depth may be NULL as arrays or per view.  c2w[v] is the
 * row-major [3,4] pose (the first 12 floats of a gsgen_frame_geometry camera block). */
int gsgen_project_gaussians_backward_batch(uint32_t n_views, uint32_t N, const float *mean, const float *qvec,
                                           const float *svec, const float *const *c2w, int detach_depth,
                                           const uint8_t *const *mask, const float *const *g_mean2d,
                                           const float *const *g_cov2d, const float *const *g_depth,
                                           float *g_mean, float *g_qvec, float *g_svec, gsgen_stream_t stream);
/* torch.optim.Adam step (no weight decay, no amsgrad: gs/gaussian_splatting.py:398-419,
 * conf/base.yaml:8-11) over one flat fp32 vector that holds every parameter field back to back,
 * in place.  group_end[k] (HOST array, ascending, last == n) closes parameter group k, group_lr[k]
 * (HOST) is its learning rate for this step; step counts from 1.  At most 8 groups. */
/* gsgen_project_gaussians_backward_batch behind the fused RGB + heads compositing (gsgen_vol_render_rgbd_backward_batch): takes
 * each view's channel gradients g_chan6[v] [N,6] = d L / d (r, g, b, depth, 1, depth^2) and its depths depth[v] [N] and forms
 * d L / d depth = g3 + 2 depth g5 itself (the depth head and the depth^2 head, gs/gaussian_splatting.py:1334-1403), and writes
 * the colour gradient g_color [N,3] = sum over the views of g_chan6[v][:, 0:3] in the same pass -- the two torch kernels the
 * caller ran between the compositing backward and the projection backward.  Rows a view's mask excludes contribute nothing. */
int gsgen_project_gaussians_backward_batch_heads(uint32_t n_views, uint32_t N, const float *mean, const float *qvec,
                                                 const float *svec, const float *const *c2w, int detach_depth,
                                                 const uint8_t *const *mask, const float *const *g_mean2d,
                                                 const float *const *g_cov2d, const float *const *g_chan6,
                                                 const float *const *depth, float *g_mean, float *g_qvec, float *g_svec,
                                                 float *g_color, gsgen_stream_t stream);
/* The same behind the MOMENT form of that compositing backward (gsgen_vol_render_rgbd_backward_batch_moments, round 6):
 * g_mom2[v] [N,2] = (Mu, Mv) and g_mom4[v] [N,4] = (Muu, Muv, Mvv, -) are sums over the view's pixels of the per-pixel weight
 * g = d L / d (a G) * a G against the whitened offsets (u, v) of the Cholesky-form Gaussian; with Sigma^-1 d = (k0 u, k1 u + k2 v),
 * k from the view's cov2d[v] [N,2,2] exactly as the compositing kernels prepare it, the reference's sums (kernels.h:394-418)
 *   d mean2d = (k0 Mu, k1 Mu + k2 Mv),  d cov2d = 0.5 [k0^2 Muu, k0 (k1 Muu + k2 Muv); same, k1^2 Muu + 2 k1 k2 Muv + k2^2 Mvv]
 * are formed here in fp64, once per (view, Gaussian), and chained through the projection as above.  g_chan6[v][:, 3] holds the
 * folded d L / d depth of the three depth heads (columns 4, 5 stay zero).  g_mom2[v] is OVERWRITTEN with the view's
 * d L / d mean2d (what gsgen_densify_update_batch reads, gs/gaussian_splatting.py:464-469). */
/* ... and behind the SH kernels' moment form (gsgen_vol_render_backward_sh_batch_routed_moments): g_mom2[v] = sum g (tx, ty),
 * g_mom4[v] = sum g (tx tx, tx ty, ty ty, -) with (tx, ty) = det Sigma^-1 d; d mean2d = M1 / det, d cov2d = 0.5 M2 / det^2 with the
 * fp32 determinant of cov2d[v] the kernels use (kernels.h:179).  g_mom2[v] is overwritten with d L / d mean2d as above. */
int gsgen_project_gaussians_backward_batch_moments_sh(uint32_t n_views, uint32_t N, const float *mean, const float *qvec,
                                                      const float *svec, const float *const *c2w, int detach_depth,
                                                      const uint8_t *const *mask, float *const *g_mom2,
                                                      const float *const *g_mom4, const float *const *cov2d, float *g_mean,
                                                      float *g_qvec, float *g_svec,
                                                      float *stat_grad_accum /* or NULL: [N], += sum over the views of |d L / d mean2d| */,
                                                      float *stat_cnt /* or NULL: [N], += the views that saw the Gaussian (gs/gaussian_splatting.py:464-469) */,
                                                      gsgen_stream_t stream);
int gsgen_project_gaussians_backward_batch_heads_moments(uint32_t n_views, uint32_t N, const float *mean, const float *qvec,
                                                         const float *svec, const float *const *c2w, int detach_depth,
                                                         const uint8_t *const *mask, float *const *g_mom2,
                                                         const float *const *g_mom4, const float *const *g_chan6,
                                                         const float *const *depth, const float *const *cov2d,
                                                         const float *const *chol /* or NULL: the views' gsgen_geometry_view::chol */,
                                                         float *g_mean, float *g_qvec, float *g_svec, float *g_color,
                                                         float *stat_grad_accum /* or NULL, as above */, float *stat_cnt /* or NULL */,
                                                         gsgen_stream_t stream);
/* The model's parameter activations (utils/activations.py:36-57; svec / alpha / color = act(raw), gs/gaussian_splatting.py:113-124) for
 * all three fields in one launch, and their backward in one (g_* hold d L / d activated on entry, d L / d raw on return).  Codes:
 * 0 nothing, 1 exp, 2 sigmoid, 3 abs, 4 relu, 5 softplus, 6 biased_relu, 7 biased_abs.  svec / color are [N,3], alpha [N].  The
 * reference runs three torch kernels forward and three autograd nodes backward per step: host time, not device time. */
int gsgen_activate_fields(uint32_t N, const float *svec_raw, const float *alpha_raw, const float *color_raw, int svec_act,
                          int alpha_act, int color_act, float *svec, float *alpha, float *color, gsgen_stream_t stream);
int gsgen_activate_fields_backward(uint32_t N, const float *svec_raw, const float *alpha_raw, const float *color_raw,
                                   const float *svec, const float *alpha, const float *color, int svec_act, int alpha_act,
                                   int color_act, float *g_svec, float *g_alpha, float *g_color, gsgen_stream_t stream);
int gsgen_adam_step(uint64_t n, float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                    uint32_t n_groups, const uint64_t *group_end, const float *group_lr, float beta1,
                    float beta2, float eps, uint32_t step, gsgen_stream_t stream);
/* The same step with its per-step scalars read from DEVICE memory, so that a captured hipGraph of an optimisation step replays with
 * the learning rates and bias corrections of the step it is replayed for (gsgen_amd.graph.CapturedStep; gsgen_adam_step bakes them
 * into its kernel arguments).  scalars [9] (device): step_size of group 0 .. 7 (lr / (1 - beta1^step), unused groups ignored) and, at
 * [8], sqrt(1 - beta2^step) -- the nine floats gsgen_adam_step_scalars computes on the HOST (same double arithmetic as
 * gsgen_adam_step) into out9 for the caller to copy there. */
int gsgen_adam_step_scalars(uint32_t n_groups, const float *group_lr, float beta1, float beta2, uint32_t step, float *out9);
int gsgen_adam_step_device_scalars(uint64_t n, float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                                   uint32_t n_groups, const uint64_t *group_end, float beta1, float beta2, float eps,
                                   const float *scalars, gsgen_stream_t stream);
/* Densification statistics of one camera (gs/gaussian_splatting.py:1240-1245, :464-469), rows
 * aligned with mask [N] (NULL = all rows):
 *   max_radii2d[i] = max(max_radii2d[i], m + sqrt(max(m^2 - det(cov2d_i), 0))), m = tr/2
 *   grad_accum[i] += |grad_mean2d_i|_2 ;  cnt[i] += 1
 * Either pair (cov2d, max_radii2d) / (grad_mean2d, grad_accum[, cnt]) may be NULL together.
 * torch.det's LU rounding is not reproduced: det = c00*c11 - c01*c10 in fp32. */
int gsgen_densify_update(uint32_t N, const float *cov2d, const float *grad_mean2d,
                         const uint8_t *mask, float *max_radii2d, float *grad_accum, float *cnt,
                         gsgen_stream_t stream);
/* The same for the cameras of a batch in ONE launch (maximum / sum / count folded over the views in
 * registers, then one atomic each per Gaussian).  cov2d / grad_mean2d / mask: HOST arrays of n_views DEVICE
 * pointers, read before the call returns; the NULL conventions of gsgen_densify_update apply to the arrays
 * (mask may also be NULL per view). */
int gsgen_densify_update_batch(uint32_t n_views, uint32_t N, const float *const *cov2d,
                               const float *const *grad_mean2d, const uint8_t *const *mask, float *max_radii2d,
                               float *grad_accum, float *cnt, gsgen_stream_t stream);
/* AABB tile rectangles + pair count on the device (gs/culling.py:8-37 without the .item()
 * host sync): writes aabb_topleft/bottomright int32 [N,2] and *total (device uint32, zeroed
 * inside the call) = N_with_dub. */
int gsgen_tile_culling_aabb_count(uint32_t N, const float *mean2d, const float *cov2d,
                                  uint32_t tile_size, float fx, float fy, float cx, float cy,
                                  uint32_t w, uint32_t h, float D, int *aabb_topleft,
                                  int *aabb_bottomright, uint32_t *total, gsgen_stream_t stream);

/* ---- the reference's older binning pipeline (compat: GaussianRenderer, gs/debug.py, gs/benchmarks.py) ----
 * mode 0: a Gaussian belongs to a tile when its value at one of the four tile corners exceeds thresh
 * (count_num_gaussians_each_tile / image_sort: render.cu:46-70,139-176, tile_ops.h, kernels.h:253-272;
 * cov_or_radius = cov [N,2,2]); mode 1: when its bounding circle reaches the tile
 * (count_num_gaussians_each_tile_bcircle / prepare_image_sort: render.cu:73-137, culling.h:48-130,
 * kernels.h:306-350; cov_or_radius = radius [N]).
 * count: num_gaussians[tile] += hits (int32 [T], caller-zeroed).
 * image_sort: offset[T] = exclusive scan of the incoming tile_n_gaussians; tiledepth [N_with_dub] receives
 * the {lo = float depth, hi = tile} keys in fill order (per tile, ascending Gaussian index) and is left
 * unsorted, as in the reference; gaussian_ids [N_with_dub] receives the ids sorted per tile by depth;
 * mode 0 rewrites tile_n_gaussians with its own count.  Workspace from the caller. */
int gsgen_legacy_count_tiles(uint32_t mode, uint32_t N, const float *mean, const float *cov_or_radius,
                             const float *topleft, uint32_t tile_size, uint32_t n_tiles_h, uint32_t n_tiles_w,
                             float pixel_size_x, float pixel_size_y, float thresh, int *num_gaussians,
                             gsgen_stream_t stream);
size_t gsgen_legacy_sort_workspace_bytes(uint32_t N_with_dub, uint32_t n_tiles);
int gsgen_legacy_image_sort(uint32_t mode, uint32_t N, uint32_t N_with_dub, int *gaussian_ids,
                            unsigned long long *tiledepth, const float *depth, int *tile_n_gaussians, int *offset,
                            const float *mean, const float *cov_or_radius, const float *topleft, uint32_t tile_size,
                            uint32_t n_tiles_h, uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y,
                            float thresh, void *workspace, size_t workspace_bytes, gsgen_stream_t stream);

/* Fused frame: frustum cull + projection + AABB + binning + per-tile sort in one enqueue with
 * no host round trip (gs/gaussian_splatting.py:1208-1295 without the mask gathers and syncs).
 * `cam` (DEVICE, 56 floats): [0..11] c2w row-major 3x4; [12..15] fx, fy, cx, cy;
 * [16] frustum_culling_radius (<= 0: skip the cull); [17] tile_culling_radius (the reference's
 * D = 6.0); [18..19] spare; [20..37] frustum plane normals [6,3]; [38..55] plane points [6,3]
 * (both as utils/camera.py:260-294 computes them; gsgen_pack_camera fills the block on the host).
 * Culled Gaussians keep their index and emit no pairs.  gaussian_ids has capacity D_cap; the
 * true pair count is written to *total (device); if it exceeds D_cap nothing is binned,
 * start/end are all GSGEN_LIST_OVERFLOW (-2) and *total still holds the required size.  A frame that does not fit is never
 * rendered as a finite blank image: every compositing FORWARD of this library writes NaN into out (all channels) and T for a
 * tile whose start is GSGEN_LIST_OVERFLOW -- loss and image are visibly dead --, every backward skips such a tile.  (The
 * reference sizes gaussian_ids exactly after a host sync, gs/culling.py:34, and has no such state; -1 stays "empty tile",
 * aabb_culling.h:248-249.)
 * pair_report (gsgen_frame_geometry_report, gsgen_geometry_view::pair_report; optional): two uint32 in HOST-visible memory
 * (pinned + device-mapped, e.g. hipHostMalloc) that the geometry launch writes with system-scope stores: [0] = this frame's
 * pair count (every frame: a caller can grow the list BEFORE it overflows), [1] = max(count) over the frames that did not
 * fit (kept until the host clears it).  No copy, no event, no sync; a hipGraph replay reports the same way. */
#define GSGEN_LIST_OVERFLOW (-2)
size_t gsgen_frame_workspace_bytes(uint32_t N, uint32_t D_cap, uint32_t n_tiles);
/* Small host -> device upload THROUGH KERNEL ARGUMENTS (per-render constants: camera blocks, pixel origins): the bytes
 * travel in the dispatch packets of one-workgroup kernels (3 584 bytes each), so the host buffer may be reused as soon
 * as the call returns, nothing is staged or pinned, the call does not wait for the stream and can be captured into a
 * hipGraph.  dst (device) and bytes must be multiples of 4.  No counterpart in the reference (it uploads with
 * torch's `.to(device)` per camera). */
int gsgen_upload_small(void *dst, const void *host_src, size_t bytes, gsgen_stream_t stream);
/* HOST helper: the 68-float per-camera blocks this library's fused paths upload (gsgen_upload_small) for n cameras in
 * one call: cam[56] as gsgen_pack_camera | pixel origin (-cx/fx, -cy/fy) | the 9 rotation entries of c2w, row-major |
 * pad.  c2w: n poses of >= 12 floats (rows of [R|t]), c2w_stride floats apart; intr: [n,8] doubles fx fy cx cy w h near
 * far. */
int gsgen_pack_camera_blocks(uint32_t n, const float *c2w, uint32_t c2w_stride, const double *intr,
                             float frustum_radius, float tile_radius, float *blocks);
/* HOST helper (no device work): fills cam[56] from a host c2w [3,4] and the CameraInfo fields
 * (utils/camera.py:219-259; yfov = 2 atan(h / 2fy), aspect = w / h). */
int gsgen_pack_camera(const float *c2w, float fx, float fy, float cx, float cy, uint32_t w, uint32_t h,
                      double near_plane, double far_plane, float frustum_radius, float tile_radius,
                      float *cam);
int gsgen_frame_geometry(uint32_t N, const float *mean, const float *qvec, const float *svec,
                         const float *cam, uint32_t W, uint32_t H, uint32_t D_cap, float *mean2d,
                         float *cov2d, float *depth, uint8_t *mask, int *gaussian_ids, int *start,
                         int *end, uint32_t *total, void *workspace, size_t workspace_bytes,
                         gsgen_stream_t stream);
/* HOST helper: the device-side address of a pinned (hipHostMalloc / torch pin_memory) host block, for pair_report; NULL when the
 * block is not mapped into the device's address space. */
void *gsgen_host_device_pointer(void *pinned_host);
int gsgen_frame_geometry_report(uint32_t N, const float *mean, const float *qvec, const float *svec,
                                const float *cam, uint32_t W, uint32_t H, uint32_t D_cap, float *mean2d,
                                float *cov2d, float *depth, uint8_t *mask, int *gaussian_ids, int *start,
                                int *end, uint32_t *total, uint32_t *pair_report, void *workspace,
                                size_t workspace_bytes, gsgen_stream_t stream);

/* gsgen_frame_geometry for the B cameras of a batch in ONE enqueue of the same eight kernels
 * (gridDim.y / .z = view) instead of eight small launches per camera: a single view's launches are
 * latency-bound (100k Gaussians, 2.5k tiles do not fill 256 CUs; 108 us end to end on cfg2, ~45 us per
 * view however many streams overlap them), B views per launch fill the chip.  Per view the outputs are
 * bit-identical to gsgen_frame_geometry.  `views` is HOST memory, read before the call returns; each
 * view brings its own outputs, pair capacity and gsgen_frame_workspace_bytes(N, D_cap, n_tiles)
 * workspace (gsgen_frame_tile_order applies to it unchanged).  batch_workspace: device,
 * gsgen_frame_batch_workspace_bytes(n_views) bytes, not reused before the enqueued work has run. */
typedef struct gsgen_geometry_view {
  const float *cam;                        /* DEVICE, 56 floats as for gsgen_frame_geometry */
  float *mean2d, *cov2d, *depth;           /* [N,2], [N,2,2], [N] */
  uint8_t *mask;                           /* [N] */
  int *gaussian_ids, *start, *end;         /* [D_cap], [n_tiles], [n_tiles] */
  uint32_t *total;                         /* [1] */
  void *workspace;
  size_t workspace_bytes;
  uint32_t D_cap;
  /* optional (NULL = leave alone): this view's gradient accumulators of the step's backward -- d L / d mean2d [N,2] (8-byte
   * aligned), d L / d cov2d [N,2,2] (16-byte aligned), d L / d (r, g, b, depth, 1, depth^2) [N,6] (8-byte aligned) -- ZERO-FILLED
   * by the projection launch, which writes a record per Gaussian of the view anyway: the compositing backward kernels
   * accumulate into caller-zeroed arrays (as the reference's, vol_render.h:866-992), and the fill between forward and
   * backward was a launch of its own in every step's chain */
  float *zero_grad_mean2d, *zero_grad_cov2d, *zero_grad_chan6;
  uint32_t *pair_report;                   /* optional, HOST-visible [2]: see "pair_report" above */
  /* optional (round 6): [N,4] floats, 16-byte aligned -- the projection launch also prepares the evaluation record of the RGB / scalar /
   * RGB + heads compositing kernels once per Gaussian of the view: (p0, p1, p2) = sqrt(0.5 log2 e) * the Cholesky factor of the
   * symmetrised Sigma^-1, formed in fp64 from cov2d and rounded to fp32, and a validity flag (0 for a degenerate or non-finite
   * covariance).  gsgen_rgbd_view::chol hands it to the compositing launches. */
  float *chol;
  /* optional (round 6): [N] floats shared by the views of every batch -- the projection launch raises them to each visible Gaussian's
   * screen-space radius m + sqrt(max(m^2 - det, 0)), m = (c0 + c3) / 2 of this view's cov2d: the `max_radii2d` statistic the trainer's prune
   * step reads (gs/gaussian_splatting.py:1240-1245), what gsgen_densify_update_batch(cov2d, ...) computes in a launch of its own */
  float *max_radii2d;
} gsgen_geometry_view;
size_t gsgen_frame_batch_workspace_bytes(uint32_t n_views);
int gsgen_frame_geometry_batch(uint32_t n_views, const gsgen_geometry_view *views, uint32_t N, const float *mean,
                               const float *qvec, const float *svec, uint32_t W, uint32_t H,
                               void *batch_workspace, gsgen_stream_t stream);
/* The same, and the launch also zero-fills `zero_shared` (zero_shared_floats floats: a multiple of 4, 16-byte aligned; 0 = none):
 * the gradient accumulators the views of the batch SHARE (d L / d alpha [N] and d L / d sh [N,3,C*C] or d L / d colour, laid out
 * back to back by the caller).  With the per-view targets above no fill kernel is left between a step's forward and backward. */
int gsgen_frame_geometry_batch_zero(uint32_t n_views, const gsgen_geometry_view *views, uint32_t N, const float *mean,
                                    const float *qvec, const float *svec, uint32_t W, uint32_t H, float *zero_shared,
                                    size_t zero_shared_floats, void *batch_workspace, gsgen_stream_t stream);

/* Launch order for the compositing kernels produced by gsgen_frame_geometry: a device array
 * of n_tiles tile indices, longest list first (pointer into `workspace`; pure host arithmetic).
 * The *_ordered variants of the SH compositing entry points take it (NULL = spatial order);
 * they are otherwise identical to gsgen_vol_render_sh / gsgen_vol_render_backward_sh. */
const uint32_t *gsgen_frame_tile_order(void *workspace, uint32_t N, uint32_t D_cap, uint32_t n_tiles);
int gsgen_vol_render_sh_ordered(uint32_t N, uint32_t D, const float *mean, const float *cov,
                                const float *sh_coeffs, const float *alpha, const int *start,
                                const int *end, const int *gaussian_ids, float *out, const float *topleft,
                                const float *c2w, uint32_t tile_size, uint32_t n_tiles_h,
                                uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y, uint32_t H,
                                uint32_t W, uint32_t C, float thresh, const float *bg_rgb, float *T,
                                const uint32_t *tile_order, gsgen_stream_t stream);
int gsgen_vol_render_backward_sh_ordered(uint32_t N, uint32_t D, const float *mean, const float *cov,
                                         const float *sh_coeffs, const float *alpha, const int *start,
                                         const int *end, const int *gaussian_ids, const float *out,
                                         float *grad_mean, float *grad_cov, float *grad_sh_coeffs,
                                         float *grad_alpha, const float *grad_out, const float *topleft,
                                         const float *c2w, uint32_t tile_size, uint32_t n_tiles_h,
                                         uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y,
                                         uint32_t H, uint32_t W, uint32_t C, float thresh,
                                         const float *bg_rgb, const uint32_t *tile_order,
                                         gsgen_stream_t stream);

/* Segmented backward.  The compositing backward normally runs one workgroup per tile, and the
 * launch ends on the few heaviest tiles.  Given a segment workspace, the FORWARD additionally
 * stores the per-pixel state (transmittance, prefix colour) in front of every 32nd list entry and
 * the entry at which each pixel saturated; the BACKWARD then runs one workgroup per (tile,
 * 32-entry segment) -- uniform work units, n_segments per tile (the last one takes whatever
 * remains).  Same results up to the rounding of `final - prefix`.  n_segments <= 1 or a NULL
 * workspace: exactly the *_ordered entry points.  The workspace written by the forward must be
 * handed unchanged to the backward of the same frame. */
size_t gsgen_segment_workspace_bytes(uint32_t n_tiles, uint32_t n_segments);
int gsgen_vol_render_sh_segmented(uint32_t N, uint32_t D, const float *mean, const float *cov,
                                  const float *sh_coeffs, const float *alpha, const int *start,
                                  const int *end, const int *gaussian_ids, float *out, const float *topleft,
                                  const float *c2w, uint32_t tile_size, uint32_t n_tiles_h,
                                  uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y, uint32_t H,
                                  uint32_t W, uint32_t C, float thresh, const float *bg_rgb, float *T,
                                  const uint32_t *tile_order, void *segment_workspace, uint32_t n_segments,
                                  gsgen_stream_t stream);
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
                                           gsgen_stream_t stream);

/* Batched cameras in ONE launch (the reference loops over the cameras of a batch in
 * Python, gs/gaussian_splatting.py:1423-1466, calling render_sh / its backward once per camera).  A
 * single 800x800 launch ends on a tail of long sparse tiles that leaves most of the chip idle;
 * one launch per (at most 8) views pays that tail once per batch.  Every view has its own projected records,
 * lists and outputs; sh_coeffs / alpha and their gradients are shared, the gradients accumulating
 * atomically over views exactly as they do over tiles.  Per view the result is bit-identical to
 * gsgen_vol_render_sh_segmented / gsgen_vol_render_backward_sh_segmented on the same inputs
 * (gradient sums up to fp32 atomic order).  `views` is HOST memory, read before the call returns.
 * batch_workspace: device, gsgen_sh_batch_workspace_bytes(n_views) bytes, one per batch in flight.  (Since round 4 the per-view
 * kernel parameters travel in the kernel arguments, 8 views per launch, and the plain batched entry points no longer write
 * to it; the routed ones keep their per-tile routing bytes behind it -- size and argument stay for ABI stability.)
 * The batched forward writes EVERY pixel of out / T: empty tiles receive bg_rgb (0 without one) and T = 1, so the caller
 * need not pre-initialise the images (the per-camera entry points keep the reference's contract, vol_render.h:1006-1013). */
typedef struct gsgen_sh_view {
  const float *mean, *cov;                 /* [N,2], [N,2,2] of this view */
  const int *start, *end, *gaussian_ids;   /* [n_tiles], [n_tiles], [D] */
  const uint32_t *tile_order;              /* [n_tiles] or NULL (then NULL in every view) */
  const float *topleft, *c2w;              /* [2], [9] */
  const float *bg_rgb;                     /* [3] or NULL */
  float pixel_size_x, pixel_size_y;
  float *out, *T;                          /* [H,W,3], [H,W]; out is read by the backward */
  void *segment_workspace;                 /* gsgen_segment_workspace_bytes, or NULL if n_segments <= 1 */
  const float *grad_out;                   /* backward: [H,W,3] */
  float *grad_mean, *grad_cov;             /* backward: [N,2], [N,2,2] of this view, accumulated into */
  /* Round 6, optional (zero = as before), the *_routed launches with per-splat bounds (sh_row_bounds) only -- the persistent exact
   * fallback kernels are launched only when they may have something to do:
   *   route_report: one uint32 in HOST-visible memory (pinned + device-mapped, as pair_report): the polynomial forward stores 1 there
   *     whenever a tile's staged batch holds more than a quarter of splats beyond the bound -- the condition that hands the tile to the
   *     exact fallback.  No copy, no event, no sync; the host reads (and clears) it before a later batch.
   *   no_fallback = 1 (the same in every view, forward and backward of a batch alike): the two fallback launches are NOT enqueued and
   *     the polynomial kernels never hand a tile over -- every splat beyond the bound takes the per-entry exact tier instead, however
   *     many a tile holds.  Always correct (the same colours within the routing's 1e-5), slower on tiles crowded with such splats:
   *     meant for callers whose earlier batches reported none (gsgen_amd.BatchRenderer, bench.py: three clean reports in a row). */
  uint32_t *route_report;
  uint32_t no_fallback;
} gsgen_sh_view;
size_t gsgen_sh_batch_workspace_bytes(uint32_t n_views);
int gsgen_vol_render_sh_batch(uint32_t n_views, const gsgen_sh_view *views, uint32_t N,
                              const float *sh_coeffs, const float *alpha, uint32_t tile_size,
                              uint32_t n_tiles_h, uint32_t n_tiles_w, uint32_t H, uint32_t W, uint32_t C,
                              float thresh, uint32_t n_segments, void *batch_workspace,
                              gsgen_stream_t stream);
int gsgen_vol_render_backward_sh_batch(uint32_t n_views, const gsgen_sh_view *views, uint32_t N,
                                       const float *sh_coeffs, const float *alpha, float *grad_sh_coeffs,
                                       float *grad_alpha, uint32_t tile_size, uint32_t n_tiles_h,
                                       uint32_t n_tiles_w, uint32_t H, uint32_t W, uint32_t C, float thresh,
                                       uint32_t n_segments, void *batch_workspace, gsgen_stream_t stream);

/* ---- the coefficient bound: the fast form of the per-pixel SH basis, decided ON THE DEVICE ----------------------------
 * The reference evaluates the SH basis per pixel for dir = normalize(R (qx, qy, 1)) (vol_render_sh.h:48-65).  Inside a
 * 16 x 16 tile that is a degree-2 polynomial in the pixel offsets to within 1.0 delta^3 of a basis value (delta = the tile's
 * half diagonal in camera space; the fit that ships reaches 0.93 delta^3 -- tests/test_poly_fit_bound.py sweeps it over rotations,
 * tile positions and focal lengths): six-term contractions against coefficients transformed once per (tile, splat), +56 %
 * renders/s on BASELINE configs[1].  The colour error is <= 0.25 * S * delta^3 with
 *     S = max over splats and channels of sum_{k >= 1} |sh[i][c][k]|,
 * and the polynomial form is used only where that stays <= 1e-5 - 8.7e-7.  The 8.7e-7 belong to the Taylor tier (round 5): a
 * splat whose logits move little enough across the tile -- remainder bound 0.00694 (l + q)^3 + 0.02312 q (2 l + q) <= 8.7e-7
 * for the linear and quadratic parts l, q of its row's coefficient sum, the ordinary case by a factor of hundreds -- gets its
 * COLOUR as a quadratic in the pixel offsets, formed once per (tile, splat): no exponential per pixel.  Together: routed colours
 * within 1e-5 of the exact kernels', a tenth of the 1e-4 image tolerance.  (Rounds 2-4 routed on 0.7 delta^3, a calibration
 * of a different interpolation: 1.4e-5.)
 *
 * S lives in DEVICE memory and never visits the host: gsgen_sh_l1_bound writes it (one coalesced pass over the
 * coefficients, ~5 us for 100 k splats; enqueue it on the render's stream whenever the coefficients may have changed, i.e.
 * every optimiser step), the *_bounded entry points below take its device address.  With a bound (SH degree 3 only) they
 * enqueue the polynomial-basis kernel AND the exact one over the same grid; every workgroup reads S and its view's pixel
 * size and exactly one of the two kernels renders the view -- no host decision, no synchronisation, hipGraph-capturable, and
 * a bound that is too large only costs speed, never accuracy.  A forward and its backward must see the same value (pass the
 * same address and do not rewrite it in between).  sh_l1_bound == NULL: the exact kernels only (gsgen_vol_render_sh_batch,
 * gsgen_vol_render_sh_segmented, ... are exactly that). */
int gsgen_sh_l1_bound(uint32_t N, const float *sh_coeffs, uint32_t C, float *out /* device, 1 float, OVERWRITTEN */,
                      gsgen_stream_t stream);
/* Debug verification of a bound some other pass produced (e.g. a fused optimiser step): *n_violations (device uint32,
 * OVERWRITTEN) = the number of (splat, channel) rows whose sum_{k >= 1} |sh| exceeds *bound (device). */
int gsgen_sh_l1_bound_check(uint32_t N, const float *sh_coeffs, uint32_t C, const float *bound, uint32_t *n_violations,
                            gsgen_stream_t stream);
/* HOST arithmetic only: 1 if a view of this pixel size (max of pixel_size_x / _y) takes the polynomial form under the
 * bound value S -- the device's own rule, for reports */
int gsgen_sh_poly_applies(float sh_l1_bound, float max_pixel_size, uint32_t C);
int gsgen_vol_render_sh_batch_bounded(uint32_t n_views, const gsgen_sh_view *views, uint32_t N,
                                      const float *sh_coeffs, const float *alpha, uint32_t tile_size,
                                      uint32_t n_tiles_h, uint32_t n_tiles_w, uint32_t H, uint32_t W, uint32_t C,
                                      float thresh, uint32_t n_segments, const float *sh_l1_bound /* device or NULL */,
                                      void *batch_workspace, gsgen_stream_t stream);
int gsgen_vol_render_backward_sh_batch_bounded(uint32_t n_views, const gsgen_sh_view *views, uint32_t N,
                                               const float *sh_coeffs, const float *alpha, float *grad_sh_coeffs,
                                               float *grad_alpha, uint32_t tile_size, uint32_t n_tiles_h,
                                               uint32_t n_tiles_w, uint32_t H, uint32_t W, uint32_t C, float thresh,
                                               uint32_t n_segments, const float *sh_l1_bound /* device or NULL */,
                                               void *batch_workspace, gsgen_stream_t stream);
/* One camera (gsgen_vol_render_sh_segmented / gsgen_vol_render_backward_sh_segmented + the bound): what the `_gs` SH names
 * (tile_based_vol_rendering_sh / _backward_sh and their _with_bg forms, gs/src/render.h:83-127) run on -- the binding
 * computes S into a scratch float in front of each call, so the reference's call shape gets the fast kernels too. */
int gsgen_vol_render_sh_bounded(uint32_t N, uint32_t D, const float *mean, const float *cov,
                                const float *sh_coeffs, const float *alpha, const int *start,
                                const int *end, const int *gaussian_ids, float *out, const float *topleft,
                                const float *c2w, uint32_t tile_size, uint32_t n_tiles_h,
                                uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y, uint32_t H,
                                uint32_t W, uint32_t C, float thresh, const float *bg_rgb, float *T,
                                const uint32_t *tile_order, void *segment_workspace, uint32_t n_segments,
                                const float *sh_l1_bound /* device or NULL */, gsgen_stream_t stream);
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
                                         const float *sh_l1_bound /* device or NULL */, gsgen_stream_t stream);

/* ---- per-TILE routing of the polynomial SH basis (round 4) ------------------------------------------------------------------
 * The *_bounded entry points above route per VIEW on one number, the scene's largest per-row coefficient sum: one splat with large
 * higher-band coefficients, or a wide camera, sends the whole view to the exact kernels.  The *_routed entry points take the
 * bound per SPLAT -- sh_row_bounds[i] = max over the three channels of sum_{k >= 1} |sh[i][c][k]|, DEVICE memory, [N], measured per
 * step by gsgen_sh_l1_bound_rows (one coalesced pass, no atomics but one per workgroup; out_max, optional, receives the global
 * maximum the per-view rule uses) -- and decide per ENTRY and per TILE: a splat that satisfies 0.25 * S_i * delta^3 <= 1e-5 - 8.7e-7
 * for its view's pixel size (the same rule, per splat) is evaluated through the tile's polynomial form; one beyond it is evaluated
 * exactly, entry by entry, inside the same kernel (the pixel's own SH basis against its raw coefficients); a staged batch of 32
 * records with more than a quarter of such splats sends its tile -- and only that tile -- to the exact kernel.  Splats behind the
 * point where the tile's pixels saturate are never staged and never count.  Batched launches record the decision in one byte per (view, tile)
 * inside batch_workspace (gsgen_sh_batch_workspace_bytes_routed): the polynomial forward writes it, the exact fallback behind it
 * and both backward kernels read it, so forward and backward agree by construction; per-camera launches decide before they touch
 * a tile, forward and backward alike: with sh_l1_bound given as well (the maximum gsgen_sh_l1_bound_rows leaves in out_max) a
 * view wholly within its bound -- the normal case -- takes the polynomial form without a look at the lists, otherwise the
 * tile's list is scanned against the per-splat bounds (any splat beyond them: the exact form for that tile).  sh_row_bounds == NULL: exactly the *_bounded behaviour on sh_l1_bound.  SH degree 3 only
 * (C != 4: the exact kernels).  No counterpart in the reference (vol_render_sh.h:48-65 evaluates the basis per pixel). */
int gsgen_sh_l1_bound_rows(uint32_t N, const float *sh_coeffs, uint32_t C, float *out_max /* device, 1 float, or NULL */,
                           float *out_rows /* device, [N] */, gsgen_stream_t stream);
/* The same without the 4-byte fill in front of the pass (one launch less per step): running_max is only ever RAISED -- an upper
 * bound of the current coefficients' maximum as long as the caller zeroed it at some point, tight again whenever the caller
 * zeroes it (the per-view shortcut it feeds is conservative under a stale larger value, never wrong). */
int gsgen_sh_l1_bound_rows_running(uint32_t N, const float *sh_coeffs, uint32_t C, float *running_max /* device, 1 float, or NULL */,
                                   float *out_rows /* device, [N] */, gsgen_stream_t stream);
size_t gsgen_sh_batch_workspace_bytes_routed(uint32_t n_views, uint32_t n_tiles);
int gsgen_vol_render_sh_batch_routed(uint32_t n_views, const gsgen_sh_view *views, uint32_t N,
                                     const float *sh_coeffs, const float *alpha, uint32_t tile_size,
                                     uint32_t n_tiles_h, uint32_t n_tiles_w, uint32_t H, uint32_t W, uint32_t C,
                                     float thresh, uint32_t n_segments, const float *sh_l1_bound /* device or NULL */,
                                     const float *sh_row_bounds /* device [N] or NULL */, void *batch_workspace,
                                     gsgen_stream_t stream);
int gsgen_vol_render_backward_sh_batch_routed(uint32_t n_views, const gsgen_sh_view *views, uint32_t N,
                                              const float *sh_coeffs, const float *alpha, float *grad_sh_coeffs,
                                              float *grad_alpha, uint32_t tile_size, uint32_t n_tiles_h,
                                              uint32_t n_tiles_w, uint32_t H, uint32_t W, uint32_t C, float thresh,
                                              uint32_t n_segments, const float *sh_l1_bound, const float *sh_row_bounds,
                                              void *batch_workspace, gsgen_stream_t stream);
/* The MOMENT form of gsgen_vol_render_backward_sh_batch_routed (round 6; vol_render_sh.h:353-455 / vol_render_bg.h:131-242 for the
 * cameras of a batch): same arguments and routing; the views' grad_mean [N,2] receive sum g (tx, ty), grad_cov [N,4] sum g
 * (tx tx, tx ty, ty ty; the fourth float is not written), g = d L / d (a G) * a G per pixel, (tx, ty) = det Sigma^-1 d as the
 * Gaussian is evaluated (kernels.h:172-193) -- seven packed operations per pixel pair instead of thirteen and one atomic per
 * (tile, Gaussian) less.  gsgen_project_gaussians_backward_batch_moments_sh expands them per (view, Gaussian). */
int gsgen_vol_render_backward_sh_batch_routed_moments(uint32_t n_views, const gsgen_sh_view *views, uint32_t N,
                                                      const float *sh_coeffs, const float *alpha, float *grad_sh_coeffs,
                                                      float *grad_alpha, uint32_t tile_size, uint32_t n_tiles_h,
                                                      uint32_t n_tiles_w, uint32_t H, uint32_t W, uint32_t C, float thresh,
                                                      uint32_t n_segments, const float *sh_l1_bound, const float *sh_row_bounds,
                                                      void *batch_workspace, gsgen_stream_t stream);
int gsgen_vol_render_sh_routed(uint32_t N, uint32_t D, const float *mean, const float *cov,
                               const float *sh_coeffs, const float *alpha, const int *start,
                               const int *end, const int *gaussian_ids, float *out, const float *topleft,
                               const float *c2w, uint32_t tile_size, uint32_t n_tiles_h,
                               uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y, uint32_t H,
                               uint32_t W, uint32_t C, float thresh, const float *bg_rgb, float *T,
                               const uint32_t *tile_order, void *segment_workspace, uint32_t n_segments,
                               const float *sh_l1_bound, const float *sh_row_bounds, gsgen_stream_t stream);
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
                                        const float *sh_l1_bound, const float *sh_row_bounds, gsgen_stream_t stream);

/* Fused RGB + auxiliary heads: what render_one does in four compositing passes
 * (gs/gaussian_splatting.py:1304-1403: rgb, depth, opacity = scalar 1, depth^2) in one.
 * out6 / grad_out6 are [H,W,6] = (r, g, b, depth, opacity, depth^2); grad_chan6 [N,6] receives the
 * gradients of the per-Gaussian channel values (r, g, b, d, 1, d*d) -- the caller folds columns
 * 3 and 5 into d/d depth (g3 + 2 d g5).  Semantics per channel are those of the single-head
 * entry points above; T as in gsgen_vol_render_start_end_with_T. */
int gsgen_vol_render_rgbd(uint32_t N, uint32_t D, const float *mean, const float *cov, const float *color,
                          const float *depth, const float *alpha, const int *start, const int *end,
                          const int *gaussian_ids, float *out6, const float *topleft, uint32_t tile_size,
                          uint32_t n_tiles_h, uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y,
                          uint32_t H, uint32_t W, float thresh, float *T, const uint32_t *tile_order,
                          gsgen_stream_t stream);
int gsgen_vol_render_rgbd_backward(uint32_t N, uint32_t D, const float *mean, const float *cov, const float *color,
                                   const float *depth, const float *alpha, const int *start, const int *end,
                                   const int *gaussian_ids, const float *out6, float *grad_mean, float *grad_cov,
                                   float *grad_chan6, float *grad_alpha, const float *grad_out6,
                                   const float *topleft, uint32_t tile_size, uint32_t n_tiles_h,
                                   uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y, uint32_t H,
                                   uint32_t W, float thresh, const uint32_t *tile_order, gsgen_stream_t stream);

/* The same for the B cameras of a batch in ONE launch each way (as gsgen_vol_render_sh_batch): per-view
 * records, depths, lists, out6 / T and per-view gradients (grad_chan6 holds the view's own depth-head
 * gradients, so it is not shared); color / alpha are shared and grad_alpha accumulates over the views.
 * batch_workspace: gsgen_sh_batch_workspace_bytes(n_views), one per batch in flight. */
typedef struct gsgen_rgbd_view {
  const float *mean, *cov, *depth;         /* [N,2], [N,2,2], [N] of this view */
  const int *start, *end, *gaussian_ids;
  const uint32_t *tile_order;              /* or NULL (then NULL in every view) */
  const float *topleft;                    /* [2] */
  float pixel_size_x, pixel_size_y;
  float *out6, *T;                         /* [H,W,6], [H,W]; out6 is read by the backward */
  const float *grad_out6;                  /* backward: [H,W,6]; or NULL (heads only): the four images below */
  float *grad_mean, *grad_cov, *grad_chan6; /* backward: [N,2], [N,2,2], [N,6] of this view, accumulated into */
  /* backward of the heads with grad_out6 == NULL: d L / d rgb [H,W,3], / d depth, / d opacity, / d depth^2 [H,W] as an
   * autograd engine delivers them (one tensor per output; any may be NULL = zero) -- no [H,W,6] image to assemble */
  const float *grad_rgb, *grad_depth, *grad_opacity, *grad_depth2;
  /* Round 6, optional (zero-initialised = the layout above), RGB + heads launches only -- what the reference's model computes in
   * torch around its four compositing passes, folded into the two launches:
   *   out_rgb [H,W,3], out_depth, out_opacity, out_depth2 [H,W]: the four heads as separate contiguous images instead of out6 (all
   *     four or none; out6 may then be NULL).  The backward reads them as the forward's final image; grad_out6 must be NULL (the
   *     four-image gradient form).
   *   bg_rgb [3] (device): the forward writes rgb + T * bg (gs/renderer.py:1182; empty tiles show bg); the backward takes that sum
   *     as the final image, as the reference's does.
   *   grad_bg [64][4] (device, zeroed by the caller), backward: row (tile % 64) accumulates the tile's sum over pixels of
   *     nan_to_num(grad_rgb * T) (gs/renderer.py:1283: d L / d bg), T the forward's; the caller adds the 64 rows.
   *   depth_variance = 1: the sixth head is z_var = depth2 - depth^2 (gs/gaussian_splatting.py:1397) -- written so by the forward,
   *     and grad_depth2 is d L / d z_var (the backward forms d depth2 = g and d depth -= 2 depth g itself). */
  float *out_rgb, *out_depth, *out_opacity, *out_depth2;
  const float *bg_rgb;
  float *grad_bg;
  uint32_t depth_variance;
  /* optional (rgbd and rgb launches): the view's prepared evaluation records [N,4] as gsgen_geometry_view::chol delivers them -- staging
   * then reads 16 bytes per record instead of running the fp64 Cholesky preparation (kernels.h:195-224 evaluates the quadratic form in
   * fp64; here its factor is prepared in fp64) per staged (tile, Gaussian) record in the forward and again in the backward */
  const float *chol;
  /* optional (batched launches): DEVICE address of {pixel_size_x, pixel_size_y} of this view.  When set the kernels take the two from
   * there and ignore the floats above: nothing that changes from step to step then travels in kernel arguments, so a captured
   * hipGraph of the step can be replayed for other cameras (poses AND intrinsics: the reference samples a focal length per step,
   * data/__init__.py:194) once their camera blocks and these floats are in place (gsgen_amd.graph.CapturedStep). */
  const float *pixel_size_dev;
} gsgen_rgbd_view;
/* The batched forwards (rgbd and rgb) write EVERY pixel of out6 / T, empty tiles included (channels 0, T = 1): the caller need
 * not pre-initialise the images (the per-camera entry points above keep the reference's contract: empty tiles are left alone). */
int gsgen_vol_render_rgbd_batch(uint32_t n_views, const gsgen_rgbd_view *views, uint32_t N, const float *color,
                                const float *alpha, uint32_t tile_size, uint32_t n_tiles_h, uint32_t n_tiles_w,
                                uint32_t H, uint32_t W, float thresh, void *batch_workspace,
                                gsgen_stream_t stream);
int gsgen_vol_render_rgbd_backward_batch(uint32_t n_views, const gsgen_rgbd_view *views, uint32_t N,
                                         const float *color, const float *alpha, float *grad_alpha,
                                         uint32_t tile_size, uint32_t n_tiles_h, uint32_t n_tiles_w, uint32_t H,
                                         uint32_t W, float thresh, void *batch_workspace, gsgen_stream_t stream);
/* The MOMENT form of gsgen_vol_render_rgbd_backward_batch (round 6; replaces vol_render.h:866-992 + vol_render_scalar.h:104-234
 * for the trainer's default outputs, gs/gaussian_splatting.py:1304-1416).  Same arguments; what the per-view accumulators
 * receive differs: grad_mean [N,2] the two first moments (Mu, Mv), grad_cov [N,4] the three second moments (Muu, Muv, Mvv; the
 * fourth float is not written) of the per-pixel weight d L / d (a G) * a G against the whitened offsets u = p0 x + p1 y,
 * v = p2 y of the Cholesky-form Gaussian, grad_chan6 [N,6] = (d r, d g, d b, d L / d depth with the depth^2 head folded in
 * -- sum w (go_d + 2 d go_dd) --, untouched, untouched).  gsgen_project_gaussians_backward_batch_heads_moments expands them: the
 * entry loop sums 10 components per (tile, Gaussian) instead of 13 and spends 14 packed operations on the geometric part
 * instead of 29.  Gradients agree with the plain form to fp32 rounding (tests/test_cpu_host.py, tests/test_gpu_api.py). */
int gsgen_vol_render_rgbd_backward_batch_moments(uint32_t n_views, const gsgen_rgbd_view *views, uint32_t N,
                                                 const float *color, const float *alpha, float *grad_alpha,
                                                 uint32_t tile_size, uint32_t n_tiles_h, uint32_t n_tiles_w, uint32_t H,
                                                 uint32_t W, float thresh, void *batch_workspace, gsgen_stream_t stream);
/* Post-activation RGB only (gsgen_vol_render_start_end_with_T / gsgen_vol_render_backward_start_end for the
 * cameras of a batch): the same view array with out6 / grad_out6 read as [H,W,3] images, depth and grad_chan6
 * unused (may be NULL); the colour gradient [N,3] is shared and accumulates over the views like grad_alpha. */
int gsgen_vol_render_rgb_batch(uint32_t n_views, const gsgen_rgbd_view *views, uint32_t N, const float *color,
                               const float *alpha, uint32_t tile_size, uint32_t n_tiles_h, uint32_t n_tiles_w,
                               uint32_t H, uint32_t W, float thresh, void *batch_workspace, gsgen_stream_t stream);
int gsgen_vol_render_rgb_backward_batch(uint32_t n_views, const gsgen_rgbd_view *views, uint32_t N,
                                        const float *color, const float *alpha, float *grad_color,
                                        float *grad_alpha, uint32_t tile_size, uint32_t n_tiles_h,
                                        uint32_t n_tiles_w, uint32_t H, uint32_t W, float thresh,
                                        void *batch_workspace, gsgen_stream_t stream);

/* Self test of the wave64 cross-lane reduce-scatter used by the backward (tests only):
 * in [64 lanes, P components]; out[0..64) = per-lane result, out[64..128) = the component index that
 * lane owns (-1: duplicate holder); P in {8,16,32,64}. */
int gsgen_selftest_reduce_scatter(uint32_t P, const float *in, float *out, gsgen_stream_t stream);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* GSGEN_HIP_H */
