// legacy.hip -- the reference's older binning pipeline, for gfx950 (compiled with -ffp-contract=off).
//
// Replaces (paths relative to /root/reference/gs/src): count_num_gaussians_each_tile{,_bcircle}
// (render.cu:46-97 -> include/tile_ops.h:11-169), image_sort / prepare_image_sort (render.cu:99-176 ->
// tile_ops.h:171-506, culling.h:48-130) and their membership tests (kernels.h:253-350).  Only the
// reference's GaussianRenderer (gs/renderer.py:1353-1564), gs/debug.py and gs/benchmarks.py call them;
// the live path bins by AABB (binning.hip).  Kept faithful, not fast: the algorithm tests every
// (tile, Gaussian) pair.  Here one wavefront owns a tile and sweeps the Gaussians 64 at a time
// (ballot + popcount instead of the reference's one thread per tile looping over all of them), the
// scan is one workgroup, and the per-tile depth sort is binning.hip's register bitonic sort instead of
// a global 64-bit radix sort: within a tile the order is (depth bits, Gaussian index), which is what the
// reference's stable sort of {tile, depth} keys filled in index order produces.
#include "common.hpp"
#include "../../include/gsgen_hip.h"

extern "C" int gsgen_internal_sort_segments(uint32_t T, const uint32_t *tile_off, const uint32_t *ctrl,
                                            unsigned long long *keys, int *ids, int *start, int *end,
                                            gsgen_stream_t stream);

namespace gs {

// kernels.h:226-251 (fp64)
__device__ __forceinline__ float legacy_gauss(const float *mean, const float *cov, float qx, float qy) {
  const double c0 = cov[0], c1 = cov[1], c2 = cov[2], c3 = cov[3];
  const double det = c0 * c3 - c1 * c2;
  const double x = (double)(qx - mean[0]), y = (double)(qy - mean[1]);
  const double tx = x * c3 - y * c2, ty = -x * c1 + y * c0;
  double radial = (tx * x + ty * y) / det;
  if (radial < 0.0) radial = 1000.0;
  return (float)exp(-0.5 * radial);
}
// kernels.h:274-304
__device__ __forceinline__ float legacy_dist_seg(float x, float y, float x1, float x2, float y1, float y2) {
  const float A = x - x1, B = y - y1, C = x2 - x1, D = y2 - y1;
  const float dot = A * C + B * D, len_sq = C * C + D * D;
  float param = -1.0f, xx, yy;
  if (len_sq != 0) param = dot / len_sq;
  if (param < 0) { xx = x1; yy = y1; }
  else if (param > 1) { xx = x2; yy = y2; }
  else { xx = x1 + param * C; yy = y1 + param * D; }
  const float dx = x - xx, dy = y - yy;
  return sqrtf(dx * dx + dy * dy);
}
// MODE 0: value at a tile corner above thresh (kernels.h:253-272); 1: bounding circle reaches the tile (:306-350)
template <int MODE>
__device__ __forceinline__ bool legacy_hit(float tlx, float tly, uint32_t tile_size, float psx, float psy,
                                           const float *mean, const float *shape, uint32_t i, float thresh) {
  if constexpr (MODE == 0) {
    const float xr = tlx + (float)tile_size * psx, yb = tly + (float)tile_size * psy;
    const float *m = mean + 2 * (size_t)i, *c = shape + 4 * (size_t)i;
    float v = fmaxf(0.0f, legacy_gauss(m, c, tlx, tly));
    v = fmaxf(v, legacy_gauss(m, c, xr, tly));
    v = fmaxf(v, legacy_gauss(m, c, tlx, yb));
    v = fmaxf(v, legacy_gauss(m, c, xr, yb));
    return v > thresh;
  } else {
    const float rx = mean[2 * (size_t)i] - tlx, ry = mean[2 * (size_t)i + 1] - tly;
    const float px = psx * (float)tile_size, py = psy * (float)tile_size;
    if (rx >= 0 && rx <= px && ry >= 0 && ry <= py) return true;
    const float d1 = legacy_dist_seg(rx, ry, 0.0f, px, 0.0f, 0.0f);
    const float d2 = legacy_dist_seg(rx, ry, 0.0f, px, py, py);
    const float d3 = legacy_dist_seg(rx, ry, 0.0f, 0.0f, 0.0f, py);
    const float d4 = legacy_dist_seg(rx, ry, px, px, 0.0f, py);
    return fminf(fminf(d1, d2), fminf(d3, d4)) < shape[i];
  }
}
// topleft[0] + pixel_size_x * tile_x * tile_size, left to right (tile_ops.h:51-53)
__device__ __forceinline__ void legacy_tile_origin(const float *topleft, uint32_t tile, uint32_t ntw, uint32_t tile_size,
                                                   float psx, float psy, float &tlx, float &tly) {
  const int tx = (int)(tile % ntw), ty = (int)(tile / ntw);
  tlx = topleft[0] + psx * (float)tx * (float)tile_size;
  tly = topleft[1] + psy * (float)ty * (float)tile_size;
}

template <int MODE>
__global__ void __launch_bounds__(64)
k_legacy_count(uint32_t N, const float *__restrict__ mean, const float *__restrict__ shape,
               const float *__restrict__ topleft, uint32_t tile_size, uint32_t ntw, float psx, float psy,
               float thresh, int *__restrict__ num_gaussians) {
  const uint32_t tile = blockIdx.x, lane = threadIdx.x;
  float tlx, tly;
  legacy_tile_origin(topleft, tile, ntw, tile_size, psx, psy, tlx, tly);
  int cnt = 0;
  for (uint32_t i0 = 0; i0 < N; i0 += 64) {
    const uint32_t i = i0 + lane;
    const bool hit = i < N && legacy_hit<MODE>(tlx, tly, tile_size, psx, psy, mean, shape, i, thresh);
    cnt += __popcll(__ballot(hit));
  }
  if (lane == 0) num_gaussians[tile] += cnt;  // the reference accumulates too (tile_ops.h:63)
}

// offset = exclusive scan of tile_n (also as the sort's uint32 [T+1] segment table); one workgroup
__global__ void __launch_bounds__(256)
k_legacy_scan(uint32_t T, const int *__restrict__ tile_n, int *__restrict__ offset, uint32_t *__restrict__ tile_off,
              uint32_t *__restrict__ ctrl) {
  __shared__ uint32_t part[256];
  const uint32_t t = threadIdx.x;
  const uint32_t per = (T + 255) / 256;
  const uint32_t b = t * per, e = min(T, b + per);
  uint32_t sum = 0;
  for (uint32_t i = b; i < e; ++i) sum += (uint32_t)tile_n[i];
  part[t] = sum;
  __syncthreads();
  if (t == 0) {
    uint32_t run = 0;
    for (int k = 0; k < 256; ++k) { const uint32_t v = part[k]; part[k] = run; run += v; }
    tile_off[T] = run;
    ctrl[0] = run; ctrl[1] = 0u;
  }
  __syncthreads();
  uint32_t run = part[t];
  for (uint32_t i = b; i < e; ++i) {
    offset[i] = (int)run;
    tile_off[i] = run;
    run += (uint32_t)tile_n[i];
  }
}

template <int MODE>
__global__ void __launch_bounds__(64)
k_legacy_fill(uint32_t N, uint32_t cap, const float *__restrict__ mean, const float *__restrict__ shape,
              const float *__restrict__ depth, const float *__restrict__ topleft, uint32_t tile_size, uint32_t ntw,
              float psx, float psy, float thresh, const int *__restrict__ offset, int *__restrict__ tile_n,
              unsigned long long *__restrict__ tiledepth, unsigned long long *__restrict__ keys) {
  const uint32_t tile = blockIdx.x, lane = threadIdx.x;
  float tlx, tly;
  legacy_tile_origin(topleft, tile, ntw, tile_size, psx, psy, tlx, tly);
  const uint32_t off0 = (uint32_t)offset[tile];
  uint32_t off = off0;
  for (uint32_t i0 = 0; i0 < N; i0 += 64) {
    const uint32_t i = i0 + lane;
    const bool hit = i < N && legacy_hit<MODE>(tlx, tly, tile_size, psx, psy, mean, shape, i, thresh);
    const unsigned long long m = __ballot(hit);
    const uint32_t pos = off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (hit && pos < cap) {
      const uint32_t bits = __float_as_uint(depth[i]);
      tiledepth[pos] = ((unsigned long long)tile << 32) | bits;  // {lo = depth, hi = tile}: tile_ops.h:228-229
      keys[pos] = ((unsigned long long)bits << 32) | i;
    }
    off += (uint32_t)__popcll(m);
  }
  if (MODE == 0 && lane == 0) tile_n[tile] = (int)(off - off0);  // fill_tiledepth_cuda recounts (tile_ops.h:333-352)
}

}  // namespace gs

using namespace gs;

extern "C" {

int gsgen_legacy_count_tiles(uint32_t mode, uint32_t N, const float *mean, const float *cov_or_radius,
                             const float *topleft, uint32_t tile_size, uint32_t n_tiles_h, uint32_t n_tiles_w,
                             float pixel_size_x, float pixel_size_y, float thresh, int *num_gaussians,
                             gsgen_stream_t stream) {
  const uint32_t T = n_tiles_h * n_tiles_w;
  if (T == 0 || N == 0) return 0;
  if (!mean || !cov_or_radius || !topleft || !num_gaussians || mode > 1) return GSGEN_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (mode == 0)
    hipLaunchKernelGGL((k_legacy_count<0>), dim3(T), dim3(64), 0, s, N, mean, cov_or_radius, topleft, tile_size,
                       n_tiles_w, pixel_size_x, pixel_size_y, thresh, num_gaussians);
  else
    hipLaunchKernelGGL((k_legacy_count<1>), dim3(T), dim3(64), 0, s, N, mean, cov_or_radius, topleft, tile_size,
                       n_tiles_w, pixel_size_x, pixel_size_y, thresh, num_gaussians);
  return (int)hipGetLastError();
}

size_t gsgen_legacy_sort_workspace_bytes(uint32_t N_with_dub, uint32_t n_tiles) {
  return 256 + sizeof(unsigned long long) * (size_t)N_with_dub + sizeof(uint32_t) * ((size_t)n_tiles + 8) +
         2 * sizeof(int) * (size_t)n_tiles + 256;
}

int gsgen_legacy_image_sort(uint32_t mode, uint32_t N, uint32_t N_with_dub, int *gaussian_ids,
                            unsigned long long *tiledepth, const float *depth, int *tile_n_gaussians, int *offset,
                            const float *mean, const float *cov_or_radius, const float *topleft, uint32_t tile_size,
                            uint32_t n_tiles_h, uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y,
                            float thresh, void *workspace, size_t workspace_bytes, gsgen_stream_t stream) {
  const uint32_t T = n_tiles_h * n_tiles_w;
  if (T == 0) return 0;
  if (!tile_n_gaussians || !offset || !topleft || !workspace || mode > 1) return GSGEN_EINVAL;
  if (N_with_dub && (!gaussian_ids || !tiledepth)) return GSGEN_EINVAL;
  if (N && (!mean || !cov_or_radius || !depth)) return GSGEN_EINVAL;
  if (workspace_bytes < gsgen_legacy_sort_workspace_bytes(N_with_dub, T)) return GSGEN_EWORKSPACE;
  char *w = (char *)workspace;
  w += (256 - ((uintptr_t)w & 255)) & 255;
  unsigned long long *keys = (unsigned long long *)w; w += sizeof(unsigned long long) * (size_t)N_with_dub;
  uint32_t *tile_off = (uint32_t *)w; w += sizeof(uint32_t) * ((size_t)T + 4);
  uint32_t *ctrl = (uint32_t *)w; w += sizeof(uint32_t) * 4;
  int *start = (int *)w; w += sizeof(int) * (size_t)T;
  int *end = (int *)w;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_legacy_scan, dim3(1), dim3(256), 0, s, T, tile_n_gaussians, offset, tile_off, ctrl);
  if (N) {
    if (mode == 0)
      hipLaunchKernelGGL((k_legacy_fill<0>), dim3(T), dim3(64), 0, s, N, N_with_dub, mean, cov_or_radius, depth, topleft,
                         tile_size, n_tiles_w, pixel_size_x, pixel_size_y, thresh, offset, tile_n_gaussians, tiledepth,
                         keys);
    else
      hipLaunchKernelGGL((k_legacy_fill<1>), dim3(T), dim3(64), 0, s, N, N_with_dub, mean, cov_or_radius, depth, topleft,
                         tile_size, n_tiles_w, pixel_size_x, pixel_size_y, thresh, offset, tile_n_gaussians, tiledepth,
                         keys);
  }
  if (int e = (int)hipGetLastError()) return e;
  return gsgen_internal_sort_segments(T, tile_off, ctrl, keys, gaussian_ids, start, end, stream);
}

}  // extern "C"
