// binning.hip -- 16x16 tile binning and per-tile depth sort for gfx950.
//
// Replaces gs/src/include/aabb_culling.h:15-41 (key emission with one global atomic slot
// counter), :235-241 (one global 64-bit cub::DeviceRadixSort over all D pairs, 8 passes) and
// :70-103 (start/end from key boundaries), plus the 5 cudaMalloc/cudaFree and the blocking
// D2H copy around them (:204-229, :256-259).
//
// MI355X design: the tile id is not sorted at all.
//   1. histogram   per-tile pair counts with integer L2 atomics            (D atomics)
//   2. scan        exclusive scan over the T tiles -> segment offsets       (one workgroup)
//   3. emit        each Gaussian drops (depth_bits<<32 | id) into its tiles' segments; the
//                  slot inside a segment comes from a per-tile atomic cursor
//   4. sort        one workgroup per tile sorts its segment IN LDS on the 64-bit key
//                  (depth bits, then Gaussian id), writes ids back, fills start/end
// so sort traffic is one read + one write of the pairs instead of 8 global radix passes, and
// the order inside a tile is deterministic although the emission order is not: keys are
// unique.  Ordering semantics are the reference's: ascending UNSIGNED float bits of depth
// (the low word of its int64 key), so negative depths sort after positive ones.
// Everything is enqueued on the caller's stream with no host synchronisation.
#include "common.hpp"
#include "../../include/gsgen_hip.h"

namespace gs {

constexpr int kThreads = 256;
constexpr int kSortThreads = 256;
constexpr int kSortLds = 4096;  // keys sorted in LDS per tile (32 KiB); longer segments sort in global memory

__global__ void __launch_bounds__(kThreads)
k_count_rects(uint32_t N, const int *__restrict__ tl, const int *__restrict__ br, int ntw, int nth,
              uint32_t *__restrict__ tile_count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int2 a = *reinterpret_cast<const int2 *>(tl + 2 * (size_t)i);
  const int2 b = *reinterpret_cast<const int2 *>(br + 2 * (size_t)i);
  // rectangles handed in through the C ABI are clamped to the grid: an out-of-grid index
  // would be an out-of-bounds write in the reference
  const int x0 = max(a.x, 0), y0 = max(a.y, 0), x1 = min(b.x, ntw - 1), y1 = min(b.y, nth - 1);
  for (int ty = y0; ty <= y1; ++ty)
    for (int tx = x0; tx <= x1; ++tx) atomicAdd(&tile_count[ty * ntw + tx], 1u);
}

// exclusive scan of tile_count[T] -> tile_off[T+1]; ctrl[0] = total, ctrl[1] = overflow flag
__global__ void __launch_bounds__(1024)
k_scan_tiles(uint32_t T, const uint32_t *__restrict__ tile_count, uint32_t *__restrict__ tile_off,
             uint32_t *__restrict__ ctrl, uint32_t cap, uint32_t *__restrict__ total_out) {
  __shared__ uint32_t s_part[1024];
  const uint32_t t = threadIdx.x;
  const uint32_t per = (T + 1023u) / 1024u;
  const uint32_t b = t * per, e = min(b + per, T);
  uint32_t sum = 0;
  for (uint32_t i = b; i < e; ++i) sum += tile_count[i];
  s_part[t] = sum;
  __syncthreads();
  // Hillis-Steele inclusive scan over 1024 partials
  for (uint32_t d = 1; d < 1024u; d <<= 1) {
    const uint32_t v = (t >= d) ? s_part[t - d] : 0u;
    __syncthreads();
    s_part[t] += v;
    __syncthreads();
  }
  uint32_t run = s_part[t] - sum;  // exclusive prefix of this thread's chunk
  for (uint32_t i = b; i < e; ++i) {
    tile_off[i] = run;
    run += tile_count[i];
  }
  if (t == 1023u) {
    const uint32_t total = s_part[1023];
    tile_off[T] = total;
    ctrl[0] = total;
    ctrl[1] = (total > cap) ? 1u : 0u;
    if (total_out != nullptr) *total_out = total;
  }
}

__global__ void __launch_bounds__(kThreads)
k_emit(uint32_t N, const int *__restrict__ tl, const int *__restrict__ br,
       const float *__restrict__ depth, int ntw, int nth, const uint32_t *__restrict__ tile_off,
       uint32_t *__restrict__ tile_fill, const uint32_t *__restrict__ ctrl,
       unsigned long long *__restrict__ keys) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  if (ctrl[1] != 0u) return;  // capacity exceeded: bin nothing
  const int2 a = *reinterpret_cast<const int2 *>(tl + 2 * (size_t)i);
  const int2 b = *reinterpret_cast<const int2 *>(br + 2 * (size_t)i);
  const int x0 = max(a.x, 0), y0 = max(a.y, 0), x1 = min(b.x, ntw - 1), y1 = min(b.y, nth - 1);
  if (x1 < x0 || y1 < y0) return;
  const unsigned long long key =
      ((unsigned long long)__float_as_uint(depth[i]) << 32) | (unsigned long long)i;
  for (int ty = y0; ty <= y1; ++ty)
    for (int tx = x0; tx <= x1; ++tx) {
      const int tile = ty * ntw + tx;
      const uint32_t pos = tile_off[tile] + atomicAdd(&tile_fill[tile], 1u);
      keys[pos] = key;
    }
}

// normalised bitonic network (every comparator puts the smaller key at the lower index), so
// a segment of arbitrary length n behaves as if padded with +inf up to the next power of two.
template <typename KeyPtr>
__device__ __forceinline__ void bitonic_sort(KeyPtr k, uint32_t n, uint32_t tid, uint32_t nthreads) {
  uint32_t p2 = 1;
  while (p2 < n) p2 <<= 1;
  for (uint32_t size = 2; size <= p2; size <<= 1) {
    // flip step: partner = i ^ (size - 1)
    for (uint32_t i = tid; i < p2; i += nthreads) {
      const uint32_t l = i ^ (size - 1u);
      if (l > i && l < n) {
        const unsigned long long a = k[i], b = k[l];
        if (a > b) { k[i] = b; k[l] = a; }
      }
    }
    __syncthreads();
    for (uint32_t j = size >> 2; j > 0; j >>= 1) {
      for (uint32_t i = tid; i < p2; i += nthreads) {
        const uint32_t l = i ^ j;
        if (l > i && l < n) {
          const unsigned long long a = k[i], b = k[l];
          if (a > b) { k[i] = b; k[l] = a; }
        }
      }
      __syncthreads();
    }
  }
}

__global__ void __launch_bounds__(kSortThreads)
k_sort_tiles(uint32_t T, const uint32_t *__restrict__ tile_off, const uint32_t *__restrict__ ctrl,
             unsigned long long *__restrict__ keys, int *__restrict__ ids, int *__restrict__ start,
             int *__restrict__ end) {
  __shared__ unsigned long long s_keys[kSortLds];
  const uint32_t tile = xcd_swizzle(blockIdx.x, gridDim.x);
  const uint32_t tid = threadIdx.x;
  if (ctrl[1] != 0u) {
    if (tid == 0) { start[tile] = -1; end[tile] = -1; }
    return;
  }
  const uint32_t b = tile_off[tile], e = tile_off[tile + 1];
  const uint32_t n = e - b;
  if (tid == 0) {  // empty tiles stay -1 (aabb_culling.h:248-249)
    start[tile] = n ? (int)b : -1;
    end[tile] = n ? (int)e : -1;
  }
  if (n == 0) return;
  if (n <= (uint32_t)kSortLds) {
    for (uint32_t i = tid; i < n; i += kSortThreads) s_keys[i] = keys[b + i];
    __syncthreads();
    bitonic_sort(s_keys, n, tid, kSortThreads);
    for (uint32_t i = tid; i < n; i += kSortThreads) ids[b + i] = (int)(uint32_t)(s_keys[i] & 0xffffffffull);
  } else {
    // oversized segment: same network directly on the global segment (one workgroup, so
    // __syncthreads + L2-coherent stores of this CU order the passes)
    unsigned long long *k = keys + b;
    __syncthreads();
    bitonic_sort(k, n, tid, kSortThreads);
    for (uint32_t i = tid; i < n; i += kSortThreads) ids[b + i] = (int)(uint32_t)(k[i] & 0xffffffffull);
  }
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct BinWs {
  uint32_t *tile_count, *tile_fill, *tile_off, *ctrl;
  unsigned long long *keys;
  int *tl, *br;  // only in the frame workspace
  size_t bytes;
};
static BinWs carve(void *base, uint32_t N, uint32_t D, uint32_t T, bool with_rects) {
  BinWs w{};
  size_t off = 0;
  char *p = (char *)base;
  auto take = [&](size_t bytes) { void *r = p ? p + off : nullptr; off += align_up(bytes, 256); return r; };
  // tile_count | tile_fill | ctrl are contiguous so that one memset clears them
  w.tile_count = (uint32_t *)take(sizeof(uint32_t) * (2 * (size_t)T + 4));
  w.tile_fill = w.tile_count ? w.tile_count + T : nullptr;
  w.ctrl = w.tile_count ? w.tile_count + 2 * (size_t)T : nullptr;
  w.tile_off = (uint32_t *)take(sizeof(uint32_t) * ((size_t)T + 1));
  w.keys = (unsigned long long *)take(sizeof(unsigned long long) * (size_t)(D ? D : 1));
  if (with_rects) {
    w.tl = (int *)take(sizeof(int) * 2 * (size_t)(N ? N : 1));
    w.br = (int *)take(sizeof(int) * 2 * (size_t)(N ? N : 1));
  }
  w.bytes = off;
  return w;
}

static int bin_and_sort(uint32_t N, uint32_t cap, uint32_t nth, uint32_t ntw, const int *tl,
                        const int *br, const float *depth, int *ids, int *start, int *end,
                        const BinWs &w, bool counted, uint32_t *total_out, hipStream_t s) {
  const uint32_t T = nth * ntw;
  const dim3 gN((N + kThreads - 1) / kThreads);
  if (!counted) {
    if (hipError_t e = hipMemsetAsync(w.tile_count, 0, sizeof(uint32_t) * (2 * (size_t)T + 4), s)) return (int)e;
    if (N) hipLaunchKernelGGL(k_count_rects, gN, dim3(kThreads), 0, s, N, tl, br, (int)ntw, (int)nth, w.tile_count);
  }
  hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, s, T, w.tile_count, w.tile_off, w.ctrl, cap, total_out);
  if (N) hipLaunchKernelGGL(k_emit, gN, dim3(kThreads), 0, s, N, tl, br, depth, (int)ntw, (int)nth, w.tile_off, w.tile_fill, w.ctrl, w.keys);
  hipLaunchKernelGGL(k_sort_tiles, dim3(T), dim3(kSortThreads), 0, s, T, w.tile_off, w.ctrl, w.keys, ids, start, end);
  return (int)hipGetLastError();
}

}  // namespace gs

using namespace gs;

extern "C" {

int gsgen_internal_frame_project(uint32_t N, const float *mean, const float *qvec, const float *svec,
                                 const float *cam, int w, int h, int ntw, float *mean2d, float *cov2d,
                                 float *depth, uint8_t *mask, int *tl, int *br, uint32_t *tile_count,
                                 gsgen_stream_t stream);

const char *gsgen_version(void) { return "gsgen_hip 0.1 (gfx950)"; }

const char *gsgen_error_string(int code) {
  if (code == 0) return "success";
  if (code == GSGEN_EUNSUPPORTED) return "unsupported configuration (tile_size must be 16, C in 1..4)";
  if (code == GSGEN_EINVAL) return "invalid argument (null pointer or inconsistent sizes)";
  if (code == GSGEN_EWORKSPACE) return "workspace too small";
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "unknown error";
}

size_t gsgen_tile_culling_workspace_bytes(uint32_t N, uint32_t D, uint32_t n_tiles) {
  return carve(nullptr, N, D, n_tiles, false).bytes;
}

int gsgen_tile_culling_aabb_start_end(uint32_t N, uint32_t D, uint32_t n_tiles_h,
                                      uint32_t n_tiles_w, const int *aabb_topleft,
                                      const int *aabb_bottomright, const float *depth,
                                      int *gaussian_ids, int *start, int *end, void *workspace,
                                      size_t workspace_bytes, gsgen_stream_t stream) {
  const uint32_t T = n_tiles_h * n_tiles_w;
  if (T == 0) return 0;
  if (!start || !end || !workspace) return GSGEN_EINVAL;
  if (N && (!aabb_topleft || !aabb_bottomright || !depth)) return GSGEN_EINVAL;
  if (D && !gaussian_ids) return GSGEN_EINVAL;
  const BinWs w = carve(workspace, N, D, T, false);
  if (w.bytes > workspace_bytes) return GSGEN_EWORKSPACE;
  return bin_and_sort(N, D, n_tiles_h, n_tiles_w, aabb_topleft, aabb_bottomright, depth, gaussian_ids,
                      start, end, w, false, nullptr, (hipStream_t)stream);
}

size_t gsgen_frame_workspace_bytes(uint32_t N, uint32_t D_cap, uint32_t n_tiles) {
  return carve(nullptr, N, D_cap, n_tiles, true).bytes;
}

int gsgen_frame_geometry(uint32_t N, const float *mean, const float *qvec, const float *svec,
                         const float *cam, uint32_t W, uint32_t H, uint32_t D_cap, float *mean2d,
                         float *cov2d, float *depth, uint8_t *mask, int *gaussian_ids, int *start,
                         int *end, uint32_t *total, void *workspace, size_t workspace_bytes,
                         gsgen_stream_t stream) {
  const uint32_t ntw = (W + kTile - 1) / kTile, nth = (H + kTile - 1) / kTile;
  const uint32_t T = ntw * nth;
  if (T == 0) return 0;
  if (!cam || !start || !end || !workspace || !total) return GSGEN_EINVAL;
  if (N && (!mean || !qvec || !svec || !mean2d || !cov2d || !depth || !mask)) return GSGEN_EINVAL;
  const BinWs w = carve(workspace, N, D_cap, T, true);
  if (w.bytes > workspace_bytes) return GSGEN_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  if (hipError_t e = hipMemsetAsync(w.tile_count, 0, sizeof(uint32_t) * (2 * (size_t)T + 4), s)) return (int)e;
  if (int e = gsgen_internal_frame_project(N, mean, qvec, svec, cam, (int)W, (int)H, (int)ntw, mean2d,
                                           cov2d, depth, mask, w.tl, w.br, w.tile_count, stream))
    return e;
  return bin_and_sort(N, D_cap, nth, ntw, w.tl, w.br, depth, gaussian_ids, start, end, w, true, total, s);
}

}  // extern "C"
