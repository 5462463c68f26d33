// binning.hip -- 16x16 tile binning and per-tile depth sort for gfx950.
//
// Replaces gs/src/include/aabb_culling.h:15-41 (key emission with one global atomic slot
// counter), :235-241 (one global 64-bit cub::DeviceRadixSort over all D pairs, 8 passes) and
// :70-103 (start/end from key boundaries), plus the 5 cudaMalloc/cudaFree and the blocking
// D2H copy around them (:204-229, :256-259).
//
// MI355X design: the tile id is not sorted at all, and there is not a single global atomic
// (device-scope atomics leave the XCD-private L2 and cost ~0.7 ns each on this chip: a
// scatter with one atomic per (Gaussian, tile) pair measured 480 us for 0.7 M pairs).
// Binning is a PULL: the tile grid is cut into groups of 8x8 tiles (one tile per lane of a
// wave64), the Gaussians into chunks of 2048; wave (group, chunk) streams its chunk's
// rectangles 64 at a time, ballots the ones that touch the group at all (~5 %), and for those
// every lane tests its own tile.
//   1. count       cnt[chunk][tile] = hits                                  (no atomics)
//   2. scan        per tile over chunks, then over the T tiles -> segment offsets
//   3. emit        the same walk again; lane writes (depth_bits<<32 | id) at
//                  tile_off[tile] + cnt_prefix[chunk][tile] + running   (id-ascending order)
//   4. sort        one workgroup of four wavefronts per tile sorts its segment on the 64-bit key (depth bits, then Gaussian
//                  id) in REGISTERS -- quarters per wavefront, three LDS passes for the distances that span quarters -- writes
//                  ids back, fills start/end; lists longer than 2048 entries: register blocks + merge passes over the
//                  (L2-resident) segment
// so sort traffic is one read + one write of the pairs instead of 8 global radix passes, and
// the result is deterministic: keys are unique.  Ordering semantics are the reference's: ascending UNSIGNED float bits of depth
// (the low word of its int64 key), so negative depths sort after positive ones.
// Everything is enqueued on the caller's stream with no host synchronisation.
#include "common.hpp"
#include "../../include/gsgen_hip.h"
#include <cstdlib>
#include <vector>

namespace gs {

constexpr int kGroup = 8;      // tiles per group side: 8x8 tiles <-> 64 lanes
constexpr int kChunk = 2048;   // Gaussians per chunk (round 6: 1 024 and 512 measured -- twice / four times the workgroups for the push kernels -- no gain: profiles/r06_s10_*)

struct GroupGeom {
  int gx0, gy0, tx, ty, tile;
  bool in_grid;
};
__device__ __forceinline__ GroupGeom group_geom(int ntw, int nth) {
  const int ngw = (ntw + kGroup - 1) / kGroup;
  const int g = (int)blockIdx.x;
  GroupGeom q;
  q.gx0 = (g % ngw) * kGroup;
  q.gy0 = (g / ngw) * kGroup;
  const int lane = lane_id();
  q.tx = q.gx0 + (lane & 7);
  q.ty = q.gy0 + (lane >> 3);
  q.in_grid = (q.tx < ntw) && (q.ty < nth);
  q.tile = q.ty * ntw + q.tx;
  return q;
}

// EMIT = false: cnt[chunk][tile] = number of rectangles of the chunk covering the tile, and
//               wcnt[chunk][wave][tile] = the share of each of the workgroup's 4 wavefronts.
// EMIT = true : write the keys at tile_off[tile] + cnt[chunk][tile] (now a prefix over chunks) + the
//               counts of the lower wavefronts + running.
// Workgroup = 4 waves on one (group, chunk): wave w walks sub-chunk w (kChunk/4 Gaussians),
// with all of its kIter rectangle loads in flight at once (the walk is latency-bound: one
// dependent L2 round trip per 64 Gaussians otherwise).
// The walk over the ballot broadcasts the four rectangle words of a hit (readlane) and every lane tests its
// tile.  (Tried: rectangle packed into 12 bits of group-relative coordinates, one readlane per hit and the
// covered lanes as a 64-bit mask built with scalar instructions -- the ~15 scalar instructions per hit cost
// more than the 3 readlanes + 4 compares they replace: count pass 78 -> 88 us per 8 cfg2 views.)  The count
// pass leaves the per-wavefront counts in wcnt, so the emit pass walks once (it used to count again to find
// each wavefront's base).
constexpr int kPullWaves = 4;
constexpr int kIter = kChunk / (64 * kPullWaves);  // 64-Gaussian slices per wave

__device__ __forceinline__ int rd_lane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }

struct RectRegs {
  int x0[kIter], y0[kIter], x1[kIter], y1[kIter];
};

template <bool EMIT>
__device__ __forceinline__ void
bin_pull_body(uint32_t N, const int *__restrict__ tl, const int *__restrict__ br,
           const float *__restrict__ depth, int ntw, int nth, uint32_t T,
           uint32_t *__restrict__ cnt, uint32_t *__restrict__ wcnt, const uint32_t *__restrict__ tile_off,
           const uint32_t *__restrict__ ctrl, unsigned long long *__restrict__ keys) {
  __shared__ uint32_t s_cnt[kPullWaves][64];
  if (EMIT && ctrl[1] != 0u) return;  // capacity exceeded: bin nothing
  const GroupGeom q = group_geom(ntw, nth);
  const uint32_t chunk = blockIdx.y;
  const int lane = lane_id();
  const int wave = (int)(threadIdx.x >> 6);
  const uint32_t begin = chunk * (uint32_t)kChunk + (uint32_t)wave * (uint32_t)(64 * kIter);
  const uint32_t stop = min(N, chunk * (uint32_t)kChunk + (uint32_t)kChunk);

  RectRegs R;
  unsigned dbits[EMIT ? kIter : 1];
  unsigned long long touch[kIter];
#pragma unroll
  for (int it = 0; it < kIter; ++it) {
    const uint32_t i = begin + (uint32_t)(it * 64 + lane);
    int x0 = 0, y0 = 0, x1 = -1, y1 = -1;
    if (i < stop) {
      const int2 a = *reinterpret_cast<const int2 *>(tl + 2 * (size_t)i);
      const int2 c = *reinterpret_cast<const int2 *>(br + 2 * (size_t)i);
      // rectangles handed in through the C ABI are clamped to the grid (an out-of-grid tile
      // index would be an out-of-bounds write in the reference)
      x0 = max(a.x, 0); y0 = max(a.y, 0); x1 = min(c.x, ntw - 1); y1 = min(c.y, nth - 1);
      if (EMIT) dbits[it] = __float_as_uint(depth[i]);
    } else if (EMIT) {
      dbits[it] = 0u;
    }
    const bool touches = (x1 >= x0) && (y1 >= y0) && (x1 >= q.gx0) && (x0 <= q.gx0 + kGroup - 1) &&
                         (y1 >= q.gy0) && (y0 <= q.gy0 + kGroup - 1);
    touch[it] = __ballot(touches);
    R.x0[it] = x0; R.y0[it] = y0; R.x1[it] = x1; R.y1[it] = y1;
  }
  if (!EMIT) {
    uint32_t running = 0;
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
      unsigned long long m = touch[it];
      while (m != 0ull) {
        const int src = __ffsll((long long)m) - 1;
        m &= (m - 1ull);
        const int rx0 = rd_lane(R.x0[it], src), rx1 = rd_lane(R.x1[it], src);
        const int ry0 = rd_lane(R.y0[it], src), ry1 = rd_lane(R.y1[it], src);
        const bool hit = (q.tx >= rx0) && (q.tx <= rx1) && (q.ty >= ry0) && (q.ty <= ry1);
        running += hit ? 1u : 0u;
      }
    }
    if (q.in_grid) wcnt[((size_t)chunk * kPullWaves + (size_t)wave) * T + q.tile] = running;
    s_cnt[wave][lane] = running;
    __syncthreads();
    if (wave == 0 && q.in_grid)
      cnt[(size_t)chunk * T + q.tile] = s_cnt[0][lane] + s_cnt[1][lane] + s_cnt[2][lane] + s_cnt[3][lane];
    return;
  }
  uint32_t pos = 0;
  if (q.in_grid) {
    pos = tile_off[q.tile] + cnt[(size_t)chunk * T + q.tile];
    for (int w = 0; w < wave; ++w) pos += wcnt[((size_t)chunk * kPullWaves + (size_t)w) * T + q.tile];
  }
#pragma unroll
  for (int it = 0; it < kIter; ++it) {
    unsigned long long m = touch[it];
    const uint32_t i0 = begin + (uint32_t)(it * 64);
    while (m != 0ull) {
      const int src = __ffsll((long long)m) - 1;
      m &= (m - 1ull);
      const int rx0 = rd_lane(R.x0[it], src), rx1 = rd_lane(R.x1[it], src);
      const int ry0 = rd_lane(R.y0[it], src), ry1 = rd_lane(R.y1[it], src);
      const bool hit = (q.tx >= rx0) && (q.tx <= rx1) && (q.ty >= ry0) && (q.ty <= ry1);
      const unsigned db = (unsigned)rd_lane((int)dbits[EMIT ? it : 0], src);
      if (hit) {
        keys[pos] = ((unsigned long long)db << 32) | (unsigned long long)(i0 + (uint32_t)src);
        ++pos;
      }
    }
  }
}

// per tile: exclusive scan of cnt[.][tile] over the chunks (in place), total -> tile_count (scan_chunks_body below).
// Round 5: lanes <-> TILES (64 consecutive tiles per workgroup), the chunks walked serially -- every load and store of a
// wavefront is one contiguous 256-byte run of cnt[chunk][tile0 .. tile0 + 63].  (Until round 4: one wavefront per tile, lanes
// <-> chunks: each lane's 4 bytes lay T words from its neighbour's, a cache line per lane -- 209 MB moved per 4-view launch of
// BASELINE configs[2] for 8 T chunks = 32 MB of scan, profiles/r04_traffic.json.)  Four wavefronts per workgroup split the
// chunk range: each sums its quarter, the quarters' sums meet in LDS, each then rewrites its quarter with the exclusive
// prefixes: 2 reads (the second from L2) + 1 write per counter.
// NW wavefronts per workgroup: 4 in a camera batch (hundreds of workgroups), 16 for a lone camera (40 workgroups at 800 x 800:
// the chain of dependent chunk steps per wavefront is what a lone render waits for).
constexpr uint32_t kScanChunkTiles = 64;
template <uint32_t NW>
__device__ __forceinline__ void
scan_chunks_body(uint32_t T, uint32_t nchunks, uint32_t *__restrict__ cnt, uint32_t *__restrict__ tile_count) {
  __shared__ uint32_t s_part[NW][kScanChunkTiles];
  const uint32_t lane = (uint32_t)lane_id(), w = threadIdx.x >> 6;
  const uint32_t t = blockIdx.x * kScanChunkTiles + lane;
  const uint32_t q = (nchunks + NW - 1u) / NW;
  const uint32_t c0 = min(w * q, nchunks), c1 = min(c0 + q, nchunks);
  uint32_t sum = 0;
  if (t < T)
    for (uint32_t c = c0; c < c1; ++c) sum += cnt[(size_t)c * T + t];
  s_part[w][lane] = sum;
  __syncthreads();
  uint32_t run = 0;
  for (uint32_t k = 0; k < w; ++k) run += s_part[k][lane];
  if (t < T) {
    for (uint32_t c = c0; c < c1; ++c) {
      const uint32_t v = cnt[(size_t)c * T + t];
      cnt[(size_t)c * T + t] = run;
      run += v;
    }
    if (w == NW - 1u) tile_count[t] = run;  // (the last part ends on the tile's total, also when its chunk range is empty)
  }
}

// exclusive scan of tile_count[T] -> tile_off[T+1]; ctrl[0] = total, ctrl[1] = overflow flag.
// Sums SATURATE at 2^32 - 1: a pair count beyond 32 bits (a diverged scene: millions of Gaussians each covering every tile)
// must read as "does not fit", never wrap round to a small number that does (the emit pass would then write past the
// pair buffer).  The offsets of an overflowing frame are not used: ctrl[1] makes every later stage bin nothing.
__device__ __forceinline__ uint32_t sat_add_u32(uint32_t a, uint32_t b) {
  const uint32_t s = a + b;
  return s < a ? 0xffffffffu : s;
}

// One workgroup of kScanThreads = 256 (four wavefronts, one per SIMD): with 1024 threads this link of the chain needed 16
// free wave slots on ONE compute unit and waited 0.3 ms for them whenever another batch's compositing launch filled the chip
// (profiles/r03_notes.md: 7.9 us alone, 299 us average with three steps in flight).
constexpr uint32_t kScanThreads = 256;
// report (optional): two words the HOST can read while the stream runs (pinned, device-mapped host memory): [0] takes every
// frame's pair count (so the host can grow a buffer BEFORE it overflows), [1] the largest count of a frame that did not fit
// (never overwritten by a later frame that does: the host clears it when it has dealt with it).  One 4-byte store over the
// fabric per view and frame; nothing is copied, no event, nothing the host waits for, and a hipGraph replay reports the same way.
__device__ __forceinline__ void report_pair_count(uint32_t *report, uint32_t total, bool overflow) {
  if (report == nullptr) return;
#if defined(__HIP_DEVICE_COMPILE__)
  __hip_atomic_store(report, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (overflow) __hip_atomic_fetch_max(report + 1, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#else
  report[0] = total;
  if (overflow && report[1] < total) report[1] = total;
#endif
}
__device__ __forceinline__ void
scan_tiles_body(uint32_t T, const uint32_t *__restrict__ tile_count, uint32_t *__restrict__ tile_off,
             uint32_t *__restrict__ ctrl, uint32_t cap, uint32_t *__restrict__ total_out, uint32_t *report = nullptr) {
  __shared__ uint32_t s_part[kScanThreads];
  const uint32_t t = threadIdx.x;
  const uint32_t per = (T + kScanThreads - 1u) / kScanThreads;
  const uint32_t b = min(t * per, T), e = min(b + per, T);
  uint32_t sum = 0;
  for (uint32_t i = b; i < e; ++i) sum = sat_add_u32(sum, tile_count[i]);
  s_part[t] = sum;
  __syncthreads();
  // Hillis-Steele inclusive scan over the partials
  for (uint32_t d = 1; d < kScanThreads; d <<= 1) {
    const uint32_t v = (t >= d) ? s_part[t - d] : 0u;
    __syncthreads();
    s_part[t] = sat_add_u32(s_part[t], v);
    __syncthreads();
  }
  uint32_t run = (t > 0) ? s_part[t - 1] : 0u;  // exclusive prefix of this thread's chunk
  for (uint32_t i = b; i < e; ++i) {
    tile_off[i] = run;
    run = sat_add_u32(run, tile_count[i]);
  }
  if (t == kScanThreads - 1u) {
    const uint32_t total = s_part[kScanThreads - 1u];
    tile_off[T] = total;
    ctrl[0] = total;
    ctrl[1] = (total > cap) ? 1u : 0u;
    if (total_out != nullptr) *total_out = total;
    report_pair_count(report, total, total > cap);
  }
}

// Longest-list-first launch order for the compositing kernels (LPT scheduling): a tile's
// cost grows with its list length, the image centre carries lists several times longer than
// the border, and a compositing launch does not fit the chip all at once -- started in spatial
// order the long tiles begin late and the launch ends on a few stragglers.  Counting sort of
// the tiles by (list length / 8) descending, one workgroup, LDS atomics only.
// n_long (ctrl[2]) = the number of tiles at the front of the order whose list is kWaveSortMax entries or longer: the sort launch
// gives those a workgroup each and packs the rest four to a workgroup (sort_tiles_packed_body).
constexpr uint32_t kWaveSortMax = 512;  // a multiple of 8; lists SHORTER than this are sorted by one wavefront in registers
__device__ __forceinline__ void
order_tiles_body(uint32_t T, const uint32_t *__restrict__ tile_count, uint32_t *__restrict__ tile_order, uint32_t *__restrict__ n_long) {
  __shared__ uint32_t s_hist[256];
  __shared__ uint32_t s_base[256];
  const uint32_t t = threadIdx.x;
  if (t < 256u) s_hist[t] = 0u;
  __syncthreads();
  for (uint32_t i = t; i < T; i += kScanThreads) atomicAdd(&s_hist[255u - min(tile_count[i] >> 3, 255u)], 1u);
  __syncthreads();
  if (t == 0) {
    uint32_t run = 0;
    for (int b = 0; b < 256; ++b) {
      if (b == 256 - (int)(kWaveSortMax >> 3)) *n_long = run;  // bins 0 .. b-1: count >> 3 >= kWaveSortMax / 8
      s_base[b] = run;
      run += s_hist[b];
    }
  }
  __syncthreads();
  for (uint32_t i = t; i < T; i += kScanThreads) {
    const uint32_t b = 255u - min(tile_count[i] >> 3, 255u);
    tile_order[atomicAdd(&s_base[b], 1u)] = i;
  }
}

// normalised bitonic network (every comparator puts the smaller key at the lower index), so
// a segment of arbitrary length n behaves as if padded with +inf up to the next power of two.
// Register form: ONE wavefront sorts 64*K keys, K per lane (element e = lane*K + r), with no
// memory traffic and no barrier: comparators whose partner lies in the same lane are plain
// register compare-exchanges, the others trade registers with lane ^ m.
typedef unsigned long long u64;
__device__ __forceinline__ void cmpx(u64 &lo, u64 &hi) {
  const u64 a = lo, b = hi;
  const bool sw = a > b;
  lo = sw ? b : a;
  hi = sw ? a : b;
}
template <int K>
__device__ __forceinline__ void cross_step(u64 (&k)[K], int m, bool flip) {
  // partner lane = lane ^ m; in a flip step the partner register is K-1-r, else r
  const int lane = lane_id();
  int top = m;  // highest set bit of m decides who is the lower index
  top |= top >> 1; top |= top >> 2; top |= top >> 4;
  top = (top + 1) >> 1;
  // the lower index keeps the smaller key, the upper one the larger: ONE comparison, flipped in the upper lanes (equal keys -- the
  // +inf padding -- may trade places: the same bits).  (As `keep_min ? theirs < mine : theirs > mine` this compiled to two
  // exec-masked branches per key: twice as many scalar as vector instructions in the launch.)
  const bool upper = (lane & top) != 0;
  auto pick = [&](u64 mine, u64 theirs) {
    const bool take = (theirs < mine) != upper;
    const uint32_t lo = take ? (uint32_t)theirs : (uint32_t)mine, hi = take ? (uint32_t)(theirs >> 32) : (uint32_t)(mine >> 32);
    return ((u64)hi << 32) | lo;
  };
  if (flip) {
    // my register r meets the partner's register K-1-r: handle (r, K-1-r) together so that no
    // copy of the whole key array is needed (K = 32 would spill otherwise)
    if constexpr (K == 1) {
      k[0] = pick(k[0], __shfl_xor(k[0], m, 64));
    } else {
#pragma unroll
      for (int r = 0; r < K / 2; ++r) {
        const u64 a = k[r], b = k[K - 1 - r];
        const u64 oa = __shfl_xor(b, m, 64), ob = __shfl_xor(a, m, 64);
        k[r] = pick(a, oa);
        k[K - 1 - r] = pick(b, ob);
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < K; ++r) k[r] = pick(k[r], __shfl_xor(k[r], m, 64));
  }
}
// all comparators at distance J inside a lane (compile-time register indices)
template <int K, int J>
__device__ __forceinline__ void local_step(u64 (&k)[K]) {
#pragma unroll
  for (int r = 0; r < K; ++r) {
    if ((r & J) == 0) cmpx(k[r], k[r | J]);
  }
}
template <int K, int J>
__device__ __forceinline__ void local_tail(u64 (&k)[K]) {  // distances J, J/2, ..., 1
  if constexpr (J >= 1) {
    local_step<K, J>(k);
    local_tail<K, J / 2>(k);
  }
}
// merge stages whose blocks fit in a lane: size = 2 .. K (compile time)
template <int K, int SIZE>
__device__ __forceinline__ void local_sizes(u64 (&k)[K]) {
  if constexpr (SIZE <= K) {
#pragma unroll
    for (int r = 0; r < K; ++r) {
      const int l = r ^ (SIZE - 1);
      if (l > r) cmpx(k[r], k[l]);
    }
    local_tail<K, SIZE / 4>(k);
    local_sizes<K, SIZE * 2>(k);
  }
}
template <int K>
__device__ __forceinline__ void sort_regs(u64 (&k)[K]) {
  local_sizes<K, 2>(k);
  // merge stages spanning lanes: the bodies below are the same code for every size (only the
  // xor mask changes), so this is a RUNTIME loop -- fully unrolling 66 stages of 32 registers
  // made the compiler put the key array into scratch.
  for (int size = 2 * K; size <= 64 * K; size <<= 1) {
    cross_step<K>(k, size / K - 1, true);
    for (int j = size >> 2; j >= K; j >>= 1) cross_step<K>(k, j / K, false);
    local_tail<K, K / 2>(k);
  }
}
template <int K>
__device__ __forceinline__ void sort_segment_regs(const u64 *__restrict__ keys, int *__restrict__ ids, uint32_t n) {
  const uint32_t lane = (uint32_t)lane_id();
  u64 k[K];
#pragma unroll
  for (int r = 0; r < K; ++r) {
    const uint32_t e = lane * K + r;
    k[r] = (e < n) ? keys[e] : ~0ull;  // +inf padding
  }
  sort_regs<K>(k);
#pragma unroll
  for (int r = 0; r < K; ++r) {
    const uint32_t e = lane * K + r;
    if (e < n) ids[e] = (int)(uint32_t)(k[r] & 0xffffffffull);
  }
}
// Lists beyond the four-wavefront path (more than 2 048 entries: dense clusters; nothing on the BASELINE workloads): ONE
// wavefront sorts blocks of 64 K entries in registers and runs the merge stages of the network that span blocks -- distance
// >= a block -- as passes over the (L2-resident) global segment, four comparators per lane in flight, and every stage's
// remaining distances in registers again, block by block.  The workgroup-wide network in global memory this replaces (its own
// launch, one more link in every frame's chain, 91 barrier-separated passes for an 8 192-entry list) cost every frame a launch
// that normally found nothing to do.
__device__ __forceinline__ void long_pass(u64 *__restrict__ k, uint32_t n, uint32_t p2, uint32_t dist, bool flip) {
  // comparator c of p2/2: i = (c / dist) * 2 dist + c % dist; partner i ^ (2 dist - 1) (flip) or i + dist
  const uint32_t lane = (uint32_t)lane_id();
  const uint32_t half = p2 >> 1;
  for (uint32_t cbase = 0; cbase < half; cbase += 256u) {
    u64 a[4], b[4];
    uint32_t ia[4], ib[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t c = cbase + (uint32_t)u * 64u + lane;
      const uint32_t off = c & (dist - 1u);
      ia[u] = ((c - off) << 1) + off;
      ib[u] = flip ? (ia[u] ^ (2u * dist - 1u)) : (ia[u] + dist);
      const bool live = c < half && ib[u] < n;  // ia < ib always; a partner beyond n is +inf padding: nothing moves
      if (!live) ib[u] = 0xffffffffu;
      a[u] = live ? k[ia[u]] : 0ull;
      b[u] = live ? k[ib[u]] : ~0ull;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (ib[u] != 0xffffffffu && a[u] > b[u]) { k[ia[u]] = b[u]; k[ib[u]] = a[u]; }
  }
  wave_mem_fence();  // ONE wavefront runs these passes (the others of the workgroup may have left): orders this pass's stores
                     // before the next pass's loads without a workgroup barrier
}
template <int K>
__device__ __forceinline__ void sort_segment_long(u64 *__restrict__ keys, int *__restrict__ ids, uint32_t n) {
  constexpr uint32_t kBlock = 64u * K;
  const uint32_t lane = (uint32_t)lane_id();
  const uint32_t nblk = (n + kBlock - 1u) / kBlock;
  u64 k[K];
  for (uint32_t blk = 0; blk < nblk; ++blk) {
    const uint32_t e0 = blk * kBlock + lane * K;
#pragma unroll
    for (int r = 0; r < K; ++r) k[r] = (e0 + r < n) ? keys[e0 + r] : ~0ull;
    sort_regs<K>(k);
    if (nblk == 1u) {
#pragma unroll
      for (int r = 0; r < K; ++r)
        if (e0 + r < n) ids[e0 + r] = (int)(uint32_t)(k[r] & 0xffffffffull);
      return;
    }
#pragma unroll
    for (int r = 0; r < K; ++r)
      if (e0 + r < n) keys[e0 + r] = k[r];
  }
  wave_mem_fence();
  uint32_t p2 = kBlock;
  while (p2 < n) p2 <<= 1;
  for (uint32_t size = 2u * kBlock; size <= p2; size <<= 1) {
    long_pass(keys, n, p2, size >> 1, true);
    for (uint32_t j = size >> 2; j >= kBlock; j >>= 1) long_pass(keys, n, p2, j, false);
    const bool last = size == p2;
    for (uint32_t blk = 0; blk < nblk; ++blk) {
      const uint32_t e0 = blk * kBlock + lane * K;
#pragma unroll
      for (int r = 0; r < K; ++r) k[r] = (e0 + r < n) ? keys[e0 + r] : ~0ull;
      for (int j = (int)kBlock / 2; j >= K; j >>= 1) cross_step<K>(k, j / K, false);
      local_tail<K, K / 2>(k);
#pragma unroll
      for (int r = 0; r < K; ++r) {
        if (e0 + r < n) {
          if (last) ids[e0 + r] = (int)(uint32_t)(k[r] & 0xffffffffull);
          else keys[e0 + r] = k[r];
        }
      }
    }
    wave_mem_fence();
  }
}

// The per-tile sort launch: one workgroup of FOUR wavefronts per tile.  Up to 256 entries: wavefront 0 sorts them in registers
// (K = 1 / 2 / 4 keys per lane; the others leave at once).  257 .. 2 048: each wavefront sorts a quarter in registers (K = 2 / 4 / 8),
// the two merge stages that span quarters run their cross-quarter distances as three passes over LDS (256 threads, a barrier
// each) and every stage's distances inside a quarter in registers again.  Beyond 2 048: wavefront 0's block sort + merge passes
// over the segment (sort_segment_long).  One wavefront per tile with up to 32 keys per lane was the shape until round 3: a lone
// view's 2 500 wavefronts all fit the chip at once, nothing balances them, and the launch lasted as long as the SIMD that drew
// three 1 024-key tiles (43 us against 28 now); in the 8-view launches (longest list first, dynamically placed) the two shapes
// take the same time alone (114 / 117 us) and this one -- 46 registers instead of 145 -- finds room beside the compositing
// kernels sooner: 0.60 instead of 0.80 ms average in flight, +0.7 .. 1.6 % renders/s (profiles/r03_notes.md, session y).
constexpr int kCoopWaves = 4;
constexpr uint32_t kCoopMax = 2048;
__device__ __forceinline__ void lds_pass(u64 *k, uint32_t p2, uint32_t dist, bool flip) {
  for (uint32_t c = threadIdx.x; c < (p2 >> 1); c += 64u * kCoopWaves) {
    const uint32_t off = c & (dist - 1u);
    const uint32_t i = ((c - off) << 1) + off;
    const uint32_t l = flip ? (i ^ (2u * dist - 1u)) : (i + dist);
    const u64 a = k[i], b = k[l];  // (the segment is padded with +inf up to p2: no bounds to test)
    if (a > b) { k[i] = b; k[l] = a; }
  }
  __syncthreads();
}
template <int KW>
__device__ __forceinline__ void sort_segment_coop(const u64 *__restrict__ keys, int *__restrict__ ids, uint32_t n, u64 *s_keys) {
  constexpr uint32_t kBlk = 64u * KW;  // a wavefront's quarter; p2 = 4 kBlk
  const uint32_t wave = threadIdx.x >> 6, lane = (uint32_t)lane_id();
  const uint32_t e0 = wave * kBlk + lane * KW;
  u64 k[KW];
#pragma unroll
  for (int r = 0; r < KW; ++r) k[r] = (e0 + r < n) ? keys[e0 + r] : ~0ull;
  sort_regs<KW>(k);
  auto to_lds = [&]() {
#pragma unroll
    for (int r = 0; r < KW; ++r) s_keys[e0 + r] = k[r];
    __syncthreads();
  };
  auto tail_in_registers = [&]() {  // a stage's distances kBlk / 2 .. 1: inside the quarter
#pragma unroll
    for (int r = 0; r < KW; ++r) k[r] = s_keys[e0 + r];
    for (int j = (int)kBlk / 2; j >= KW; j >>= 1) cross_step<KW>(k, j / KW, false);
    local_tail<KW, KW / 2>(k);
  };
  to_lds();
  lds_pass(s_keys, 4u * kBlk, kBlk, true);         // stage 2 kBlk: its flip step
  tail_in_registers();
  __syncthreads();                                 // (everyone has read its quarter before it is overwritten)
  to_lds();
  lds_pass(s_keys, 4u * kBlk, 2u * kBlk, true);    // stage 4 kBlk: flip, then distance kBlk
  lds_pass(s_keys, 4u * kBlk, kBlk, false);
  tail_in_registers();
#pragma unroll
  for (int r = 0; r < KW; ++r)
    if (e0 + r < n) ids[e0 + r] = (int)(uint32_t)(k[r] & 0xffffffffull);
}
__device__ __forceinline__ void
sort_tiles_coop_body(uint32_t tile, const uint32_t *__restrict__ tile_off, const uint32_t *__restrict__ ctrl,
                     unsigned long long *__restrict__ keys, int *__restrict__ ids, int *__restrict__ start,
                     int *__restrict__ end, u64 *s_keys) {
  const uint32_t tid = threadIdx.x;
  if (ctrl[1] != 0u) {  // the frame's pairs do not fit the caller's list: NOT "every tile empty" (that is a valid, finite, blank
    // image) but kListOverflow -- the compositing forwards write NaN into such a tile, the backwards skip it
    if (tid == 0) { start[tile] = kListOverflow; end[tile] = kListOverflow; }
    return;
  }
  const uint32_t b = tile_off[tile], e = tile_off[tile + 1];
  const uint32_t n = e - b;
  if (tid == 0) {  // empty tiles stay -1 (aabb_culling.h:248-249)
    start[tile] = n ? (int)b : -1;
    end[tile] = n ? (int)e : -1;
  }
  if (n == 0) return;
  if (n <= 256u || n > kCoopMax) {  // (uniform over the workgroup)
    if (tid >= 64u) return;
    if (n <= 64u) sort_segment_regs<1>(keys + b, ids + b, n);
    else if (n <= 128u) sort_segment_regs<2>(keys + b, ids + b, n);
    else if (n <= 256u) sort_segment_regs<4>(keys + b, ids + b, n);
    else sort_segment_long<8>(keys + b, ids + b, n);  // (512-entry blocks: the kernel stays at its quarter sorts' registers)
    return;
  }
  if (n <= 512u) sort_segment_coop<2>(keys + b, ids + b, n, s_keys);
  else if (n <= 1024u) sort_segment_coop<4>(keys + b, ids + b, n, s_keys);
  else sort_segment_coop<8>(keys + b, ids + b, n, s_keys);
}
// With a launch order (order_tiles_body: longest list first, ctrl[2] = how many lists reach kWaveSortMax): workgroup `slot` of a
// view takes the slot-th tile while those last, and FOUR of the shorter ones after that, a wavefront each -- a list one
// wavefront sorts in registers leaves the other three with nothing to do, and the workgroup's 16 KB of LDS (ten workgroups a
// compute unit) then held the short half of a frame's tiles to ten wavefronts a compute unit on a launch that waits on
// cross-lane round trips.  (The surplus workgroups at the end of the grid leave at once.)
__device__ __forceinline__ void
sort_tiles_packed_body(uint32_t slot, uint32_t T, const uint32_t *__restrict__ tile_order, const uint32_t *__restrict__ tile_off,
                       const uint32_t *__restrict__ ctrl, unsigned long long *__restrict__ keys, int *__restrict__ ids,
                       int *__restrict__ start, int *__restrict__ end, u64 *s_keys) {
  const uint32_t n_long = min(ctrl[2], T);
  if (slot < n_long) {
    sort_tiles_coop_body(tile_order[slot], tile_off, ctrl, keys, ids, start, end, s_keys);
    return;
  }
  const uint32_t at = n_long + (slot - n_long) * (uint32_t)kCoopWaves + (threadIdx.x >> 6);
  if (at >= T) return;
  const uint32_t tile = tile_order[at];
  const bool first = lane_id() == 0;
  if (ctrl[1] != 0u) {  // (see sort_tiles_coop_body)
    if (first) { start[tile] = kListOverflow; end[tile] = kListOverflow; }
    return;
  }
  const uint32_t b = tile_off[tile], e = tile_off[tile + 1];
  const uint32_t n = e - b;
  if (first) {
    start[tile] = n ? (int)b : -1;
    end[tile] = n ? (int)e : -1;
  }
  if (n == 0) return;
  if (n <= 64u) sort_segment_regs<1>(keys + b, ids + b, n);
  else if (n <= 128u) sort_segment_regs<2>(keys + b, ids + b, n);
  else if (n <= 256u) sort_segment_regs<4>(keys + b, ids + b, n);
  else sort_segment_regs<8>(keys + b, ids + b, n);  // (kWaveSortMax = 512)
}
static_assert(kWaveSortMax <= 512 && kWaveSortMax % 8 == 0 && kWaveSortMax <= 255 * 8, "sort_tiles_packed_body");
__global__ void __launch_bounds__(64 * kCoopWaves)
k_sort_tiles(uint32_t T, const uint32_t *__restrict__ tile_off, const uint32_t *__restrict__ ctrl,
                  unsigned long long *__restrict__ keys, int *__restrict__ ids, int *__restrict__ start,
                  int *__restrict__ end, const uint32_t *__restrict__ tile_order) {
  __shared__ u64 s_keys[kCoopMax];
  if (tile_order != nullptr) sort_tiles_packed_body(blockIdx.x, T, tile_order, tile_off, ctrl, keys, ids, start, end, s_keys);
  else sort_tiles_coop_body(blockIdx.x, tile_off, ctrl, keys, ids, start, end, s_keys);  // (legacy.hip's segments: no order)
}
// 1-D grid of T x B workgroups, view = id % B, tiles in order_tiles_body order (longest list first): every view's long sorts
// start at once and the launch ends on the short ones.  (View-major order measured 134 us for 8 cfg2 views with one wavefront
// per tile: ~2.5 views resident at a time, each waiting on its longest tile.)
__global__ void __launch_bounds__(64 * kCoopWaves)
k_sort_tiles_views(uint32_t T, uint32_t B, const GeoView *__restrict__ views) {
  __shared__ u64 s_keys[kCoopMax];
  const uint32_t rank = blockIdx.x / B;
  const GeoView v = views[blockIdx.x - rank * B];
  sort_tiles_packed_body(rank, T, v.tile_order, v.tile_off, v.ctrl, v.keys, v.ids, v.start, v.end, s_keys);
}

// ---- self test of the cross-lane primitives (tests/ only; exported for the parity suite) ----
// ---- kernels: one view per launch (arguments in the kernel arguments) and B views per launch
// (gridDim.y or .z = view, the per-view pointers read from a GeoView table in device memory) --------
template <bool EMIT>
__global__ void __launch_bounds__(64 * kPullWaves)
k_bin_pull(uint32_t N, const int *__restrict__ tl, const int *__restrict__ br,
           const float *__restrict__ depth, int ntw, int nth, uint32_t T,
           uint32_t *__restrict__ cnt, uint32_t *__restrict__ wcnt, const uint32_t *__restrict__ tile_off,
           const uint32_t *__restrict__ ctrl, unsigned long long *__restrict__ keys) {
  bin_pull_body<EMIT>(N, tl, br, depth, ntw, nth, T, cnt, wcnt, tile_off, ctrl, keys);
}
template <bool EMIT>
__global__ void __launch_bounds__(64 * kPullWaves)
k_bin_pull_views(uint32_t N, int ntw, int nth, uint32_t T, const GeoView *__restrict__ views) {
  const GeoView v = views[blockIdx.z];
  bin_pull_body<EMIT>(N, v.tl, v.br, v.depth, ntw, nth, T, v.cnt, v.wcnt, v.tile_off, v.ctrl, v.keys);
}
__global__ void __launch_bounds__(1024)
k_scan_chunks(uint32_t T, uint32_t nchunks, uint32_t *__restrict__ cnt, uint32_t *__restrict__ tile_count) {
  scan_chunks_body<16>(T, nchunks, cnt, tile_count);
}
__global__ void __launch_bounds__(256)
k_scan_chunks_views(uint32_t T, uint32_t nchunks, const GeoView *__restrict__ views) {
  const GeoView v = views[blockIdx.y];
  scan_chunks_body<4>(T, nchunks, v.cnt, v.tile_count);
}
// tile offsets and the longest-first launch order in ONE launch: both read tile_count only, both are one workgroup --
// as two kernels they were two ~5 us links in a lone render's chain of dependent launches
__global__ void __launch_bounds__(kScanThreads)
k_scan_order_tiles(uint32_t T, const uint32_t *__restrict__ tile_count, uint32_t *__restrict__ tile_off,
                   uint32_t *__restrict__ ctrl, uint32_t cap, uint32_t *__restrict__ total_out,
                   uint32_t *__restrict__ tile_order, uint32_t *report) {
  scan_tiles_body(T, tile_count, tile_off, ctrl, cap, total_out, report);
  order_tiles_body(T, tile_count, tile_order, ctrl + 2);
}
__global__ void __launch_bounds__(kScanThreads)
k_scan_order_tiles_views(uint32_t T, const GeoView *__restrict__ views) {
  const GeoView v = views[blockIdx.y];
  scan_tiles_body(T, v.tile_count, v.tile_off, v.ctrl, v.cap, v.total, v.report);
  order_tiles_body(T, v.tile_count, v.tile_order, v.ctrl + 2);
}
// ---- push binning of a camera batch (round 4) ------------------------------------------------------------------------------
// The pull kernels above -- every 8 x 8-tile group scans every 2 048-Gaussian chunk -- issue 25 + 31 M vector instructions per
// 8 cfg2 views for 6 M (Gaussian, tile) pairs: a fifth of the RGB + heads step's.  Here a workgroup takes one chunk of ONE view,
// every thread walks the rectangles of its Gaussians, and the per-tile counters live in LDS (T words: LDS atomics, a few dozen
// per counter and chunk): COUNT leaves cnt[chunk][tile] exactly as the pull kernel does (same numbers, same layout, the same
// k_scan_chunks behind it), EMIT starts each counter at the tile's segment offset + the chunk's prefix and takes slots from
// it.  The slots of one chunk inside a segment come out in no particular order -- the sort behind it orders by (depth, id),
// keys are unique, so the lists are the same bits as ever.  A rectangle of more than kPushOwn tiles is finished by the whole
// wavefront (ballot, broadcast, 64 tiles at a time): one near Gaussian does not make 63 lanes wait for its 200 tiles.
// (Counters in GLOBAL memory -- one atomic per pair on tile_count, no chunks -- were measured first: 3 620 instead of 5 440
// renders/s, the hot tiles' counters serialise a thousand returning atomics each.)  Images beyond kPushMaxTiles tiles keep
// the pull kernels.
constexpr int kPushOwn = 12;
constexpr uint32_t kPushMaxTiles = 8192;  // 32 KB of LDS: 2 048 x 1 024 pixels and the like
constexpr int kPushThreads = 512;  // (round 6: 256 -> 512: the count launch 25 -> 18 us, the emit 72 -> 66 us per 8 cfg2 views alone; 1 024: 17 / 64, no better in flight: profiles/r06_s22_*)
constexpr uint32_t kPushMinWorkgroups = 128;  // (chunks x views) below which the pull kernels are the faster launch
// A camera batch is a throughput launch (other steps' kernels fill the chip around it): there the fewer instructions win from
// a quarter of that on.
constexpr uint32_t kPushMinWorkgroupsBatch = 32;
static uint32_t push_min_workgroups(bool batch) {
#if defined(GSGEN_EMU_KNOBS)  // the CPU emulator build of the tests only (oracle/Makefile): their scenes are one or two chunks, the
  // variable lets them reach both forms; the product library reads nothing from the environment
  if (const char *e = getenv("GSGEN_BIN_PUSH_MIN_WORKGROUPS")) return (uint32_t)strtoul(e, nullptr, 10);
#endif
  return batch ? kPushMinWorkgroupsBatch : kPushMinWorkgroups;
}
// the counters: T words of dynamic LDS (10 KB at 800 x 800: a workgroup finds room beside the compositing kernels' blocks
// sooner than with the 32 KB of the largest image); the CPU emulator build has no dynamic LDS and takes the maximum
#if defined(__HIP_DEVICE_COMPILE__)
#define GS_PUSH_COUNTERS(name) extern __shared__ uint32_t name[]
#else
#define GS_PUSH_COUNTERS(name) __shared__ uint32_t name[kPushMaxTiles]
#endif
template <bool EMIT>
__device__ __forceinline__ void
bin_push_body(uint32_t N, const int *__restrict__ tl, const int *__restrict__ br, const float *__restrict__ depth, int ntw, int nth,
              uint32_t T, uint32_t *__restrict__ cnt, const uint32_t *__restrict__ tile_off, const uint32_t *__restrict__ ctrl,
              unsigned long long *__restrict__ keys, uint32_t *s_tile) {
  if (EMIT && ctrl[1] != 0u) return;  // capacity exceeded: bin nothing
  const uint32_t chunk = blockIdx.x;
  uint32_t *const crow = cnt + (size_t)chunk * T;
  for (uint32_t t = threadIdx.x; t < T; t += (uint32_t)kPushThreads) s_tile[t] = EMIT ? tile_off[t] + crow[t] : 0u;
  __syncthreads();
  const int lane = lane_id();
  const uint32_t stop = min(N, chunk * (uint32_t)kChunk + (uint32_t)kChunk);
  auto visit = [&](int tile, uint32_t id, unsigned dbits) {
    const uint32_t pos = atomicAdd(&s_tile[tile], 1u);
    if (EMIT) keys[pos] = ((unsigned long long)dbits << 32) | (unsigned long long)id;
  };
  for (uint32_t i0 = chunk * (uint32_t)kChunk; i0 < stop; i0 += (uint32_t)kPushThreads) {  // (uniform trip count: ballots inside)
    const uint32_t i = i0 + threadIdx.x;
    int x0 = 0, y0 = 0, x1 = -1, y1 = -1;
    unsigned db = 0u;
    if (i < stop) {
      const int2 a = *reinterpret_cast<const int2 *>(tl + 2 * (size_t)i);
      const int2 c = *reinterpret_cast<const int2 *>(br + 2 * (size_t)i);
      // rectangles clamped to the grid, as the pull kernels take them
      x0 = max(a.x, 0); y0 = max(a.y, 0); x1 = min(c.x, ntw - 1); y1 = min(c.y, nth - 1);
      if (EMIT) db = __float_as_uint(depth[i]);
    }
    const int rw = x1 - x0 + 1, rh = y1 - y0 + 1;
    const int n = (rw > 0 && rh > 0) ? rw * rh : 0;
    {
      int tx = x0, ty = y0;
      const int own = min(n, kPushOwn);
      for (int k = 0; k < own; ++k) {
        visit(ty * ntw + tx, i, db);
        if (++tx > x1) { tx = x0; ++ty; }
      }
    }
    unsigned long long big = __ballot(n > kPushOwn);
    while (big != 0ull) {  // (wave-uniform)
      const int src = __ffsll((long long)big) - 1;
      big &= (big - 1ull);
      const int bx0 = rd_lane(x0, src), by0 = rd_lane(y0, src), bw = rd_lane(rw, src), bn = rd_lane(n, src);
      const uint32_t bid = (uint32_t)rd_lane((int)i, src);
      const unsigned bdb = (unsigned)rd_lane((int)db, src);
      for (int k = kPushOwn + lane; k < bn; k += 64) {
        const int ry = k / bw;
        visit((by0 + ry) * ntw + bx0 + (k - ry * bw), bid, bdb);
      }
    }
  }
  if (!EMIT) {
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < T; t += (uint32_t)kPushThreads) crow[t] = s_tile[t];
  }
}

template <bool EMIT>
__global__ void __launch_bounds__(kPushThreads)
k_bin_push(uint32_t N, const int *__restrict__ tl, const int *__restrict__ br, const float *__restrict__ depth, int ntw, int nth,
           uint32_t T, uint32_t *__restrict__ cnt, const uint32_t *__restrict__ tile_off, const uint32_t *__restrict__ ctrl,
           unsigned long long *__restrict__ keys) {
  GS_PUSH_COUNTERS(s_tile);
  bin_push_body<EMIT>(N, tl, br, depth, ntw, nth, T, cnt, tile_off, ctrl, keys, s_tile);
}
template <bool EMIT>
__global__ void __launch_bounds__(kPushThreads)
k_bin_push_views(uint32_t N, int ntw, int nth, uint32_t T, const GeoView *__restrict__ views) {
  GS_PUSH_COUNTERS(s_tile);
  const GeoView v = views[blockIdx.y];
  bin_push_body<EMIT>(N, v.tl, v.br, v.depth, ntw, nth, T, v.cnt, v.tile_off, v.ctrl, v.keys, s_tile);
}

template <int P>
__global__ void __launch_bounds__(64) k_selftest_reduce_scatter(const float *__restrict__ in, float *__restrict__ out) {
  float v[P];
  const int lane = lane_id();
#pragma unroll
  for (int i = 0; i < P; ++i) v[i] = in[lane * P + i];
  wave_reduce_scatter<P>(v);
  out[lane] = v[0];
  out[64 + lane] = (float)(scatter_owner<P>(lane) ? scatter_comp<P>(lane) : -1);
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct BinWs {
  uint32_t *tile_count, *tile_off, *ctrl, *cnt, *wcnt, *tile_order;
  unsigned long long *keys;
  int *tl, *br;  // only in the frame workspace
  uint32_t nchunks;
  size_t bytes;
};
static BinWs carve(void *base, uint32_t N, uint32_t D, uint32_t T, bool with_rects) {
  BinWs w{};
  size_t off = 0;
  char *p = (char *)base;
  auto take = [&](size_t bytes) { void *r = p ? p + off : nullptr; off += align_up(bytes, 256); return r; };
  w.nchunks = (N + kChunk - 1) / kChunk;
  if (w.nchunks == 0) w.nchunks = 1;
  w.tile_count = (uint32_t *)take(sizeof(uint32_t) * ((size_t)T + 4));
  w.ctrl = w.tile_count ? w.tile_count + T : nullptr;
  w.tile_off = (uint32_t *)take(sizeof(uint32_t) * ((size_t)T + 1));
  w.tile_order = (uint32_t *)take(sizeof(uint32_t) * (size_t)(T ? T : 1));
  w.cnt = (uint32_t *)take(sizeof(uint32_t) * (size_t)w.nchunks * T);
  w.wcnt = (uint32_t *)take(sizeof(uint32_t) * (size_t)w.nchunks * kPullWaves * T);
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
                        const BinWs &w, uint32_t *total_out, hipStream_t s, uint32_t *report = nullptr) {
  const uint32_t T = nth * ntw;
  if (cap > 0x7fffffffu) return GSGEN_EINVAL;  // start / end / the list positions are int32 (the reference's layout)
  const uint32_t ngroups = ((ntw + kGroup - 1) / kGroup) * ((nth + kGroup - 1) / kGroup);
  const dim3 gpull(ngroups, w.nchunks);
  const dim3 bpull(64 * kPullWaves);
  // per-tile counters in LDS (bin_push_body) where there are enough chunks to fill the chip; a lone view of 100 k Gaussians is
  // 49 workgroups -- there the pull kernels' 2 401 finish sooner (0.44 vs 0.38 ms per render with one render in flight)
  const bool push = T <= kPushMaxTiles && w.nchunks >= push_min_workgroups(false);
  if (N == 0) {
    if (hipError_t e = hipMemsetAsync(w.cnt, 0, sizeof(uint32_t) * (size_t)w.nchunks * T, s)) return (int)e;
  } else if (push) {
    hipLaunchKernelGGL((k_bin_push<false>), dim3(w.nchunks), dim3(kPushThreads), sizeof(uint32_t) * T, s, N, tl, br, depth, (int)ntw, (int)nth, T, w.cnt,
                       (const uint32_t *)nullptr, (const uint32_t *)nullptr, (unsigned long long *)nullptr);
  } else {
    hipLaunchKernelGGL((k_bin_pull<false>), gpull, bpull, 0, s, N, tl, br, depth, (int)ntw, (int)nth, T,
                       w.cnt, w.wcnt, (const uint32_t *)nullptr, (const uint32_t *)nullptr,
                       (unsigned long long *)nullptr);
  }
  hipLaunchKernelGGL(k_scan_chunks, dim3((T + kScanChunkTiles - 1) / kScanChunkTiles), dim3(1024), 0, s, T, w.nchunks,
                     w.cnt, w.tile_count);
  hipLaunchKernelGGL(k_scan_order_tiles, dim3(1), dim3(kScanThreads), 0, s, T, w.tile_count, w.tile_off, w.ctrl, cap, total_out,
                     w.tile_order, report);
  if (N && push)
    hipLaunchKernelGGL((k_bin_push<true>), dim3(w.nchunks), dim3(kPushThreads), sizeof(uint32_t) * T, s, N, tl, br, depth, (int)ntw, (int)nth, T, w.cnt,
                       (const uint32_t *)w.tile_off, (const uint32_t *)w.ctrl, w.keys);
  else if (N)
    hipLaunchKernelGGL((k_bin_pull<true>), gpull, bpull, 0, s, N, tl, br, depth, (int)ntw, (int)nth, T,
                       w.cnt, w.wcnt, w.tile_off, w.ctrl, w.keys);
  hipLaunchKernelGGL(k_sort_tiles, dim3(T), dim3(64 * kCoopWaves), 0, s, T, w.tile_off, w.ctrl, w.keys, ids, start, end,
                     (const uint32_t *)w.tile_order);
  return (int)hipGetLastError();
}

}  // namespace gs

using namespace gs;

extern "C" {

int gsgen_internal_frame_project(uint32_t N, const float *mean, const float *qvec, const float *svec,
                                 const float *cam, int w, int h, int ntw, float *mean2d, float *cov2d,
                                 float *depth, uint8_t *mask, int *tl, int *br, gsgen_stream_t stream);

int gsgen_internal_frame_project_views(uint32_t N, const float *mean, const float *qvec, const float *svec,
                                       int w, int h, int ntw, const GeoView *host_views, GeoView *dev_views,
                                       uint32_t B, float *zero_shared, size_t zero_shared_floats, gsgen_stream_t stream);

// used by legacy.hip: per-segment sort of (depth bits << 32 | id) keys, ids out (ctrl[1] must be 0)
int gsgen_internal_sort_segments(uint32_t T, const uint32_t *tile_off, const uint32_t *ctrl,
                                 unsigned long long *keys, int *ids, int *start, int *end, gsgen_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (T == 0) return 0;
  hipLaunchKernelGGL(k_sort_tiles, dim3(T), dim3(64 * kCoopWaves), 0, s, T, tile_off, ctrl, keys, ids, start, end,
                     (const uint32_t *)nullptr);
  return (int)hipGetLastError();
}

const char *gsgen_version(void) { return "gsgen_hip 0.1 (gfx950)"; }

int gsgen_selftest_reduce_scatter(uint32_t P, const float *in /*[64,P]*/, float *out /*[64]*/,
                                  gsgen_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  switch (P) {
    case 8: hipLaunchKernelGGL((k_selftest_reduce_scatter<8>), dim3(1), dim3(64), 0, s, in, out); break;
    case 16: hipLaunchKernelGGL((k_selftest_reduce_scatter<16>), dim3(1), dim3(64), 0, s, in, out); break;
    case 32: hipLaunchKernelGGL((k_selftest_reduce_scatter<32>), dim3(1), dim3(64), 0, s, in, out); break;
    case 64: hipLaunchKernelGGL((k_selftest_reduce_scatter<64>), dim3(1), dim3(64), 0, s, in, out); break;
    default: return GSGEN_EUNSUPPORTED;
  }
  return (int)hipGetLastError();
}

const char *gsgen_error_string(int code) {
  if (code == 0) return "success";
  if (code == GSGEN_EUNSUPPORTED) return "unsupported configuration (tile_size 1 .. 32 -- 16 only for the batched / segmented / fused entry points --, C in 1..4)";
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
  if (D > 0x7fffffffu) return GSGEN_EINVAL;  // int32 list positions (the reference's start / end layout)
  const BinWs w = carve(workspace, N, D, T, false);
  if (w.bytes > workspace_bytes) return GSGEN_EWORKSPACE;
  return bin_and_sort(N, D, n_tiles_h, n_tiles_w, aabb_topleft, aabb_bottomright, depth, gaussian_ids,
                      start, end, w, nullptr, (hipStream_t)stream);
}

size_t gsgen_frame_batch_workspace_bytes(uint32_t n_views) { return (size_t)n_views * sizeof(GeoView); }

int gsgen_frame_geometry_batch(uint32_t n_views, const gsgen_geometry_view *views, uint32_t N, const float *mean,
                               const float *qvec, const float *svec, uint32_t W, uint32_t H,
                               void *batch_workspace, gsgen_stream_t stream) {
  return gsgen_frame_geometry_batch_zero(n_views, views, N, mean, qvec, svec, W, H, nullptr, 0, batch_workspace, stream);
}

int gsgen_frame_geometry_batch_zero(uint32_t n_views, const gsgen_geometry_view *views, uint32_t N, const float *mean,
                                    const float *qvec, const float *svec, uint32_t W, uint32_t H, float *zero_shared,
                                    size_t zero_shared_floats, void *batch_workspace, gsgen_stream_t stream) {
  if (zero_shared_floats && (!zero_shared || (zero_shared_floats & 3u) || (reinterpret_cast<uintptr_t>(zero_shared) & 15u) ||
                             zero_shared_floats / 4 > 0xffffffffull))
    return GSGEN_EINVAL;
  const uint32_t ntw = (W + kTile - 1) / kTile, nth = (H + kTile - 1) / kTile;
  const uint32_t T = ntw * nth;
  if (T == 0 || n_views == 0) return 0;
  if (!views || !batch_workspace) return GSGEN_EINVAL;
  if (n_views > 65535) return GSGEN_EINVAL;  // gridDim.y / .z
  if (N && (!mean || !qvec || !svec)) return GSGEN_EINVAL;
  std::vector<GeoView> gv(n_views);
  uint32_t nchunks = 1;
  for (uint32_t b = 0; b < n_views; ++b) {
    const gsgen_geometry_view &v = views[b];
    if (!v.cam || !v.start || !v.end || !v.workspace || !v.total || !v.gaussian_ids) return GSGEN_EINVAL;
    if (N && (!v.mean2d || !v.cov2d || !v.depth || !v.mask)) return GSGEN_EINVAL;
    if (v.D_cap > 0x7fffffffu) return GSGEN_EINVAL;  // int32 list positions
    const BinWs w = carve(v.workspace, N, v.D_cap, T, true);
    if (w.bytes > v.workspace_bytes) return GSGEN_EWORKSPACE;
    nchunks = w.nchunks;
    GeoView &g = gv[b];
    g.cam = v.cam; g.mean2d = v.mean2d; g.cov2d = v.cov2d; g.depth = v.depth; g.mask = v.mask;
    g.tl = w.tl; g.br = w.br; g.cnt = w.cnt; g.wcnt = w.wcnt; g.tile_count = w.tile_count; g.tile_off = w.tile_off;
    g.ctrl = w.ctrl; g.tile_order = w.tile_order; g.keys = w.keys;
    g.ids = v.gaussian_ids; g.start = v.start; g.end = v.end; g.total = v.total; g.cap = v.D_cap; g.report = v.pair_report;
    g.z_mean2d = v.zero_grad_mean2d; g.z_cov2d = v.zero_grad_cov2d; g.z_chan6 = v.zero_grad_chan6;
    g.chol = v.chol; g.max_r = v.max_radii2d;
    if ((reinterpret_cast<uintptr_t>(g.z_mean2d) & 7u) || (reinterpret_cast<uintptr_t>(g.z_cov2d) & 15u) ||
        (reinterpret_cast<uintptr_t>(g.z_chan6) & 7u) || (reinterpret_cast<uintptr_t>(g.chol) & 15u))
      return GSGEN_EINVAL;
  }
  hipStream_t s = (hipStream_t)stream;
  GeoView *dv = reinterpret_cast<GeoView *>(batch_workspace);
  if (int e = gsgen_internal_frame_project_views(N, mean, qvec, svec, (int)W, (int)H, (int)ntw, gv.data(), dv,
                                                 n_views, zero_shared_floats ? zero_shared : nullptr, zero_shared_floats,
                                                 stream))
    return e;
  const uint32_t B = n_views;
  const uint32_t ngroups = ((ntw + kGroup - 1) / kGroup) * ((nth + kGroup - 1) / kGroup);
  const dim3 gpull(ngroups, nchunks, B), bpull(64 * kPullWaves);
  const bool push = T <= kPushMaxTiles && nchunks * B >= push_min_workgroups(true);  // (per-tile counters in LDS: k_bin_push_views)
  const dim3 gpush(nchunks, B), bpush(kPushThreads);
  if (N == 0) {
    for (uint32_t b = 0; b < B; ++b)
      if (hipError_t e = hipMemsetAsync(gv[b].cnt, 0, sizeof(uint32_t) * (size_t)nchunks * T, s)) return (int)e;
  } else if (push) {
    hipLaunchKernelGGL((k_bin_push_views<false>), gpush, bpush, sizeof(uint32_t) * T, s, N, (int)ntw, (int)nth, T, (const GeoView *)dv);
  } else {
    hipLaunchKernelGGL((k_bin_pull_views<false>), gpull, bpull, 0, s, N, (int)ntw, (int)nth, T, (const GeoView *)dv);
  }
  hipLaunchKernelGGL(k_scan_chunks_views, dim3((T + kScanChunkTiles - 1) / kScanChunkTiles, B), dim3(256), 0, s, T, nchunks, (const GeoView *)dv);
  hipLaunchKernelGGL(k_scan_order_tiles_views, dim3(1, B), dim3(kScanThreads), 0, s, T, (const GeoView *)dv);
  if (N && push)
    hipLaunchKernelGGL((k_bin_push_views<true>), gpush, bpush, sizeof(uint32_t) * T, s, N, (int)ntw, (int)nth, T, (const GeoView *)dv);
  else if (N)
    hipLaunchKernelGGL((k_bin_pull_views<true>), gpull, bpull, 0, s, N, (int)ntw, (int)nth, T, (const GeoView *)dv);
  hipLaunchKernelGGL(k_sort_tiles_views, dim3(T * B), dim3(64 * kCoopWaves), 0, s, T, B, (const GeoView *)dv);
  return (int)hipGetLastError();
}

void *gsgen_host_device_pointer(void *pinned_host) {
  void *dev = nullptr;
  if (pinned_host == nullptr || hipHostGetDevicePointer(&dev, pinned_host, 0) != hipSuccess) {
    (void)hipGetLastError();  // (not sticky: an unmapped block is an answer, not a fault)
    return nullptr;
  }
  return dev;
}

const uint32_t *gsgen_frame_tile_order(void *workspace, uint32_t N, uint32_t D_cap, uint32_t n_tiles) {
  return carve(workspace, N, D_cap, n_tiles, true).tile_order;
}

size_t gsgen_frame_workspace_bytes(uint32_t N, uint32_t D_cap, uint32_t n_tiles) {
  return carve(nullptr, N, D_cap, n_tiles, true).bytes;
}

int gsgen_frame_geometry(uint32_t N, const float *mean, const float *qvec, const float *svec,
                         const float *cam, uint32_t W, uint32_t H, uint32_t D_cap, float *mean2d,
                         float *cov2d, float *depth, uint8_t *mask, int *gaussian_ids, int *start,
                         int *end, uint32_t *total, void *workspace, size_t workspace_bytes,
                         gsgen_stream_t stream) {
  return gsgen_frame_geometry_report(N, mean, qvec, svec, cam, W, H, D_cap, mean2d, cov2d, depth, mask, gaussian_ids, start, end,
                                     total, nullptr, workspace, workspace_bytes, stream);
}

int gsgen_frame_geometry_report(uint32_t N, const float *mean, const float *qvec, const float *svec,
                                const float *cam, uint32_t W, uint32_t H, uint32_t D_cap, float *mean2d,
                                float *cov2d, float *depth, uint8_t *mask, int *gaussian_ids, int *start,
                                int *end, uint32_t *total, uint32_t *pair_report, void *workspace, size_t workspace_bytes,
                                gsgen_stream_t stream) {
  const uint32_t ntw = (W + kTile - 1) / kTile, nth = (H + kTile - 1) / kTile;
  const uint32_t T = ntw * nth;
  if (T == 0) return 0;
  if (!cam || !start || !end || !workspace || !total) return GSGEN_EINVAL;
  if (N && (!mean || !qvec || !svec || !mean2d || !cov2d || !depth || !mask)) return GSGEN_EINVAL;
  if (D_cap > 0x7fffffffu) return GSGEN_EINVAL;
  const BinWs w = carve(workspace, N, D_cap, T, true);
  if (w.bytes > workspace_bytes) return GSGEN_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  if (int e = gsgen_internal_frame_project(N, mean, qvec, svec, cam, (int)W, (int)H, (int)ntw, mean2d,
                                           cov2d, depth, mask, w.tl, w.br, stream))
    return e;
  return bin_and_sort(N, D_cap, nth, ntw, w.tl, w.br, depth, gaussian_ids, start, end, w, total, s, pair_report);
}

}  // extern "C"
