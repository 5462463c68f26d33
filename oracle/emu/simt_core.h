// simt_core.h -- a small SIMT execution emulator for the CPU (TEST INFRASTRUCTURE).
//
// Runs a GPU kernel written in the CUDA/HIP single-source style on the host: every GPU
// thread of a workgroup is a ucontext fiber, fibers of one workgroup are scheduled round
// robin on ONE OS thread and switch only at __syncthreads() / wave collectives, workgroups
// run one after another.  That is enough to execute, unmodified,
//   (a) the reference's CUDA kernels from /root/reference/gs/src/include (through
//       oracle/emu/cuda/*.h) -> oracle/_ref/libgs_ref.so, the "real reference" the C oracle
//       is pinned against, and
//   (b) this repo's own .hip kernels (through oracle/emu/hip/hip_runtime.h) for debugging
//       in a container without a GPU (tests/test_emu_*.py).  The product never loads either.
//
// Semantics: 64-wide waves (wave = threads [64k, 64k+63] of the linearised workgroup),
// collectives assume every not-yet-exited lane of the wave takes part (wave-uniform control
// flow around collectives), exited threads are dropped from barrier counts (Volta+/CDNA
// behaviour).  Not thread-safe across OS threads by design.
#pragma once
#include <ucontext.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

namespace simt {

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_ { unsigned x, y, z; };

constexpr int kWave = 64;
constexpr size_t kStack = 128 * 1024;

struct WaveState {
  int live = 0, count = 0;
  unsigned gen = 0;
  uint64_t buf[2][kWave];
  uint64_t valid[2] = {0, 0};
};

struct Fiber {
  ucontext_t ctx;
  bool done = false;
  uint3_ tidx;
  unsigned linear = 0;
};

struct Block {
  std::vector<Fiber> fibers;
  std::vector<WaveState> waves;
  int live = 0;
  int bar_count = 0;
  unsigned bar_gen = 0;
  int bar_or[2] = {0, 0};
  int bar_cnt[2] = {0, 0};
  ucontext_t sched;
  int current = -1;
  const std::function<void()> *body = nullptr;
};

inline Block *&cur_block() { static thread_local Block *b = nullptr; return b; }
inline uint3_ &tls_threadIdx() { static thread_local uint3_ v{0, 0, 0}; return v; }
inline uint3_ &tls_blockIdx() { static thread_local uint3_ v{0, 0, 0}; return v; }
inline dim3 &tls_blockDim() { static thread_local dim3 v; return v; }
inline dim3 &tls_gridDim() { static thread_local dim3 v; return v; }

inline std::vector<char *> &stack_pool() { static thread_local std::vector<char *> p; return p; }

inline void yield_() {
  Block *b = cur_block();
  Fiber &f = b->fibers[b->current];
  swapcontext(&f.ctx, &b->sched);
}

inline void release_block_barrier(Block *b) {
  b->bar_count = 0;
  unsigned g = b->bar_gen;
  b->bar_or[(g + 1) & 1] = 0;
  b->bar_cnt[(g + 1) & 1] = 0;
  b->bar_gen = g + 1;
}
inline void release_wave(WaveState &w) {
  w.count = 0;
  w.valid[(w.gen + 1) & 1] = 0;
  w.gen = w.gen + 1;
}

inline int sync_impl(int pred, bool want_count) {
  Block *b = cur_block();
  unsigned g = b->bar_gen;
  int slot = g & 1;
  if (pred) { b->bar_or[slot] = 1; b->bar_cnt[slot] += 1; }
  b->bar_count++;
  if (b->bar_count == b->live) release_block_barrier(b);
  else while (b->bar_gen == g) yield_();
  return want_count ? b->bar_cnt[slot] : b->bar_or[slot];
}
inline void syncthreads() { (void)sync_impl(0, false); }
inline int syncthreads_or(int p) { return sync_impl(p, false); }
inline int syncthreads_count(int p) { return sync_impl(p, true); }

// wave-wide exchange: every live lane deposits a 64-bit value; returns the slot to read from
inline const uint64_t *wave_exchange(uint64_t v, uint64_t *valid_mask) {
  Block *b = cur_block();
  unsigned lin = b->fibers[b->current].linear;
  WaveState &w = b->waves[lin / kWave];
  int lane = lin % kWave;
  unsigned g = w.gen;
  int slot = g & 1;
  w.buf[slot][lane] = v;
  w.valid[slot] |= (1ull << lane);
  w.count++;
  if (w.count == w.live) release_wave(w);
  else while (w.gen == g) yield_();
  if (valid_mask) *valid_mask = w.valid[slot];
  return w.buf[slot];
}
inline int lane_of_current() {
  Block *b = cur_block();
  return (int)(b->fibers[b->current].linear % kWave);
}

inline uint64_t ballot(int pred) {
  uint64_t valid;
  const uint64_t *buf = wave_exchange(pred ? 1u : 0u, &valid);
  uint64_t m = 0;
  for (int l = 0; l < kWave; ++l)
    if (((valid >> l) & 1ull) && buf[l]) m |= (1ull << l);
  return m;
}
template <typename T>
inline T shfl_idx(T v, int src_lane) {
  static_assert(sizeof(T) <= 8, "shfl of <= 8 byte types only");
  uint64_t raw = 0;
  std::memcpy(&raw, &v, sizeof(T));
  uint64_t valid;
  const uint64_t *buf = wave_exchange(raw, &valid);
  src_lane &= (kWave - 1);
  uint64_t r = ((valid >> src_lane) & 1ull) ? buf[src_lane] : raw;  // inactive source: own value
  T out;
  std::memcpy(&out, &r, sizeof(T));
  return out;
}
template <typename T>
inline T shfl_xor(T v, int mask) { return shfl_idx(v, lane_of_current() ^ mask); }
template <typename T>
inline T shfl_down(T v, int d) { int l = lane_of_current(); return shfl_idx(v, l + d < kWave ? l + d : l); }
template <typename T>
inline T shfl_up(T v, int d) { int l = lane_of_current(); return shfl_idx(v, l - d >= 0 ? l - d : l); }

inline void fiber_entry() {
  Block *b = cur_block();
  (*b->body)();
  Fiber &f = b->fibers[b->current];
  f.done = true;
  b->live--;
  WaveState &w = b->waves[f.linear / kWave];
  w.live--;
  // an exiting thread may complete a barrier / collective the others are waiting in
  if (b->live > 0 && b->bar_count == b->live && b->bar_count > 0) release_block_barrier(b);
  if (w.live > 0 && w.count == w.live && w.count > 0) release_wave(w);
  swapcontext(&f.ctx, &b->sched);
}

inline void run_block(Block &b, const std::function<void()> &body, dim3 block, uint3_ bidx) {
  const unsigned nt = block.x * block.y * block.z;
  b.fibers.assign(nt, Fiber());
  b.waves.assign((nt + kWave - 1) / kWave, WaveState());
  b.live = (int)nt;
  b.bar_count = 0; b.bar_gen = 0;
  b.bar_or[0] = b.bar_or[1] = 0; b.bar_cnt[0] = b.bar_cnt[1] = 0;
  b.body = &body;
  auto &pool = stack_pool();
  while (pool.size() < nt) pool.push_back((char *)std::malloc(kStack));
  cur_block() = &b;
  tls_blockIdx() = bidx;
  for (unsigned i = 0; i < nt; ++i) {
    Fiber &f = b.fibers[i];
    f.linear = i;
    f.tidx.x = i % block.x;
    f.tidx.y = (i / block.x) % block.y;
    f.tidx.z = i / (block.x * block.y);
    b.waves[i / kWave].live++;
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = pool[i];
    f.ctx.uc_stack.ss_size = kStack;
    f.ctx.uc_link = nullptr;
    makecontext(&f.ctx, (void (*)())fiber_entry, 0);
  }
  int remaining = (int)nt;
  long idle_passes = 0;
  while (remaining > 0) {
    int before_live = b.live;
    unsigned before_gen = b.bar_gen;
    unsigned wsum = 0;
    for (auto &w : b.waves) wsum += w.gen;
    for (unsigned i = 0; i < nt; ++i) {
      Fiber &f = b.fibers[i];
      if (f.done) continue;
      b.current = (int)i;
      tls_threadIdx() = f.tidx;
      swapcontext(&b.sched, &f.ctx);
      if (f.done) --remaining;
    }
    unsigned wsum2 = 0;
    for (auto &w : b.waves) wsum2 += w.gen;
    if (b.live == before_live && b.bar_gen == before_gen && wsum2 == wsum) {
      if (++idle_passes > 4) {
        std::fprintf(stderr, "simt: deadlock in block (%u,%u,%u): divergent barrier/collective?\n",
                     bidx.x, bidx.y, bidx.z);
        std::abort();
      }
    } else idle_passes = 0;
  }
  cur_block() = nullptr;
}

inline void launch(dim3 grid, dim3 block, const std::function<void()> &body) {
  static thread_local Block b;
  tls_blockDim() = block;
  tls_gridDim() = grid;
  for (unsigned z = 0; z < grid.z; ++z)
    for (unsigned y = 0; y < grid.y; ++y)
      for (unsigned x = 0; x < grid.x; ++x) run_block(b, body, block, uint3_{x, y, z});
}

// ---- atomics (single OS thread: plain read-modify-write) ------------------------------------
template <typename T, typename U>
inline T atomic_add(T *p, U v) { T old = *p; *p = (T)(old + (T)v); return old; }
template <typename T, typename U>
inline T atomic_max(T *p, U v) { T old = *p; if ((T)v > old) *p = (T)v; return old; }
template <typename T, typename U>
inline T atomic_min(T *p, U v) { T old = *p; if ((T)v < old) *p = (T)v; return old; }

}  // namespace simt
