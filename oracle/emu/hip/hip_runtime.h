// Stand-in for <hip/hip_runtime.h> when this repo's .hip kernels are compiled with g++ on
// top of oracle/emu/simt_core.h (TEST INFRASTRUCTURE: CPU debugging of the kernel logic in a
// container without a GPU; never part of the product build, which uses the real header).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>

#include "../simt_core.h"

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __shared__ static thread_local

using simt::dim3;
#define threadIdx (simt::tls_threadIdx())
#define blockIdx (simt::tls_blockIdx())
#define blockDim (simt::tls_blockDim())
#define gridDim (simt::tls_gridDim())

using std::max;
using std::min;

typedef void *hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
inline hipError_t hipGetLastError() { return 0; }
inline hipError_t hipHostGetDevicePointer(void **dev, void *host, unsigned) { *dev = host; return 0; }  // (one address space)
inline const char *hipGetErrorString(hipError_t) { return "emulated"; }
inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return 0; }

struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline int2 make_int2(int x, int y) { return int2{x, y}; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  simt::launch((grid), (block), [&]() { kernel(__VA_ARGS__); })

inline void __syncthreads() { simt::syncthreads(); }
inline int __syncthreads_or(int p) { return simt::syncthreads_or(p); }
inline int __syncthreads_count(int p) { return simt::syncthreads_count(p); }
inline unsigned long long __ballot(int p) { return simt::ballot(p); }
template <typename T> inline T __shfl_xor(T v, int m, int = 64) { return simt::shfl_xor(v, m); }
template <typename T> inline T __shfl(T v, int l, int = 64) { return simt::shfl_idx(v, l); }
template <typename T> inline T __shfl_down(T v, int d, int = 64) { return simt::shfl_down(v, d); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }

inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
inline int __float_as_int(float f) { int u; std::memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
inline float __int_as_float(int u) { float f; std::memcpy(&f, &u, 4); return f; }

// gfx950 cross-lane builtins used by common.hpp (semantics per the CDNA4 ISA: DPP row = 16
// lanes, bank = 4 lanes; v_permlane32_swap exchanges vdst's upper 32 lanes with vsrc's lower
// 32; v_permlane16_swap exchanges vdst's odd rows with vsrc's even rows).
struct emu_u2 { unsigned v[2]; unsigned operator[](int i) const { return v[i]; } };
inline emu_u2 emu_permlane_swap(unsigned d, unsigned s, int width) {
  const int lane = simt::lane_of_current();
  emu_u2 r;
  if (width == 32) {
    // new vdst[l>=32] = vsrc[l-32]; new vsrc[l<32] = vdst[l+32]
    const unsigned s_from = simt::shfl_idx(s, lane - 32 >= 0 ? lane - 32 : lane);
    const unsigned d_from = simt::shfl_idx(d, lane + 32 < 64 ? lane + 32 : lane);
    r.v[0] = (lane >= 32) ? s_from : d;
    r.v[1] = (lane < 32) ? d_from : s;
  } else {
    const int row = lane >> 4;
    const unsigned s_from = simt::shfl_idx(s, (row & 1) ? lane - 16 : lane);  // odd row of vdst <- even row of vsrc
    const unsigned d_from = simt::shfl_idx(d, (row & 1) ? lane : lane + 16);  // even row of vsrc <- odd row of vdst
    r.v[0] = (row & 1) ? s_from : d;
    r.v[1] = (row & 1) ? s : d_from;
  }
  return r;
}
#define __builtin_amdgcn_permlane32_swap(d, s, fi, bc) emu_permlane_swap((d), (s), 32)
#define __builtin_amdgcn_permlane16_swap(d, s, fi, bc) emu_permlane_swap((d), (s), 16)
inline int emu_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  const int lane = simt::lane_of_current();
  const int row = lane >> 4, in_row = lane & 15, bank = in_row >> 2;
  int from = -1;
  if (ctrl < 0x100) from = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
  else if (ctrl > 0x100 && ctrl <= 0x10F) { int n = ctrl - 0x100; from = (in_row + n < 16) ? lane + n : -1; }
  else if (ctrl > 0x110 && ctrl <= 0x11F) { int n = ctrl - 0x110; from = (in_row - n >= 0) ? lane - n : -1; }
  else if (ctrl > 0x120 && ctrl <= 0x12F) { int n = ctrl - 0x120; from = (row << 4) | ((in_row - n) & 15); }
  else if (ctrl == 0x138) from = lane - 1;  // wave_shr:1 (lane 0 has no source)
  else if (ctrl == 0x142) from = (row >= 1) ? ((row - 1) << 4) | 15 : -1;  // row_bcast:15
  else if (ctrl == 0x143) from = (row >= 2) ? 31 : -1;                       // row_bcast:31
  else if (ctrl == 0x140) from = (row << 4) | (15 - in_row);
  else if (ctrl == 0x141) from = (lane & ~7) | (7 - (lane & 7));
  const int got = simt::shfl_idx(src, from >= 0 ? from : lane);
  const bool enabled = ((row_mask >> row) & 1) && ((bank_mask >> bank) & 1);
  if (!enabled) return old;
  if (from < 0) return bound_ctrl ? 0 : old;
  return got;
}
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) emu_update_dpp((old), (src), (ctrl), (rm), (bm), (bc))

#define __builtin_amdgcn_mov_dpp(src, ctrl, rm, bm, bc) emu_update_dpp((src), (src), (ctrl), (rm), (bm), (bc))
#define __builtin_amdgcn_readfirstlane(v) (v)
#define __builtin_amdgcn_sched_barrier(m) ((void)0)
#define __builtin_amdgcn_readlane(v, l) simt::shfl_idx((int)(v), (l))
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))

template <typename T, typename U> inline T atomicAdd(T *p, U v) { return simt::atomic_add(p, v); }
template <typename T, typename U> inline T atomicMax(T *p, U v) { return simt::atomic_max(p, v); }
