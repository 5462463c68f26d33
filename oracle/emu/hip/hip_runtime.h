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
inline const char *hipGetErrorString(hipError_t) { return "emulated"; }
inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return 0; }

struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
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

#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))

template <typename T, typename U> inline T atomicAdd(T *p, U v) { return simt::atomic_add(p, v); }
template <typename T, typename U> inline T atomicMax(T *p, U v) { return simt::atomic_max(p, v); }
