// Stand-in for the CUDA runtime headers, on top of oracle/emu/simt_core.h, so that the
// REFERENCE's own CUDA sources (/root/reference/gs/src/include/*.h) compile with g++ and
// execute on the CPU -- see oracle/ref_build.py.  TEST INFRASTRUCTURE; all code here is
// this repo's, nothing is taken from the CUDA toolkit.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "../simt_core.h"

#ifndef __CUDACC__
#define __CUDACC__ 1  // makes the reference's helper_math.h skip its host re-definitions of fminf & co
#endif
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define __restrict__ __restrict

using simt::dim3;
#define threadIdx (simt::tls_threadIdx())
#define blockIdx (simt::tls_blockIdx())
#define blockDim (simt::tls_blockDim())
#define gridDim (simt::tls_gridDim())

#define SIMT_VEC2(T, N) struct N##2 { T x, y; }; static inline N##2 make_##N##2(T x, T y) { return N##2{x, y}; }
#define SIMT_VEC3(T, N) struct N##3 { T x, y, z; }; static inline N##3 make_##N##3(T x, T y, T z) { return N##3{x, y, z}; }
#define SIMT_VEC4(T, N) struct N##4 { T x, y, z, w; }; static inline N##4 make_##N##4(T x, T y, T z, T w) { return N##4{x, y, z, w}; }
SIMT_VEC2(float, float) SIMT_VEC3(float, float) SIMT_VEC4(float, float)
SIMT_VEC2(int, int) SIMT_VEC3(int, int) SIMT_VEC4(int, int)
SIMT_VEC2(unsigned int, uint) SIMT_VEC3(unsigned int, uint) SIMT_VEC4(unsigned int, uint)
SIMT_VEC2(double, double)

static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
#define __expf(x) expf(x)  /* CUDA fast-math intrinsic; glibc declares but does not export the name */
static inline float __fdividef(float a, float b) { return a / b; }
// CUDA's overloaded min/max accept mixed integer types
template <typename A, typename B> static inline auto min(A a, B b) -> decltype(a + b) { using R = decltype(a + b); return (R)a < (R)b ? (R)a : (R)b; }
template <typename A, typename B> static inline auto max(A a, B b) -> decltype(a + b) { using R = decltype(a + b); return (R)a > (R)b ? (R)a : (R)b; }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }

static inline void __syncthreads() { simt::syncthreads(); }
#define FULL_MASK_SIMT 0xffffffffu
// 32-wide warp collectives of CUDA on the 64-wide emulated wave: only the (unused) experimental
// kernels of vol_render_sh.h reference them; semantics restricted to the low/high 32 lanes.
static inline unsigned __activemask() { return 0xffffffffu; }
template <typename T> static inline T __shfl_xor_sync(unsigned, T v, int m, int = 32) { return simt::shfl_xor(v, m); }
template <typename T> static inline T __shfl_down_sync(unsigned, T v, int d, int = 32) { return simt::shfl_down(v, d); }
template <typename T> static inline T __shfl_sync(unsigned, T v, int l, int = 32) { return simt::shfl_idx(v, l); }
static inline unsigned __ballot_sync(unsigned, int p) { return (unsigned)simt::ballot(p); }

template <typename T, typename U> static inline T atomicAdd(T *p, U v) { return simt::atomic_add(p, v); }
template <typename T, typename U> static inline T atomicMax(T *p, U v) { return simt::atomic_max(p, v); }
template <typename T, typename U> static inline T atomicMin(T *p, U v) { return simt::atomic_min(p, v); }

typedef int cudaError_t;
typedef void *cudaStream_t;
typedef void *cudaEvent_t;
#define cudaSuccess 0
enum cudaMemcpyKind { cudaMemcpyHostToDevice, cudaMemcpyDeviceToHost, cudaMemcpyDeviceToDevice, cudaMemcpyHostToHost };
static inline cudaError_t cudaGetLastError() { return 0; }
static inline const char *cudaGetErrorString(cudaError_t) { return "emulated"; }
static inline cudaError_t cudaMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return 0; }
static inline cudaError_t cudaFree(void *p) { free(p); return 0; }
static inline cudaError_t cudaMemset(void *p, int v, size_t n) { memset(p, v, n); return 0; }
static inline cudaError_t cudaMemcpy(void *d, const void *s, size_t n, cudaMemcpyKind) { memcpy(d, s, n); return 0; }
static inline cudaError_t cudaDeviceSynchronize() { return 0; }
static inline cudaError_t cudaEventCreate(cudaEvent_t *e) { *e = nullptr; return 0; }
static inline cudaError_t cudaEventDestroy(cudaEvent_t) { return 0; }
static inline cudaError_t cudaEventRecord(cudaEvent_t, cudaStream_t = nullptr) { return 0; }
static inline cudaError_t cudaEventSynchronize(cudaEvent_t) { return 0; }
static inline cudaError_t cudaEventElapsedTime(float *ms, cudaEvent_t, cudaEvent_t) { *ms = 0.f; return 0; }

// dynamic shared memory (`extern __shared__ T name[]`, rewritten by oracle/ref_build.py): the fibers
// of a block run on one OS thread, so a thread_local buffer is per block, like the static ones
inline void *simt_dyn_smem() {
  alignas(16) static thread_local unsigned char buf[64 * 1024];
  return buf;
}

// kernel<<<grid, block[, shmem[, stream]]>>>(args...) is rewritten by oracle/ref_build.py into
// SIMT_LAUNCH((kernel), grid, block[, ...])(args...)
namespace simt {
template <typename K>
struct Launcher {
  K k; dim3 g, b;
  template <typename... A> void operator()(A... a) const {
    K kk = k;
    launch(g, b, [&]() { kk(a...); });
  }
};
template <typename K> Launcher<K> make_launcher(K k, dim3 g, dim3 b, size_t = 0, void * = nullptr) { return Launcher<K>{k, g, b}; }
}  // namespace simt
#define SIMT_LAUNCH(k, ...) simt::make_launcher(k, __VA_ARGS__)
