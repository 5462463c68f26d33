// microbenchmark: cost of the wave64 reduce-scatter variants (cycles per call, one wave / many waves)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../gsgen_amd/csrc/common.hpp"
using namespace gs;

template <int P>
__device__ __forceinline__ void rs_bperm(float (&v)[P]) {
  const int lane = lane_id();
#pragma unroll
  for (int h = P / 2; h >= 1; h >>= 1) {
    const bool up = (lane & h) != 0;
#pragma unroll
    for (int i = 0; i < h; ++i) {
      const float lo = v[i], hi = v[i + h];
      v[i] = (up ? hi : lo) + __shfl_xor(up ? lo : hi, h, 64);
    }
  }
}

// in-row levels first with fused, bank-masked v_add_f32_dpp (2 instructions per exchange), then
// the two cross-row levels on the 4 surviving components
template <int H>
__device__ __forceinline__ float xchg_add_asm(float lo, float hi) {
  float d;
  if constexpr (H == 8) {
    asm volatile("s_nop 1\n v_add_f32_dpp %0, %1, %1 row_shl:8 row_mask:0xf bank_mask:0x3\n v_add_f32_dpp %0, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xc" : "=&v"(d) : "v"(lo), "v"(hi));
  } else if constexpr (H == 4) {
    asm volatile("s_nop 1\n v_add_f32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n v_add_f32_dpp %0, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa" : "=&v"(d) : "v"(lo), "v"(hi));
  } else if constexpr (H == 2) {
    float a, b;
    asm volatile("s_nop 1\n v_add_f32_dpp %0, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=&v"(a), "=&v"(b) : "v"(lo), "v"(hi));
    d = (lane_id() & 2) ? b : a;
  } else {
    float a, b;
    asm volatile("s_nop 1\n v_add_f32_dpp %0, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=&v"(a), "=&v"(b) : "v"(lo), "v"(hi));
    d = (lane_id() & 1) ? b : a;
  }
  return d;
}
__device__ __forceinline__ void rs_rowfirst(float (&v)[64]) {
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = xchg_add_asm<8>(v[i], v[i + 32]);
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = xchg_add_asm<4>(v[i], v[i + 16]);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = xchg_add_asm<2>(v[i], v[i + 8]);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = xchg_add_asm<1>(v[i], v[i + 4]);
  v[0] = xchg_add<16>(v[0], v[2]);
  v[1] = xchg_add<16>(v[1], v[3]);
  v[0] = xchg_add<32>(v[0], v[1]);
}

template <int VAR>
__global__ void __launch_bounds__(64) k(float *out, const float *in, int iters, long long *cyc) {
  float v[64];
  for (int i = 0; i < 64; ++i) v[i] = in[(threadIdx.x * 64 + i) & 4095];
  long long t0 = clock64();
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    float w[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) w[i] = v[i] + acc;
    if (VAR == 0) wave_reduce_scatter<64>(w);
    else if (VAR == 1) rs_bperm<64>(w);
    else if (VAR == 5) rs_rowfirst(w);
    else if (VAR == 2) { // only the two swap levels
#pragma unroll
      for (int i = 0; i < 32; ++i) w[i] = xchg_add<32>(w[i], w[i + 32]);
#pragma unroll
      for (int i = 0; i < 16; ++i) w[i] = xchg_add<16>(w[i], w[i + 16]);
    } else if (VAR == 3) { // only dpp levels on 16 comps
#pragma unroll
      for (int i = 0; i < 8; ++i) w[i] = xchg_add<8>(w[i], w[i + 8]);
#pragma unroll
      for (int i = 0; i < 4; ++i) w[i] = xchg_add<4>(w[i], w[i + 4]);
#pragma unroll
      for (int i = 0; i < 2; ++i) w[i] = xchg_add<2>(w[i], w[i + 2]);
      w[0] = xchg_add<1>(w[0], w[1]);
    } else if (VAR == 4) { // 63 plain adds baseline
#pragma unroll
      for (int i = 1; i < 64; ++i) w[0] += w[i];
    }
    acc += w[0] * 1e-9f;
  }
  long long t1 = clock64();
  out[blockIdx.x * 64 + threadIdx.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
  float *in, *out; long long *cyc;
  hipMalloc(&in, 4096 * 4); hipMalloc(&out, 64 * 4096 * 4); hipMalloc(&cyc, 8);
  std::vector<float> h(4096, 1.0f);
  hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  const int iters = 2000;
  const char *names[] = {"swap+dpp reduce_scatter<64>", "bpermute reduce_scatter<64>", "swap levels only (48 xchg)", "dpp levels only (15 xchg)", "63 plain adds", "row-first fused-dpp asm"};
  for (int blocks : {1, 1024, 4096}) {
    for (int var : {0, 4, 5}) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      auto launch = [&](int v) {
        switch (v) {
          case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, out, in, iters, cyc); break;
          case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, out, in, iters, cyc); break;
          case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, out, in, iters, cyc); break;
          case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, out, in, iters, cyc); break;
          case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(64), 0, 0, out, in, iters, cyc); break;
          default: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(64), 0, 0, out, in, iters, cyc); break;
        }
      };
      launch(var); hipDeviceSynchronize();
      hipEventRecord(e0); launch(var); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      printf("blocks=%5d  %-34s  %8.1f clk/iter (wave0)   %8.3f ms total  -> %.2f ns/iter/wave-slot\n", blocks, names[var], (double)c / iters, ms, ms * 1e6 / iters / blocks);
    }
  }
  return 0;
}
