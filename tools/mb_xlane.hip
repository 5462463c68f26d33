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
  const char *names[] = {"swap+dpp reduce_scatter<64>", "bpermute reduce_scatter<64>", "swap levels only (48 xchg)", "dpp levels only (15 xchg)", "63 plain adds"};
  for (int blocks : {1, 1024, 4096}) {
    for (int var = 0; var < 5; ++var) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      auto launch = [&](int v) {
        switch (v) {
          case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, out, in, iters, cyc); break;
          case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, out, in, iters, cyc); break;
          case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, out, in, iters, cyc); break;
          case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, out, in, iters, cyc); break;
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
