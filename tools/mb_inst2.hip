#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int OP>
__global__ void __launch_bounds__(64) k(float *out, int iters, long long *cyc) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b0 = 1.0001f, b1 = 31.5f;
  unsigned long long msk = (threadIdx.x & 1) ? ~0ull : 0x5555555555555555ull;
  msk = __builtin_amdgcn_readfirstlane((int)msk) | ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(msk>>32))<<32);
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (OP == 0) { REP8(asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %1, %0, %1, vcc\n v_cndmask_b32 %3, %2, %3, vcc\n v_cndmask_b32 %5, %4, %5, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b1) : "vcc");) }
    if (OP == 1) { REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %1, %8\n v_cndmask_b32_e64 %2, %2, %3, %8\n v_cndmask_b32_e64 %4, %4, %5, %8\n v_cndmask_b32_e64 %6, %6, %7, %8\n v_cndmask_b32_e64 %1, %0, %1, %8\n v_cndmask_b32_e64 %3, %2, %3, %8\n v_cndmask_b32_e64 %5, %4, %5, %8\n v_cndmask_b32_e64 %7, %6, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(msk));) }
    if (OP == 2) { REP8(asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_gt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_gt_f32 vcc, %4, %8\n v_cndmask_b32 %4, %4, %5, vcc\n v_cmp_gt_f32 vcc, %6, %8\n v_cndmask_b32 %6, %6, %7, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b1) : "vcc");) }
    if (OP == 3) { REP8(asm volatile("v_max_f32 %0, %0, %1\n v_max_f32 %2, %2, %3\n v_max_f32 %4, %4, %5\n v_max_f32 %6, %6, %7\n v_min_f32 %1, %0, %1\n v_min_f32 %3, %2, %3\n v_min_f32 %5, %4, %5\n v_min_f32 %7, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (OP == 4) { REP8(asm volatile("v_fmac_f32 %0, %1, %8\n v_fmac_f32 %2, %3, %8\n v_fmac_f32 %4, %5, %8\n v_fmac_f32 %6, %7, %8\n v_fmac_f32 %1, %0, %8\n v_fmac_f32 %3, %2, %8\n v_fmac_f32 %5, %4, %8\n v_fmac_f32 %7, %6, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));) }
    if (OP == 5) { REP8(asm volatile("v_readlane_b32 s20, %0, 5\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 5\n v_readlane_b32 s23, %3, 5\n v_readlane_b32 s24, %4, 5\n v_readlane_b32 s25, %5, 5\n v_readlane_b32 s26, %6, 5\n v_readlane_b32 s27, %7, 5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "s20","s21","s22","s23","s24","s25","s26","s27");) }
    if (OP == 6) { REP8(asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mov_b32_dpp %1, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_mov_b32_dpp %3, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_mov_b32_dpp %5, %4 row_bcast:31 row_mask:0xc bank_mask:0xf\n v_mov_b32_dpp %7, %6 row_bcast:31 row_mask:0xc bank_mask:0xf\n s_nop 1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (OP == 7) { REP8(asm volatile("v_fma_f32 %0, %1, s20, %0\n v_fma_f32 %2, %3, s21, %2\n v_fma_f32 %4, %5, s22, %4\n v_fma_f32 %6, %7, s23, %6\n v_fma_f32 %1, %0, s24, %1\n v_fma_f32 %3, %2, s25, %3\n v_fma_f32 %5, %4, s26, %5\n v_fma_f32 %7, %6, s27, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
  }
  long long t1 = clock64();
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float *out; long long *cyc; hipMalloc(&out, 64 * 8192 * 4); hipMalloc(&cyc, 8);
  const char *names[] = {"1 v_cmp + 7 v_cndmask(vcc)", "v_cndmask_e64 sgpr-pair mask", "v_cmp+v_cndmask pairs", "v_max/min_f32", "v_fmac_f32 (vgpr)", "v_readlane_b32", "v_mov_b32_dpp (shr/bcast)", "v_fma_f32 sgpr operand"};
  const int iters = 200;
  for (int blocks : {1, 4096}) for (int op = 0; op < 8; ++op) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto L = [&]() { switch (op) {
      case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc); break; case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc); break;
      case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc); break; case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc); break;
      case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc); break; case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc); break;
      case 6: hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc); break; default: hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc); break; } };
    L(); hipDeviceSynchronize(); hipEventRecord(e0); L(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double n = (double)iters * 64;
    printf("blocks=%4d %-30s %6.2f clk/inst (1 wave)  %7.3f ms  -> %5.2f SIMD-clk/inst @2.1GHz\n", blocks, names[op], c / n, ms, blocks == 1 ? 0.0 : ms * 1e-3 * 2.1e9 / (n * blocks / 1024.0));
  }
}
