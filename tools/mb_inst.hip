// per-instruction issue cost on gfx950 (cycles per wave64 instruction, independent ops, one wave)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int OP>
__global__ void __launch_bounds__(64) k(float *out, int iters, long long *cyc) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b0 = 1.0001f, b1 = 0.5f;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (OP == 0) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));) }
    if (OP == 1) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&b0), "v"(*(double*)&b0));) }
    if (OP == 2) { REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (OP == 3) { REP8(asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (OP == 4) { REP8(asm volatile("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (OP == 5) { REP8(asm volatile("s_nop 1\n v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %4, %5, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %6, %7, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_f32_dpp %1, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (OP == 6) { REP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %1, %0, %1, vcc\n v_cndmask_b32 %3, %2, %3, vcc\n v_cndmask_b32 %5, %4, %5, vcc\n v_cndmask_b32 %7, %6, %7, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");) }
    if (OP == 7) { REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (OP == 8) { REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));) }
    if (OP == 9) { REP8(asm volatile("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"((int)((threadIdx.x ^ 32) * 4)));) }
  }
  long long t1 = clock64();
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float *out; long long *cyc; hipMalloc(&out, 64 * 8192 * 4); hipMalloc(&cyc, 8);
  const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_permlane32_swap", "v_permlane16_swap", "v_add_f32_dpp row_shr (+nop)", "v_cndmask_b32", "v_rcp_f32", "v_mul/add_f32", "ds_bpermute_b32"};
  const int iters = 200;
  for (int blocks : {1, 4096}) for (int op = 0; op < 10; ++op) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto L = [&]() { switch (op) {
      case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc); break; case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc); break;
      case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc); break; case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc); break;
      case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc); break; case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc); break;
      case 6: hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc); break; case 7: hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc); break;
      case 8: hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc); break; default: hipLaunchKernelGGL(k<9>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc); break; } };
    L(); hipDeviceSynchronize(); hipEventRecord(e0); L(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double n = (double)iters * 64;  // instructions per wave
    // SIMD-cycles per instruction when 4 waves share a SIMD (4096 blocks = 16 waves/CU)
    printf("blocks=%4d %-30s %6.2f clk/inst (wave0 latency-view)  %7.3f ms  -> %5.2f SIMD-clk/inst @2.1GHz\n", blocks, names[op], c / n, ms, blocks == 1 ? 0.0 : ms * 1e-3 * 2.1e9 / (n * blocks / 1024.0));
  }
}
