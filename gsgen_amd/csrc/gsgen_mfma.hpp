// gsgen_mfma.hpp -- the matrix-core primitives the compositing backward uses (gfx950).
//
// Included as <gsgen_mfma.hpp>: the CPU SIMT emulator under oracle/emu (test infrastructure)
// puts a same-named header with scalar restatements of these three functions first on its
// include path, exactly as it does for <hip/hip_runtime.h>.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gs {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // one 128-bit MFMA operand fragment (8 bf16)

__device__ __forceinline__ f32x4 f32x4_zero() { return f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }

// two floats -> two bf16, round to nearest even (v_cvt_pk_bf16_f32): a in bits 0..15
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const bf16x2 v = __builtin_convertvector(f32x2{a, b}, bf16x2);
  return __builtin_bit_cast(uint32_t, v);
}

// D = A * B + C, one wavefront, v_mfma_f32_16x16x32_bf16.  Lane l supplies row i = l & 15 of A
// and column j = l & 15 of B, the 8 elements of its 128-bit operand are the SAME 8 k-indices
// (a function of l >> 4 and the element number) in A and in B; lane l receives
// D[4 * (l >> 4) + r][l & 15] in element r.
__device__ __forceinline__ f32x4 mfma_16x16x32_bf16(u32x4 a, u32x4 b, f32x4 c) {
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ u32x4 u32x4_zero() { return u32x4{0u, 0u, 0u, 0u}; }

// ---- discipline around an MFMA chain ------------------------------------------------------------
// Observed on MI355X / ROCm 7.2 (profiles/r01_notes.md, "MFMA chain hazard"): with two or more
// wavefronts sharing a matrix core, a register that an already-issued v_mfma still has to read
// (as A, B or C) can be overwritten by a later LDS return or vector write before the MFMA reads it
// -- the compiler's wait-state counts assume the MFMA starts when it issues.  The chain in
// composite.hip therefore follows three rules, enforced with these helpers:
//   1. every operand fragment of the chain is loaded into its OWN registers before the first MFMA
//      (mfma_operands_ready pins them all live at once and waits for the loads),
//   2. between that point and the drain the instruction stream contains nothing but the MFMAs
//      (scheduling barriers on both sides: no other write can land in a register an MFMA reads),
//   3. the accumulators are consumed by real vector instructions (mfma_drain) before anything else
//      runs: a vector read of an MFMA result waits for it, MFMAs complete in order, so after the
//      drain every source register of the chain is free to be reused.
// The waits are software waits (s_nop): nothing in the hardware stalls a vector read of a register an MFMA in
// flight will still write, nor a write to one it still has to read.  Their length was first set for the steady
// state (lgkmcnt(0) before the chain, 16 wait states after it): green for thousands of back-to-back launches, but
// the FIRST launches of a process came out wrong on some boxes (20-100 % of process starts there; all four
// gradient outputs off by 1e-5 .. 1e-2 of their maximum; only seen in the two-wavefronts-per-tile build at SH
// degree 0) -- tools/mfma_stress.py, profiles/r01_notes.md "first-launch hazard".  Measured on such boxes, fresh
// processes: short waits 3/8, 1/5, 3/3 starts wrong; 8 wait states before + 48 after 0/16; 32 + 128: 0/2.
// Built with 16 before + 64 after (cfg2: -1 % against the short waits).  GSGEN_MFMA_SHORT_WAITS builds the old
// lengths for such experiments.
#ifndef GSGEN_MFMA_SHORT_WAITS
#define GSGEN_MFMA_PRE "s_waitcnt lgkmcnt(0)\n\ts_nop 15"
#define GSGEN_MFMA_POST "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
#else
#define GSGEN_MFMA_PRE "s_waitcnt lgkmcnt(0)"
#define GSGEN_MFMA_POST "s_nop 7\n\ts_nop 7\n\t"
#endif
__device__ __forceinline__ void mfma_operands_ready(u32x4 &a, u32x4 &b, u32x4 &c, u32x4 &d) {
  asm volatile(GSGEN_MFMA_PRE : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "memory");
}
__device__ __forceinline__ void mfma_operands_ready(u32x4 &a, u32x4 &b) {
  asm volatile(GSGEN_MFMA_PRE : "+v"(a), "+v"(b) : : "memory");
}
__device__ __forceinline__ void mfma_drain(f32x4 &a, f32x4 &b, f32x4 &c) {
  float a3 = a[3], b3 = b[3], c3 = c[3], sink;
  asm volatile(GSGEN_MFMA_POST "v_or_b32 %0, %1, %2\n\tv_or_b32 %0, %0, %3"
               : "=v"(sink) : "v"(a3), "v"(b3), "v"(c3) : "memory");
  asm volatile("" : "+v"(a), "+v"(b), "+v"(c) : : "memory");
}

// LDS written by some lanes of this wavefront is about to be read by others (or the reverse).  The
// hardware executes one wavefront's LDS operations in order; this only stops the compiler from
// moving them across.  (Not a workgroup barrier: wavefronts of a workgroup pass independently.)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// the value, but opaque to common-subexpression elimination
__device__ __forceinline__ float opaque(float v) {
  asm volatile("" : "+v"(v));
  return v;
}

}  // namespace gs
