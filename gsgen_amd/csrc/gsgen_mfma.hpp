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
__device__ __forceinline__ f32x4 mfma_16x16x32_bf16(uint4 a, uint4 b, f32x4 c) {
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Scheduling fence after one k-step of a multi-accumulator MFMA chain: the accumulators stay in
// their registers and no memory operation moves across (see composite.hip, flush()).
__device__ __forceinline__ void mfma_step_fence(f32x4 &a, f32x4 &b, f32x4 &c) {
  asm volatile("s_nop 1" : "+v"(a), "+v"(b), "+v"(c) : : "memory");
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
