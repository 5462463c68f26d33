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
// Rule 3 needs a wait for the chain, and nothing in the hardware provides one: a vector read of a register an MFMA
// in flight will still write is not stalled, nor is a write to one it still has to read.  The first version
// waited a fixed 16 wait states (s_nop), tuned in the steady state: green for thousands of back-to-back launches,
// but the FIRST launches of a process came out wrong on some boxes (20-100 % of process starts there; all four
// gradient outputs off by 1e-5 .. 1e-2 of their maximum; only caught in the two-wavefronts-per-tile build at SH
// degree 0) -- tools/mfma_stress.py, profiles/r01_notes.md "first-launch hazard".  Fresh processes on such boxes,
// wrong starts / starts: 16 wait states 3/3, 1/5, 3/8, 3/8; 8 before the chain + 48 after it 0/16; the
// completion poll below 0/10.  Default build: the poll (mfma_wait_chain) -- it waits as long as the chain really
// takes, whatever the clocks and the other wavefronts on the matrix core do -- at the cost of one extra MFMA per
// chain and four registers (cfg2: same throughput as 16 + 64 fixed wait states, 1 % below the unsafe 16).
// GSGEN_MFMA_FIXED_WAITS builds 16 + 64 fixed wait states instead, GSGEN_MFMA_SHORT_WAITS the original lengths
// (the hazard detector of tools/mfma_stress.py experiments).
#if defined(GSGEN_MFMA_PLAIN)
// experiment build: no discipline at all -- the chain exactly as hipcc schedules and pads it (tools/stress)
#define GSGEN_MFMA_PRE ""
#define GSGEN_MFMA_POST ""
#elif defined(GSGEN_MFMA_SHORT_WAITS)
#define GSGEN_MFMA_PRE "s_waitcnt lgkmcnt(0)"
#define GSGEN_MFMA_POST "s_nop 7\n\ts_nop 7\n\t"
#elif defined(GSGEN_MFMA_FIXED_WAITS)
#define GSGEN_MFMA_PRE "s_waitcnt lgkmcnt(0)\n\ts_nop 15"
#define GSGEN_MFMA_POST "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
#else
#define GSGEN_MFMA_POLL 1
#define GSGEN_MFMA_PRE "s_waitcnt lgkmcnt(0)\n\ts_nop 15"
#define GSGEN_MFMA_POST "s_nop 1\n\t"
#endif
#if defined(GSGEN_MFMA_PLAIN)
__device__ __forceinline__ void mfma_operands_ready(u32x4 &, u32x4 &, u32x4 &, u32x4 &) {}
__device__ __forceinline__ void mfma_operands_ready(u32x4 &, u32x4 &) {}
#else
__device__ __forceinline__ void mfma_operands_ready(u32x4 &a, u32x4 &b, u32x4 &c, u32x4 &d) {
  asm volatile(GSGEN_MFMA_PRE : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "memory");
}
__device__ __forceinline__ void mfma_operands_ready(u32x4 &a, u32x4 &b) {
  asm volatile(GSGEN_MFMA_PRE : "+v"(a), "+v"(b) : : "memory");
}
#endif
#ifdef GSGEN_MFMA_POLL
// The completion wait: one more MFMA is issued behind the chain into four registers this block owns, the last of
// them preloaded with a pattern no product of bf16 values can produce (a NaN with a payload below bf16's
// mantissa); MFMAs complete in order, so when that register changes the whole chain has read its operands and
// written its results.  The four registers are chosen per kernel variant (pixels per lane) so that the kernel's
// register count, hence its occupancy, stays what it was.
#define GSGEN_MFMA_PROBE_ASM(R0, R1, R2, R3)                                                  \
  asm volatile("v_mov_b32 v" R3 ", 0xffffdead\n\t"                                            \
               "s_nop 4\n\t"                                                                  \
               "v_mfma_f32_16x16x32_bf16 v[" R0 ":" R3 "], %3, %4, 0\n"                        \
               ".Lgsgen_poll%=:\n\t"                                                          \
               "s_nop 7\n\t"                                                                  \
               "v_cmp_eq_u32_e32 vcc, 0xffffdead, v" R3 "\n\t"                                 \
               "s_cbranch_vccnz .Lgsgen_poll%=\n\t"                                           \
               "s_nop 1"                                                                       \
               : "+v"(c0), "+v"(c1), "+v"(c2) : "v"(pa), "v"(pb) : "v" R0, "v" R1, "v" R2, "v" R3, "vcc", "memory")
// c0..c2: the chain's accumulators -- only named so that the block is ordered behind the MFMAs producing them
template <int PPL>
__device__ __forceinline__ void mfma_wait_chain(const u32x4 &pa, const u32x4 &pb, f32x4 &c0, f32x4 &c1, f32x4 &c2) {
  if constexpr (PPL == 4) GSGEN_MFMA_PROBE_ASM("252", "253", "254", "255");
  else if constexpr (PPL == 2) GSGEN_MFMA_PROBE_ASM("160", "161", "162", "163");
  else GSGEN_MFMA_PROBE_ASM("124", "125", "126", "127");
}
#else
template <int PPL>
__device__ __forceinline__ void mfma_wait_chain(const u32x4 &, const u32x4 &, f32x4 &, f32x4 &, f32x4 &) {}
#endif
#if defined(GSGEN_MFMA_PLAIN)
__device__ __forceinline__ void mfma_drain(f32x4 &, f32x4 &, f32x4 &) {}
#else
__device__ __forceinline__ void mfma_drain(f32x4 &a, f32x4 &b, f32x4 &c) {
  float a3 = a[3], b3 = b[3], c3 = c[3], sink;
  asm volatile(GSGEN_MFMA_POST "v_or_b32 %0, %1, %2\n\tv_or_b32 %0, %0, %3"
               : "=v"(sink) : "v"(a3), "v"(b3), "v"(c3) : "memory");
  asm volatile("" : "+v"(a), "+v"(b), "+v"(c) : : "memory");
}
#endif

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
