// common.hpp -- shared device helpers for the gfx950 rasterizer kernels.
//
// Execution model used throughout (MI355X / CDNA4): wave64.  A 16x16 pixel tile is owned by
// ONE workgroup of 256/PPL threads; with the default PPL = 4 that is exactly one wavefront,
// each lane owning 4 pixels of one column (rows ly0, ly0+4, ly0+8, ly0+12), so a Gaussian
// record staged in LDS is read once per wave and amortised over 256 pixels, and every
// "is anybody still alive / does anybody contribute" decision is a single 64-bit ballot.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// occupancy hint for the register allocator (the CPU emulator build compiles these files with g++, which does not know it)
#if defined(__clang__)
#define GS_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n)))
#else
#define GS_WAVES_PER_EU(n)
#endif

namespace gs {

constexpr int kTile = 16;
constexpr int kListOverflow = -2;  // start / end of every tile of a frame whose pairs did not fit the list (GSGEN_LIST_OVERFLOW)
constexpr float kMinAlpha = 0.00392156862745098f;  // 1/255  (reference common.h:89)
constexpr float kAlphaClamp = 0.99f;               // reference vol_render.h:212
constexpr float kLog2e = 1.4426950408889634f;
// relative half-width of the window around a*G == 1/255 inside which the fast fp32 Gaussian is
// re-evaluated with the reference's own arithmetic, so that the skip decision (a discontinuity
// of size ~1/255 in the image) is the reference's decision.
constexpr float kGuardTol = 2.0e-4f;
// The Cholesky-form record of the RGB / scalar modes holds p = L * sqrt(0.5 log2 e) with L L^T = Sigma^-1 (prepared in
// fp64 per record), so Sigma^-1 d = (p0 u, p1 u + p2 v) * kInvSc2 with u = p0 x + p1 y, v = p2 y -- the backward's
// v = Sigma^-1 d without the fp32 determinant (whose cancellation costs eps * cond on large, elongated splats).
constexpr float kInvSc2 = 1.3862943611198906f;  // 1 / (0.5 log2 e) = 2 ln 2

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// The evaluation record of an RGB / scalar / RGB + heads Gaussian: p = L * sqrt(0.5 log2 e), L L^T = Sigma^-1 (symmetrised), in fp64
// from the fp32 cov2d, rounded to fp32; ok = false for a degenerate or non-finite covariance (such a record never contributes).
// ONE function with contraction off for every translation unit that forms it -- the compositing kernels' staging (prep_record), the
// projection launch that prepares it once per (view, Gaussian) for the batched launches (round 6: gsgen_geometry_view::chol), the
// projection backward's moment expansion -- so that all of them hold the same bits.
struct CholRec { float p0, p1, p2; bool ok; };
__device__ __forceinline__ CholRec chol_prep(float c0, float c1, float c2, float c3) {
#pragma clang fp contract(off)
  const double d0 = c0, d1 = c1, d2 = c2, d3 = c3;
  const double det = d0 * d3 - d1 * d2;
  bool ok = (fabsf(c0) <= 3.402823466e+38f) && (fabsf(c1) <= 3.402823466e+38f) && (fabsf(c2) <= 3.402823466e+38f) &&
            (fabsf(c3) <= 3.402823466e+38f) && (det > 0.0) && (d3 > 0.0);
  const double sdet = ok ? det : 1.0, s3 = ok ? d3 : 1.0;
  const double qa = s3 / sdet, qb = -0.5 * (d1 + d2) / sdet, qc = d0 / sdet;
  const double l11 = sqrt(qa), l21 = qb / l11;
  const double l22s = qc - l21 * l21;
  ok = ok && (l22s > 0.0);
  const double l22 = sqrt(ok ? l22s : 1.0);
  const double sc = 0.84932180028801904;  // sqrt(0.5 log2 e)
  CholRec r{0.0f, 0.0f, 0.0f, ok};
  if (ok) { r.p0 = (float)(l11 * sc); r.p1 = (float)(l21 * sc); r.p2 = (float)(l22 * sc); }
  return r;
}

// two packed fp32 values: arithmetic on v2f lowers to v_pk_mul / v_pk_add / v_pk_fma_f32
typedef float v2f __attribute__((vector_size(8)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// a value every lane of the wave holds identically (e.g. an LDS broadcast read), moved to a scalar register
// "does any lane of the wavefront say yes": the ballot of a BOOLEAN (already a lane mask in scalar registers) -- __ballot(int)
// first materialises the flag as an integer per lane and compares it again (two vector instructions per test)
__device__ __forceinline__ bool wave_any(bool pred) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_ballot_w64(pred) != 0ull;
#else
  return __ballot((int)pred) != 0ull;
#endif
}
// Orders the memory operations (LDS and global) of THIS wavefront's lanes: what any lane stored before the call is visible to
// every lane of the same wavefront after it.  No s_barrier: correct wherever only one wavefront is involved -- including code
// that the other wavefronts of the workgroup have already LEFT (a __syncthreads() there relies on the hardware dropping
// terminated wavefronts from the barrier count, which HIP's model does not promise).
__device__ __forceinline__ void wave_mem_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#else
  (void)__ballot(1);  // (the CPU emulator runs lanes as fibers: a wave collective is its wave-level rendezvous)
#endif
}
__device__ __forceinline__ float wave_uniform(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

// ---- cross-lane exchange on the VALU (no LDS traffic) ---------------------------------------
// xchg_add<H>(lo, hi): with partner = lane ^ H, lanes with (lane & H) == 0 return
// lo + partner.lo, the others return hi + partner.hi.  One reduce-scatter step for a pair of
// components.  H = 32 / 16 use the gfx950 v_permlane{32,16}_swap instructions (one swap + one
// add), H = 8 / 4 two DPP row shifts restricted by bank masks (+ one add), H = 2 / 1 quad
// permutes and a select.  (The generic __shfl_xor path lowers to ds_bpermute_b32 through the
// LDS crossbar; the backward spent half of its time there.)
__device__ __forceinline__ float as_f(unsigned u) { return __uint_as_float(u); }
__device__ __forceinline__ unsigned as_u(float f) { return __float_as_uint(f); }

template <int CTRL, int BANK_MASK>
__device__ __forceinline__ float dpp_mov(float old, float src) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xf,
                                                    BANK_MASK, false));
}
constexpr int kDppQuadXor1 = 0xB1;  // quad_perm:[1,0,3,2]
constexpr int kDppQuadXor2 = 0x4E;  // quad_perm:[2,3,0,1]
constexpr int kDppRowShl = 0x100;   // row_shl:n -> lane i reads lane i+n of its 16-lane row
constexpr int kDppRowShr = 0x110;   // row_shr:n -> lane i reads lane i-n

// quad permute of a value every lane of which is read: no "old" operand, so the compiler folds the permute into the
// instruction that consumes it (v_add_f32_dpp) instead of copy + v_mov_b32_dpp + add
template <int CTRL>
__device__ __forceinline__ float quad_perm(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
// a + b as ONE scalar add whatever the vectoriser thinks: horizontal sums of packed accumulators -- (a0 + a1, b0 + b1)
// written as a vector expression costs three register moves and a packed add
__device__ __forceinline__ float add_scalar(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
  float r;
  asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
#else
  return a + b;
#endif
}

template <int H>
__device__ __forceinline__ float xchg_add(float lo, float hi) {
  if constexpr (H == 32) {
    const auto r = __builtin_amdgcn_permlane32_swap(as_u(lo), as_u(hi), false, false);
    return as_f(r[0]) + as_f(r[1]);
  } else if constexpr (H == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(as_u(lo), as_u(hi), false, false);
    return as_f(r[0]) + as_f(r[1]);
  } else if constexpr (H == 8) {
    // banks (4-lane groups of a row) 0,1 are class 0, banks 2,3 class 1
    const float t = dpp_mov<kDppRowShl + 8, 0x3>(hi, lo);  // class 0: partner.lo ; class 1: own hi
    const float u = dpp_mov<kDppRowShr + 8, 0xC>(lo, hi);  // class 1: partner.hi ; class 0: own lo
    return t + u;
  } else if constexpr (H == 4) {
    const float t = dpp_mov<kDppRowShl + 4, 0x5>(hi, lo);
    const float u = dpp_mov<kDppRowShr + 4, 0xA>(lo, hi);
    return t + u;
  } else {
    static_assert(H == 2 || H == 1, "H must be a power of two <= 32");
    constexpr int ctrl = (H == 2) ? kDppQuadXor2 : kDppQuadXor1;
    const float a = lo + quad_perm<ctrl>(lo);
    const float b = hi + quad_perm<ctrl>(hi);
    return (lane_id() & H) ? b : a;
  }
}

// all-lanes xor-butterfly step on a single value: v + v[lane ^ H]
template <int H>
__device__ __forceinline__ float xor_add(float v) { return xchg_add<H>(v, v); }

// In-row exchange with the add fused into the DPP instruction and the two classes selected by
// bank masks: 2 instructions per exchange instead of 2 movs + add (device pass only: the host
// pass and the CPU emulator take the builtin form above).
template <int H>
__device__ __forceinline__ float xchg_add_row(float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
  float d;
  if constexpr (H == 8) {
    asm volatile("s_nop 1\n v_add_f32_dpp %0, %1, %1 row_shl:8 row_mask:0xf bank_mask:0x3\n"
                 " v_add_f32_dpp %0, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xc" : "=&v"(d) : "v"(lo), "v"(hi));
    return d;
  } else if constexpr (H == 4) {
    asm volatile("s_nop 1\n v_add_f32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n"
                 " v_add_f32_dpp %0, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa" : "=&v"(d) : "v"(lo), "v"(hi));
    return d;
  } else {
    return xchg_add<H>(lo, hi);
  }
#else
  return xchg_add<H>(lo, hi);
#endif
}

// Reduce-scatter over the 64 lanes of a wave.  v[0..P) are per-lane partial sums of P
// components (P a power of two <= 64).  The exchange levels run over the lane distances of
// kLaneDist (32, 16: v_permlane swaps; 8, 4, 2, 1: DPP inside a 16-lane row); the first log2(P)
// levels halve the live components, the remaining ones are plain all-reduce steps on the one
// surviving value.  On return v[0] in lane l holds the wave-wide total of component
// scatter_comp<P>(l); lanes with scatter_owner<P>(l) hold each component exactly once.
// Cost: P-1 exchanges instead of 6*P for P independent butterflies.
// Order chosen by kernel-level A/B on MI355X (profiles/r01_notes.md): cross-row swaps first
// (413 us backward) vs in-row first (437 us builtin, 419 us with the fused-DPP asm).
constexpr int kLaneDist[6] = {32, 16, 8, 4, 2, 1};
constexpr int ilog2c(int v) { return v <= 1 ? 0 : 1 + ilog2c(v >> 1); }

template <int P>
__device__ __forceinline__ int scatter_comp(int lane) {
  int comp = 0;
#pragma unroll
  for (int k = 0; k < ilog2c(P); ++k) comp |= ((lane & kLaneDist[k]) ? (P >> (k + 1)) : 0);
  return comp;
}
template <int P>
__device__ __forceinline__ bool scatter_owner(int lane) {
  int rest = 0;
#pragma unroll
  for (int k = ilog2c(P); k < 6; ++k) rest |= kLaneDist[k];
  return (lane & rest) == 0;
}

template <int L>
__device__ __forceinline__ float xchg_any(float lo, float hi) {
  if constexpr (L == 8 || L == 4) return xchg_add_row<L>(lo, hi);  // two fused v_add_f32_dpp instead of two moves + an add
  return xchg_add<L>(lo, hi);
}
template <int P, int K>
__device__ __forceinline__ void reduce_scatter_level(float (&v)[P]) {
  if constexpr (K < 6) {
    constexpr int L = kLaneDist[K];
    if constexpr (K < ilog2c(P)) {
      constexpr int S = P >> (K + 1);
#pragma unroll
      for (int i = 0; i < S; ++i) v[i] = xchg_any<L>(v[i], v[i + S]);
    } else {
      v[0] = xchg_any<L>(v[0], v[0]);
    }
    reduce_scatter_level<P, K + 1>(v);
  }
}
template <int P>
__device__ __forceinline__ void wave_reduce_scatter(float (&v)[P]) {
  static_assert(P >= 1 && P <= 64 && (P & (P - 1)) == 0, "P must be a power of two <= 64");
  reduce_scatter_level<P, 0>(v);
}

// The same reduce-scatter on components held as (even, odd) pairs: v2[i] = (v[2i], v[2i+1]).  The exchanges are
// per register as above, but the adds of a pair are one v_pk_add_f32 (packed fp32 adds issue at the rate of scalar
// ones on gfx950), which halves the add count of the levels that still hold two or more components per lane.
template <int H>
__device__ __forceinline__ v2f xchg_add2(v2f lo, v2f hi) {
  if constexpr (H == 32) {
    const auto a = __builtin_amdgcn_permlane32_swap(as_u(lo[0]), as_u(hi[0]), false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(as_u(lo[1]), as_u(hi[1]), false, false);
    return v2f{as_f(a[0]), as_f(b[0])} + v2f{as_f(a[1]), as_f(b[1])};
  } else if constexpr (H == 16) {
    const auto a = __builtin_amdgcn_permlane16_swap(as_u(lo[0]), as_u(hi[0]), false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(as_u(lo[1]), as_u(hi[1]), false, false);
    return v2f{as_f(a[0]), as_f(b[0])} + v2f{as_f(a[1]), as_f(b[1])};
  } else if constexpr (H == 8) {
    const v2f t = {dpp_mov<kDppRowShl + 8, 0x3>(hi[0], lo[0]), dpp_mov<kDppRowShl + 8, 0x3>(hi[1], lo[1])};
    const v2f u = {dpp_mov<kDppRowShr + 8, 0xC>(lo[0], hi[0]), dpp_mov<kDppRowShr + 8, 0xC>(lo[1], hi[1])};
    return t + u;
  } else if constexpr (H == 4) {
    const v2f t = {dpp_mov<kDppRowShl + 4, 0x5>(hi[0], lo[0]), dpp_mov<kDppRowShl + 4, 0x5>(hi[1], lo[1])};
    const v2f u = {dpp_mov<kDppRowShr + 4, 0xA>(lo[0], hi[0]), dpp_mov<kDppRowShr + 4, 0xA>(lo[1], hi[1])};
    return t + u;
  } else {
    static_assert(H == 2 || H == 1, "H must be a power of two <= 32");
    constexpr int ctrl = (H == 2) ? kDppQuadXor2 : kDppQuadXor1;
    const v2f a = lo + v2f{dpp_mov<ctrl, 0xf>(lo[0], lo[0]), dpp_mov<ctrl, 0xf>(lo[1], lo[1])};
    const v2f b = hi + v2f{dpp_mov<ctrl, 0xf>(hi[0], hi[0]), dpp_mov<ctrl, 0xf>(hi[1], hi[1])};
    return (lane_id() & H) ? b : a;
  }
}
// v2[0 .. P/2): on return v2[0][0] in lane l holds the wave-wide total of component scatter_comp<P>(l), exactly as
// wave_reduce_scatter<P> leaves it in v[0] (same exchange order, same owners).
template <int P, int K, int KEND = 6>
__device__ __forceinline__ void reduce_scatter2_level(v2f (&v2)[P / 2]) {
  if constexpr (K < KEND) {
    constexpr int L = kLaneDist[K];
    if constexpr (K < ilog2c(P)) {
      constexpr int S = P >> (K + 1);  // live components after this level
      if constexpr (S >= 2) {
#pragma unroll
        for (int i = 0; i < S / 2; ++i) v2[i] = xchg_add2<L>(v2[i], v2[i + S / 2]);
      } else {
        v2[0][0] = xchg_any<L>(v2[0][0], v2[0][1]);
      }
    } else {
      v2[0][0] = xchg_any<L>(v2[0][0], v2[0][0]);
    }
    reduce_scatter2_level<P, K + 1, KEND>(v2);
  }
}
template <int P>
__device__ __forceinline__ void wave_reduce_scatter2(v2f (&v2)[P / 2]) {
  static_assert(P >= 2 && P <= 64 && (P & (P - 1)) == 0, "P must be a power of two, 2..64");
  reduce_scatter2_level<P, 0>(v2);
}
// The first four levels only (lane distances 32, 16, 8, 4; P <= 16): v2[0][0] in lane l is the total of component
// scatter_comp<P>(l) over the 16 lanes that share l's two low bits.  Four such partial results (say of four
// independent vectors) are then finished by ONE 4-component reduce-scatter over the distances 2 and 1
// (quad_reduce_scatter4): the lane's low bits select the vector, so 4 x 16 components end up one per lane.
template <int P>
__device__ __forceinline__ float wave_reduce_scatter2_rows(v2f (&v2)[P / 2]) {
  static_assert(P >= 2 && P <= 16 && (P & (P - 1)) == 0, "P must be a power of two, 2..16");
  reduce_scatter2_level<P, 0, 4>(v2);
  return v2[0][0];
}
// lane l returns the sum over its quad (lanes l ^ 1, l ^ 2, l ^ 3 and itself) of r[l & 3]
__device__ __forceinline__ float quad_reduce_scatter4(float r0, float r1, float r2, float r3) {
  r0 = xchg_add<2>(r0, r2);
  r1 = xchg_add<2>(r1, r3);
  return xchg_add<1>(r0, r1);
}
// lanes that hold a component of wave_reduce_scatter2_rows<P> exactly once within their 16-lane class
template <int P>
__device__ __forceinline__ bool scatter_rows_owner(int lane) {
  int rest = 0;
#pragma unroll
  for (int k = ilog2c(P); k < 4; ++k) rest |= kLaneDist[k];
  return (lane & rest) == 0;
}

// Reduce-scatter of TEN components (round 6: the moment form of the RGB + heads backward).  c[0], c[1] = pairs A, B and c[3], c[4]
// = pairs C, D are exchanged whole at lane distance 32 (A <-> C, B <-> D), the two singles e, f = c[2] against each other; at
// distance 16 the surviving pairs are exchanged against each other and the single is all-reduced; distances 8 and 4 finish
// (W0, W1, Z, Z) like a four-component scatter, 2 and 1 are all-reduce steps.  8 lane swaps + 5 adds at the two cross-row
// levels instead of the 12 + 6 of a 16-wide scatter with six empty slots.  Returns the lane's total; component numbering
// (scatter10_comp): 0 1 = A, 2 3 = B, 4 5 = C, 6 7 = D, 8 = e, 9 = f, i.e. c = {(0,1), (2,3), (8,9), (4,5), (6,7)}.
__device__ __forceinline__ float wave_reduce_scatter10(v2f (&c)[5]) {
  const v2f X = xchg_add2<32>(c[0], c[3]);                 // lanes < 32: A, lanes >= 32: C
  const v2f Y = xchg_add2<32>(c[1], c[4]);                 // B | D
  float Z = xchg_any<32>(c[2][0], c[2][1]);                // e | f
  const v2f W = xchg_add2<16>(X, Y);                       // bit 16 clear: X's pair, set: Y's pair
  Z = xchg_any<16>(Z, Z);
  const float P1 = xchg_any<8>(W[0], Z), P2 = xchg_any<8>(W[1], Z);  // bit 8 clear: (W0, W1), set: (Z, Z)
  float Q = xchg_any<4>(P1, P2);
  Q = xchg_any<2>(Q, Q);
  return xchg_any<1>(Q, Q);
}
__device__ __forceinline__ int scatter10_comp(int lane) {
  if (lane & 8) return (lane & 32) ? 9 : 8;
  return ((lane & 32) ? 4 : 0) + ((lane & 16) ? 2 : 0) + ((lane & 4) ? 1 : 0);
}
__device__ __forceinline__ bool scatter10_owner(int lane) {
  if (lane & 3) return false;
  return (lane & 8) ? (lane & (16 | 4)) == 0 : true;
}

// Only the log2(P) halving levels: every lane ends with the partial sum of component
// scatter_comp<P>(lane) over the lanes that agree with it on the remaining lane bits (64 / P
// partials per component, to be combined by the caller -- e.g. by the atomics that follow anyway).
template <int P, int K>
__device__ __forceinline__ void reduce_scatter_halving_level(float (&v)[P]) {
  if constexpr (K < ilog2c(P)) {
    constexpr int L = kLaneDist[K];
    constexpr int S = P >> (K + 1);
#pragma unroll
    for (int i = 0; i < S; ++i) v[i] = xchg_any<L>(v[i], v[i + S]);
    reduce_scatter_halving_level<P, K + 1>(v);
  }
}
template <int P>
__device__ __forceinline__ void wave_reduce_scatter_partial(float (&v)[P]) {
  reduce_scatter_halving_level<P, 0>(v);
}

// ---- wave64 prefix scans on DPP (GCN/CDNA row_shr + row_bcast idiom, 7 fused steps) --------
constexpr int kDppRowBcast15 = 0x142;  // lane 15 of each row -> every lane of the next row
constexpr int kDppRowBcast31 = 0x143;  // lane 31 -> rows 2 and 3
constexpr int kDppWaveShr1 = 0x138;    // lane i reads lane i-1 across the whole wave

template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_get(float identity, float src) {
  // lanes masked off, or whose source lane does not exist, receive `identity`
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(src), CTRL,
                                                    ROW_MASK, BANK_MASK, false));
}
// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ float wave_scan_add(float x) {
  float v = x + dpp_get<kDppRowShr + 1, 0xf, 0xf>(0.0f, x);
  v += dpp_get<kDppRowShr + 2, 0xf, 0xf>(0.0f, x);
  v += dpp_get<kDppRowShr + 3, 0xf, 0xf>(0.0f, x);
  v += dpp_get<kDppRowShr + 4, 0xf, 0xe>(0.0f, v);
  v += dpp_get<kDppRowShr + 8, 0xf, 0xc>(0.0f, v);
  v += dpp_get<kDppRowBcast15, 0xa, 0xf>(0.0f, v);
  v += dpp_get<kDppRowBcast31, 0xc, 0xf>(0.0f, v);
  return v;
}
// inclusive prefix product over the 64 lanes
__device__ __forceinline__ float wave_scan_mul(float x) {
  float v = x * dpp_get<kDppRowShr + 1, 0xf, 0xf>(1.0f, x);
  v *= dpp_get<kDppRowShr + 2, 0xf, 0xf>(1.0f, x);
  v *= dpp_get<kDppRowShr + 3, 0xf, 0xf>(1.0f, x);
  v *= dpp_get<kDppRowShr + 4, 0xf, 0xe>(1.0f, v);
  v *= dpp_get<kDppRowShr + 8, 0xf, 0xc>(1.0f, v);
  v *= dpp_get<kDppRowBcast15, 0xa, 0xf>(1.0f, v);
  v *= dpp_get<kDppRowBcast31, 0xc, 0xf>(1.0f, v);
  return v;
}
// Sum over each aligned group of 8 lanes, delivered to the group's last lane ((lane & 7) == 7):
// three dependent row-shift adds (the xor-butterfly all-reduce costs nine instructions and gives
// every lane a copy nobody needs).
__device__ __forceinline__ float group8_sum_to_last(float v) {
  v += dpp_get<kDppRowShr + 4, 0xf, 0xf>(0.0f, v);
  v += dpp_get<kDppRowShr + 2, 0xf, 0xf>(0.0f, v);
  v += dpp_get<kDppRowShr + 1, 0xf, 0xf>(0.0f, v);
  return v;
}

// value of the previous lane (lane 0 receives `identity`)
__device__ __forceinline__ float wave_shift_up(float identity, float x) {
  return dpp_get<kDppWaveShr1, 0xf, 0xf>(identity, x);
}
__device__ __forceinline__ float read_lane(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// two packed fp32 values: arithmetic on v2f lowers to v_pk_mul/add/fma_f32.  Kernel-level A/B
// on MI355X (profiles/r01_notes.md): packed SH dot products beat scalar FMA pairs by ~11 % on
// the compositing backward (fewer issue slots), although an isolated dependent-chain
// microbenchmark of v_pk_fma_f32 suggests otherwise.
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return a * b + c; }
__device__ __forceinline__ v2f splat2(float a) { return v2f{a, a}; }
// Explicitly fused multiply-adds (one rounding), scalar and packed: used where forward and backward kernels must
// produce the same bits from the same inputs whatever the surrounding code looks like (the Gaussian evaluation
// that decides "skip" and "saturated"), instead of leaving the choice of what to contract to the compiler.  The CPU
// emulator build (g++, -ffp-contract=off) takes the fmaf form, which is the same arithmetic.
__device__ __forceinline__ float ffma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ v2f ffma2(v2f a, v2f b, v2f c) {
#if defined(__clang__)
  return __builtin_elementwise_fma(a, b, c);
#else
  return v2f{__builtin_fmaf(a[0], b[0], c[0]), __builtin_fmaf(a[1], b[1], c[1])};
#endif
}

// 1 - x with x taken AS ROUNDED, whatever produced it: under -ffp-contract=fast a subtraction whose operand is a visible
// product a * b becomes fma(-a, b, 1) in some instantiations of a kernel and stays two roundings in others (and
// fma(-x, 1, 1) is first folded to that subtraction: no help).  The transmittance T (1 - a G) decides "saturated" in the forward
// AND the backward, in every shape of them -- they must agree to the bit.
__device__ __forceinline__ v2f one_minus2(v2f x) {
#pragma clang fp contract(off)
  return v2f{1.0f, 1.0f} - x;
}

// Workgroup -> tile map that is both XCD-local and XCD-balanced.  Workgroup b runs on XCD
// b % 8 (observed dispatch, speed only).  The tile grid is cut into 4x4-tile super-tiles;
// super-tile s belongs to XCD s % 8, so every XCD owns small clusters spread over the whole
// image (the image centre carries lists several times longer than the border -- handing each
// XCD a contiguous band of rows leaves the XCDs with the border rows idle), while the 16
// neighbouring tiles of a cluster share most of their Gaussians in that XCD's private L2.
// Launch tile_map_blocks() workgroups; some map outside the grid and simply exit.
__host__ __device__ __forceinline__ uint32_t tile_map_blocks(int ntw, int nth) {
  const uint32_t S = (uint32_t)(((ntw + 3) >> 2) * ((nth + 3) >> 2));
  return ((S + 7u) >> 3) * 8u * 16u;
}
__device__ __forceinline__ bool tile_of_block(uint32_t b, int ntw, int nth, int &tx, int &ty) {
  const uint32_t xcd = b & 7u, idx = b >> 3;
  const uint32_t j = idx >> 4, k = idx & 15u;
  const uint32_t sw = (uint32_t)((ntw + 3) >> 2), sh = (uint32_t)((nth + 3) >> 2);
  const uint32_t sidx = xcd + 8u * j;
  if (sidx >= sw * sh) return false;
  tx = (int)((sidx % sw) * 4u + (k & 3u));
  ty = (int)((sidx / sw) * 4u + (k >> 2));
  return tx < ntw && ty < nth;
}

// Bijective remap of the linear workgroup index so that each XCD (workgroup b lands on XCD
// b % 8 on MI355X) walks a contiguous range of tiles: neighbouring tiles share Gaussians, and
// each XCD has a private 4 MiB L2.  Speed only, never correctness.
__device__ __forceinline__ uint32_t xcd_swizzle(uint32_t b, uint32_t nb) {
  const uint32_t xcd = b & 7u, idx = b >> 3;
  const uint32_t full = nb >> 3, rem = nb & 7u;
  return xcd * full + (xcd < rem ? xcd : rem) + idx;
}

// pixel centre in normalised camera space: topleft + g * pixel_size, the product rounded
// before the add as written in the reference (vol_render.h:186-187) and in the oracle.
__device__ __forceinline__ float pixel_coord(float origin, int g, float pixel_size) {
#pragma clang fp contract(off)
  const float m = (float)g * pixel_size;
  return origin + m;
}

__device__ __forceinline__ bool finite_f(float v) { return fabsf(v) <= 3.402823466e+38f; }

// One camera of a batched geometry enqueue (gsgen_frame_geometry_batch): everything that differs per
// view; the kernels take the table in device memory and pick their view with blockIdx.y / .z.
struct GeoView {
  const float *cam;
  float *mean2d, *cov2d, *depth;
  uint8_t *mask;
  int *tl, *br;
  uint32_t *cnt, *wcnt, *tile_count, *tile_off, *ctrl, *tile_order;
  unsigned long long *keys;
  int *ids, *start, *end;
  uint32_t *total;
  uint32_t cap, pad_;
  uint32_t *report;  // optional, HOST-visible (pinned) memory: [0] = the view's pair count, [1] = max over the overflowing frames
  // optional per-view gradient accumulators of the step's backward ([N,2], [N,2,2], [N,6]; any may be NULL): zero-filled by
  // the projection launch, which touches every Gaussian of the view anyway (gsgen_frame_geometry_batch_zero)
  float *z_mean2d, *z_cov2d, *z_chan6;
  float *chol;  // optional [N,4]: (p0, p1, p2, ok) of chol_prep per Gaussian of the view, for the batched RGB / RGB + heads compositing launches
  float *max_r;  // optional [N], shared by the views: running maximum of the screen-space radius (gs/gaussian_splatting.py:1240-1245)
};

}  // namespace gs
