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

namespace gs {

constexpr int kTile = 16;
constexpr float kMinAlpha = 0.00392156862745098f;  // 1/255  (reference common.h:89)
constexpr float kAlphaClamp = 0.99f;               // reference vol_render.h:212
constexpr float kLog2e = 1.4426950408889634f;
// relative half-width of the window around a*G == 1/255 inside which the fast fp32 Gaussian is
// re-evaluated with the reference's own arithmetic, so that the skip decision (a discontinuity
// of size ~1/255 in the image) is the reference's decision.
constexpr float kGuardTol = 2.0e-4f;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// Reduce-scatter over the 64 lanes of a wave.  v[0..P) are per-lane partial sums of P
// components (P a power of two <= 64).  On return v[0] in lane l holds the wave-wide total of
// component (l & (P-1)).  Costs P-1 exchanges for the scatter phase instead of 6*P for P
// independent butterflies.
template <int P>
__device__ __forceinline__ void wave_reduce_scatter(float (&v)[P]) {
  static_assert(P >= 1 && P <= 64 && (P & (P - 1)) == 0, "P must be a power of two <= 64");
  const int lane = lane_id();
  // scatter phase: masks P/2, P/4, ... 1
#pragma unroll
  for (int h = P / 2; h >= 1; h >>= 1) {
    const bool up = (lane & h) != 0;
#pragma unroll
    for (int i = 0; i < h; ++i) {
      const float lo = v[i], hi = v[i + h];
      const float send = up ? lo : hi;
      const float keep = up ? hi : lo;
      v[i] = keep + __shfl_xor(send, h, 64);
    }
  }
  // remaining lanes-with-equal-(l & (P-1)) hold partials of the same component
#pragma unroll
  for (int m = 32; m >= P; m >>= 1) v[0] += __shfl_xor(v[0], m, 64);
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

}  // namespace gs
