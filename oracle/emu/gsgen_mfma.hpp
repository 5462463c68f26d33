// CPU restatement of gsgen_amd/csrc/gsgen_mfma.hpp for the SIMT emulator (test infrastructure).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

namespace gs {

struct f32x4 {
  float v[4];
  float &operator[](int i) { return v[i]; }
  float operator[](int i) const { return v[i]; }
};
inline f32x4 f32x4_zero() { return f32x4{{0.0f, 0.0f, 0.0f, 0.0f}}; }
typedef uint4 u32x4;
inline u32x4 u32x4_zero() { return u32x4{0u, 0u, 0u, 0u}; }

inline uint32_t emu_bf16_rne(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  if ((u & 0x7f800000u) == 0x7f800000u) return u >> 16;  // inf / nan: truncate
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
inline uint32_t pack_bf16x2(float a, float b) { return emu_bf16_rne(a) | (emu_bf16_rne(b) << 16); }

inline float emu_bf16_to_f32(uint32_t h) {
  const uint32_t u = h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// v_mfma_f32_16x16x32_bf16: A[i][k] lives in lane i + 16 * (k / 8), element k % 8 (same for
// B[k][j] with j in place of i); D[4 * (l >> 4) + r][l & 15] comes back in element r of lane l.
inline f32x4 mfma_16x16x32_bf16(uint4 a, uint4 b, f32x4 c) {
  const int lane = simt::lane_of_current();
  const unsigned aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
  f32x4 d = c;
  unsigned bcol[4][4];  // B column j = lane & 15: [k group][dword]
  for (int g = 0; g < 4; ++g)
    for (int w = 0; w < 4; ++w) bcol[g][w] = simt::shfl_idx(bw[w], (lane & 15) + 16 * g);
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * (lane >> 4) + r;
    double acc = 0.0;
    for (int g = 0; g < 4; ++g)
      for (int w = 0; w < 4; ++w) {
        const unsigned av = simt::shfl_idx(aw[w], i + 16 * g);
        const unsigned bv = bcol[g][w];
        acc += (double)emu_bf16_to_f32(av & 0xffffu) * (double)emu_bf16_to_f32(bv & 0xffffu);
        acc += (double)emu_bf16_to_f32(av >> 16) * (double)emu_bf16_to_f32(bv >> 16);
      }
    d.v[r] = (float)((double)c.v[r] + acc);
  }
  return d;
}

// lanes are fibers that only switch at collectives: make this one, so that every lane's LDS
// writes are done before any lane reads
inline void wave_lds_sync() { (void)simt::shfl_idx(0, 0); }
inline void mfma_operands_ready(uint4 &, uint4 &, uint4 &, uint4 &) {}
inline void mfma_operands_ready(uint4 &, uint4 &) {}
inline void mfma_drain(f32x4 &, f32x4 &, f32x4 &) {}
template <int PPL>
inline void mfma_wait_chain(const uint4 &, const uint4 &, f32x4 &, f32x4 &, f32x4 &) {}
inline float opaque(float v) { return v; }

}  // namespace gs
