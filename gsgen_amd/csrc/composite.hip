// composite.hip -- front-to-back alpha compositing, forward and backward, for gfx950.
//
// Replaces (reference paths relative to /root/reference/gs/src/include):
//   RGB     fwd vol_render.h:994-1062 (+ body :169-265), bwd :866-973 (+ body :318-418)
//   scalar  fwd vol_render_scalar.h:14-102, bwd :104-234
//   SH      fwd vol_render_sh.h:97-248 / vol_render_bg.h:12-110,
//           bwd vol_render_sh.h:268-455 / vol_render_bg.h:131-242, basis shencoder.h:13-62
//
// Design (see common.hpp): one workgroup per 16x16 tile, 256/PPL threads, PPL pixels per
// lane; the tile's depth-sorted Gaussian list is staged 64 records at a time into LDS with
// coalesced loads (one record's coefficients are contiguous), then every lane walks the
// staged records with broadcast LDS reads.  Early termination (T < thresh) and the
// "nobody contributes" skip are wave ballots.  The backward re-walks the list front to back
// exactly as the reference does (suffix = final - prefix), reduces the per-pixel gradient
// contributions of a Gaussian across the wave with a register reduce-scatter, and issues
// ONE vector atomic (<= 55 consecutive floats) per (tile, Gaussian) instead of the
// reference's one LDS atomic per (pixel, Gaussian, component).  The SH kernels that run by default
// (k_composite_fwd_sh_vec / k_composite_bwd_sh_vec) carry the per-pixel arithmetic as packed fp32 pixel
// pairs; the backward can be launched per (tile, list segment) from checkpoints the forward leaves
// behind; given the device-resident bound on the coefficients (gsgen_sh_l1_bound) the SH degree-3 launches
// evaluate the per-pixel SH basis through a per-tile polynomial fit (NB = 6, composite_common.hpp) for the views
// whose error bound holds -- routed per view on the device (poly_route), the exact kernel renders the others.
//
// Numerics: the per-(pixel, Gaussian) Gaussian is evaluated in fp32 on the fast path (the
// reference uses fp64 for RGB/scalar, fp32 for SH).  For RGB/scalar the quadratic form is
// evaluated through a Cholesky factor computed once per staged record in fp64, which keeps
// the fp32 error at eps*sqrt(cond) instead of eps*cond.  Whenever a*G lands within kGuardTol
// of the 1/255 skip threshold the value is recomputed with the reference's own arithmetic
// (fp64 / uncontracted fp32), so the discontinuous decision is the reference's.  Forward and
// backward share eval code, so the backward's recomputed prefix equals the forward bit for
// bit (the invariant the reference asserts at vol_render_sh.h:452-454).
#include "composite_common.hpp"
#include <stdio.h>
#include <string.h>
#include <string>
#include <type_traits>
#include <vector>

namespace gs {

// ---- LDS staging ---------------------------------------------------------------------------
// WITH_COL = false: no staged coefficients (the polynomial-basis kernels transform them straight from HBM, poly_transform)
template <int MODE, int CB, int KB = kBatch, bool WITH_COL = true>
struct Stage {
  using TR = Traits<MODE, CB>;
  float mx[KB], my[KB], a[KB];
  float c0[KB], c1[KB], c2[KB], c3[KB];
  float p0[KB], p1[KB], p2[KB];  // RGB/scalar: scaled Cholesky factor; SH: p0 = kk, p1 = 1/det
  int id[KB];
  alignas(16) float col[WITH_COL ? KB * TR::NCOLP : 4];
};

// SCALE: the staged SH coefficients are pre-multiplied by -log2(e), so that the colour evaluation's
// sigmoid(s) = 1 / (1 + exp2(-log2(e) s)) needs no multiply per (pixel, channel)
template <int MODE, int CB, int NT, int KB = kBatch, bool SCALE = false, bool WITH_COL = true>
__device__ __forceinline__ void stage_batch(Stage<MODE, CB, KB, WITH_COL> &S, const CompParams &p, int list_base,
                                            int nb) {
  using TR = Traits<MODE, CB>;
  const int t = (int)threadIdx.x;
  static_assert(KB <= NT, "one thread per staged record");
  if (t < KB) {
    int id = 0;
    float mx = 0.f, my = 0.f, a = 0.f, c0 = 1.f, c1 = 0.f, c2 = 0.f, c3 = 1.f, p0 = 0.f, p1 = 0.f,
          p2 = 0.f;
    bool prepared = false;
    if constexpr (MODE != MODE_SH) prepared = p.chol != nullptr;  // (uniform over the launch)
    if (t < nb) {
      id = p.ids[list_base + t];
      const float2 m = *reinterpret_cast<const float2 *>(p.mean + 2 * (size_t)id);
      mx = m.x; my = m.y;
      if (prepared) {
        // the record the projection launch prepared for this (view, Gaussian) (gsgen_geometry_view::chol): no fp64 chain here
        const float4 q = *reinterpret_cast<const float4 *>(p.chol + 4 * (size_t)id);
        const float al = fminf(p.alpha[id], kAlphaClamp);
        const bool ok = q.w != 0.0f && finite_f(mx) && finite_f(my) && finite_f(al);
        a = ok ? al : 0.0f; p0 = ok ? q.x : 0.0f; p1 = ok ? q.y : 0.0f; p2 = ok ? q.z : 0.0f;
      } else {
        const float4 c = *reinterpret_cast<const float4 *>(p.cov + 4 * (size_t)id);
        c0 = c.x; c1 = c.y; c2 = c.z; c3 = c.w;
        const GRec r = prep_record<MODE>(mx, my, c0, c1, c2, c3, p.alpha[id]);
        a = r.a; p0 = r.p0; p1 = r.p1; p2 = r.p2;
      }
    }
    S.id[t] = id; S.mx[t] = mx; S.my[t] = my; S.a[t] = a;
    if (!prepared) { S.c0[t] = c0; S.c1[t] = c1; S.c2[t] = c2; S.c3[t] = c3; }  // (prepared: the guard path reads cov from memory)
    S.p0[t] = p0; S.p1[t] = p1; S.p2[t] = p2;
  }
  if constexpr (MODE == MODE_SH) __syncthreads();  // S.id is consumed below by other lanes
  // colour / scalar / SH coefficients: NCOL contiguous floats per record in HBM
  if constexpr (!WITH_COL) {
    return;
  } else if constexpr (MODE == MODE_SH && TR::CCP == TR::CC) {
    constexpr int Q = TR::NCOL / 4;
    for (int e = t; e < nb * Q; e += NT) {
      const int g = e / Q, k = e - g * Q;
      float4 v = *reinterpret_cast<const float4 *>(p.col + (size_t)S.id[g] * TR::NCOL + 4 * k);
      if constexpr (SCALE) { v.x *= -kLog2e; v.y *= -kLog2e; v.z *= -kLog2e; v.w *= -kLog2e; }
      *reinterpret_cast<float4 *>(&S.col[g * TR::NCOLP + 4 * k]) = v;
    }
  } else if constexpr (MODE == MODE_SH) {
    // channel stride CC in HBM -> CCP in LDS; the pad lanes were zeroed once at kernel start
    for (int e = t; e < nb * TR::NCOL; e += NT) {
      const int g = e / TR::NCOL, k = e - g * TR::NCOL;
      const int c = k / TR::CC, kk = k - c * TR::CC;
      const float v = p.col[(size_t)S.id[g] * TR::NCOL + k];
      S.col[g * TR::NCOLP + c * TR::CCP + kk] = SCALE ? -kLog2e * v : v;
    }
  } else {
    if (t < nb) {
      const int id = S.id[t];  // own record: written by this same thread above
      load_channels<MODE, TR::NCOL>(p, id, &S.col[t * TR::NCOL]);
    }
  }
}

// the raw covariance of staged record g for the threshold guard's reference arithmetic: staged in LDS, or -- launches that stage the
// projection's prepared records (CompParams::chol) -- from memory (a handful of pixels per frame take this path)
template <int MODE, int CB, int KB, bool WC>
__device__ __forceinline__ void guard_cov(const Stage<MODE, CB, KB, WC> &S, const CompParams &p, int g, float &c0, float &c1,
                                          float &c2, float &c3) {
  if (MODE != MODE_SH && p.chol != nullptr) {
    const float4 c = *reinterpret_cast<const float4 *>(p.cov + 4 * (size_t)S.id[g]);
    c0 = c.x; c1 = c.y; c2 = c.z; c3 = c.w;
  } else {
    c0 = S.c0[g]; c1 = S.c1[g]; c2 = S.c2[g]; c3 = S.c3[g];
  }
}
// per-Gaussian values broadcast from LDS into registers
template <int MODE, int CB, int KB, bool WC>
__device__ __forceinline__ GRec load_rec(const Stage<MODE, CB, KB, WC> &S, int g) {
  GRec r;
  r.mx = S.mx[g]; r.my = S.my[g]; r.a = S.a[g];
  r.c0 = S.c0[g]; r.c1 = S.c1[g]; r.c2 = S.c2[g]; r.c3 = S.c3[g];
  r.p0 = S.p0[g]; r.p1 = S.p1[g]; r.p2 = S.p2[g];
  return r;
}

// ============================================================================================
// forward
// ============================================================================================
// "a tile crowded with splats beyond the bound was seen" into the caller's host-visible word (pinned, device-mapped: a system-scope store)
__device__ __forceinline__ void report_crowded_tile(uint32_t *word) {
#if defined(__HIP_DEVICE_COMPILE__)
  __hip_atomic_store(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#else
  *word = 1u;
#endif
}
// A tile of a frame whose (tile, Gaussian) pairs did not fit the caller's list (start == kListOverflow, binning.hip): NaN in
// every channel and in T.  A blank-but-finite view would train on silently; this one kills the loss.
template <int NCH>
__device__ __forceinline__ void poison_pixel(const CompParams &p, size_t pix) {
  const float nan = __builtin_nanf("");
  if (NCH == 6 && p.pl_rgb != nullptr) {  // (the heads as separate images: gsgen_rgbd_view::out_rgb ...)
    p.pl_rgb[3 * pix] = nan; p.pl_rgb[3 * pix + 1] = nan; p.pl_rgb[3 * pix + 2] = nan;
    p.pl_d[pix] = nan; p.pl_o[pix] = nan; p.pl_z[pix] = nan;
  } else {
#pragma unroll
    for (int c = 0; c < NCH; ++c) p.out[NCH * pix + c] = nan;
  }
  if (p.T != nullptr) p.T[pix] = nan;
}
// RGB + heads: the epilogue of a pixel of the batched forward -- the background behind what is left of the transmittance
// (gs/renderer.py:1182: out + T * bg, product rounded before the sum as torch forms it) and, where the caller asks for it, the depth
// variance in place of the second moment (gs/gaussian_splatting.py:1397: z_var = depth2 - depth * depth)
__device__ __forceinline__ void heads_epilogue(const CompParams &p, const float (&bg)[3], float (&e)[6], float T) {
#pragma clang fp contract(off)
  if (p.bg != nullptr) {  // (bg: the three floats, loaded ONCE at the top of the kernel -- read here, behind the image stores of the
    // lane's previous pixel, the compiler must assume they alias and fetch them again per pixel: a round trip to memory each)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float tb = T * bg[c];
      e[c] = e[c] + tb;
    }
  }
  if (p.zvar) {
    const float dd = e[3] * e[3];
    e[5] = e[5] - dd;
  }
}
// the tile's three background-gradient sums into row (tile % 64) of the view's partial block: one 4-component reduce-scatter on the
// vector ALU's cross-lane instructions (three ds_bpermute butterflies cost 18 LDS round trips in front of every tile's list walk)
__device__ __forceinline__ void bg_grad_atomics(const CompParams &p, int tile, int lane, const float (&sums)[3]) {
  float v[4] = {sums[0], sums[1], sums[2], 0.0f};
  wave_reduce_scatter<4>(v);
  const int comp = scatter_comp<4>(lane);
  if (scatter_owner<4>(lane) && comp < 3) atomicAdd(p.g_bg + 4 * (tile & 63) + comp, v[0]);
}
// one pixel's channels to the image(s): [H,W,NCH] interleaved, or -- RGB + heads with separate images -- rgb [H,W,3] + three [H,W]
template <int MODE, int NCH>
__device__ __forceinline__ void store_heads(const CompParams &p, size_t pix, const float (&e)[NCH]) {
  if constexpr (MODE == MODE_RGBD) {
    if (p.pl_rgb != nullptr) {  // (uniform over the launch)
      p.pl_rgb[3 * pix] = e[0]; p.pl_rgb[3 * pix + 1] = e[1]; p.pl_rgb[3 * pix + 2] = e[2];
      p.pl_d[pix] = e[3]; p.pl_o[pix] = e[4]; p.pl_z[pix] = e[5];
      return;
    }
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) p.out[NCH * pix + c] = e[c];
}
// The parameter blocks of a batched launch travel in the kernel arguments themselves (<= kPackMax views per launch; larger
// batches are launched in chunks): no table in device memory, no launch that writes one (until round 4 a one-workgroup
// k_write_params launch in front of every batched forward and backward).
constexpr int kPackMax = 8;
template <bool BATCH> struct ViewPack {
  CompParams v[kPackMax];
  __host__ __device__ const CompParams *table() const { return v; }
};
template <> struct ViewPack<false> {
  __host__ __device__ const CompParams *table() const { return nullptr; }
};
// BATCH: one launch renders B cameras, each workgroup taking its camera's parameters from plist[view]
// (the kernel-argument table) instead of p_arg -- a single launch's tail (the never-saturating
// sparse tiles) is then paid once per batch instead of once per camera.  The grid is 1-D, B x the
// per-camera grid, camera-major (all blocks of camera 0, then camera 1, ...; measured against two interleaved orders in
// round 1: as fast on cfg2, faster on the 64 random cameras of cfg4 -- an interleaved B = 8 pins each camera to one XCD);
// p_arg only carries B (n_lo).
// total: the size of the grid the map is taken over -- gridDim.x, or the virtual grid a persistent launch strides through
__device__ __forceinline__ uint32_t batch_view(const CompParams &p_arg, uint32_t &bid, uint32_t *grid, uint32_t total) {
  const uint32_t B = (uint32_t)p_arg.n_lo, per = total / B;
  const uint32_t view = bid / per;
  bid -= view * per;
  if (grid) *grid = per;
  return view;
}
__device__ __forceinline__ uint32_t batch_view(const CompParams &p_arg, uint32_t &bid, uint32_t *grid = nullptr) {
  return batch_view(p_arg, bid, grid, gridDim.x);
}

// TS: tile side.  16 everywhere in this library's own pipeline; 8 and 32 (= every other side from 1 to 32) exist for callers of the `_gs` entry points
// that configure another `tile_size` (the reference takes it as a parameter, conf/base.yaml:132; its launch is
// tile_size x tile_size threads, vol_render.h:1001-1004): TS = 8 runs one wavefront per tile (PPL = 1), TS = 32 four
// wavefronts at 4 pixels per lane.  Per-pixel results do not depend on the tile size.
template <int MODE, int CB, int PPL, int TS = 16>
__global__ void __launch_bounds__(TS * TS / PPL) k_composite_fwd(CompParams p) {
  // by value: read once with scalar loads; through a reference every use in the entry loop would be
  // re-read from memory (the kernel's own stores may alias it as far as the compiler knows)
  const uint32_t bid = blockIdx.x;
  using TR = Traits<MODE, CB>;
  constexpr int NT = TS * TS / PPL;
  constexpr int ROWS = NT / TS;
  static_assert(TS == 8 || TS == 16 || TS == 32, "tile side");
  static_assert(NT >= kBatch && NT % 64 == 0, "a staging round needs one thread per record");
  constexpr int NCH = TR::NCH;
  __shared__ Stage<MODE, CB> S;

  int tx, ty;
  if (!block_tile(p, tx, ty, bid)) return;  // uniform over the workgroup
  const int tile = ty * p.ntw + tx;
  const int st = p.start[tile];
  const int n = (st < 0) ? 0 : (p.end[tile] - st);
  const int t = (int)threadIdx.x;
  const int lx = t % TS, ly0 = t / TS;
  // TS = 32 serves every caller tile side that is not 8 or 16 (1 .. 32): the workgroup covers a 32 x 32 patch of which the
  // caller's side x side tile is the top-left part; the threads beyond it own no pixel
  const int side = (TS == 32) ? p.tile_side : TS;
  const int gx = tx * side + lx;

  bool valid[PPL];
  int gy[PPL];
#pragma unroll
  for (int j = 0; j < PPL; ++j) {
    gy[j] = ty * side + ly0 + j * ROWS;
    valid[j] = (gx < p.W) && (gy[j] < p.H) && (TS != 32 || (lx < side && ly0 + j * ROWS < side));
  }

  if (st == kListOverflow) {  // the frame's pairs did not fit the list (GSGEN_LIST_OVERFLOW): never a finite blank tile
#pragma unroll
    for (int j = 0; j < PPL; ++j)
      if (valid[j]) poison_pixel<NCH>(p, (size_t)gy[j] * p.W + gx);
    return;
  }
  if (n == 0) {  // uniform over the workgroup
    if constexpr (MODE == MODE_SH) {
      if (p.bg != nullptr || p.fill_empty) {  // vol_render_bg.h:34-53: empty tiles show the background
#pragma unroll
        for (int j = 0; j < PPL; ++j)
          if (valid[j]) {
            const size_t pix = (size_t)gy[j] * p.W + gx;
            float *o = p.out + 3 * pix;
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c] = p.bg != nullptr ? p.bg[c] : 0.0f;
            if (p.fill_empty && p.T != nullptr) p.T[pix] = 1.0f;  // batched launches write the whole image (CompParams::fill_empty)
          }
      }
    }
    return;  // otherwise the caller's pre-initialised out / T stand (vol_render.h:1006-1013)
  }

  const float px = pixel_coord(p.topleft[0], gx, p.psx);
  float py[PPL];
#pragma unroll
  for (int j = 0; j < PPL; ++j) py[j] = pixel_coord(p.topleft[1], gy[j], p.psy);

  // per-pixel SH basis (vol_render_sh.h:48-65, 210-216), kept as (k, k+1) pairs
  v2f Yp[MODE == MODE_SH ? PPL : 1][MODE == MODE_SH ? TR::NPAIR : 1];
  if constexpr (MODE == MODE_SH) {
    if constexpr (TR::CCP != TR::CC) {  // zero the pad lanes of the staged coefficients once
      for (int e = t; e < kBatch * TR::NCOLP; e += NT) S.col[e] = 0.0f;
      __syncthreads();
    }
    float R[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = p.rot[i];
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      float Yf[TR::CCP];
      sh_basis_of_pixel<CB>(R, px, py[j], Yf);
#pragma unroll
      for (int k = 0; k < TR::NPAIR; ++k) Yp[j][k] = v2f{Yf[2 * k], Yf[2 * k + 1]};
      __builtin_amdgcn_sched_barrier(0);  // one pixel's temporaries at a time
    }
  }

  float acc[PPL][NCH];
  float Tr[PPL];
  bool alive[PPL];
  int stop[PPL];  // first list index the pixel does not process (segmented backward)
#pragma unroll
  for (int j = 0; j < PPL; ++j) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[j][c] = 0.0f;
    Tr[j] = 1.0f;
    alive[j] = valid[j];
    stop[j] = valid[j] ? n : 0;
  }
  const bool seg_out = (MODE == MODE_SH) && p.nseg > 1 && p.ckpt != nullptr;

  for (int base = 0; base < n; base += kBatch) {
    const int nb = min(kBatch, n - base);
    if (base > 0) __syncthreads();  // everyone is done with the previous batch
    stage_batch<MODE, CB, NT, kBatch, MODE == MODE_SH>(S, p, st + base, nb);  // SH: coefficients x -log2(e)
    __syncthreads();

    for (int g = 0; g < nb; ++g) {
      bool any_alive = false;
#pragma unroll
      for (int j = 0; j < PPL; ++j) any_alive |= alive[j];
      if (!wave_any(any_alive)) break;  // this wave's 64*PPL pixels are saturated
      if constexpr (MODE == MODE_SH) {
        const int e = base + g;
        if (seg_out && (e % kSegLen) == 0 && e > 0 && e / kSegLen < p.nseg) {  // wave-uniform
#pragma unroll
          for (int j = 0; j < PPL; ++j)
            p.ckpt[((size_t)tile * p.nseg + e / kSegLen) * (TS * TS) + (ly0 + j * ROWS) * TS + lx] =
                make_float4(Tr[j], acc[j][0], acc[j][1], acc[j][2]);
        }
      }

      const GRec r = load_rec(S, g);
      const float x = px - r.mx;
      float G[PPL];
      bool con[PPL];
      bool any_con = false;
#pragma unroll
      for (int j = 0; j < PPL; ++j) {
        G[j] = gauss_eval<MODE>(r, x, py[j] - r.my, px, py[j], alive[j]);
        con[j] = alive[j] && !(r.a * G[j] < kMinAlpha);
        any_con |= con[j];
      }
      if (!wave_any(any_con)) continue;  // nobody in the wave sees this Gaussian

      const float *cg = &S.col[g * TR::NCOLP];
      if constexpr (MODE == MODE_SH) {
        // Every lane evaluates all of its pixels; a pixel that does not contribute carries
        // weight 0.  76 % of the evaluated pairs contribute on the headline workload, so the
        // branch-free form wastes little and lets the 3 x CC dot products run as packed FMAs.
        float w[PPL];
#pragma unroll
        for (int j = 0; j < PPL; ++j) w[j] = con[j] ? (r.a * Tr[j]) * G[j] : 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          v2f q[TR::NPAIR];
#pragma unroll
          for (int k = 0; k < TR::NPAIR; ++k) q[k] = *reinterpret_cast<const v2f *>(cg + c * TR::CCP + 2 * k);
#pragma unroll
          for (int j = 0; j < PPL; ++j) {
            // the colour arithmetic of k_composite_fwd_sh_vec, operation for operation (pre-scaled coefficients,
            // explicit fused accumulate): a camera renders to the same bits through either kernel
            v2f s2 = q[0] * Yp[j][0];
#pragma unroll
            for (int k = 1; k < TR::NPAIR; ++k) s2 = ffma2(q[k], Yp[j][k], s2);
            const float yv = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(s2[0] + s2[1]));
            acc[j][c] = ffma(w[j], yv, acc[j][c]);
          }
        }
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
          // explicit: 1 - round(a G) in every kernel (left to the compiler, `1 - a*G` becomes a fused -a*G + 1 in
          // some instantiations and not in others, and forward and backward would disagree on T in the last bit)
          const float om = ffma(-(r.a * G[j]), con[j] ? 1.0f : 0.0f, 1.0f);
          Tr[j] *= om;
          const bool still = alive[j] && !(Tr[j] < p.thresh);
          if (alive[j] && !still) stop[j] = base + g + 1;
          alive[j] = still;
        }
      } else {
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
          if (con[j]) {
            const float ag = r.a * G[j];
            const float coeff = (r.a * Tr[j]) * G[j];
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc[j][c] = ffma(cg[c], coeff, acc[j][c]);  // explicit: as k_composite_fwd_chan_vec
            Tr[j] *= ffma(-ag, 1.0f, 1.0f);  // 1 - round(a G), never a fused -a*G + 1 (see the SH branch)
            alive[j] = !(Tr[j] < p.thresh);
          }
        }
      }
    }
    bool any_alive = false;
#pragma unroll
    for (int j = 0; j < PPL; ++j) any_alive |= alive[j];
    if (__syncthreads_or((int)any_alive) == 0) break;  // whole tile saturated: stop staging
  }

#pragma unroll
  for (int j = 0; j < PPL; ++j) {
    if (!valid[j]) continue;
    const size_t pix = (size_t)gy[j] * p.W + gx;
    if constexpr (MODE == MODE_SCALAR) {
      p.out[pix] = acc[j][0];
    } else if constexpr (MODE == MODE_RGBD) {
      acc[j][4] = 1.0f - Tr[j];  // opacity = sum w = 1 - T, as k_composite_fwd_chan_vec forms it (the same T, hence the same bits)
#pragma unroll
      for (int c = 0; c < NCH; ++c) p.out[NCH * pix + c] = acc[j][c];
    } else {
      float *o = p.out + 3 * pix;
      if (p.bg != nullptr) {  // vol_render_bg.h:95-100
        o[0] = acc[j][0] + p.bg[0] * Tr[j];
        o[1] = acc[j][1] + p.bg[1] * Tr[j];
        o[2] = acc[j][2] + p.bg[2] * Tr[j];
      } else {
        o[0] = acc[j][0]; o[1] = acc[j][1]; o[2] = acc[j][2];
      }
    }
    if (p.T != nullptr) p.T[pix] = Tr[j];
  }
  if constexpr (MODE == MODE_SH) {
    if (p.stop != nullptr) {
#pragma unroll
      for (int j = 0; j < PPL; ++j) p.stop[(size_t)tile * (TS * TS) + (ly0 + j * ROWS) * TS + lx] = stop[j];
    }
  }
}

// ============================================================================================
// forward, SH, packed per-pixel arithmetic (2 or 4 pixels per lane)
// ============================================================================================
// k_composite_fwd<MODE_SH> with the lane's pixels taken two at a time (v2f): Gaussian, weights, colour sums and the
// transmittance update are packed fp32 instructions (full rate on gfx950, see k_composite_bwd_sh_vec), the
// coefficients are staged pre-scaled by -log2(e), and the two pixels' dot products are interleaved.  Decisions (skip,
// saturated) come from gauss_sh_pair / the explicit T update, bit for bit those of every other kernel.  Same launch
// shape and outputs (image, T, segment checkpoints and stop indices) as k_composite_fwd.
// NB > 0 (= kPolyNB, SH degree 3 only): the tile-local polynomial form of the per-pixel basis, see composite_common.hpp --
// the same kernel with 6-term contractions against coefficients transformed once per (tile, splat).
// NB = kRouted (SH degree 3, launches that were given the coefficient bound): ONE launch holds both forms; every workgroup
// reads the bound and its view's pixel size (poly_route: two scalar loads, uniform over the workgroup) and runs the
// polynomial form where the error bound holds, the exact form elsewhere.  The two forms share one block of LDS (a union) and
// the register budget is the larger of the two -- the same occupancy class as either alone.
constexpr int kRouted = -1;
// PERSIST (the tile bodies below, called from a loop over tiles): the thread index is re-read through an opaque statement in
// every iteration, so that the lane's per-tile constants are NOT hoisted out of the loop and kept in registers across it (the
// persistent backward came out at 194 registers -- two wavefronts per SIMD -- with them hoisted)
template <bool PERSIST>
__device__ __forceinline__ int tile_thread_index() {
  int t = (int)threadIdx.x;
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (PERSIST) asm volatile("" : "+v"(t));
#endif
  return t;
}
// ---- the per-entry exact tier of the polynomial kernels ---------------------------------------------------------------------
// A splat beyond the coefficient bound of this view's pixel size (poly_row_ok fails) inside a tile that is otherwise the polynomial
// kernel's: its three logits are evaluated EXACTLY at the lane's pixels -- the pixel's SH basis (the table of sh_basis) against the
// raw coefficients, read through a wave-uniform address -- and enter the same denominators 1 + exp2(s) the polynomial values
// would have.  Wave-uniform branch per entry (a bit of the staged batch's mask: scalar instructions only for every other entry),
// about three times an ordinary entry's instructions; a batch with more than a quarter of such splats sends the tile to the exact
// kernel instead (kExactTierShare).  The backward takes the splat's gradient through the tile's polynomial basis like any other
// entry's: that error is the basis's own (<= 0.175 delta^3 per unit of d L / d s: 5e-7 at the headline pixel size), whatever the
// coefficients.
constexpr int kExactTierShare = 4;
struct Logits3 { float s0, s1, s2; };
#if defined(__HIP_DEVICE_COMPILE__)
typedef const float __attribute__((address_space(4))) *uniform_floats;  // (constant address space + a scalar address: s_load)
__device__ __forceinline__ uniform_floats uniform_pointer(const float *q) {
  const uintptr_t u = (uintptr_t)q;
  return (uniform_floats)(((uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                          (uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u));
}
__device__ __forceinline__ float rsqrt_fast(float v) { return __builtin_amdgcn_rsqf(v); }
#else
typedef const float *uniform_floats;  // (the host pass of hipcc and the CPU emulator build)
__device__ __forceinline__ uniform_floats uniform_pointer(const float *q) { return q; }
__device__ __forceinline__ float rsqrt_fast(float v) { return 1.0f / sqrtf(v); }
#endif
// One pixel's three logits -log2(e) * (sh_c . Y(pixel)), |s| <= 40 (as poly_transform keeps the polynomial's: the product of three
// denominators stays finite).  NOT inlined, and written to stay small: the polynomial kernels' register budgets are their entry
// loops' (inlined with a basis table, the backward went from 114 to 170 registers).  Both addresses are wave-uniform: scalar
// loads; every basis value goes straight into the three dot products.
__device__ __attribute__((noinline)) Logits3 exact_tier_logits(const float *rot_, const float *q_, float qx, float qy) {
  const uniform_floats rot = uniform_pointer(rot_), q = uniform_pointer(q_);
  float x = rot[0] * qx + rot[1] * qy + rot[2];
  float y = rot[3] * qx + rot[4] * qy + rot[5];
  float z = rot[6] * qx + rot[7] * qy + rot[8];
  const float rl = rsqrt_fast(x * x + y * y + z * z);
  x *= rl; y *= rl; z *= rl;
  float a0 = 0.28209479177387814f * q[0], a1 = 0.28209479177387814f * q[16], a2 = 0.28209479177387814f * q[32];
  auto add = [&](int k, float v) { a0 = fmaf(v, q[k], a0); a1 = fmaf(v, q[16 + k], a1); a2 = fmaf(v, q[32 + k], a2); };
  add(1, -0.48860251190291987f * y);
  add(2, 0.48860251190291987f * z);
  add(3, -0.48860251190291987f * x);
  __builtin_amdgcn_sched_barrier(0);  // (a few coefficients in flight at a time: scalar registers are the caller's parameters)
  const float x2 = x * x, y2 = y * y, z2 = z * z;
  add(4, 1.0925484305920792f * (x * y));
  add(5, -1.0925484305920792f * (y * z));
  add(6, 0.94617469575755997f * z2 - 0.31539156525251999f);
  add(7, -1.0925484305920792f * (x * z));
  add(8, 0.54627421529603959f * x2 - 0.54627421529603959f * y2);
  __builtin_amdgcn_sched_barrier(0);
  add(9, 0.59004358992664352f * y * (-3.0f * x2 + y2));
  add(10, 2.8906114426405538f * (x * y) * z);
  add(11, 0.45704579946446572f * y * (1.0f - 5.0f * z2));
  add(12, 0.3731763325901154f * z * (5.0f * z2 - 3.0f));
  __builtin_amdgcn_sched_barrier(0);
  add(13, 0.45704579946446572f * x * (1.0f - 5.0f * z2));
  add(14, 1.4453057213202769f * z * (x2 - y2));
  add(15, 0.59004358992664352f * x * (-x2 + 3.0f * y2));
  // (a NaN coefficient stays a NaN in the image, as in the exact kernels and the reference: fminf / fmaxf would drop it)
  auto fin = [](float a) { return a == a ? fminf(fmaxf(-kLog2e * a, -40.0f), 40.0f) : a; };
  return Logits3{fin(a0), fin(a1), fin(a2)};
}
template <int PPL>
__device__ __forceinline__ void exact_tier_logits_all(const CompParams &p, int id_uniform, float px, const v2f (&py2)[PPL / 2],
                                                      v2f (&lg)[3][PPL / 2]) {
  const float *q = p.col + (size_t)id_uniform * 48u;
#pragma unroll
  for (int j = 0; j < PPL; ++j) {
    const Logits3 l = exact_tier_logits(p.rot, q, px, py2[j >> 1][j & 1]);
    lg[0][j >> 1][j & 1] = l.s0;
    lg[1][j >> 1][j & 1] = l.s1;
    lg[2][j >> 1][j & 1] = l.s2;
  }
}
// A staged polynomial row at the lane's pixel pairs: five packed FMAs per pair and channel, explicitly fused (every shape of the
// kernel agrees).  The Taylor tier's rows deliver the colour, all others the logit (composite_common.hpp).
template <int NP>
__device__ __forceinline__ void poly_rows_at_pixels(const float *cg, v2f pu2, const v2f (&pv2)[NP], v2f (&yv)[3][NP]) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float *cw = cg + c * kPolyStride;  // (w0, w1 | w2, w4 | w5, 0 | w3, -)
    const v2f ab = ffma2(pu2, ffma2(pu2, *reinterpret_cast<const v2f *>(cw + 4), *reinterpret_cast<const v2f *>(cw + 2)),
                         *reinterpret_cast<const v2f *>(cw));  // w0 + u (w2 + u w5) | w1 + u w4
    const float Cc = cw[6];
#pragma unroll
    for (int jp = 0; jp < NP; ++jp) yv[c][jp] = ffma2(pv2[jp], ffma2(pv2[jp], splat2(Cc), splat2(ab[1])), splat2(ab[0]));
  }
}
// the staged batch's mask of such splats (bit g = entry g of the batch; KB <= 32).  NT threads, the first nb of them hold an entry.
template <int NT>
__device__ __forceinline__ uint32_t exact_tier_mask(const CompParams &p, const int *ids, int nb, int t, uint32_t *slot) {
  bool bad = false;
  if (p.sh_rows != nullptr && t < nb) bad = !poly_row_ok(p.sh_rows[ids[t]], fmaxf(fabsf(p.psx), fabsf(p.psy)));
  const uint32_t m = (uint32_t)__ballot((int)bad);
  if constexpr (NT == 64) {
    (void)slot;
    return m;
  } else {  // the entries sit in the first wavefront: through LDS to the others (the caller's barrier follows)
    if (t == 0) *slot = m;
    return 0u;
  }
}
template <int CB, bool POLY>
struct FwdShVecShared {
  // POLY: 32 records per round -- 4.8 KB of LDS per workgroup, so that 20 one-wavefront workgroups (5 per SIMD) fit a CU
  static constexpr int KB = POLY ? 32 : kBatch;
  Stage<MODE_SH, CB, KB, !POLY> S;
  alignas(16) float Vs[POLY ? kPolyNB * 16 : 4];          // POLY: V of this tile
  alignas(16) float Ws[POLY ? KB * 3 * kPolyStride : 4];  // POLY: transformed coefficients of the staged batch
                                                          // (and, before the first batch, the nine node bases)
  uint32_t exact_mask;                                    // POLY, two wavefronts per tile: exact_tier_mask of the staged batch
  float tay_ok[POLY ? KB * 3 : 1];                        // POLY: rows of the staged batch that may take the Taylor tier (poly_transform)
};
// TRACK = false: a launch known (on the host) to carry no stop list -- four registers of running state less
template <int CB, int PPL, int NB, bool PERSIST = false, bool TRACK = true>
__device__ __forceinline__ void composite_fwd_sh_vec_tile(const CompParams &p, uint32_t bid, FwdShVecShared<CB, (NB > 0)> &sm) {
  static_assert(PPL == 4 || PPL == 2, "pixel pairs: 2 or 4 pixels per lane");
  constexpr bool POLY = NB > 0;
  static_assert(!POLY || (NB == kPolyNB && CB == 4), "polynomial basis: SH degree 3");
  constexpr int MODE = MODE_SH;
  using TR = Traits<MODE, CB>;
  constexpr int NT = 256 / PPL, ROWS = NT / 16, NP = PPL / 2;
  constexpr int CCP = POLY ? NB : TR::CCP, NPAIR = CCP / 2;
  constexpr int KB = FwdShVecShared<CB, POLY>::KB;
  auto &S = sm.S;
  float *const Vs = sm.Vs;
  float *const Ws = sm.Ws;
  int tx, ty;
  if (!block_tile(p, tx, ty, bid)) return;  // uniform over the workgroup
  const int tile = ty * p.ntw + tx;
  const int st = p.start[tile];
  const int n = (st < 0) ? 0 : (p.end[tile] - st);
  const int t = tile_thread_index<PERSIST>();
  const int lx = t & 15, ly0 = t >> 4;
  const int gx = tx * kTile + lx;

  bool valid[PPL];
  int gy[PPL], stop[PPL];
#pragma unroll
  for (int j = 0; j < PPL; ++j) {
    gy[j] = ty * kTile + ly0 + j * ROWS;
    valid[j] = (gx < p.W) && (gy[j] < p.H);
    stop[j] = valid[j] ? n : 0;
  }
  if constexpr (POLY) {  // (per-tile routing: this tile is the polynomial kernel's until a staged batch says otherwise)
    if (t == 0 && p.tile_flags != nullptr) p.tile_flags[tile] = 0;
  }
  // the background colour in three registers, loaded before any store of this tile (round 6): read in the epilogue, behind the image
  // stores of the lane's previous pixel, the compiler has to assume they alias and fetches them again -- twelve dependent round trips
  // to memory at the end of every tile
  const bool has_bg = p.bg != nullptr;
  float bgv[3] = {0.0f, 0.0f, 0.0f};
  if (has_bg) { bgv[0] = p.bg[0]; bgv[1] = p.bg[1]; bgv[2] = p.bg[2]; }
  if (st == kListOverflow) {  // the frame's pairs did not fit the list (GSGEN_LIST_OVERFLOW): never a finite blank tile
#pragma unroll
    for (int j = 0; j < PPL; ++j)
      if (valid[j]) poison_pixel<3>(p, (size_t)gy[j] * p.W + gx);
    return;
  }
  if (n == 0) {  // uniform over the workgroup
    if (p.bg != nullptr || p.fill_empty) {  // vol_render_bg.h:34-53: empty tiles show the background
#pragma unroll
      for (int j = 0; j < PPL; ++j)
        if (valid[j]) {
          const size_t pix = (size_t)gy[j] * p.W + gx;
          float *o = p.out + 3 * pix;
#pragma unroll
          for (int c = 0; c < 3; ++c) o[c] = bgv[c];
          if (p.fill_empty && p.T != nullptr) p.T[pix] = 1.0f;  // batched launches write the whole image (CompParams::fill_empty)
        }
    }
    return;  // otherwise the caller's pre-initialised out / T stand (vol_render.h:1006-1013)
  }
  if constexpr (TR::CCP != TR::CC) {  // zero the pad lanes of the staged coefficients once
    for (int e = t; e < KB * TR::NCOLP; e += NT) S.col[e] = 0.0f;
    __syncthreads();
  }

  const float px = pixel_coord(p.topleft[0], gx, p.psx);
  v2f py2[NP];
#pragma unroll
  for (int j = 0; j < PPL; ++j) py2[j >> 1][j & 1] = pixel_coord(p.topleft[1], gy[j], p.psy);
  v2f Yp[POLY ? 1 : PPL][POLY ? 1 : NPAIR];
  // POLY: the lane's column offset and its pixel pairs' row offsets -- the six monomials (1, v, u, v^2, uv, u^2) are never
  // materialised: s = (w0 + u (w2 + u w5)) + v ((w1 + u w4) + v w3), the bracketed terms once per lane and channel
  const float pu = poly_offset(lx);
  const v2f pu2 = splat2(pu);
  v2f pv2[NP];
#pragma unroll
  for (int jp = 0; jp < NP; ++jp) pv2[jp] = v2f{poly_offset(ly0 + (2 * jp) * ROWS), poly_offset(ly0 + (2 * jp + 1) * ROWS)};
  if constexpr (POLY) {
    static_assert(KB * 3 * kPolyNB >= kPolyNodes * 16, "node scratch fits the coefficient buffer");
    poly_tile_setup<NT>(p, tx, ty, Ws, Vs);
  } else {
    float R[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = p.rot[i];
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      const float pyj = py2[j >> 1][j & 1];
      float Yf[CCP];
      sh_basis_of_pixel<CB>(R, px, pyj, Yf);
#pragma unroll
      for (int k = 0; k < NPAIR; ++k) Yp[j][k] = v2f{Yf[2 * k], Yf[2 * k + 1]};
      __builtin_amdgcn_sched_barrier(0);  // one pixel's temporaries at a time
    }
  }

  v2f acc2[NP][3], Tr2[NP];
#pragma unroll
  for (int jp = 0; jp < NP; ++jp) {
#pragma unroll
    for (int c = 0; c < 3; ++c) acc2[jp][c] = v2f{0.0f, 0.0f};
    // a pixel is alive while T >= thresh (T only decreases); pixels outside the image start at -1: never alive,
    // never written
    Tr2[jp] = v2f{valid[2 * jp] ? 1.0f : -1.0f, valid[2 * jp + 1] ? 1.0f : -1.0f};
  }
  auto alive = [&](int j) { return !(Tr2[j >> 1][j & 1] < p.thresh); };
  const bool seg_out = p.nseg > 1 && p.ckpt != nullptr;
  const bool track_stop = TRACK && p.stop != nullptr;

  uint32_t exact_mask = 0u;  // POLY: the staged batch's splats of the per-entry exact tier
  uint32_t tay_mask = 0u;    // POLY: ... and those of the Taylor tier (no exponential per pixel: composite_common.hpp)
  for (int base = 0; base < n; base += KB) {
    const int nb = min(KB, n - base);
    if (base > 0) __syncthreads();  // everyone is done with the previous batch
    stage_batch<MODE, CB, NT, KB, true, !POLY>(S, p, st + base, nb);
    if constexpr (POLY) {
      // per-tile routing: splats beyond the bound for this view's pixel size take the per-entry exact tier
      // (exact_tier_logits_all); more than a quarter of a staged batch sends the WHOLE tile to the exact kernel (nothing has
      // been written yet; CompParams::tile_flags).
      if (p.sh_rows != nullptr) {  // (uniform over the launch; known to be false where the caller proved every splat within the bound)
        exact_mask = exact_tier_mask<NT>(p, S.id, nb, t, &sm.exact_mask);
        __syncthreads();
        if constexpr (NT != 64) exact_mask = sm.exact_mask;
        exact_mask = (uint32_t)__builtin_amdgcn_readfirstlane((int)exact_mask);
        if (__popcll((unsigned long long)exact_mask) * kExactTierShare > nb) {  // (uniform over the workgroup)
          if (t == 0 && p.route_report != nullptr) report_crowded_tile(p.route_report);  // (the host's hint: CompParams::route_report)
          if (!p.no_fallback) {
            if (t == 0 && p.tile_flags != nullptr) p.tile_flags[tile] = 1;
            return;
          }
          // no_fallback: no exact kernel runs behind this one -- the tile stays, its splats beyond the bound through the per-entry tier
        }
      } else {
        exact_mask = 0u;
        __syncthreads();
      }
      poly_transform<NT, KB>(p.col, S.id, Vs, Ws, nb, sm.tay_ok);
      __syncthreads();
      tay_mask = (uint32_t)__builtin_amdgcn_readfirstlane((int)(taylor_mask(sm.tay_ok, nb) & ~exact_mask));
      taylor_convert<NT>(Ws, tay_mask, nb);
      __syncthreads();
    } else {
      __syncthreads();
    }

    for (int g = 0; g < nb; ++g) {
      bool any_alive = false;
#pragma unroll
      for (int j = 0; j < PPL; ++j) any_alive |= alive(j);
      if (!wave_any(any_alive)) break;  // this wave's 64*PPL pixels are saturated
      const int e_idx = base + g;
      if (seg_out && (e_idx % kSegLen) == 0 && e_idx > 0 && e_idx / kSegLen < p.nseg) {  // wave-uniform
#pragma unroll
        for (int j = 0; j < PPL; ++j)
          p.ckpt[((size_t)tile * p.nseg + e_idx / kSegLen) * 256 + (ly0 + j * ROWS) * 16 + lx] =
              make_float4(Tr2[j >> 1][j & 1], acc2[j >> 1][0][j & 1], acc2[j >> 1][1][j & 1], acc2[j >> 1][2][j & 1]);
      }

      const float r_mx = S.mx[g], r_my = S.my[g], r_a = S.a[g], r_c0 = S.c0[g], r_c1 = S.c1[g], r_c2 = S.c2[g],
                  r_c3 = S.c3[g], r_p0 = S.p0[g];
      const float x = px - r_mx;
      // G2 / ag2: the Gaussian and a G, ZEROED where the pixel does not take part (skip threshold, or not alive)
      v2f G2[NP], ag2[NP];
      bool any_con = false;
      float guard_dist = 0.0f;
#pragma unroll
      for (int jp = 0; jp < NP; ++jp) {
        G2[jp] = gauss_sh_pair(r_c0, r_c1, r_c2, r_c3, r_p0, x, py2[jp] - splat2(r_my));
        ag2[jp] = splat2(r_a) * G2[jp];
        {
          // the lane's smallest distance to the threshold (dead pixels included: a spurious trip re-tests per pixel)
          const v2f dist = ag2[jp] - splat2(kMinAlpha);
          const float dmin = fminf(fabsf(dist[0]), fabsf(dist[1]));
          guard_dist = jp == 0 ? dmin : fminf(guard_dist, dmin);
        }
      }
      const bool any_guard = guard_dist <= kMinAlpha * kGuardTol;
      // within rounding of the skip threshold the reference's arithmetic decides (as gauss_eval); one wave-uniform
      // test for all the lane's pixels, almost never taken
      if (wave_any(any_guard)) {
#pragma unroll
        for (int jp = 0; jp < NP; ++jp)
#pragma unroll
          for (int e = 0; e < 2; ++e)
            if (alive(2 * jp + e) && fabsf(ag2[jp][e] - kMinAlpha) <= kMinAlpha * kGuardTol) {
              G2[jp][e] = gauss_ref_f32(r_mx, r_my, r_c0, r_c1, r_c2, r_c3, px, py2[jp][e]);
              ag2[jp][e] = r_a * G2[jp][e];
            }
      }
#pragma unroll
      for (int jp = 0; jp < NP; ++jp)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const bool con = alive(2 * jp + e) && !(ag2[jp][e] < kMinAlpha);
          G2[jp][e] = con ? G2[jp][e] : 0.0f;
          if constexpr (!POLY) ag2[jp][e] = con ? ag2[jp][e] : 0.0f;
          any_con |= con;
        }
      if constexpr (POLY) {  // the same products again, from the masked G: two packed multiplies instead of four selects
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) {
          ag2[jp] = splat2(r_a) * G2[jp];
        }
      }
      if (!wave_any(any_con)) continue;  // nobody in the wave sees this Gaussian

      const float *cg = POLY ? &Ws[g * 3 * kPolyStride] : &S.col[g * TR::NCOLP];
      v2f w2[NP];
      if constexpr (!POLY) {
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) w2[jp] = (splat2(r_a) * Tr2[jp]) * G2[jp];  // (a T) G, or 0
      }
      if constexpr (POLY) {
        // the staged rows at the lane's pixels -- or, for the staged batch's few splats of the exact tier (a wave-uniform bit:
        // scalar instructions only on the ordinary path), the logits from the pixel's own basis
        v2f yv[3][NP];
        if ((exact_mask >> g) & 1u) exact_tier_logits_all<PPL>(p, __builtin_amdgcn_readfirstlane(S.id[g]), px, py2, yv);
        else poly_rows_at_pixels<NP>(cg, pu2, pv2, yv);
        if (!((tay_mask >> g) & 1u)) {
          // not the Taylor tier (wave-uniform; the rare case): yv holds logits.  The three channels' denominators 1 + exp2(s_c)
          // first, then ONE reciprocal per pixel for all of them: 1 / d_c = (1 / (d_0 d_1 d_2)) * (the other two).
          // (poly_transform keeps |s| <= 40: the product stays finite.)
#pragma unroll
          for (int jp = 0; jp < NP; ++jp) {
            v2f den[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) den[c] = splat2(1.0f) + v2f{__builtin_amdgcn_exp2f(yv[c][jp][0]), __builtin_amdgcn_exp2f(yv[c][jp][1])};
            const v2f d01 = den[0] * den[1];
            const v2f d = d01 * den[2];
            const v2f r = v2f{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
            const v2f r01 = r * den[2];  // 1 / (d0 d1)
            yv[0][jp] = r01 * den[1];
            yv[1][jp] = r01 * den[0];
            yv[2][jp] = r * d01;
          }
        }
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) {
          w2[jp] = Tr2[jp] * ag2[jp];  // T (a G), or 0 (formed BEHIND the exact tier's calls: four registers less across them)
#pragma unroll
          for (int c = 0; c < 3; ++c) acc2[jp][c] = ffma2(w2[jp], yv[c][jp], acc2[jp][c]);
        }
      }
      if constexpr (!POLY) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          v2f q[NPAIR];
#pragma unroll
          for (int k = 0; k < NPAIR; ++k) q[k] = *reinterpret_cast<const v2f *>(cg + c * CCP + 2 * k);
#pragma unroll
          for (int jp = 0; jp < NP; ++jp) {
            v2f sa = q[0] * Yp[2 * jp][0], sb = q[0] * Yp[2 * jp + 1][0];
#pragma unroll
            for (int k = 1; k < NPAIR; ++k) {
              sa = ffma2(q[k], Yp[2 * jp][k], sa);
              sb = ffma2(q[k], Yp[2 * jp + 1][k], sb);
            }
            const v2f sp = v2f{add_scalar(sa[0], sa[1]), add_scalar(sb[0], sb[1])};
            const v2f den = splat2(1.0f) + v2f{__builtin_amdgcn_exp2f(sp[0]), __builtin_amdgcn_exp2f(sp[1])};
            const v2f yv = v2f{__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
            acc2[jp][c] = ffma2(w2[jp], yv, acc2[jp][c]);
          }
        }
      }
      if (track_stop) {  // (wave-uniform; a launch without the segmented backward's stop list skips 12 vector instructions)
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) {
          const bool was[2] = {alive(2 * jp), alive(2 * jp + 1)};
          Tr2[jp] = Tr2[jp] * one_minus2(ag2[jp]);  // T (1 - a G) if it contributed (explicit)
#pragma unroll
          for (int e = 0; e < 2; ++e)
            if (was[e] && !alive(2 * jp + e)) stop[2 * jp + e] = base + g + 1;
        }
      } else {
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) Tr2[jp] = Tr2[jp] * one_minus2(ag2[jp]);
      }
    }
    bool any_alive = false;
#pragma unroll
    for (int j = 0; j < PPL; ++j) any_alive |= alive(j);
    if (__syncthreads_or((int)any_alive) == 0) break;  // whole tile saturated: stop staging
  }

#pragma unroll
  for (int j = 0; j < PPL; ++j) {
    if (!valid[j]) continue;
    const size_t pix = (size_t)gy[j] * p.W + gx;
    const float Tj = Tr2[j >> 1][j & 1];
    float *o = p.out + 3 * pix;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float a = acc2[j >> 1][c][j & 1];
      o[c] = has_bg ? a + bgv[c] * Tj : a;  // vol_render_bg.h:95-100
    }
    if (p.T != nullptr) p.T[pix] = Tj;
  }
  if (p.stop != nullptr) {
#pragma unroll
    for (int j = 0; j < PPL; ++j) p.stop[(size_t)tile * 256 + (ly0 + j * ROWS) * 16 + lx] = stop[j];
  }
}
// How a launch that was given the coefficient bound is made of these instantiations (launch helpers below):
//   one camera   NB = kRouted   -- ONE kernel holding both forms (a lone launch does not fill the chip: register-limited
//                                  occupancy is irrelevant, an extra launch is not)
//   camera batch NB = kPolyNB   -- the polynomial form alone: 76-120 registers instead of 96-168, i.e. 4 wavefronts per SIMD in
//                                  the backward instead of 3 (+7-11 % renders/s, profiles/r03_ab_pairskip_polyonly.txt); a
//                                  workgroup whose view fails the bound leaves at once
//              + NB = kFallback -- the exact form for exactly those views: a PERSISTENT launch of a few thousand workgroups
//                                  that first asks whether any view of the batch needs it (normally none: the launch is over
//                                  after a handful of scalar loads per workgroup) and otherwise strides over the batch's
//                                  (view, tile) grid, skipping the views the polynomial kernel took.
constexpr int kFallback = -2;
// Per-camera launches that route per TILE (CompParams::sh_rows without tile_flags): does every splat of the tile's list satisfy
// the bound for this view's pixel size?  All threads of the workgroup call it (it ends in a barrier); forward and backward scan
// the same list, hence take the same form.  `bid`: the workgroup's index in the view's tile grid.
__device__ __forceinline__ bool tile_list_within_bound(const CompParams &p, uint32_t bid) {
  int tx, ty;
  if (!block_tile(p, tx, ty, bid)) return true;  // (no tile: the body leaves at once either way)
  const int tile = ty * p.ntw + tx;
  const int st = p.start[tile];
  const int n = (st < 0) ? 0 : (p.end[tile] - st);
  const float ps = fmaxf(fabsf(p.psx), fabsf(p.psy));
  bool bad = false;
  for (int k = (int)threadIdx.x; k < n; k += (int)blockDim.x) bad |= !poly_row_ok(p.sh_rows[p.ids[st + k]], ps);
  return __syncthreads_or((int)bad) == 0;
}
// which tile does workgroup `bid` of a view's grid render, and is it flagged for the exact kernel?  (persistent fallback, per-tile
// routing)
__device__ __forceinline__ bool tile_flagged(const CompParams &p, uint32_t bid) {
  int tx, ty;
  if (!block_tile(p, tx, ty, bid)) return false;
  return p.tile_flags[ty * p.ntw + tx] != 0;
}
// per-camera routed launches (both forms in one kernel): the form of tile bid.  The view's bound first -- a scene wholly within it,
// the normal case, needs no look at the lists (scanning them cost the one-render-in-flight path 10 %: a scan covers the whole list,
// the walk a quarter of it) --, then the tile's list against the per-splat bounds.
__device__ __forceinline__ bool routed_tile_is_polynomial(const CompParams &p, uint32_t bid) {
  if (p.sh_bound != nullptr && poly_route(p.sh_bound, p.psx, p.psy)) return true;
  return p.sh_rows != nullptr && tile_list_within_bound(p, bid);
}
// TRACK = false (the batched polynomial kernel of an unsegmented launch only): no stop list to keep -- with the per-entry exact
// tier's calls in the entry loop, those four registers decide whether the loop fits 96 without reloading spilled values per entry
template <int CB, int PPL, bool BATCH = false, int NB = 0, bool TRACK = true>
__global__ void __launch_bounds__(256 / PPL)
GS_WAVES_PER_EU((NB == kPolyNB && BATCH && PPL == 4 && !TRACK) ? 5 : 1)  // that kernel: five wavefronts per SIMD (<= 96 registers)
k_composite_fwd_sh_vec(CompParams p_arg, ViewPack<BATCH> pack) {
  const CompParams *plist = pack.table();  // (kernel-argument memory: scalar loads, no table in device memory)
  uint32_t bid = blockIdx.x;
  if constexpr (NB == kRouted) {
    static_assert(CB == 4, "routing exists for SH degree 3");
    union Shared {
      FwdShVecShared<4, true> poly;
      FwdShVecShared<4, false> exact;
    };
    __shared__ Shared sm;
    const CompParams *pp = BATCH ? &plist[batch_view(p_arg, bid)] : &p_arg;  // (see k_composite_bwd_sh_vec)
    const bool poly = routed_tile_is_polynomial(*pp, bid);
    if (poly) {
      CompParams p = *pp;
      p.sh_rows = nullptr;  // (every splat of this tile is within the bound: no per-entry tests, no exact tier)
      composite_fwd_sh_vec_tile<4, PPL, kPolyNB>(p, bid, sm.poly);
    } else {
      const CompParams p = *pp;
      composite_fwd_sh_vec_tile<4, PPL, 0>(p, bid, sm.exact);
    }
  } else if constexpr (NB == kFallback) {
    static_assert(CB == 4 && BATCH, "the persistent exact fallback of a bounded batch");
    __shared__ FwdShVecShared<4, false> sm;
    const uint32_t B = (uint32_t)p_arg.n_lo, total = p_arg.vgrid;
    const bool per_tile = plist[0].sh_rows != nullptr;  // (one mode per launch)
    if (!per_tile) {
      bool any = false;
      for (uint32_t v = 0; v < B; ++v) any |= !poly_route(plist[v].sh_bound, plist[v].psx, plist[v].psy);
      if (!any) return;  // every view of the batch took the polynomial form
    }
    const uint32_t per = total / B;  // camera-major
    uint32_t base = 0;  // first block of the view b lies in (b only grows: no division in the loop)
    uint32_t view = 0;  // (an index, not a walking pointer: the table stays in kernel-argument memory)
    for (uint32_t b = blockIdx.x; b < total; b += gridDim.x) {
      while (b >= base + per) { base += per; ++view; }
      const CompParams *pp = &plist[view];
      // per-view routing: the views the polynomial kernel left; per-tile routing (this launch runs BEHIND the polynomial
      // kernel): the tiles it flagged
      if (per_tile ? !tile_flagged(*pp, b - base) : poly_route(pp->sh_bound, pp->psx, pp->psy)) continue;
      const CompParams p = *pp;
      composite_fwd_sh_vec_tile<4, PPL, 0, true>(p, b - base, sm);
      __syncthreads();  // the LDS block is reused by the next tile
    }
  } else if constexpr (NB == kPolyNB && BATCH) {
    __shared__ FwdShVecShared<4, true> sm;
    const CompParams *pp = &plist[batch_view(p_arg, bid)];
    const bool view_ok = pp->sh_bound != nullptr && poly_route(pp->sh_bound, pp->psx, pp->psy);
    if (pp->sh_rows == nullptr && !view_ok) return;  // per-view routing: this view is the exact fallback's
    CompParams p = *pp;
    if (view_ok) p.sh_rows = nullptr;  // the whole scene is within this view's bound: no per-entry tests (tile_flags stay 0)
    composite_fwd_sh_vec_tile<4, PPL, kPolyNB, false, TRACK>(p, bid, sm);
  } else {
    const CompParams p = BATCH ? plist[batch_view(p_arg, bid)] : p_arg;  // see k_composite_fwd
    __shared__ FwdShVecShared<CB, (NB > 0)> sm;
    composite_fwd_sh_vec_tile<CB, PPL, NB>(p, bid, sm);
  }
}

// ============================================================================================
// backward
// ============================================================================================
// The unpacked vector form of the backward: the shape behind the `_gs` entry points for callers that configure a tile side of 8
// or 32 (see k_composite_fwd); 16 x 16 tiles run the packed kernels below.
template <int MODE, int CB, int PPL, int TS>  // TS: tile side, see k_composite_fwd
__global__ void __launch_bounds__(TS * TS / PPL)
k_composite_bwd_pixel(CompParams p) {
  const uint32_t bid = blockIdx.x, grid = gridDim.x;
  using TR = Traits<MODE, CB>;
  constexpr int NT = TS * TS / PPL;
  static_assert(NT >= kBatch && NT % 64 == 0, "a staging round needs one thread per record");
  constexpr int ROWS = NT / TS;
  constexpr int NCH = TR::NCH;
  constexpr int P = TR::P;
  __shared__ Stage<MODE, CB> S;

  // Segmented launch (SH only): workgroup = (tile, segment of kSegLen list entries) starting from the
  // state the forward left in front of the segment (CompParams::ckpt / stop); segment-major order: all
  // first segments, then all second ones, ... (the heavy tiles' segments spread over the launch).
  const int nseg = (MODE == MODE_SH && p.nseg > 1) ? p.nseg : 1;
  const uint32_t tiles_grid = grid / (uint32_t)nseg;
  const int seg = (int)(bid / tiles_grid);
  int tx, ty;
  if (!block_tile(p, tx, ty, bid % tiles_grid)) return;  // uniform over the workgroup
  const int tile = ty * p.ntw + tx;
  const int st = p.start[tile];
  const int n = (st < 0) ? 0 : (p.end[tile] - st);
  if (n == 0 || n < p.n_lo || n >= p.n_hi) return;
  const int e_lo = seg * kSegLen;
  const int e_hi = (seg == nseg - 1) ? n : min(n, e_lo + kSegLen);  // the last segment takes the rest
  if (e_lo >= n) return;
  const int t = (int)threadIdx.x;
  const int lane = t & 63;
  const int lx = t % TS, ly0 = t / TS;
  const int side = (TS == 32) ? p.tile_side : TS;  // (TS = 32: any caller tile side that is not 8 or 16, see k_composite_fwd)
  const int gx = tx * side + lx;

  bool valid[PPL];
  int gy[PPL];
  float py[PPL];
  const float px = pixel_coord(p.topleft[0], gx, p.psx);
#pragma unroll
  for (int j = 0; j < PPL; ++j) {
    gy[j] = ty * side + ly0 + j * ROWS;
    valid[j] = (gx < p.W) && (gy[j] < p.H) && (TS != 32 || (lx < side && ly0 + j * ROWS < side));
    py[j] = pixel_coord(p.topleft[1], gy[j], p.psy);
  }

  v2f Yp[MODE == MODE_SH ? PPL : 1][MODE == MODE_SH ? TR::NPAIR : 1];
  if constexpr (MODE == MODE_SH) {
    if constexpr (TR::CCP != TR::CC) {
      for (int e = t; e < kBatch * TR::NCOLP; e += NT) S.col[e] = 0.0f;
      __syncthreads();
    }
    float R[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = p.rot[i];
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      float Yf[TR::CCP];
      sh_basis_of_pixel<CB>(R, px, py[j], Yf);
#pragma unroll
      for (int k = 0; k < TR::NPAIR; ++k) Yp[j][k] = v2f{Yf[2 * k], Yf[2 * k + 1]};
    }
  }

  float go[PPL][NCH], fin[PPL][NCH], pre[PPL][NCH], Tr[PPL];
  bool alive[PPL];
#pragma unroll
  for (int j = 0; j < PPL; ++j) {
    alive[j] = valid[j];
    if constexpr (MODE == MODE_SH) {
      if (nseg > 1) alive[j] = alive[j] && (p.stop[(size_t)tile * (TS * TS) + (ly0 + j * ROWS) * TS + lx] > e_lo);
    }
  }
  if constexpr (MODE == MODE_SH) {
    if (nseg > 1) {  // every pixel of the tile stopped before this segment?
      bool any = false;
#pragma unroll
      for (int j = 0; j < PPL; ++j) any |= alive[j];
      if (__syncthreads_or((int)any) == 0) return;
    }
  }
#pragma unroll
  for (int j = 0; j < PPL; ++j) {
    const size_t pix = valid[j] ? ((size_t)gy[j] * p.W + gx) : 0;
    float4 ck = make_float4(1.0f, 0.0f, 0.0f, 0.0f);  // state in front of entry e_lo: T, prefix rgb
    if constexpr (MODE == MODE_SH) {
      if (seg > 0 && alive[j]) ck = p.ckpt[((size_t)tile * nseg + seg) * (TS * TS) + (ly0 + j * ROWS) * TS + lx];
    }
    const float pre0[3] = {ck.y, ck.z, ck.w};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      go[j][c] = valid[j] ? load_grad_out<MODE, NCH>(p, pix, c) : 0.0f;
      fin[j][c] = valid[j] ? p.final_img[NCH * pix + c] : 0.0f;
      pre[j][c] = (MODE == MODE_SH) ? pre0[c < 3 ? c : 0] : 0.0f;
    }
    Tr[j] = ck.x;
  }

  for (int base = e_lo; base < e_hi; base += kBatch) {
    const int nb = min(kBatch, e_hi - base);
    if (base > e_lo) __syncthreads();
    stage_batch<MODE, CB, NT>(S, p, st + base, nb);
    __syncthreads();

    for (int g = 0; g < nb; ++g) {
      bool any_alive = false;
#pragma unroll
      for (int j = 0; j < PPL; ++j) any_alive |= alive[j];
      if (!wave_any(any_alive)) break;

      const GRec r = load_rec(S, g);
      const float x = px - r.mx;
      float G[PPL];
      bool con[PPL];
      bool any_con = false;
#pragma unroll
      for (int j = 0; j < PPL; ++j) {
        G[j] = gauss_eval<MODE>(r, x, py[j] - r.my, px, py[j], alive[j]);
        con[j] = alive[j] && !(r.a * G[j] < kMinAlpha);
        any_con |= con[j];
      }
      if (!wave_any(any_con)) continue;

      // per-lane partial gradient of this Gaussian over the lane's pixels
      // layout: 0,1 mean | 2 c00 3 c01 4 c10 5 c11 | 6 alpha | 7.. colour/scalar/sh
      float gr[P];
#pragma unroll
      for (int i = 0; i < P; ++i) gr[i] = 0.0f;

      const float inv_det = r.p1;  // SH: 1 / det (fp32, as the reference); the other modes do not use it
      const float *cg = &S.col[g * TR::NCOLP];
      float pAG[PPL], ag[PPL];
      if constexpr (MODE == MODE_SH) {
        // branch-free over the lane's pixels (weight 0 when a pixel does not contribute);
        // the colour dot products and the d/d(sh) accumulation are (k, k+1)-packed FMAs
        float w[PPL], inv1m[PPL];
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
          ag[j] = r.a * G[j];
          w[j] = con[j] ? (r.a * Tr[j]) * G[j] : 0.0f;
          inv1m[j] = __builtin_amdgcn_rcpf(1.0f - ag[j]);
          pAG[j] = 0.0f;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          v2f q[TR::NPAIR], gq[TR::NPAIR];
#pragma unroll
          for (int k = 0; k < TR::NPAIR; ++k) {
            q[k] = *reinterpret_cast<const v2f *>(cg + c * TR::CCP + 2 * k);
            gq[k] = v2f{0.0f, 0.0f};
          }
#pragma unroll
          for (int j = 0; j < PPL; ++j) {
            v2f s2 = q[0] * Yp[j][0];
#pragma unroll
            for (int k = 1; k < TR::NPAIR; ++k) s2 = fma2(q[k], Yp[j][k], s2);
            const float yv = sigmoid_fast(s2[0] + s2[1]);
            pre[j][c] += w[j] * yv;
            const float gs = w[j] * (yv * (1.0f - yv)) * go[j][c];
            const v2f gs2 = splat2(gs);
#pragma unroll
            for (int k = 0; k < TR::NPAIR; ++k) gq[k] = fma2(gs2, Yp[j][k], gq[k]);
            pAG[j] += go[j][c] * (yv * Tr[j] - (fin[j][c] - pre[j][c]) * inv1m[j]);
          }
#pragma unroll
          for (int k = 0; k < TR::CC; ++k) gr[7 + c * TR::CC + k] = gq[k >> 1][k & 1];
        }
      } else {
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
          ag[j] = r.a * G[j];
          const float coeff = (r.a * Tr[j]) * G[j];
          const float inv1m = __builtin_amdgcn_rcpf(1.0f - ag[j]);
          pAG[j] = 0.0f;
          if (con[j]) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
              pre[j][c] += cg[c] * coeff;
              gr[7 + c] += coeff * go[j][c];
              pAG[j] += (cg[c] * Tr[j] - (fin[j][c] - pre[j][c]) * inv1m) * go[j][c];
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < PPL; ++j) {
        // kernel_gaussian_2d_backward (kernels.h:394-418): v = Sigma^-T d
        const float pa = con[j] ? pAG[j] : 0.0f;
        const float y = py[j] - r.my;
        const float gg = pa * ag[j];
        float vx, vy;
        if constexpr (MODE == MODE_SH) {  // the reference's fp32 formula (kernels.h:394-418)
          vx = (x * r.c3 - y * r.c2) * inv_det;
          vy = (y * r.c0 - x * r.c1) * inv_det;
        } else {  // from the Cholesky factor (see kInvSc2): no fp32 determinant
          const float u = r.p0 * x + r.p1 * y, v = r.p2 * y;
          vx = kInvSc2 * (r.p0 * u);
          vy = kInvSc2 * (r.p1 * u + r.p2 * v);
        }
        gr[0] += gg * vx;
        gr[1] += gg * vy;
        const float h = 0.5f * gg;
        gr[2] += h * vx * vx;
        gr[3] += h * vx * vy;
        gr[5] += h * vy * vy;
        gr[6] += pa * G[j];
        const float om = ffma(-ag[j], con[j] ? 1.0f : 0.0f, 1.0f);  // as the forward: 1 - round(a G), explicit
        Tr[j] *= om;
        alive[j] = alive[j] && !(Tr[j] < p.thresh);
      }
      gr[4] = gr[3];  // grad_cov[1] and grad_cov[2] receive the same value (kernels.h:414-415)

      wave_reduce_scatter<P>(gr);
      const int comp = scatter_comp<P>(lane);
      if (scatter_owner<P>(lane) && comp < TR::NCOMP) {
        const size_t id = (size_t)S.id[g];
        float *dst;
        if (comp < 2) dst = p.g_mean + 2 * id + comp;
        else if (comp < 6) dst = p.g_cov + 4 * id + (comp - 2);
        else if (comp == 6) dst = p.g_alpha + id;
        else dst = p.g_col + (size_t)TR::NCOL * id + (comp - 7);
        atomicAdd(dst, gr[0]);
      }
    }
    bool any_alive = false;
#pragma unroll
    for (int j = 0; j < PPL; ++j) any_alive |= alive[j];
    if (__syncthreads_or((int)any_alive) == 0) break;
  }
}


// ============================================================================================
// backward, SH, vector ALUs, packed per-pixel arithmetic (the default SH backward)
// ============================================================================================
// k_composite_bwd_pixel<MODE_SH> spends 890 VALU quad-cycles per (wavefront, list entry) at 4 pixels per lane
// (profiles/r02_*): ~310 of them are the per-pixel scalar arithmetic (Gaussian, weights, suffix colour,
// d/d(alpha G), mean / covariance terms) which is identical for the lane's pixels, 32 are transcendental (4 quads
// each).  On gfx950 a packed fp32 instruction (v_pk_mul / v_pk_add / v_pk_fma) issues at the rate of a scalar one
// (SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.11 quads for a stream that is a quarter packed), so this kernel
//   * carries every per-pixel quantity as a PAIR of pixels (v2f) and does that arithmetic packed;
//   * replaces the three selects per pixel (weight, 1 - aG, d/d(aG)) by one 0/1 mask multiplied in (exact);
//   * keeps the suffix colour as one running value per channel (final - prefix) instead of final and prefix;
//   * reads the coefficients pre-scaled by -log2(e) (stage_batch<SCALE>): no multiply in front of the exp2;
//   * lays the 3 x CCP SH components out FIRST in the reduction vector, so that the (k, k+1) accumulator pairs of
//     the packed FMAs are the (even, odd) pairs of wave_reduce_scatter2 (packed adds, no register moves).
// The Gaussian evaluation is gauss_sh_pair: bit for bit gauss_eval<MODE_SH>, so "skip" and "saturated" decisions
// equal the forward's.  Same launch shapes as k_composite_bwd_pixel (one workgroup per tile or per (tile, segment),
// 256 / PPL threads); PPL = 4 (one wavefront per tile) or 2.
// The gradient vector is reduced CHANNEL BY CHANNEL (wave_reduce_scatter2_rows on the 3 x CCP SH components as
// soon as a channel's accumulators are complete, the 7 geometric components likewise, one quad_reduce_scatter4 at the
// end) instead of as one 64-component vector after the third channel: the same number of exchanges, but the finished
// channels no longer sit in 32 registers while the next one is computed (190 -> 164 registers in round 2: the single-reduction
// form, `CHRED = false`, is in the history up to round 3).
// NB > 0 (= kPolyNB; SH degree 3, one wavefront per tile): the tile-local polynomial form of the per-pixel basis
// (composite_common.hpp) -- 6-term contractions, 3 x 6 SH gradient components across the lanes, expanded by the tile's V
// (through 24 floats of LDS) in front of the 48 atomics.
// NB = kRouted: as k_composite_fwd_sh_vec -- one launch, the form chosen per workgroup from the device-resident bound.
template <int CB, int PPL, bool POLY>
struct BwdShVecShared {
  static constexpr int KB = 32;  // records per staging round: 10.6 KB of LDS per workgroup, 12+ workgroups per CU
  static constexpr int NT = 256 / PPL, NP = PPL / 2;
  Stage<MODE_SH, CB, KB, !POLY> S;
  v2f go_s[!POLY ? 3 * NP * NT : 1];                      // exact form: grad_out of the lane's pixel pairs, [channel][pair][thread]
  alignas(16) float Vs[POLY ? kPolyNB * 16 : 4];          // POLY: V of this tile
  alignas(16) float Ws[POLY ? KB * 3 * kPolyStride : 4];    // POLY: transformed coefficients of the staged batch
                                                          // (and, before the first batch, the nine node bases)
  float gw_s[POLY ? 3 * 8 : 1];                           // POLY: a splat's reduced gradient in the tile's basis
  float tay_ok[POLY ? KB * 3 : 1];                        // POLY: rows of the staged batch that may take the Taylor tier (poly_transform)
};
// MOM (round 6; the batched launches of gsgen_vol_render_backward_sh_batch_routed_moments): the geometric gradients leave the kernel
// as MOMENTS of the per-pixel weight g = d L / d (a G) * a G against (tx, ty) = det * Sigma^-1 d, the offsets gauss_sh_pair forms
// anyway -- (Mx, My) into grad_mean, (Mxx, Mxy, Myy) into grad_cov[0..2] -- and the projection backward scales them by 1 / det and
// 0.5 / det^2 per (view, Gaussian) (geometry.hip, moments_to_grads_sh): 7 packed operations per pixel pair instead of 13, one
// atomic per entry less (grad_cov[2] = grad_cov[1] is formed there too).
template <int CB, int PPL, int NB, bool PERSIST = false, bool MOM = false>
__device__ __forceinline__ void composite_bwd_sh_vec_tile(const CompParams &p, uint32_t bid, uint32_t grid,
                                                          BwdShVecShared<CB, PPL, (NB > 0)> &sm) {
  static_assert(PPL == 4 || PPL == 2, "pixel pairs: 2 or 4 pixels per lane");
  constexpr bool POLY = NB > 0;
  static_assert(!POLY || (NB == kPolyNB && CB == 4 && PPL == 4), "polynomial basis: SH degree 3, one wavefront per tile");
  constexpr int MODE = MODE_SH;
  using TR = Traits<MODE, CB>;
  constexpr int NT = 256 / PPL, ROWS = NT / 16, NP = PPL / 2;
  constexpr int CCP = POLY ? NB : TR::CCP, NPAIR = CCP / 2, NSH = 3 * CCP;  // SH components incl. padding
  static_assert(NSH % 2 == 0, "component layout");
  constexpr int KB = BwdShVecShared<CB, PPL, POLY>::KB;
  auto &S = sm.S;
  v2f *const go_s = sm.go_s;
  float *const Vs = sm.Vs;
  float *const Ws = sm.Ws;
  float *const gw_s = sm.gw_s;
  const int nseg = p.nseg > 1 ? p.nseg : 1;
  const uint32_t tiles_grid = grid / (uint32_t)nseg;
  const int seg = (int)(bid / tiles_grid);
  int tx, ty;
  if (!block_tile(p, tx, ty, bid % tiles_grid)) return;  // uniform over the workgroup
  const int tile = ty * p.ntw + tx;
  const int st = p.start[tile];
  const int n = (st < 0) ? 0 : (p.end[tile] - st);
  if (n == 0 || n < p.n_lo || n >= p.n_hi) return;
  if constexpr (POLY) {  // per-tile routing: the forward sent this tile to the exact kernel (CompParams::tile_flags)
    if (p.tile_flags != nullptr && p.tile_flags[tile] != 0) return;
  }
  const int e_lo = seg * kSegLen;
  const int e_hi = (seg == nseg - 1) ? n : min(n, e_lo + kSegLen);  // the last segment takes the rest
  if (e_lo >= n) return;
  const int t = tile_thread_index<PERSIST>();
  const int lane = t & 63;
  const int lx = t & 15, ly0 = t >> 4;
  const int gx = tx * kTile + lx;
  const float px = pixel_coord(p.topleft[0], gx, p.psx);

  // A pixel is "alive" while its transmittance has not dropped below the threshold: T only decreases, so the flag is
  // T itself (pixels outside the image, or stopped before this segment, start at T = -1: never alive, and every
  // product they enter is masked by the 0/1 contribution mask).
  bool valid[PPL], alive0[PPL];
  int gy[PPL];
  v2f py2[NP];
#pragma unroll
  for (int j = 0; j < PPL; ++j) {
    gy[j] = ty * kTile + ly0 + j * ROWS;
    valid[j] = (gx < p.W) && (gy[j] < p.H);
    py2[j >> 1][j & 1] = pixel_coord(p.topleft[1], gy[j], p.psy);
    alive0[j] = valid[j];
    if (nseg > 1) alive0[j] = alive0[j] && (p.stop[(size_t)tile * 256 + (ly0 + j * ROWS) * 16 + lx] > e_lo);
  }
  if (nseg > 1) {  // every pixel of the tile stopped before this segment?
    bool any = false;
#pragma unroll
    for (int j = 0; j < PPL; ++j) any |= alive0[j];
    if (__syncthreads_or((int)any) == 0) return;
  }
  if constexpr (TR::CCP != TR::CC) {  // zero the pad lanes of the staged coefficients once
    for (int e = t; e < KB * TR::NCOLP; e += NT) S.col[e] = 0.0f;
    __syncthreads();
  }

  // per-pixel SH basis as (k, k+1) pairs
  v2f Yp[POLY ? 1 : PPL][POLY ? 1 : NPAIR];
  // POLY: where this lane's geometric component goes (see the reduction at the end of the entry loop): its base address and its
  // stride per splat, nullptr for lanes that hold none -- three registers instead of the selects, shifts and spilled lane masks
  // that formed the address per entry (13 vector instructions per (wavefront, entry) less)
  float *geo_base = nullptr;
  uint32_t geo_stride = 0u;
  // POLY: the lane's column offset and its pixel pairs' row offsets stand for the six monomials (see the forward)
  const float pu = poly_offset(lx);
  const v2f pu2 = splat2(pu);
  v2f pv2[NP];
#pragma unroll
  for (int jp = 0; jp < NP; ++jp) pv2[jp] = v2f{poly_offset(ly0 + (2 * jp) * ROWS), poly_offset(ly0 + (2 * jp + 1) * ROWS)};
  if constexpr (POLY) {
    static_assert(!POLY || KB * 3 * kPolyNB >= kPolyNodes * 16, "node scratch fits the coefficient buffer");
    poly_tile_setup<NT>(p, tx, ty, Ws, Vs);
    {
      // lane (m, slot e = comp - 6) -> m 0: mean[e] | 1: cov[e] | 2: cov[3], alpha | 3: -, cov[2]
      const int m = lane & 3, e = scatter_comp<8>(lane) - 6;
      if constexpr (MOM) {  // m 0: (Mx, My) -> mean[e] | 1: (Mxx, Mxy) -> cov[e] | 2: (Myy, alpha) -> cov[2], alpha | 3: -
        if (scatter_rows_owner<8>(lane) && e >= 0 && m < 3) {
          geo_base = m == 0 ? p.g_mean + e : (m == 1 ? p.g_cov + e : (e == 0 ? p.g_cov + 2 : p.g_alpha));
          geo_stride = m == 0 ? 2u : ((m == 2 && e == 1) ? 1u : 4u);
        }
      } else if (scatter_rows_owner<8>(lane) && e >= 0 && !(m == 3 && e == 0)) {
        geo_base = m == 0 ? p.g_mean + e : ((m == 2 && e == 1) ? p.g_alpha : p.g_cov + (m == 1 ? e : (m == 2 ? 3 : 2)));
        geo_stride = m == 0 ? 2u : ((m == 2 && e == 1) ? 1u : 4u);
      }
    }
  } else {
    float R[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = p.rot[i];
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      const float pyj = py2[j >> 1][j & 1];
      float Yf[CCP];
      sh_basis_of_pixel<CB>(R, px, pyj, Yf);
#pragma unroll
      for (int k = 0; k < NPAIR; ++k) Yp[j][k] = v2f{Yf[2 * k], Yf[2 * k + 1]};
      __builtin_amdgcn_sched_barrier(0);  // one pixel's temporaries at a time
    }
  }

  // pixel pairs: element e of pair jp is pixel j = 2 jp + e.  rem = final - (prefix colour incl. the current
  // splat): the suffix the reference forms as final - Cpre_incl (vol_render_sh.h:328-333)
  v2f go2[NP][3], rem2[NP][3], Tr2[NP];
#pragma unroll
  for (int j = 0; j < PPL; ++j) {
    const size_t pix = valid[j] ? ((size_t)gy[j] * p.W + gx) : 0;
    float4 ck = make_float4(1.0f, 0.0f, 0.0f, 0.0f);  // state in front of entry e_lo: T, prefix rgb
    if (seg > 0 && alive0[j]) ck = p.ckpt[((size_t)tile * nseg + seg) * 256 + (ly0 + j * ROWS) * 16 + lx];
    const float pre0[3] = {ck.y, ck.z, ck.w};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      go2[j >> 1][c][j & 1] = valid[j] ? p.grad_out[3 * pix + c] : 0.0f;
      rem2[j >> 1][c][j & 1] = valid[j] ? p.final_img[3 * pix + c] - pre0[c] : 0.0f;
    }
    Tr2[j >> 1][j & 1] = (alive0[j] && !(ck.x < p.thresh)) ? ck.x : -1.0f;
  }
  // The suffix enters d L / d (a G) only as  sum_c grad_out_c * suffix_c,  and every splat lowers that sum by
  // w * sum_c grad_out_c * colour_c: ONE running value per pixel instead of three (two packed operations less per channel and
  // pixel pair, eight registers less)
  v2f R2[NP], pvsq2[NP];
#pragma unroll
  for (int jp = 0; jp < NP; ++jp) {
    R2[jp] = fma2(go2[jp][2], rem2[jp][2], fma2(go2[jp][1], rem2[jp][1], go2[jp][0] * rem2[jp][0]));
    pvsq2[jp] = pv2[jp] * pv2[jp];
  }
  if constexpr (!POLY) {  // each thread reads back only what it wrote: no barrier
#pragma unroll
    for (int jp = 0; jp < NP; ++jp)
#pragma unroll
      for (int c = 0; c < 3; ++c) go_s[(c * NP + jp) * NT + t] = go2[jp][c];
  }
  auto alive = [&](int j) { return !(Tr2[j >> 1][j & 1] < p.thresh); };

  uint32_t exact_mask = 0u, tay_mask = 0u;
  for (int base = e_lo; base < e_hi; base += KB) {
    const int nb = min(KB, e_hi - base);
    if (base > e_lo) __syncthreads();
    stage_batch<MODE, CB, NT, KB, true, !POLY>(S, p, st + base, nb);
    __syncthreads();
    if constexpr (POLY) {
      // (the forward's per-entry exact tier: the same splats, the same test -- a tile the forward gave up is not walked here)
      exact_mask = (uint32_t)__builtin_amdgcn_readfirstlane((int)exact_tier_mask<NT>(p, S.id, nb, t, nullptr));
      poly_transform<NT, KB>(p.col, S.id, Vs, Ws, nb, sm.tay_ok);
      __syncthreads();
      tay_mask = (uint32_t)__builtin_amdgcn_readfirstlane((int)(taylor_mask(sm.tay_ok, nb) & ~exact_mask));  // (as the forward)
      taylor_convert<NT>(Ws, tay_mask, nb);
      __syncthreads();
    }

    for (int g = 0; g < nb; ++g) {
      bool any_alive = false;
#pragma unroll
      for (int j = 0; j < PPL; ++j) any_alive |= alive(j);
      if (!wave_any(any_alive)) break;

      // the record as plain scalars (a struct handed around by reference makes the compiler build the packed
      // operands through scratch memory)
      // (exact form: wave-uniform values moved to scalar registers -- nine vector registers less)
      // (POLY: 108 registers leave room for the nine scalars as vector registers -- nine v_readfirstlane less per entry)
      auto uni = [&](float v) { return !POLY ? wave_uniform(v) : v; };
      const float r_mx = uni(S.mx[g]), r_my = uni(S.my[g]), r_a = uni(S.a[g]), r_c0 = uni(S.c0[g]), r_c1 = uni(S.c1[g]),
                  r_c2 = uni(S.c2[g]), r_c3 = uni(S.c3[g]), r_p0 = uni(S.p0[g]), r_p1 = uni(S.p1[g]);
      const float x = px - r_mx;
      // G2 / ag2: the Gaussian and a G, ZEROED where the pixel does not take part (skip threshold, or not alive)
      v2f y2[NP], G2[NP], ag2[NP], tx2[NP], ty2[NP];
      bool any_con = false;
      float guard_dist = 0.0f;
#pragma unroll
      for (int jp = 0; jp < NP; ++jp) {
        y2[jp] = py2[jp] - splat2(r_my);
        G2[jp] = gauss_sh_pair(r_c0, r_c1, r_c2, r_c3, r_p0, x, y2[jp], tx2[jp], ty2[jp]);
        ag2[jp] = splat2(r_a) * G2[jp];
        {
          // the lane's smallest distance to the threshold (dead pixels included: a spurious trip re-tests per pixel)
          const v2f dist = ag2[jp] - splat2(kMinAlpha);
          const float dmin = fminf(fabsf(dist[0]), fabsf(dist[1]));
          guard_dist = jp == 0 ? dmin : fminf(guard_dist, dmin);
        }
      }
      const bool any_guard = guard_dist <= kMinAlpha * kGuardTol;
      // within rounding of the skip threshold: the reference's arithmetic decides (as gauss_eval).  One wave-uniform
      // test for all the lane's pixels: the branch is almost never taken (a handful of pixels per frame)
      if (wave_any(any_guard)) {
#pragma unroll
        for (int jp = 0; jp < NP; ++jp)
#pragma unroll
          for (int e = 0; e < 2; ++e)
            if (alive(2 * jp + e) && fabsf(ag2[jp][e] - kMinAlpha) <= kMinAlpha * kGuardTol) {
              G2[jp][e] = gauss_ref_f32(r_mx, r_my, r_c0, r_c1, r_c2, r_c3, px, py2[jp][e]);
              ag2[jp][e] = r_a * G2[jp][e];
            }
      }
#pragma unroll
      for (int jp = 0; jp < NP; ++jp)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const bool con = alive(2 * jp + e) && !(ag2[jp][e] < kMinAlpha);
          G2[jp][e] = con ? G2[jp][e] : 0.0f;
          if constexpr (!POLY) ag2[jp][e] = con ? ag2[jp][e] : 0.0f;
          any_con |= con;
        }
      if constexpr (POLY) {  // the same products again, from the masked G: two packed multiplies instead of four selects
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) {
          ag2[jp] = splat2(r_a) * G2[jp];
        }
      }
      // (skipping a pixel PAIR none of whose 128 pixels takes part -- pair-major code, one wave-uniform test per pair -- was
      // measured again in round 3 with the polynomial body: 4 488 vs 4 507 renders/s, profiles/r03_ab_pairskip_polyonly.txt)
      if (!wave_any(any_con)) continue;  // nobody in the wave sees this Gaussian

      constexpr int PCH = CCP <= 2 ? 2 : (CCP <= 4 ? 4 : (CCP <= 8 ? 8 : 16));  // one channel's components in its reduction
      float chsum[3] = {0.0f, 0.0f, 0.0f};
      const float *cg = POLY ? &Ws[g * 3 * kPolyStride] : &S.col[g * TR::NCOLP];
      v2f pch[POLY ? 3 : 1][3];  // POLY: the channels' six components each, reduced after the geometric part
      v2f w2[NP], inv1m2[NP], pAG2[NP];
#pragma unroll
      for (int jp = 0; jp < NP; ++jp) {
        w2[jp] = POLY ? Tr2[jp] * ag2[jp] : (splat2(r_a) * Tr2[jp]) * G2[jp];  // the forward's (a T) G -- POLY: T (a G) --, or 0
        if constexpr (!POLY) {
          const v2f om = one_minus2(ag2[jp]);
          inv1m2[jp] = v2f{__builtin_amdgcn_rcpf(om[0]), __builtin_amdgcn_rcpf(om[1])};
        }
        pAG2[jp] = v2f{0.0f, 0.0f};
      }
      if constexpr (POLY) {
        // colours as in the forward: the staged rows at the lane's pixels (or the exact tier's logits), colours already in the
        // Taylor tier -- otherwise the three channels' denominators and ONE reciprocal per pixel for the three sigmoids and
        // 1 / (1 - a G):  1 / x_i = (1 / prod x) * prod_{j != i} x_j  (|s| <= 40 by poly_transform, 1 - a G >= 0.01)
        v2f yv[3][NP];
        if ((exact_mask >> g) & 1u) exact_tier_logits_all<PPL>(p, __builtin_amdgcn_readfirstlane(S.id[g]), px, py2, yv);
        else poly_rows_at_pixels<NP>(cg, pu2, pv2, yv);
        v2f om2[NP];  // 1 - a G (>= 0.01)
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) om2[jp] = one_minus2(ag2[jp]);
        if ((tay_mask >> g) & 1u) {  // (wave-uniform; the ordinary case)
#pragma unroll
          for (int jp = 0; jp < NP; ++jp) inv1m2[jp] = v2f{__builtin_amdgcn_rcpf(om2[jp][0]), __builtin_amdgcn_rcpf(om2[jp][1])};
        } else {
#pragma unroll
          for (int jp = 0; jp < NP; ++jp) {
            v2f den[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) den[c] = splat2(1.0f) + v2f{__builtin_amdgcn_exp2f(yv[c][jp][0]), __builtin_amdgcn_exp2f(yv[c][jp][1])};
            const v2f om = om2[jp];
            const v2f d01 = den[0] * den[1], d2o = den[2] * om;
            const v2f dd = d01 * d2o;
            const v2f r = v2f{__builtin_amdgcn_rcpf(dd[0]), __builtin_amdgcn_rcpf(dd[1])};
            const v2f r01 = r * d2o, r2o = r * d01;  // 1 / (d0 d1), 1 / (d2 (1 - a G))
            yv[0][jp] = r01 * den[1];
            yv[1][jp] = r01 * den[0];
            yv[2][jp] = r2o * om;
            inv1m2[jp] = r2o * den[2];
          }
        }
        v2f gy2[NP];  // sum_c grad_out_c * colour_c
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          // d L / d w[c][r] = sum over pixels of gs * monomial_r: the lane's sums of gs, gs v, gs v^2, then its column's u
          v2f g0, g1, g2;
#pragma unroll
          for (int jp = 0; jp < NP; ++jp) {
            const v2f y_ = yv[c][jp];
            const v2f dy = fma2(-y_, y_, y_);  // y (1 - y)
            const v2f go = go2[jp][c];  // (registers: this body has them to spare; the exact one reads its copy in LDS)
            const v2f gs = (w2[jp] * dy) * go;
            gy2[jp] = c == 0 ? go * y_ : fma2(go, y_, gy2[jp]);
            if (jp == 0) { g0 = gs; g1 = gs * pv2[jp]; g2 = gs * pvsq2[jp]; }
            else { g0 = g0 + gs; g1 = fma2(gs, pv2[jp], g1); g2 = fma2(gs, pvsq2[jp], g2); }
          }
          const float G0 = add_scalar(g0[0], g0[1]), G1 = add_scalar(g1[0], g1[1]), G2s = add_scalar(g2[0], g2[1]);
          const float uG0 = pu * G0;
          // (1, v, u, v^2, uv, u^2): six of the channel's eight reduction slots; the last two carry geometric components (below)
          pch[c][0] = v2f{G0, G1}; pch[c][1] = v2f{uG0, G2s}; pch[c][2] = v2f{pu * G1, pu * uG0};
        }
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) {
          R2[jp] = fma2(-w2[jp], gy2[jp], R2[jp]);  // the suffix behind this splat (vol_render_sh.h:328-333), grad_out-weighted
          pAG2[jp] = fma2(gy2[jp], Tr2[jp], -(R2[jp] * inv1m2[jp]));
        }
      }
      v2f gyx2[NP];
      if constexpr (!POLY) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        v2f q[NPAIR], gq[NPAIR];
#pragma unroll
        for (int k = 0; k < NPAIR; ++k) q[k] = *reinterpret_cast<const v2f *>(cg + c * CCP + 2 * k);
        // one pixel pair: colours, suffix, d/d(a G), and its share of d/d(sh) (FIRST: it initialises the accumulators)
        auto pair_colour = [&](auto JP, auto FIRST) {
          constexpr int jp = decltype(JP)::value;
          constexpr bool first = decltype(FIRST)::value;
          // -log2(e) * (sh . Y) of the pair's two pixels: the two dot products interleaved (a v_pk_fma that depends
          // on the previous instruction costs a wait state)
          v2f sa = q[0] * Yp[2 * jp][0], sb = q[0] * Yp[2 * jp + 1][0];
#pragma unroll
          for (int k = 1; k < NPAIR; ++k) {
            sa = fma2(q[k], Yp[2 * jp][k], sa);
            sb = fma2(q[k], Yp[2 * jp + 1][k], sb);
          }
          const v2f sp = v2f{add_scalar(sa[0], sa[1]), add_scalar(sb[0], sb[1])};
          const v2f den = splat2(1.0f) + v2f{__builtin_amdgcn_exp2f(sp[0]), __builtin_amdgcn_exp2f(sp[1])};
          const v2f yv = v2f{__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
          const v2f dy = fma2(-yv, yv, yv);  // y (1 - y)
          const v2f go = go_s[(c * NP + jp) * NT + t];
          const v2f gs = (w2[jp] * dy) * go;
          gyx2[jp] = c == 0 ? go * yv : fma2(go, yv, gyx2[jp]);  // sum_c grad_out_c * colour_c (see R2)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const v2f gse = splat2(gs[e]);
            if (first && e == 0) {
#pragma unroll
              for (int k = 0; k < NPAIR; ++k) gq[k] = gse * Yp[2 * jp + e][k];
            } else {
#pragma unroll
              for (int k = 0; k < NPAIR; ++k) gq[k] = fma2(gse, Yp[2 * jp + e][k], gq[k]);
            }
          }
        };
        // (skipping a pair none of whose pixels contributes -- a wave-uniform branch per pair and channel, ~1 pair in 5
        // on the headline workload -- was measured: -1.5 % on a lone launch, +1 % with two batches in flight, 32 more
        // registers; not kept)
        pair_colour(std::integral_constant<int, 0>{}, std::true_type{});
        if constexpr (NP > 1) pair_colour(std::integral_constant<int, NP - 1>{}, std::false_type{});
        {
          v2f t[PCH / 2];
#pragma unroll
          for (int k = 0; k < PCH / 2; ++k) t[k] = k < NPAIR ? gq[k < NPAIR ? k : 0] : v2f{0.0f, 0.0f};
          chsum[c] = wave_reduce_scatter2_rows<PCH>(t);
        }
      }
#pragma unroll
      for (int jp = 0; jp < NP; ++jp) {  // the grad_out-weighted suffix behind this splat, and d L / d (a G) from it (as POLY)
        R2[jp] = fma2(-w2[jp], gyx2[jp], R2[jp]);
        pAG2[jp] = fma2(gyx2[jp], Tr2[jp], -(R2[jp] * inv1m2[jp]));
      }
      }  // !POLY
      // mean2d (2) | cov2d (4) | alpha (1): kernel_gaussian_2d_backward (kernels.h:394-418), packed over the pair
      // and summed over the lane's pairs; the two halves are added at the end
      const float inv_det = r_p1;
      v2f gm0 = {0.f, 0.f}, gm1 = gm0, gc0 = gm0, gc1 = gm0, gc3 = gm0, gal = gm0;
#pragma unroll
      for (int jp = 0; jp < NP; ++jp) {
        const v2f pa = pAG2[jp];  // (its uses below are products with the masked G / a G)
        const v2f gg = pa * ag2[jp];
        if constexpr (MOM) {  // (gm0, gm1 | gc0, gc1, gc3) hold the moments (Mx, My | Mxx, Mxy, Myy)
          const v2f t_ = gg * tx2[jp], s_ = gg * ty2[jp];
          gm0 = jp == 0 ? t_ : gm0 + t_;
          gm1 = jp == 0 ? s_ : gm1 + s_;
          gc0 = jp == 0 ? t_ * tx2[jp] : fma2(t_, tx2[jp], gc0);
          gc1 = jp == 0 ? t_ * ty2[jp] : fma2(t_, ty2[jp], gc1);
          gc3 = jp == 0 ? s_ * ty2[jp] : fma2(s_, ty2[jp], gc3);
        } else {
          const v2f vx = (splat2(x * r_c3) - y2[jp] * splat2(r_c2)) * splat2(inv_det);
          const v2f vy = (y2[jp] * splat2(r_c0) - splat2(x * r_c1)) * splat2(inv_det);
          gm0 = fma2(gg, vx, gm0);
          gm1 = fma2(gg, vy, gm1);
          const v2f h = splat2(0.5f) * gg;
          const v2f hvx = h * vx;
          gc0 = fma2(hvx, vx, gc0);
          gc1 = fma2(hvx, vy, gc1);
          gc3 = fma2(h * vy, vy, gc3);
        }
        gal = fma2(pa, G2[jp], gal);
        Tr2[jp] = Tr2[jp] * one_minus2(ag2[jp]);  // T (1 - a G) if it contributed (explicit: as the forward)
      }
      const float m0 = gm0[0] + gm0[1], m1 = gm1[0] + gm1[1];
      const float c0 = gc0[0] + gc0[1], c1 = gc1[0] + gc1[1], c3 = gc3[0] + gc3[1];
      const float ga = gal[0] + gal[1];
      if constexpr (POLY) {
        // The six DISTINCT geometric components (grad_cov[1] and grad_cov[2] receive the same value, kernels.h:414-415) ride
        // in the two spare slots of the three channels' 8-wide reductions: (m0, m1) | (c0, c1) | (c3, alpha).  No fourth
        // reduction (363 -> 348 vector instructions per (wavefront, entry) when it went in; 321 now).
        v2f t0[4] = {pch[0][0], pch[0][1], pch[0][2], v2f{m0, m1}};
        v2f t1[4] = {pch[1][0], pch[1][1], pch[1][2], v2f{c0, c1}};
        v2f t2[4] = {pch[2][0], pch[2][1], pch[2][2], v2f{c3, ga}};
        const float s0 = wave_reduce_scatter2_rows<8>(t0), s1 = wave_reduce_scatter2_rows<8>(t1), s2 = wave_reduce_scatter2_rows<8>(t2);
        // (the fourth quad lane receives channel 1's vector again: its (c0, c1) slot pays for grad_cov[2] = grad_cov[1])
        const float tot = quad_reduce_scatter4(s0, s1, s2, s1);
        const int m = lane & 3;  // the vector this lane ends up with: channel 0 / 1 / 2 / 1 again
        const int comp = scatter_comp<8>(lane);
        const bool owner = scatter_rows_owner<8>(lane);
        const size_t id = (size_t)S.id[g];
        // d L / d sh[c][k] = sum_r gw[c][r] V[r][k]: the 18 reduced values go through LDS, lanes (c, k) = (lane / 16,
        // lane % 16) expand them with their column of V (one wavefront per workgroup: the barrier is a wait)
        if (owner && m < 3 && comp < kPolyNB) gw_s[m * 8 + comp] = tot;
        // the geometric components, ONE atomic instruction (geo_base / geo_stride: the lane's slot, fixed before the loop)
        if (geo_base != nullptr) atomicAdd(geo_base + id * geo_stride, tot);
        __syncthreads();
        if (lane < 48) {
          const float *gw = &gw_s[(lane >> 4) * 8];
          const float *vk = &Vs[lane & 15];  // column (lane & 15) of V, from LDS as the sums are: six registers less across the loop
          float acc = gw[0] * vk[0];
#pragma unroll
          for (int r = 1; r < kPolyNB; ++r) acc = fmaf(gw[r], vk[r * 16], acc);
          atomicAdd(p.g_col + (size_t)TR::NCOL * id + lane, acc);  // (c, k) -> 16 c + k = lane
        }
        __syncthreads();  // gw_s is consumed before the next splat overwrites it
      } else {
        // grad_cov[1] and grad_cov[2] receive the same value (kernels.h:414-415)
        // (MOM: (Mx, My) (Mxx, Mxy) (Myy, alpha) -> mean[0..1], cov[0..1], cov[2], alpha)
        v2f ex[4] = {v2f{m0, m1}, v2f{c0, c1}, MOM ? v2f{c3, ga} : v2f{c1, c3}, MOM ? v2f{0.0f, 0.0f} : v2f{ga, 0.0f}};
        const float exsum = wave_reduce_scatter2_rows<8>(ex);
        const float tot = quad_reduce_scatter4(chsum[0], chsum[1], chsum[2], exsum);
        const int m = lane & 3;  // the vector this lane ends up with: channel 0 / 1 / 2 / geometric
        const size_t id = (size_t)S.id[g];
        float *dst = nullptr;
        if (m < 3) {
          const int k = scatter_comp<PCH>(lane);
          if (scatter_rows_owner<PCH>(lane) && k < TR::CC) dst = p.g_col + (size_t)TR::NCOL * id + m * TR::CC + k;
        } else if (m == 3 && scatter_rows_owner<8>(lane)) {
          const int e = scatter_comp<8>(lane);
          if (e < 2) dst = p.g_mean + 2 * id + e;
          else if (e < (MOM ? 5 : 6)) dst = p.g_cov + 4 * id + (e - 2);
          else if (e == (MOM ? 5 : 6)) dst = p.g_alpha + id;
        }
        if (dst != nullptr) atomicAdd(dst, tot);
      }
    }
    bool any_alive = false;
#pragma unroll
    for (int j = 0; j < PPL; ++j) any_alive |= alive(j);
    if (__syncthreads_or((int)any_alive) == 0) break;
  }
}
// (the persistent fallback runs at 152 registers, the exact kernel at 148: three wavefronts per SIMD.  It only renders views whose
// coefficient bound fails while a batch mate's holds.)
template <int CB, int PPL, bool BATCH = false, int NB = 0, bool MOM = false>
__global__ void __launch_bounds__(256 / PPL)
GS_WAVES_PER_EU((NB == kPolyNB && BATCH) ? 4 : 1)  // the batched polynomial backward: four wavefronts per SIMD (<= 128 registers)
k_composite_bwd_sh_vec(CompParams p_arg, ViewPack<BATCH> pack) {
  static_assert(!MOM || (BATCH && NB != kRouted), "the moment form exists for the batched launches");
  const CompParams *plist = pack.table();  // (kernel-argument memory: scalar loads, no table in device memory)
  uint32_t bid = blockIdx.x, grid = gridDim.x;
  if constexpr (NB == kRouted) {
    static_assert(CB == 4 && PPL == 4, "routing exists for SH degree 3, one wavefront per tile");
    union Shared {
      BwdShVecShared<4, 4, true> poly;
      BwdShVecShared<4, 4, false> exact;
    };
    __shared__ Shared sm;
    // The decision reads three words of the view's parameters; each form then takes its own copy of the block, so that
    // neither carries the scalar registers of the other's fields across the branch (merged naively, the batched
    // instantiation ran out of scalar registers and came out ONE vector register over the 168 of three wavefronts per SIMD).
    const CompParams *pp = BATCH ? &plist[batch_view(p_arg, bid, &grid)] : &p_arg;
    // (segmented launches: workgroup = (tile, segment), tile index = bid modulo the view's tile grid)
    const uint32_t tiles_grid = grid / (uint32_t)(pp->nseg > 1 ? pp->nseg : 1);
    const bool poly = routed_tile_is_polynomial(*pp, bid % tiles_grid);
    if (poly) {
      CompParams p = *pp;
      p.sh_rows = nullptr;  // (as the forward)
      composite_bwd_sh_vec_tile<4, 4, kPolyNB>(p, bid, grid, sm.poly);
    } else {
      const CompParams p = *pp;
      composite_bwd_sh_vec_tile<4, 4, 0>(p, bid, grid, sm.exact);
    }
  } else if constexpr (NB == kFallback) {  // see k_composite_fwd_sh_vec
    static_assert(CB == 4 && PPL == 4 && BATCH, "the persistent exact fallback of a bounded batch");
    __shared__ BwdShVecShared<4, 4, false> sm;
    const uint32_t B = (uint32_t)p_arg.n_lo, total = p_arg.vgrid;
    const bool per_tile = plist[0].sh_rows != nullptr;
    if (!per_tile) {
      bool any = false;
      for (uint32_t v = 0; v < B; ++v) any |= !poly_route(plist[v].sh_bound, plist[v].psx, plist[v].psy);
      if (!any) return;
    }
    const uint32_t per = total / B;  // camera-major
    const uint32_t tiles_grid = per / (uint32_t)(plist[0].nseg > 1 ? plist[0].nseg : 1);
    uint32_t base = 0;  // first block of the view b lies in (b only grows: no division in the loop)
    uint32_t view = 0;
    for (uint32_t b = blockIdx.x; b < total; b += gridDim.x) {
      while (b >= base + per) { base += per; ++view; }
      const CompParams *pp = &plist[view];
      if (per_tile ? !tile_flagged(*pp, (b - base) % tiles_grid) : poly_route(pp->sh_bound, pp->psx, pp->psy)) continue;
      const CompParams p = *pp;
      composite_bwd_sh_vec_tile<4, 4, 0, true, MOM>(p, b - base, per, sm);
      __syncthreads();
    }
  } else if constexpr (NB == kPolyNB && BATCH) {
    __shared__ BwdShVecShared<4, 4, true> sm;
    const CompParams *pp = &plist[batch_view(p_arg, bid, &grid)];
    const bool view_ok = pp->sh_bound != nullptr && poly_route(pp->sh_bound, pp->psx, pp->psy);
    if (pp->sh_rows == nullptr && !view_ok) return;  // per-view routing: this view is the exact fallback's
    CompParams p = *pp;
    if (view_ok) p.sh_rows = nullptr;  // (as the forward: the same device value, the same decision)
    composite_bwd_sh_vec_tile<4, 4, kPolyNB, false, MOM>(p, bid, grid, sm);
  } else {
    const CompParams p = BATCH ? plist[batch_view(p_arg, bid, &grid)] : p_arg;  // see k_composite_fwd
    __shared__ BwdShVecShared<CB, PPL, (NB > 0)> sm;
    composite_bwd_sh_vec_tile<CB, PPL, NB, false, MOM>(p, bid, grid, sm);
  }
}

// ============================================================================================
// forward, post-activation channels, packed per-pixel arithmetic (batched RGB + heads)
// ============================================================================================
// k_composite_fwd<MODE_RGBD> at 2 pixels per lane issues ~100 vector instructions per (wavefront, list entry), two
// wavefronts per tile; with the lane's 4 pixels as two packed pairs, one wavefront per tile, the record in scalar
// registers and one wave-uniform guard branch it is ~70 per tile and entry.  Same structure as k_composite_fwd_sh_vec.
template <int MODE, bool BATCH = false>
__global__ void __launch_bounds__(64) GS_WAVES_PER_EU(6)  // RGB + heads: 98 -> 80 registers (one dword spilled outside the entry loop): six wavefronts per SIMD
k_composite_fwd_chan_vec(CompParams p_arg, ViewPack<BATCH> pack) {
  const CompParams *plist = pack.table();  // (kernel-argument memory: scalar loads, no table in device memory)
  static_assert(MODE == MODE_RGB || MODE == MODE_SCALAR || MODE == MODE_RGBD, "post-activation channel modes");
  uint32_t bid = blockIdx.x;
  const CompParams p = BATCH ? view_params(plist[batch_view(p_arg, bid)]) : p_arg;
  using TR = Traits<MODE, 1>;
  constexpr int PPL = 4, NT = 64, ROWS = NT / 16, NP = PPL / 2;
  constexpr int NCH = TR::NCH;
  __shared__ Stage<MODE, 1> S;

  int tx, ty;
  if (!block_tile(p, tx, ty, bid)) return;  // uniform over the workgroup
  const int tile = ty * p.ntw + tx;
  const int st = p.start[tile];
  const int n = (st < 0) ? 0 : (p.end[tile] - st);
  const int t = (int)threadIdx.x;
  const int lx = t & 15, ly0 = t >> 4;
  const int gx = tx * kTile + lx;
  float bgv[3] = {0.0f, 0.0f, 0.0f};  // the view's background colour (RGB + heads, gsgen_rgbd_view::bg_rgb), or zeros
  if constexpr (MODE == MODE_RGBD) {
    if (p.bg != nullptr) { bgv[0] = p.bg[0]; bgv[1] = p.bg[1]; bgv[2] = p.bg[2]; }
  }
  if (st == kListOverflow) {  // the frame's pairs did not fit the list (GSGEN_LIST_OVERFLOW): never a finite blank tile
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      const int gyj = ty * kTile + ly0 + j * ROWS;
      if (gx < p.W && gyj < p.H) poison_pixel<NCH>(p, (size_t)gyj * p.W + gx);
    }
    return;
  }
  if (n == 0) {  // uniform over the workgroup
    // per-camera entry points: the caller's pre-initialised out / T stand (vol_render.h:1006-1013); batched launches that
    // ask for it write the empty tile themselves (CompParams::fill_empty)
    if (p.fill_empty) {
#pragma unroll
      for (int j = 0; j < PPL; ++j) {
        const int gyj = ty * kTile + ly0 + j * ROWS;
        if (gx < p.W && gyj < p.H) {
          const size_t pix = (size_t)gyj * p.W + gx;
          float e[NCH];
#pragma unroll
          for (int c = 0; c < NCH; ++c) e[c] = (MODE == MODE_RGBD && c < 3) ? bgv[c] : 0.0f;  // (T = 1: the background)
          store_heads<MODE, NCH>(p, pix, e);
          if (p.T != nullptr) p.T[pix] = 1.0f;
        }
      }
    }
    return;
  }
  const float px = pixel_coord(p.topleft[0], gx, p.psx);

  bool valid[PPL];
  int gy[PPL];
  v2f py2[NP], acc2[NP][NCH], Tr2[NP];
#pragma unroll
  for (int j = 0; j < PPL; ++j) {
    gy[j] = ty * kTile + ly0 + j * ROWS;
    valid[j] = (gx < p.W) && (gy[j] < p.H);
    py2[j >> 1][j & 1] = pixel_coord(p.topleft[1], gy[j], p.psy);
    Tr2[j >> 1][j & 1] = valid[j] ? 1.0f : -1.0f;  // alive = (T >= thresh); pixels outside never are, never written
  }
#pragma unroll
  for (int jp = 0; jp < NP; ++jp)
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc2[jp][c] = v2f{0.0f, 0.0f};
  auto alive = [&](int j) { return !(Tr2[j >> 1][j & 1] < p.thresh); };

  for (int base = 0; base < n; base += kBatch) {
    const int nb = min(kBatch, n - base);
    if (base > 0) __syncthreads();
    stage_batch<MODE, 1, NT>(S, p, st + base, nb);
    __syncthreads();

    for (int g = 0; g < nb; ++g) {
      bool any_alive = false;
#pragma unroll
      for (int j = 0; j < PPL; ++j) any_alive |= alive(j);
      if (!wave_any(any_alive)) break;  // the tile's 256 pixels are saturated

      // (round 6: the record and the channel values as plain vector registers -- LDS broadcasts -- as in the backward: twelve
      // v_readfirstlane per entry less, still 80 registers; forward alone 0.247 -> 0.220 ms per 8 cfg2 views, profiles/r06_s2_*)
      const float r_mx = S.mx[g], r_my = S.my[g], r_a = S.a[g], r_p0 = S.p0[g], r_p1 = S.p1[g], r_p2 = S.p2[g];
      const float p0x = r_p0 * (px - r_mx);
      v2f G2[NP], ag2[NP];
      bool any_con = false;
      float guard_dist = 0.0f;  // the lane's smallest distance to the threshold (dead pixels included: a spurious trip re-tests per pixel)
#pragma unroll
      for (int jp = 0; jp < NP; ++jp) {
        G2[jp] = gauss_chol_pair(p0x, r_p1, r_p2, py2[jp] - splat2(r_my));
        ag2[jp] = splat2(r_a) * G2[jp];
        const v2f dist = ag2[jp] - splat2(kMinAlpha);
        const float dmin = fminf(fabsf(dist[0]), fabsf(dist[1]));
        guard_dist = jp == 0 ? dmin : fminf(guard_dist, dmin);
      }
      const bool any_guard = guard_dist <= kMinAlpha * kGuardTol;
      if (wave_any(any_guard)) {  // within rounding of the skip threshold: the reference's arithmetic decides
        float r_c0, r_c1, r_c2, r_c3;
        guard_cov(S, p, g, r_c0, r_c1, r_c2, r_c3);
#pragma unroll
        for (int jp = 0; jp < NP; ++jp)
#pragma unroll
          for (int k = 0; k < 2; ++k)
            if (alive(2 * jp + k) && fabsf(ag2[jp][k] - kMinAlpha) <= kMinAlpha * kGuardTol) {
              G2[jp][k] = gauss_ref_f64(r_mx, r_my, r_c0, r_c1, r_c2, r_c3, px, py2[jp][k]);
              ag2[jp][k] = r_a * G2[jp][k];
            }
      }
#pragma unroll
      for (int jp = 0; jp < NP; ++jp)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const bool con = alive(2 * jp + k) && !(ag2[jp][k] < kMinAlpha);
          G2[jp][k] = con ? G2[jp][k] : 0.0f;
          any_con |= con;
        }
#pragma unroll
      for (int jp = 0; jp < NP; ++jp) ag2[jp] = splat2(r_a) * G2[jp];  // the same products from the masked G (bit-identical where
                                                                       // the pixel takes part, 0 elsewhere): no second select
      if (!wave_any(any_con)) continue;  // nobody in the wave sees this Gaussian

      const float *cg = &S.col[g * TR::NCOLP];
#pragma unroll
      for (int jp = 0; jp < NP; ++jp) {
        const v2f w2 = (splat2(r_a) * Tr2[jp]) * G2[jp];  // (a T) G, or 0
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          if (MODE == MODE_RGBD && c == 4) continue;  // the opacity head's value is 1: sum w = 1 - T telescopes, formed in the epilogue
          acc2[jp][c] = ffma2(splat2(cg[c]), w2, acc2[jp][c]);
        }
        Tr2[jp] = Tr2[jp] * one_minus2(ag2[jp]);  // T (1 - a G) if it contributed (explicit)
      }
    }
    bool any_alive = false;
#pragma unroll
    for (int j = 0; j < PPL; ++j) any_alive |= alive(j);
    if (__syncthreads_or((int)any_alive) == 0) break;  // whole tile saturated: stop staging
  }

#pragma unroll
  for (int j = 0; j < PPL; ++j) {
    if (!valid[j]) continue;
    const size_t pix = (size_t)gy[j] * p.W + gx;
    // opacity (gs/gaussian_splatting.py:1352-1373: render_scalar of ones) = sum_i w_i = 1 - T: T_i+1 = T_i - w_i exactly in the
    // reals, to fp32 rounding here (<= 2e-7 of the oracle's sum, tests/test_gpu_parity.py) -- one accumulator pair per pixel pair
    // and two packed FMAs per entry less
    if constexpr (MODE == MODE_RGBD) acc2[j >> 1][4][j & 1] = 1.0f - Tr2[j >> 1][j & 1];
    float e[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) e[c] = acc2[j >> 1][c][j & 1];
    if constexpr (MODE == MODE_RGBD) heads_epilogue(p, bgv, e, Tr2[j >> 1][j & 1]);
    store_heads<MODE, NCH>(p, pix, e);
    if (p.T != nullptr) p.T[pix] = Tr2[j >> 1][j & 1];
  }
}

// ============================================================================================
// backward, post-activation channels (RGB, scalar, RGB + heads), packed per-pixel arithmetic
// ============================================================================================
// The trainer's default outputs (gs/gaussian_splatting.py:1304-1416: rgb + depth + opacity + depth^2 from
// post-activation colours) go through MODE_RGBD.  k_composite_bwd_pixel spends ~300 vector instructions per
// (wavefront, list entry) on it at 2 pixels per lane -- a third of them register moves around its per-pixel branches.
// This is k_composite_bwd_sh_vec without the SH part: one wavefront per tile, 4 pixels per lane as two pixel PAIRS,
// every per-pixel quantity packed, contribution mask folded into G and a G, `alive` = (T >= thresh), one wave-uniform
// guard branch, the NCH + 7 gradient components reduced as (even, odd) pairs.
// Round 4 (the trainer's default path measured in the driver line): as in the SH kernels the suffix colour enters
// d L / d (a G) only as  sum_c grad_out_c * suffix_c,  so ONE running grad_out-weighted suffix per pixel replaces the NCH
// per-channel suffixes (6 -> 1 register pairs per pixel pair on RGB + heads, 4 packed operations per channel and pixel pair ->
// 2); the atomics' target is base(lane) + id * stride(lane), both fixed per lane before the list walk (the select chain
// over four arrays per entry compiled to nested exec-mask branches and a flat atomic).
// Reduction vector: channels [0, NCH) | pad to even | mean 2 | cov 4 | alpha 1.
// MOM (round 6; RGB + heads, the batched launch of gsgen_vol_render_rgbd_backward_batch_moments): the MOMENT form of the
// gradients.  Every geometric gradient of a splat is a moment of ONE per-pixel weight g = d L / d (a G) * a G against the
// whitened offsets (u, v) = (p0 x + p1 y, p2 y) the Gaussian is evaluated through anyway (gauss_chol_pair):
//   d mean2d = sum g Sigma^-1 d,  d cov2d = 0.5 sum g (Sigma^-1 d)(Sigma^-1 d)^T,  Sigma^-1 d = (k0 u, k1 u + k2 v)
// are linear in (Mu, Mv) = sum g (u, v) and (Muu, Muv, Mvv) = sum g (u u, u v, v v): the kernel accumulates those five (14
// packed operations per entry instead of 26 + 3) and the projection backward, one thread per (view, Gaussian), expands them
// (geometry.hip, moments_to_grads).  The three depth heads' channels (d, 1, d d) fold into ONE gradient component,
// d L / d depth = sum w (go_d + 2 d go_dd) (the "1" channel's gradient has no consumer), and their part of the suffix weight is
// go_o + d (go_d + d go_dd): two packed FMAs per pixel pair instead of three channels' six.  Ten components instead of
// thirteen cross the lanes, through a reduce-scatter shaped for ten (wave_reduce_scatter10: 8 lane swaps instead of 12).
// Component order: r g | b dd | Mu Mv | Muu Muv | Mvv alpha  ->  grad_chan6[.][0..3], grad_mean (as [N,2] moments),
// grad_cov (as [N,4]: three moments, the fourth float untouched), grad_alpha.
template <int MODE, bool BATCH = false, bool MOM = false>
__global__ void __launch_bounds__(64) GS_WAVES_PER_EU(5)  // RGB + heads: 106 -> 96 registers (16 bytes of scratch outside the entry loop): five per SIMD
k_composite_bwd_chan_vec(CompParams p_arg, ViewPack<BATCH> pack) {
  const CompParams *plist = pack.table();  // (kernel-argument memory: scalar loads, no table in device memory)
  static_assert(MODE == MODE_RGB || MODE == MODE_SCALAR || MODE == MODE_RGBD, "post-activation channel modes");
  static_assert(!MOM || MODE == MODE_RGBD, "the moment form exists for RGB + heads");
  uint32_t bid = blockIdx.x, grid = gridDim.x;
  const CompParams p = BATCH ? view_params(plist[batch_view(p_arg, bid, &grid)]) : p_arg;
  (void)grid;
  using TR = Traits<MODE, 1>;
  constexpr int PPL = 4, NT = 64, ROWS = NT / 16, NP = PPL / 2;
  constexpr int NCH = TR::NCH;
  constexpr int G0 = (NCH + 1) & ~1;  // first geometric component
  constexpr int P = 16;
  static_assert(G0 + 7 <= P, "component layout");
  __shared__ Stage<MODE, 1> S;

  int tx, ty;
  if (!block_tile(p, tx, ty, bid)) return;  // uniform over the workgroup
  const int tile = ty * p.ntw + tx;
  const int st = p.start[tile];
  const int n = (st < 0) ? 0 : (p.end[tile] - st);
  const int t = (int)threadIdx.x;
  const int lane = t & 63;
  const int lx = t & 15, ly0 = t >> 4;
  const int gx = tx * kTile + lx;
  if (n == 0 || n < p.n_lo || n >= p.n_hi) {
    if constexpr (MODE == MODE_RGBD) {
      // an EMPTY tile still shows the background (T = 1): its pixels' share of d L / d bg (gs/renderer.py:1283)
      if (st == -1 && p.g_bg != nullptr && p.T != nullptr && p.go_rgb != nullptr) {
        float sb[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int gy = ty * kTile + ly0 + j * 4;
          if (gx < p.W && gy < p.H) {
            const size_t pix = (size_t)gy * p.W + gx;
            const float Tf = p.T[pix];
#pragma unroll
            for (int c = 0; c < 3; ++c) sb[c] += nan_to_num_f(p.go_rgb[3 * pix + c] * Tf);
          }
        }
        bg_grad_atomics(p, tile, lane, sb);
      }
    }
    return;
  }
  const float px = pixel_coord(p.topleft[0], gx, p.psx);

  // R2 = sum_c grad_out_c * (final_c - prefix_c): the grad_out-weighted suffix colour behind the current splat
  v2f py2[NP], go2[NP][NCH], R2[NP], Tr2[NP];
  float bgs[3] = {0.0f, 0.0f, 0.0f};
  (void)bgs;
#pragma unroll
  for (int j = 0; j < PPL; ++j) {
    const int gy = ty * kTile + ly0 + j * ROWS;
    const bool valid = (gx < p.W) && (gy < p.H);
    py2[j >> 1][j & 1] = pixel_coord(p.topleft[1], gy, p.psy);
    const size_t pix = valid ? ((size_t)gy * p.W + gx) : 0;
    float r = 0.0f, gq[NCH], fin[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      gq[c] = valid ? load_grad_out<MODE, NCH>(p, pix, c) : 0.0f;
      fin[c] = valid ? load_final<MODE, NCH>(p, pix, c) : 0.0f;
    }
    if constexpr (MODE == MODE_RGBD) {
      if (p.g_bg != nullptr && p.T != nullptr) {  // d L / d bg = sum nan_to_num(grad_rgb * T) (gs/renderer.py:1283), T the forward's
        const float Tf = valid ? p.T[pix] : 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) bgs[c] += nan_to_num_f(gq[c] * Tf);
      }
      if (p.zvar) {  // the sixth head is z_var = depth2 - D^2, D the composited depth: d depth2 = g, d D -= 2 D g; final depth2 = z_var + D^2
        gq[3] = fmaf(-2.0f * fin[3], gq[5], gq[3]);
        fin[5] = fmaf(fin[3], fin[3], fin[5]);
      }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      go2[j >> 1][c][j & 1] = gq[c];
      r = fmaf(gq[c], fin[c], r);  // prefix = 0 in front of the list
    }
    R2[j >> 1][j & 1] = r;
    Tr2[j >> 1][j & 1] = valid ? 1.0f : -1.0f;  // alive = (T >= thresh); pixels outside never are
  }
  if constexpr (MODE == MODE_RGBD) {
    if (p.g_bg != nullptr && p.T != nullptr) {  // the tile's sums into slot (tile % 64): 64 addresses per view share the atomics
      bg_grad_atomics(p, tile, lane, bgs);
    }
  }
  auto alive = [&](int j) { return !(Tr2[j >> 1][j & 1] < p.thresh); };

  // where this lane's component of the reduced gradient goes: dst = base + id * stride (bytes); no target: base = 0
  const int comp = MOM ? scatter10_comp(lane) : scatter_comp<P>(lane);
  char *a_base = nullptr;
  uint32_t a_stride = 0;
  if constexpr (MOM) {
    if (scatter10_owner(lane)) {
      if (comp < 4) { a_base = reinterpret_cast<char *>(p.g_col + comp); a_stride = 4u * (uint32_t)TR::NCOL; }
      else if (comp < 6) { a_base = reinterpret_cast<char *>(p.g_mean + (comp - 4)); a_stride = 8u; }
      else if (comp < 9) { a_base = reinterpret_cast<char *>(p.g_cov + (comp - 6)); a_stride = 16u; }
      else { a_base = reinterpret_cast<char *>(p.g_alpha); a_stride = 4u; }
    }
  } else if (scatter_owner<P>(lane)) {
    if (comp < NCH) { a_base = reinterpret_cast<char *>(p.g_col + comp); a_stride = 4u * (uint32_t)TR::NCOL; }
    else if (comp >= G0 && comp < G0 + 2) { a_base = reinterpret_cast<char *>(p.g_mean + (comp - G0)); a_stride = 8u; }
    else if (comp >= G0 + 2 && comp < G0 + 6) { a_base = reinterpret_cast<char *>(p.g_cov + (comp - G0 - 2)); a_stride = 16u; }
    else if (comp == G0 + 6) { a_base = reinterpret_cast<char *>(p.g_alpha); a_stride = 4u; }
  }

  for (int base = 0; base < n; base += kBatch) {
    const int nb = min(kBatch, n - base);
    if (base > 0) __syncthreads();
    stage_batch<MODE, 1, NT>(S, p, st + base, nb);
    __syncthreads();

    for (int g = 0; g < nb; ++g) {
      bool any_alive = false;
#pragma unroll
      for (int j = 0; j < PPL; ++j) any_alive |= alive(j);
      if (!wave_any(any_alive)) break;

      // (the record as plain vector registers -- LDS broadcasts: the single-suffix form leaves room for them, and the six
      // v_readfirstlane + six more for the channel values per entry are gone)
      const float r_mx = S.mx[g], r_my = S.my[g], r_a = S.a[g], r_p0 = S.p0[g], r_p1 = S.p1[g], r_p2 = S.p2[g];
      const float x = px - r_mx;
      // G2 / ag2: the Gaussian (gauss_eval's Cholesky form) and a G, ZEROED where the pixel does not take part
      v2f y2[NP], G2[NP], ag2[NP], u2[NP], v2[NP];
      bool any_con = false;
      float guard_dist = 0.0f;  // (as the forward: one distance per lane, dead pixels included)
      const float p0x = r_p0 * x;
#pragma unroll
      for (int jp = 0; jp < NP; ++jp) {
        y2[jp] = py2[jp] - splat2(r_my);
        G2[jp] = gauss_chol_pair(p0x, r_p1, r_p2, y2[jp], u2[jp], v2[jp]);
        ag2[jp] = splat2(r_a) * G2[jp];
        const v2f dist = ag2[jp] - splat2(kMinAlpha);
        const float dmin = fminf(fabsf(dist[0]), fabsf(dist[1]));
        guard_dist = jp == 0 ? dmin : fminf(guard_dist, dmin);
      }
      const bool any_guard = guard_dist <= kMinAlpha * kGuardTol;
      if (wave_any(any_guard)) {  // within rounding of the skip threshold: the reference's arithmetic decides
        float r_c0, r_c1, r_c2, r_c3;
        guard_cov(S, p, g, r_c0, r_c1, r_c2, r_c3);
#pragma unroll
        for (int jp = 0; jp < NP; ++jp)
#pragma unroll
          for (int k = 0; k < 2; ++k)
            if (alive(2 * jp + k) && fabsf(ag2[jp][k] - kMinAlpha) <= kMinAlpha * kGuardTol) {
              G2[jp][k] = gauss_ref_f64(r_mx, r_my, r_c0, r_c1, r_c2, r_c3, px, py2[jp][k]);
              ag2[jp][k] = r_a * G2[jp][k];
            }
      }
#pragma unroll
      for (int jp = 0; jp < NP; ++jp)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const bool con = alive(2 * jp + k) && !(ag2[jp][k] < kMinAlpha);
          G2[jp][k] = con ? G2[jp][k] : 0.0f;
          any_con |= con;
        }
#pragma unroll
      for (int jp = 0; jp < NP; ++jp) ag2[jp] = splat2(r_a) * G2[jp];  // the same products from the masked G (bit-identical where
                                                                       // the pixel takes part, 0 elsewhere): no second select
      if (!wave_any(any_con)) continue;  // nobody in the wave sees this Gaussian

      const float *cg = &S.col[g * TR::NCOLP];
      v2f w2[NP], om2[NP], gy2[NP];
#pragma unroll
      for (int jp = 0; jp < NP; ++jp) {
        w2[jp] = MOM ? Tr2[jp] * ag2[jp] : (splat2(r_a) * Tr2[jp]) * G2[jp];  // the forward's (a T) G -- MOM: T (a G), equal to an ulp --, or 0
        om2[jp] = one_minus2(ag2[jp]);
      }
      if constexpr (MOM) {
        v2f c5[5];
        float gch[4];
        // the three depth heads first: their share of  sum_c grad_out_c value_c  is  go_o + d (go_d + d go_dd),  their one gradient
        // component  sum w (go_d + 2 d go_dd)
        const float d = cg[3], dd2 = d + d;
        v2f gacc;
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) {
          gy2[jp] = fma2(splat2(d), fma2(splat2(d), go2[jp][5], go2[jp][3]), go2[jp][4]);
          const v2f h = fma2(splat2(dd2), go2[jp][5], go2[jp][3]);
          gacc = jp == 0 ? w2[jp] * h : fma2(w2[jp], h, gacc);
        }
        gch[3] = add_scalar(gacc[0], gacc[1]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const v2f val = splat2(cg[c]);
#pragma unroll
          for (int jp = 0; jp < NP; ++jp) {
            gacc = jp == 0 ? w2[jp] * go2[jp][c] : fma2(w2[jp], go2[jp][c], gacc);  // d / d (channel value)
            gy2[jp] = fma2(go2[jp][c], val, gy2[jp]);                                 // sum_c grad_out_c * value_c
          }
          gch[c] = add_scalar(gacc[0], gacc[1]);
        }
        v2f Mu, Mv, Muu, Muv, Mvv, gal;
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) {
          // the suffix behind this splat, grad_out-weighted, and d L / d (a G) from it (vol_render.h:379-409)
          R2[jp] = fma2(-w2[jp], gy2[jp], R2[jp]);
          const v2f inv1m = v2f{__builtin_amdgcn_rcpf(om2[jp][0]), __builtin_amdgcn_rcpf(om2[jp][1])};
          const v2f pAG = fma2(gy2[jp], Tr2[jp], -(R2[jp] * inv1m));
          const v2f gg = pAG * ag2[jp];
          const v2f t = gg * u2[jp], sv = gg * v2[jp];
          Mu = jp == 0 ? t : Mu + t;
          Mv = jp == 0 ? sv : Mv + sv;
          Muu = jp == 0 ? t * u2[jp] : fma2(t, u2[jp], Muu);
          Muv = jp == 0 ? t * v2[jp] : fma2(t, v2[jp], Muv);
          Mvv = jp == 0 ? sv * v2[jp] : fma2(sv, v2[jp], Mvv);
          gal = jp == 0 ? pAG * G2[jp] : fma2(pAG, G2[jp], gal);
          Tr2[jp] = Tr2[jp] * om2[jp];  // T (1 - a G) if it contributed (as the forward: 1 - round(a G))
        }
        // wave_reduce_scatter10's order: (r g) (b dd) (Mvv alpha) (Mu Mv) (Muu Muv)
        c5[0] = v2f{gch[0], gch[1]};
        c5[1] = v2f{gch[2], gch[3]};
        c5[2] = v2f{add_scalar(Mvv[0], Mvv[1]), add_scalar(gal[0], gal[1])};
        c5[3] = v2f{add_scalar(Mu[0], Mu[1]), add_scalar(Mv[0], Mv[1])};
        c5[4] = v2f{add_scalar(Muu[0], Muu[1]), add_scalar(Muv[0], Muv[1])};
        const float tot = wave_reduce_scatter10(c5);
        if (a_base != nullptr) atomicAdd(reinterpret_cast<float *>(a_base + (size_t)(uint32_t)S.id[g] * a_stride), tot);
      } else {
      v2f gr2[P / 2];
#pragma unroll
      for (int i = 0; i < P / 2; ++i) gr2[i] = v2f{0.0f, 0.0f};
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const v2f val = splat2(cg[c]);
        v2f gacc;
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) {
          gacc = jp == 0 ? w2[jp] * go2[jp][c] : fma2(w2[jp], go2[jp][c], gacc);             // d / d (channel value)
          gy2[jp] = c == 0 ? go2[jp][c] * val : fma2(go2[jp][c], val, gy2[jp]);            // sum_c grad_out_c * value_c
        }
        gr2[c >> 1][c & 1] = add_scalar(gacc[0], gacc[1]);
      }
      // mean2d (2) | cov2d (4) | alpha (1): kernel_gaussian_2d_backward (kernels.h:394-418), packed over the pair
      // v = Sigma^-1 d from the Cholesky factor (see kInvSc2): no fp32 determinant
      const float k0 = kInvSc2 * r_p0, k1 = kInvSc2 * r_p1, k2 = kInvSc2 * r_p2;
      v2f gm0 = {0.f, 0.f}, gm1 = gm0, gc0 = gm0, gc1 = gm0, gc3 = gm0, gal = gm0;
#pragma unroll
      for (int jp = 0; jp < NP; ++jp) {
        // the suffix behind this splat, grad_out-weighted, and d L / d (a G) from it:
        //   sum_c grad_out_c (value_c T - suffix_c / (1 - a G))      (vol_render.h:379-409)
        R2[jp] = fma2(-w2[jp], gy2[jp], R2[jp]);
        const v2f inv1m = v2f{__builtin_amdgcn_rcpf(om2[jp][0]), __builtin_amdgcn_rcpf(om2[jp][1])};
        const v2f pAG = fma2(gy2[jp], Tr2[jp], -(R2[jp] * inv1m));
        const v2f gg = pAG * ag2[jp];
        const v2f u = fma2(splat2(r_p1), y2[jp], splat2(p0x));
        const v2f vx = splat2(k0) * u;
        const v2f vy = fma2(splat2(k1), u, (splat2(k2) * splat2(r_p2)) * y2[jp]);
        gm0 = fma2(gg, vx, gm0);
        gm1 = fma2(gg, vy, gm1);
        const v2f h = splat2(0.5f) * gg;
        const v2f hvx = h * vx;
        gc0 = fma2(hvx, vx, gc0);
        gc1 = fma2(hvx, vy, gc1);
        gc3 = fma2(h * vy, vy, gc3);
        gal = fma2(pAG, G2[jp], gal);
        Tr2[jp] = Tr2[jp] * om2[jp];  // T (1 - a G) if it contributed (as the forward: 1 - round(a G))
      }
      const float c1s = add_scalar(gc1[0], gc1[1]);
      gr2[G0 / 2 + 0] = v2f{add_scalar(gm0[0], gm0[1]), add_scalar(gm1[0], gm1[1])};
      gr2[G0 / 2 + 1] = v2f{add_scalar(gc0[0], gc0[1]), c1s};
      gr2[G0 / 2 + 2] = v2f{c1s, add_scalar(gc3[0], gc3[1])};  // grad_cov[1] and [2] receive the same value (kernels.h:414-415)
      gr2[G0 / 2 + 3] = v2f{add_scalar(gal[0], gal[1]), 0.0f};

      wave_reduce_scatter2<P>(gr2);
      if (a_base != nullptr) atomicAdd(reinterpret_cast<float *>(a_base + (size_t)(uint32_t)S.id[g] * a_stride), gr2[0][0]);
      }
    }
    bool any_alive = false;
#pragma unroll
    for (int j = 0; j < PPL; ++j) any_alive |= alive(j);
    if (__syncthreads_or((int)any_alive) == 0) break;
  }
}

// ---- launch helpers ---------------------------------------------------------------------------
// One shape per job, chosen by measurement on MI355X (profiles/r01_notes.md .. r04_notes.md); the A/B shapes of rounds 1-3
// (pixels per lane 1 / 2 / 4 of every kernel, the unpacked backward for 16 x 16 tiles, the single-reduction SH backward, two
// interleaved batch orders, a run-time variant table behind gsgen_debug_set_variant) are gone from the tree with round 4:
//   per-camera forward, every mode          k_composite_fwd<MODE, CB, 1>: 4 wavefronts per tile (127 us vs 195 us at one: a lone
//                                           launch leaves 2.4 wavefronts per SIMD and ends on the centre tiles' serial chains)
//   per-camera forward, SH degree 3 + bound k_composite_fwd_sh_vec<4, 2, false, kRouted>: both forms of the basis in one kernel
//   per-camera backward                     one wavefront per tile, packed: k_composite_bwd_sh_vec<CB, 4> (routed with a bound),
//                                           k_composite_bwd_chan_vec<MODE> for RGB / scalar / RGB + heads
//   batched forward                         SH: k_composite_fwd_sh_vec<CB, 2, true> (two wavefronts per tile); with a bound the
//                                           polynomial kernel <4, 4, true, kPolyNB> + the persistent exact fallback;
//                                           RGB, RGB + heads: k_composite_fwd_chan_vec<MODE, true>
//   batched backward                        k_composite_bwd_sh_vec<CB, 4, true> (+ polynomial / fallback pair), chan_vec
//   tile sides 8 / 32 (`_gs` callers only)  the unpacked k_composite_fwd / k_composite_bwd_pixel at one fixed shape each
template <int MODE, int CB>
static int launch_fwd(const CompParams &p_, hipStream_t s) {
  CompParams p = p_;
  const uint32_t nblk = comp_grid(p);
  if (p.ntw * p.nth == 0) return 0;
  if (p.tile_side != 16) {  // a caller's own tile size (1 .. 32): the unpacked kernel at one wavefront (8) or four per tile (a 32 x 32 patch)
    if constexpr (MODE == MODE_RGBD) return GSGEN_EUNSUPPORTED;
    else {
      if (p.nseg > 1) return GSGEN_EUNSUPPORTED;
      p.sh_bound = p.sh_rows = nullptr;  // the polynomial basis is fitted to 16 x 16 tiles
      if (p.tile_side == 8) hipLaunchKernelGGL((k_composite_fwd<MODE, CB, 1, 8>), dim3(nblk), dim3(64), 0, s, p);
      else hipLaunchKernelGGL((k_composite_fwd<MODE, CB, 4, 32>), dim3(nblk), dim3(256), 0, s, p);
      return (int)hipGetLastError();
    }
  }
  if constexpr (MODE == MODE_SH && CB == 4) {
    // with the coefficient bound: ONE launch of the routed kernel -- every workgroup reads the bound and runs the polynomial
    // form of the per-pixel basis where its error bound holds, the exact form elsewhere (poly_route)
    if (p.sh_bound != nullptr || p.sh_rows != nullptr) {
      hipLaunchKernelGGL((k_composite_fwd_sh_vec<4, 2, false, kRouted>), dim3(nblk), dim3(128), 0, s, p, ViewPack<false>{});
      return (int)hipGetLastError();
    }
  } else {
    p.sh_bound = p.sh_rows = nullptr;
  }
  hipLaunchKernelGGL((k_composite_fwd<MODE, CB, 1>), dim3(nblk), dim3(256), 0, s, p);
  return (int)hipGetLastError();
}
template <int MODE, int CB>
static int launch_bwd(const CompParams &p_, hipStream_t s) {
  CompParams p = p_;
  if (p.n_hi == 0) p.n_hi = 0x7fffffff;
  const uint32_t nblk = comp_grid(p);
  if (p.ntw * p.nth == 0) return 0;
  if (p.tile_side != 16) {  // a caller's own tile size (1 .. 32): the unpacked kernel at one wavefront (8) or four per tile (a 32 x 32 patch)
    if constexpr (MODE == MODE_RGBD) return GSGEN_EUNSUPPORTED;
    else {
      if (p.nseg > 1) return GSGEN_EUNSUPPORTED;
      p.sh_bound = p.sh_rows = nullptr;
      if (p.tile_side == 8) hipLaunchKernelGGL((k_composite_bwd_pixel<MODE, CB, 1, 8>), dim3(nblk), dim3(64), 0, s, p);
      else hipLaunchKernelGGL((k_composite_bwd_pixel<MODE, CB, 4, 32>), dim3(nblk), dim3(256), 0, s, p);
      return (int)hipGetLastError();
    }
  }
  if constexpr (MODE != MODE_SH) {
    p.sh_bound = nullptr;
    hipLaunchKernelGGL((k_composite_bwd_chan_vec<MODE>), dim3(nblk), dim3(64), 0, s, p, ViewPack<false>{});
  } else {
    const uint32_t ng = nblk * (uint32_t)(p.nseg > 1 ? p.nseg : 1);
    if constexpr (CB == 4) {
      if (p.sh_bound != nullptr || p.sh_rows != nullptr) {  // as the forward: one launch, routed on the device
        hipLaunchKernelGGL((k_composite_bwd_sh_vec<4, 4, false, kRouted>), dim3(ng), dim3(64), 0, s, p, ViewPack<false>{});
        return (int)hipGetLastError();
      }
    } else {
      p.sh_bound = p.sh_rows = nullptr;
    }
    hipLaunchKernelGGL((k_composite_bwd_sh_vec<CB, 4>), dim3(ng), dim3(64), 0, s, p, ViewPack<false>{});
  }
  return (int)hipGetLastError();
}

// ---- the persistent exact fallback of a bounded batch: where it goes ---------------------------------
// It normally has nothing to do, but its workgroups (152 registers in the backward) must each FIND ROOM on a chip that other
// batches' compositing launches fill.  Measured on one box (profiles/r03_ab_fallback_order.txt, renders/s on the driver
// command): no fallback at all (views beyond the bound would be lost) 4 993; fallback IN FRONT of the polynomial kernel on the
// caller's stream 4 959 (it finds room while the previous stage of the chain drains); behind it 4 919; on a side stream forked
// from and joined into the caller's stream with events -- meant to hide the wait behind the polynomial kernel -- 4 725 (a fourth
// and fifth hardware queue in play changes the arbitration for the worse).  Hence: in front, same stream, no extra objects.
template <class Side, class Main>
static void launch_beside(hipStream_t s, Side &&side, Main &&main) {
  side(s);
  main(s);
}

// ---- batched cameras: parameters through the kernel arguments -----------------------------------
static ViewPack<true> make_pack(const CompParams *host, uint32_t n) {
  ViewPack<true> pack{};
  for (uint32_t i = 0; i < n; ++i) pack.v[i] = host[i];
  return pack;
}
// f(first view's block with n_lo = the chunk's view count, the chunk's pack, the chunk's view count) per chunk of <= kPackMax views
template <class F>
static void for_each_chunk(const CompParams *host, uint32_t B, F &&f) {
  for (uint32_t b0 = 0; b0 < B; b0 += (uint32_t)kPackMax) {
    const uint32_t n = (B - b0) < (uint32_t)kPackMax ? (B - b0) : (uint32_t)kPackMax;
    CompParams a = host[b0];
    a.n_lo = (int)n;  // (the batched grids are camera-major: batch_view)
    f(a, make_pack(host + b0, n), n);
  }
}
template <int CB>
static void launch_fwd_sh_batch_c(const CompParams &p0, const ViewPack<true> &plist, uint32_t B, uint32_t nblk, hipStream_t s, bool bounded) {
  const dim3 g(nblk * B);
  if constexpr (CB == 4) {
    // The views carry the device address of the coefficient bound: every workgroup decides from the bound and ITS view's pixel
    // size which form renders the view (poly_route; forward and backward read the same value, hence agree).
    if (bounded) {
      // the persistent exact fallback for the views beyond the bound (10 two-wavefront workgroups per compute unit at most), then
      // the polynomial kernel over the whole grid (one wavefront per tile: 96 registers, 5 per SIMD -- 5 020 vs 4 833 renders/s
      // against two wavefronts per tile in round 3, 5 448 vs 5 271 in round 4)
      CompParams pf = p0;
      pf.vgrid = nblk * B;
      const uint32_t gf = pf.vgrid < 2560u ? pf.vgrid : 2560u;
      if (p0.sh_rows != nullptr) {  // per-tile routing: the polynomial kernel flags the tiles the fallback BEHIND it renders
        if (p0.stop == nullptr) hipLaunchKernelGGL((k_composite_fwd_sh_vec<4, 4, true, kPolyNB, false>), g, dim3(64), 0, s, p0, plist);
        else hipLaunchKernelGGL((k_composite_fwd_sh_vec<4, 4, true, kPolyNB>), g, dim3(64), 0, s, p0, plist);
        // (no_fallback: the caller's earlier batches reported no crowded tile -- nothing is handed over, nothing is launched)
        if (!p0.no_fallback) hipLaunchKernelGGL((k_composite_fwd_sh_vec<4, 2, true, kFallback>), dim3(gf), dim3(128), 0, s, pf, plist);
        return;
      }
      launch_beside(
          s, [&](hipStream_t q) { hipLaunchKernelGGL((k_composite_fwd_sh_vec<4, 2, true, kFallback>), dim3(gf), dim3(128), 0, q, pf, plist); },
          [&](hipStream_t q) {
            if (p0.stop == nullptr) hipLaunchKernelGGL((k_composite_fwd_sh_vec<4, 4, true, kPolyNB, false>), g, dim3(64), 0, q, p0, plist);
            else hipLaunchKernelGGL((k_composite_fwd_sh_vec<4, 4, true, kPolyNB>), g, dim3(64), 0, q, p0, plist);
          });
      return;
    }
  }
  // exact basis: packed, two wavefronts per tile (a lone 8-view launch is 4 % slower than at four but issues fewer vector
  // instructions, which is what counts with another batch's backward in flight: 3 010 vs 2 885 renders/s, round 2)
  hipLaunchKernelGGL((k_composite_fwd_sh_vec<CB, 2, true>), g, dim3(128), 0, s, p0, plist);
}
int launch_fwd_sh_batch(int C, const CompParams *host, uint32_t B, hipStream_t s, bool bounded) {
  if (B == 0 || host[0].ntw * host[0].nth == 0) return 0;
  const uint32_t nblk = comp_grid(host[0]);
  const bool poly = C == 4 && bounded;
  for_each_chunk(host, B, [&](const CompParams &p0, const ViewPack<true> &pack, uint32_t n) {
    switch (C) {
      case 1: launch_fwd_sh_batch_c<1>(p0, pack, n, nblk, s, false); break;
      case 2: launch_fwd_sh_batch_c<2>(p0, pack, n, nblk, s, false); break;
      case 3: launch_fwd_sh_batch_c<3>(p0, pack, n, nblk, s, false); break;
      default: launch_fwd_sh_batch_c<4>(p0, pack, n, nblk, s, poly); break;
    }
  });
  return (int)hipGetLastError();
}
template <int CB, bool MOM>
static void launch_bwd_sh_batch_c(const CompParams &p0, const ViewPack<true> &plist, uint32_t B, uint32_t nblk, hipStream_t s, bool bounded) {
  const dim3 g(nblk * B);
  if constexpr (CB == 4) {
    if (bounded) {  // as the forward: the persistent exact fallback, then the polynomial kernel (120 registers: 4 wavefronts per SIMD)
      CompParams pf = p0;
      pf.vgrid = nblk * B;
      // 12 one-wavefront workgroups per compute unit (3 per SIMD: the exact body's occupancy at 152 registers) -- with per-tile
      // routing this launch may carry a large share of the tiles (2 per SIMD until round 4, when it only ever took whole views)
      const uint32_t gf = pf.vgrid < 3072u ? pf.vgrid : 3072u;
      launch_beside(
          s, [&](hipStream_t q) {
            if (p0.sh_rows != nullptr && p0.no_fallback) return;  // (as the forward: no tile was handed over)
            hipLaunchKernelGGL((k_composite_bwd_sh_vec<4, 4, true, kFallback, MOM>), dim3(gf), dim3(64), 0, q, pf, plist);
          },
          [&](hipStream_t q) { hipLaunchKernelGGL((k_composite_bwd_sh_vec<4, 4, true, kPolyNB, MOM>), g, dim3(64), 0, q, p0, plist); });
      return;
    }
  }
  // one wavefront per tile: the per-Gaussian gradient reduction costs the same per wavefront whatever the number of pixels behind
  // it (two wavefronts per tile: 2 838 vs 3 492 renders/s on the exact basis, profiles/r04_ab_shapes.txt)
  hipLaunchKernelGGL((k_composite_bwd_sh_vec<CB, 4, true, 0, MOM>), g, dim3(64), 0, s, p0, plist);
}
template <bool MOM>
static int launch_bwd_sh_batch_m(int C, const CompParams *host, uint32_t B, hipStream_t s, bool bounded) {
  if (B == 0 || host[0].ntw * host[0].nth == 0) return 0;
  const uint32_t nblk = comp_grid(host[0]) * (uint32_t)(host[0].nseg > 1 ? host[0].nseg : 1);
  const bool poly = C == 4 && bounded;
  for_each_chunk(host, B, [&](const CompParams &p0, const ViewPack<true> &pack, uint32_t n) {
    switch (C) {
      case 1: launch_bwd_sh_batch_c<1, MOM>(p0, pack, n, nblk, s, false); break;
      case 2: launch_bwd_sh_batch_c<2, MOM>(p0, pack, n, nblk, s, false); break;
      case 3: launch_bwd_sh_batch_c<3, MOM>(p0, pack, n, nblk, s, false); break;
      default: launch_bwd_sh_batch_c<4, MOM>(p0, pack, n, nblk, s, poly); break;
    }
  });
  return (int)hipGetLastError();
}
int launch_bwd_sh_batch(int C, const CompParams *host, uint32_t B, hipStream_t s, bool bounded, bool moments = false) {
  return moments ? launch_bwd_sh_batch_m<true>(C, host, B, s, bounded) : launch_bwd_sh_batch_m<false>(C, host, B, s, bounded);
}

// post-activation channels, B cameras per launch: packed, one wavefront per tile (the same operation sequence as the per-camera
// kernels: identical bits).  MODE_RGBD = fused RGB + heads, MODE_RGB = colours only.
template <int MODE, bool MOM = false>
static int launch_chan_batch(bool backward, const CompParams *host, uint32_t B, hipStream_t s) {
  if (B == 0 || host[0].ntw * host[0].nth == 0) return 0;
  const uint32_t nblk = comp_grid(host[0]);
  for_each_chunk(host, B, [&](const CompParams &p0, const ViewPack<true> &pack, uint32_t n) {
    if (backward) hipLaunchKernelGGL((k_composite_bwd_chan_vec<MODE, true, MOM>), dim3(nblk * n), dim3(64), 0, s, p0, pack);
    else hipLaunchKernelGGL((k_composite_fwd_chan_vec<MODE, true>), dim3(nblk * n), dim3(64), 0, s, p0, pack);
  });
  return (int)hipGetLastError();
}
int launch_fwd_rgbd_batch(const CompParams *host, uint32_t B, hipStream_t s) { return launch_chan_batch<MODE_RGBD>(false, host, B, s); }
int launch_bwd_rgbd_batch(const CompParams *host, uint32_t B, hipStream_t s) { return launch_chan_batch<MODE_RGBD>(true, host, B, s); }
int launch_bwd_rgbd_batch_moments(const CompParams *host, uint32_t B, hipStream_t s) { return launch_chan_batch<MODE_RGBD, true>(true, host, B, s); }
int launch_fwd_rgb_batch(const CompParams *host, uint32_t B, hipStream_t s) { return launch_chan_batch<MODE_RGB>(false, host, B, s); }
int launch_bwd_rgb_batch(const CompParams *host, uint32_t B, hipStream_t s) { return launch_chan_batch<MODE_RGB>(true, host, B, s); }

int launch_bwd_pixel_dispatch(int mode, int C, const CompParams &p, hipStream_t s) {
  if (mode == MODE_RGB) return launch_bwd<MODE_RGB, 1>(p, s);
  if (mode == MODE_SCALAR) return launch_bwd<MODE_SCALAR, 1>(p, s);
  if (mode == MODE_RGBD) return launch_bwd<MODE_RGBD, 1>(p, s);
  switch (C) {
    case 1: return launch_bwd<MODE_SH, 1>(p, s);
    case 2: return launch_bwd<MODE_SH, 2>(p, s);
    case 3: return launch_bwd<MODE_SH, 3>(p, s);
    default: return launch_bwd<MODE_SH, 4>(p, s);
  }
}

}  // namespace gs

using namespace gs;

extern "C" {

/* Name of the compositing kernel a launch of `stage` runs ("sh_fwd", "sh_bwd", "sh_fwd_batch", "sh_bwd_batch" -- with the suffix
 * "_poly" for an enqueue that carries the coefficient bound (SH degree 3) --, "rgb_fwd", "rgb_bwd", "rgbd_fwd_batch",
 * "rgbd_bwd_batch") at SH degree C-1: one fixed shape per job since round 4 (launch helpers above).  Returns the length written
 * (excluding the terminator), 0 for an unknown stage. */
int gsgen_kernel_variant(const char *stage, uint32_t C, uint32_t n_segments, char *out, size_t out_bytes) {
  if (!stage || !out || out_bytes == 0) return 0;
  const std::string st(stage);
  char buf[160];
  int n = 0;
  const bool poly_stage = st.size() > 5 && st.compare(st.size() - 5, 5, "_poly") == 0 && C == 4;
  const std::string base = (st.size() > 5 && st.compare(st.size() - 5, 5, "_poly") == 0) ? st.substr(0, st.size() - 5) : st;
  const char *seg = n_segments > 1 ? " segmented" : "";
  if (base == "sh_fwd")
    n = poly_stage ? snprintf(buf, sizeof buf, "k_composite_fwd_sh_vec<C=4,PPL=2,ROUTED:POLY6|exact>")
                   : snprintf(buf, sizeof buf, "k_composite_fwd<SH,C=%u,PPL=1>", C);
  else if (base == "sh_fwd_batch")
    n = poly_stage ? snprintf(buf, sizeof buf, "k_composite_fwd_sh_vec<C=4,PPL=4,BATCH,POLY6> + persistent exact fallback")
                   : snprintf(buf, sizeof buf, "k_composite_fwd_sh_vec<C=%u,PPL=2,BATCH>", C);
  else if (base == "sh_bwd")
    n = poly_stage ? snprintf(buf, sizeof buf, "k_composite_bwd_sh_vec<C=4,PPL=4,ROUTED:POLY6|exact>%s", seg)
                   : snprintf(buf, sizeof buf, "k_composite_bwd_sh_vec<C=%u,PPL=4>%s", C, seg);
  else if (base == "sh_bwd_batch")
    n = poly_stage ? snprintf(buf, sizeof buf, "k_composite_bwd_sh_vec<C=4,PPL=4,BATCH,POLY6>%s + persistent exact fallback", seg)
                   : snprintf(buf, sizeof buf, "k_composite_bwd_sh_vec<C=%u,PPL=4,BATCH>%s", C, seg);
  else if (st == "rgb_fwd") n = snprintf(buf, sizeof buf, "k_composite_fwd<RGB,PPL=1>");
  else if (st == "rgb_bwd") n = snprintf(buf, sizeof buf, "k_composite_bwd_chan_vec<RGB>");
  else if (st == "rgbd_fwd_batch") n = snprintf(buf, sizeof buf, "k_composite_fwd_chan_vec<RGBD,BATCH>");
  else if (st == "rgbd_bwd_batch") n = snprintf(buf, sizeof buf, "k_composite_bwd_chan_vec<RGBD,BATCH>");
  else if (st == "rgbd_bwd_batch_moments") n = snprintf(buf, sizeof buf, "k_composite_bwd_chan_vec<RGBD,BATCH,MOMENTS>");
  else return 0;
  if (n < 0) return 0;
  if ((size_t)n >= out_bytes) n = (int)out_bytes - 1;
  memcpy(out, buf, (size_t)n);
  out[n] = 0;
  return n;
}

int gsgen_vol_render_start_end_with_T(uint32_t N, uint32_t D, const float *mean, const float *cov,
                                      const float *color, const float *alpha, const int *start,
                                      const int *end, const int *gaussian_ids, float *out,
                                      const float *topleft, uint32_t tile_size, uint32_t n_tiles_h,
                                      uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y,
                                      uint32_t H, uint32_t W, float thresh, float *T,
                                      gsgen_stream_t stream) {
  if (int e = check_common(tile_size, start, end, out)) return e;
  if (N == 0 || D == 0) return 0;  // reference: zero-sized launch; out/T keep their init values
  CompParams p{};
  p.mean = mean; p.cov = cov; p.col = color; p.alpha = alpha;
  p.start = start; p.end = end; p.ids = gaussian_ids; p.topleft = topleft;
  p.out = out; p.T = T;
  p.ntw = (int)n_tiles_w; p.nth = (int)n_tiles_h; p.H = (int)H; p.W = (int)W;
  p.psx = pixel_size_x; p.psy = pixel_size_y; p.thresh = thresh; p.tile_side = (int)tile_size;
  return launch_fwd<MODE_RGB, 1>(p, (hipStream_t)stream);
}


int gsgen_vol_render_scalar(uint32_t N, uint32_t D, const float *mean, const float *cov,
                            const float *scalar, const float *alpha, const int *start,
                            const int *end, const int *gaussian_ids, float *out,
                            const float *topleft, uint32_t tile_size, uint32_t n_tiles_h,
                            uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y, uint32_t H,
                            uint32_t W, float thresh, float *T, gsgen_stream_t stream) {
  if (int e = check_common(tile_size, start, end, out)) return e;
  if (N == 0 || D == 0) return 0;
  CompParams p{};
  p.mean = mean; p.cov = cov; p.col = scalar; p.alpha = alpha;
  p.start = start; p.end = end; p.ids = gaussian_ids; p.topleft = topleft;
  p.out = out; p.T = T;
  p.ntw = (int)n_tiles_w; p.nth = (int)n_tiles_h; p.H = (int)H; p.W = (int)W;
  p.psx = pixel_size_x; p.psy = pixel_size_y; p.thresh = thresh; p.tile_side = (int)tile_size;
  return launch_fwd<MODE_SCALAR, 1>(p, (hipStream_t)stream);
}


int gsgen_vol_render_rgbd(uint32_t N, uint32_t D, const float *mean, const float *cov, const float *color,
                          const float *depth, const float *alpha, const int *start, const int *end,
                          const int *gaussian_ids, float *out6, const float *topleft, uint32_t tile_size,
                          uint32_t n_tiles_h, uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y,
                          uint32_t H, uint32_t W, float thresh, float *T, const uint32_t *tile_order,
                          gsgen_stream_t stream) {
  if (int e = check_common(tile_size, start, end, out6)) return e;
  if (N == 0 || D == 0) return 0;
  if (!depth) return GSGEN_EINVAL;
  CompParams p{};
  p.mean = mean; p.cov = cov; p.col = color; p.depth = depth; p.alpha = alpha;
  p.start = start; p.end = end; p.ids = gaussian_ids; p.topleft = topleft;
  p.out = out6; p.T = T;
  p.ntw = (int)n_tiles_w; p.nth = (int)n_tiles_h; p.H = (int)H; p.W = (int)W;
  p.psx = pixel_size_x; p.psy = pixel_size_y; p.thresh = thresh; p.tile_side = (int)tile_size;
  p.tile_order = tile_order;
  return launch_fwd<MODE_RGBD, 1>(p, (hipStream_t)stream);
}

size_t gsgen_segment_workspace_bytes(uint32_t n_tiles, uint32_t n_segments) {
  return (size_t)n_tiles * 256 * (sizeof(float4) * (size_t)(n_segments > 1 ? n_segments : 1) + sizeof(int));
}

int gsgen_vol_render_sh_ordered(uint32_t N, uint32_t D, const float *mean, const float *cov,
                                const float *sh_coeffs, const float *alpha, const int *start,
                                const int *end, const int *gaussian_ids, float *out, const float *topleft,
                                const float *c2w, uint32_t tile_size, uint32_t n_tiles_h,
                                uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y, uint32_t H,
                                uint32_t W, uint32_t C, float thresh, const float *bg_rgb, float *T,
                                const uint32_t *tile_order, gsgen_stream_t stream) {
  return gsgen_vol_render_sh_bounded(N, D, mean, cov, sh_coeffs, alpha, start, end, gaussian_ids, out, topleft,
                                     c2w, tile_size, n_tiles_h, n_tiles_w, pixel_size_x, pixel_size_y, H, W, C,
                                     thresh, bg_rgb, T, tile_order, nullptr, 0, nullptr, stream);
}

int gsgen_vol_render_sh_segmented(uint32_t N, uint32_t D, const float *mean, const float *cov,
                                  const float *sh_coeffs, const float *alpha, const int *start,
                                  const int *end, const int *gaussian_ids, float *out, const float *topleft,
                                  const float *c2w, uint32_t tile_size, uint32_t n_tiles_h,
                                  uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y, uint32_t H,
                                  uint32_t W, uint32_t C, float thresh, const float *bg_rgb, float *T,
                                  const uint32_t *tile_order, void *segment_workspace, uint32_t n_segments,
                                  gsgen_stream_t stream) {
  return gsgen_vol_render_sh_bounded(N, D, mean, cov, sh_coeffs, alpha, start, end, gaussian_ids, out, topleft,
                                     c2w, tile_size, n_tiles_h, n_tiles_w, pixel_size_x, pixel_size_y, H, W, C,
                                     thresh, bg_rgb, T, tile_order, segment_workspace, n_segments, nullptr, stream);
}

int gsgen_vol_render_sh_bounded(uint32_t N, uint32_t D, const float *mean, const float *cov,
                                const float *sh_coeffs, const float *alpha, const int *start,
                                const int *end, const int *gaussian_ids, float *out, const float *topleft,
                                const float *c2w, uint32_t tile_size, uint32_t n_tiles_h,
                                uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y, uint32_t H,
                                uint32_t W, uint32_t C, float thresh, const float *bg_rgb, float *T,
                                const uint32_t *tile_order, void *segment_workspace, uint32_t n_segments,
                                const float *sh_l1_bound, gsgen_stream_t stream) {
  return gsgen_vol_render_sh_routed(N, D, mean, cov, sh_coeffs, alpha, start, end, gaussian_ids, out, topleft, c2w, tile_size,
                                    n_tiles_h, n_tiles_w, pixel_size_x, pixel_size_y, H, W, C, thresh, bg_rgb, T, tile_order,
                                    segment_workspace, n_segments, sh_l1_bound, nullptr, stream);
}

int gsgen_vol_render_sh_routed(uint32_t N, uint32_t D, const float *mean, const float *cov,
                               const float *sh_coeffs, const float *alpha, const int *start,
                               const int *end, const int *gaussian_ids, float *out, const float *topleft,
                               const float *c2w, uint32_t tile_size, uint32_t n_tiles_h,
                               uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y, uint32_t H,
                               uint32_t W, uint32_t C, float thresh, const float *bg_rgb, float *T,
                               const uint32_t *tile_order, void *segment_workspace, uint32_t n_segments,
                               const float *sh_l1_bound, const float *sh_row_bounds, gsgen_stream_t stream) {
  if (int e = check_common(tile_size, start, end, out)) return e;
  if (n_segments > 1 && segment_workspace == nullptr) return GSGEN_EINVAL;
  if (C < 1 || C > 4) return GSGEN_EUNSUPPORTED;  // reference dispatches C = 1..4 only (render.cu:507-544)
  if (!c2w) return GSGEN_EINVAL;
  if ((N == 0 || D == 0) && bg_rgb == nullptr) return 0;
  CompParams p{};
  p.mean = mean; p.cov = cov; p.col = sh_coeffs; p.alpha = alpha;
  p.start = start; p.end = end; p.ids = gaussian_ids; p.topleft = topleft; p.rot = c2w;
  p.bg = bg_rgb; p.out = out; p.T = T;
  p.ntw = (int)n_tiles_w; p.nth = (int)n_tiles_h; p.H = (int)H; p.W = (int)W;
  p.psx = pixel_size_x; p.psy = pixel_size_y; p.thresh = thresh; p.tile_side = (int)tile_size;
  p.tile_order = tile_order;
  // (both given: the view's bound decides first -- a scene within it needs no look at the lists --, the per-splat bounds per tile otherwise)
  p.sh_bound = (C == 4) ? sh_l1_bound : nullptr;
  p.sh_rows = (C == 4) ? sh_row_bounds : nullptr;
  if (n_segments > 1) {
    p.nseg = (int)n_segments;
    p.ckpt = reinterpret_cast<float4 *>(segment_workspace);
    p.stop = reinterpret_cast<int *>(p.ckpt + (size_t)n_tiles_h * n_tiles_w * 256 * n_segments);
  }
  hipStream_t s = (hipStream_t)stream;
  switch (C) {
    case 1: return launch_fwd<MODE_SH, 1>(p, s);
    case 2: return launch_fwd<MODE_SH, 2>(p, s);
    case 3: return launch_fwd<MODE_SH, 3>(p, s);
    default: return launch_fwd<MODE_SH, 4>(p, s);
  }
}

// ---- batched cameras -----------------------------------------------------------------------------
static int fill_view_params(uint32_t n_views, const gsgen_sh_view *views, const float *sh_coeffs,
                            const float *alpha, float *g_sh, float *g_alpha, uint32_t ntw, uint32_t nth,
                            uint32_t H, uint32_t W, float thresh, uint32_t n_segments, bool backward,
                            const float *sh_bound, std::vector<CompParams> &ps, const float *sh_rows = nullptr,
                            uint8_t *tile_flags = nullptr) {
  ps.assign(n_views, CompParams{});
  for (uint32_t b = 0; b < n_views; ++b) {
    const gsgen_sh_view &v = views[b];
    if (!v.start || !v.end || !v.out || !v.c2w) return GSGEN_EINVAL;
    if ((v.tile_order == nullptr) != (views[0].tile_order == nullptr)) return GSGEN_EINVAL;
    if (n_segments > 1 && v.segment_workspace == nullptr) return GSGEN_EINVAL;
    if (backward && (!v.grad_out || !v.grad_mean || !v.grad_cov)) return GSGEN_EINVAL;
    CompParams &p = ps[b];
    p.mean = v.mean; p.cov = v.cov; p.col = sh_coeffs; p.alpha = alpha;
    p.start = v.start; p.end = v.end; p.ids = v.gaussian_ids; p.topleft = v.topleft; p.rot = v.c2w;
    p.ntw = (int)ntw; p.nth = (int)nth; p.H = (int)H; p.W = (int)W;
    p.psx = v.pixel_size_x; p.psy = v.pixel_size_y; p.thresh = thresh;
    p.tile_order = v.tile_order;
    p.n_hi = 0x7fffffff;
    p.sh_bound = sh_bound;  // (both given: the view's bound first -- a scene within it skips the per-entry tests --, then the rows)
    p.sh_rows = sh_rows;
    p.tile_flags = (sh_rows != nullptr && tile_flags != nullptr) ? tile_flags + (size_t)b * ntw * nth : nullptr;
    // (per-tile routing only: without per-splat bounds a view beyond the scene's bound is the fallback's as a whole)
    p.route_report = sh_rows != nullptr ? v.route_report : nullptr;
    p.no_fallback = (sh_rows != nullptr && views[0].no_fallback) ? 1 : 0;
    if (backward) {
      p.final_img = v.out; p.grad_out = v.grad_out;
      p.g_mean = v.grad_mean; p.g_cov = v.grad_cov; p.g_col = g_sh; p.g_alpha = g_alpha;
    } else {
      p.bg = v.bg_rgb; p.out = v.out; p.T = v.T;
      p.fill_empty = 1;  // the batched forward writes every pixel of out / T (include/gsgen_hip.h)
    }
    if (n_segments > 1) {
      p.nseg = (int)n_segments;
      p.ckpt = reinterpret_cast<float4 *>(v.segment_workspace);
      p.stop = reinterpret_cast<int *>(p.ckpt + (size_t)nth * ntw * 256 * n_segments);
    }
  }
  return 0;
}

}  // extern "C"

// S = max over splats and channels of sum_{k >= 1} |sh[i][c][k]| (non-negative floats order like their bit patterns, so the
// maximum is an unsigned atomicMax).  SH degree 3: one float4 per thread and iteration, perfectly coalesced -- four lanes share
// a row of 16 coefficients, the first of them drops the constant term.  A FEW HUNDRED workgroups stride over the array and each
// ends with at most one atomic: atomics of every wavefront on the one result word serialise in the memory system (18 750 of them
// made this pass 214 us for 100 k splats; it is a 19 MB read).  CHECK: counts the rows whose sum EXCEEDS *bound instead (the
// debug verification of a bound some other pass produced).  NaN coefficients read as "no bound" (3e38 never passes poly_ok).
constexpr uint32_t kBoundBlocks = 512;
template <bool CHECK>
__device__ __forceinline__ void sh_l1_finish(float v, uint32_t bad, float *out, uint32_t *n_bad) {
  __shared__ float s_max[4];
  __shared__ uint32_t s_bad[4];
  const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
  if constexpr (CHECK) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) bad += (uint32_t)__shfl_xor((int)bad, d);
    if (lane == 0) s_bad[wave] = bad;
  } else {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d));
    if (lane == 0) s_max[wave] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if constexpr (CHECK) {
      const uint32_t t = s_bad[0] + s_bad[1] + s_bad[2] + s_bad[3];
      if (t != 0u) atomicAdd(n_bad, t);
    } else {
      const float m = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
      // (a plain look first: once a large value is in, most workgroups have nothing to add)
      if (m > 0.0f && m > *reinterpret_cast<volatile float *>(out)) atomicMax(reinterpret_cast<unsigned int *>(out), __float_as_uint(m));
    }
  }
}
template <bool CHECK>
__global__ void __launch_bounds__(256) k_sh_l1_rows16(uint32_t n_quads, const float4 *__restrict__ sh, float *out, const float *bound,
                                                       uint32_t *n_bad) {
  float vmax = 0.0f;
  uint32_t bad = 0;
  const float lim = CHECK ? bound[0] : 0.0f;
  // (n_quads is a multiple of 4 and the stride a multiple of 256: the four lanes of a row stay together)
  for (uint32_t base = blockIdx.x * 256u; base < n_quads; base += gridDim.x * 256u) {  // (uniform trip count: shuffles inside)
    const uint32_t i = base + threadIdx.x;
    const float4 q = i < n_quads ? sh[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    float v = ((i & 3u) ? fabsf(q.x) : 0.0f) + fabsf(q.y) + fabsf(q.z) + fabsf(q.w);
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);  // every lane of a row's four now holds the row's sum
    if (!(v == v)) v = 3.0e38f;
    if constexpr (CHECK) bad += (i < n_quads && (i & 3u) == 0u && v > lim) ? 1u : 0u;
    else vmax = fmaxf(vmax, v);
  }
  sh_l1_finish<CHECK>(vmax, bad, out, n_bad);
}
// any SH degree: one row of CC coefficients per thread and iteration
template <bool CHECK>
__global__ void __launch_bounds__(256) k_sh_l1_rows(uint32_t n_rows, const float *__restrict__ sh, int CC, float *out, const float *bound,
                                                     uint32_t *n_bad) {
  float vmax = 0.0f;
  uint32_t bad = 0;
  const float lim = CHECK ? bound[0] : 0.0f;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_rows; i += gridDim.x * 256u) {
    const float *q = sh + (size_t)i * CC;
    float v = 0.0f;
    for (int k = 1; k < CC; ++k) v += fabsf(q[k]);
    if (!(v == v)) v = 3.0e38f;
    if constexpr (CHECK) bad += (v > lim) ? 1u : 0u;
    else vmax = fmaxf(vmax, v);
  }
  sh_l1_finish<CHECK>(vmax, bad, out, n_bad);
}

// The per-SPLAT form (per-tile routing): rows[i] = max over the three channels of splat i of sum_{k >= 1} |sh[i][c][k]|, and the
// global maximum as before.  SH degree 3: 192 threads take 16 splats per iteration -- one float4 per thread, perfectly coalesced
// (a splat is 12 float4s), partial sums through LDS, 16 threads finish the splats.  No atomics but the one per workgroup.
__global__ void __launch_bounds__(192) k_sh_l1_splats16(uint32_t N, const float4 *__restrict__ sh, float *out, float *__restrict__ rows) {
  __shared__ float part[192];
  __shared__ float s_max[16];
  const uint32_t t = threadIdx.x;
  float vmax = 0.0f;
  for (uint32_t base = blockIdx.x * 16u; base < N; base += gridDim.x * 16u) {  // (uniform trip count: barriers inside)
    const size_t i = (size_t)base * 12u + t;
    const float4 q = i < (size_t)N * 12u ? sh[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    part[t] = ((t & 3u) ? fabsf(q.x) : 0.0f) + fabsf(q.y) + fabsf(q.z) + fabsf(q.w);  // (12 = 3 rows of 4 quarters: t & 3 = quarter)
    __syncthreads();
    if (t < 16u) {
      const float *r = &part[12u * t];
      const float s0 = (r[0] + r[1]) + (r[2] + r[3]), s1 = (r[4] + r[5]) + (r[6] + r[7]), s2 = (r[8] + r[9]) + (r[10] + r[11]);
      float m = fmaxf(fmaxf(s0, s1), s2);
      if (!(s0 == s0) || !(s1 == s1) || !(s2 == s2)) m = 3.0e38f;  // NaN coefficients: no bound (never passes poly_row_ok)
      if (base + t < N) { rows[base + t] = m; vmax = fmaxf(vmax, m); }
    }
    __syncthreads();
  }
  if (t < 16u) s_max[t] = vmax;
  __syncthreads();
  if (t == 0 && out != nullptr) {
    float m = s_max[0];
    for (int k = 1; k < 16; ++k) m = fmaxf(m, s_max[k]);
    if (m > 0.0f && m > *reinterpret_cast<volatile float *>(out)) atomicMax(reinterpret_cast<unsigned int *>(out), __float_as_uint(m));
  }
}
// any SH degree: one splat per thread
__global__ void __launch_bounds__(256) k_sh_l1_splats(uint32_t N, const float *__restrict__ sh, int CC, float *out, float *__restrict__ rows) {
  float vmax = 0.0f;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < N; i += gridDim.x * 256u) {
    float m = 0.0f;
    for (int c = 0; c < 3; ++c) {
      const float *q = sh + ((size_t)i * 3 + c) * CC;
      float v = 0.0f;
      for (int k = 1; k < CC; ++k) v += fabsf(q[k]);
      if (!(v == v)) v = 3.0e38f;
      m = fmaxf(m, v);
    }
    rows[i] = m;
    vmax = fmaxf(vmax, m);
  }
  if (out != nullptr) sh_l1_finish<false>(vmax, 0u, out, nullptr);
}

template <bool CHECK>
static int sh_l1_pass(uint32_t N, const float *sh_coeffs, uint32_t C, float *out, const float *bound, uint32_t *n_bad,
                      hipStream_t s) {
  if (C < 1 || C > 4) return GSGEN_EUNSUPPORTED;
  if ((CHECK ? (!bound || !n_bad) : !out) || (N && !sh_coeffs)) return GSGEN_EINVAL;
  if (hipError_t e = hipMemsetAsync(CHECK ? (void *)n_bad : (void *)out, 0, 4, s)) return (int)e;
  if (N == 0) return 0;
  const uint32_t rows = 3u * N;
  if (C == 4 && (reinterpret_cast<uintptr_t>(sh_coeffs) & 15u) == 0) {
    const uint32_t quads = 4u * rows;
    const uint32_t nb = (quads + 255u) / 256u;
    hipLaunchKernelGGL((k_sh_l1_rows16<CHECK>), dim3(nb < kBoundBlocks ? nb : kBoundBlocks), dim3(256), 0, s, quads,
                       reinterpret_cast<const float4 *>(sh_coeffs), out, bound, n_bad);
  } else {
    const uint32_t nb = (rows + 255u) / 256u;
    hipLaunchKernelGGL((k_sh_l1_rows<CHECK>), dim3(nb < kBoundBlocks ? nb : kBoundBlocks), dim3(256), 0, s, rows, sh_coeffs, (int)(C * C),
                       out, bound, n_bad);
  }
  return (int)hipGetLastError();
}

extern "C" {

int gsgen_sh_l1_bound(uint32_t N, const float *sh_coeffs, uint32_t C, float *out, gsgen_stream_t stream) {
  return sh_l1_pass<false>(N, sh_coeffs, C, out, nullptr, nullptr, (hipStream_t)stream);
}

static int sh_l1_rows_pass(uint32_t N, const float *sh_coeffs, uint32_t C, float *out_max, float *out_rows, bool running, hipStream_t s);
int gsgen_sh_l1_bound_rows(uint32_t N, const float *sh_coeffs, uint32_t C, float *out_max, float *out_rows, gsgen_stream_t stream) {
  return sh_l1_rows_pass(N, sh_coeffs, C, out_max, out_rows, false, (hipStream_t)stream);
}
int gsgen_sh_l1_bound_rows_running(uint32_t N, const float *sh_coeffs, uint32_t C, float *running_max, float *out_rows,
                                   gsgen_stream_t stream) {
  return sh_l1_rows_pass(N, sh_coeffs, C, running_max, out_rows, true, (hipStream_t)stream);
}
static int sh_l1_rows_pass(uint32_t N, const float *sh_coeffs, uint32_t C, float *out_max, float *out_rows, bool running, hipStream_t s) {
  if (C < 1 || C > 4) return GSGEN_EUNSUPPORTED;
  if (N && (!sh_coeffs || !out_rows)) return GSGEN_EINVAL;
  if (out_max != nullptr && !running)
    if (hipError_t e = hipMemsetAsync(out_max, 0, 4, s)) return (int)e;
  if (N == 0) return 0;
  if (C == 4 && (reinterpret_cast<uintptr_t>(sh_coeffs) & 15u) == 0) {
    const uint32_t nb = (N + 15u) / 16u;
    hipLaunchKernelGGL(k_sh_l1_splats16, dim3(nb < 1024u ? nb : 1024u), dim3(192), 0, s, N, reinterpret_cast<const float4 *>(sh_coeffs),
                       out_max, out_rows);
  } else {
    const uint32_t nb = (N + 255u) / 256u;
    hipLaunchKernelGGL(k_sh_l1_splats, dim3(nb < kBoundBlocks ? nb : kBoundBlocks), dim3(256), 0, s, N, sh_coeffs, (int)(C * C), out_max,
                       out_rows);
  }
  return (int)hipGetLastError();
}

int gsgen_sh_l1_bound_check(uint32_t N, const float *sh_coeffs, uint32_t C, const float *bound, uint32_t *n_violations,
                            gsgen_stream_t stream) {
  return sh_l1_pass<true>(N, sh_coeffs, C, nullptr, bound, n_violations, (hipStream_t)stream);
}

int gsgen_sh_poly_applies(float sh_l1_bound, float max_pixel_size, uint32_t C) {
  return (C == 4 && poly_ok(sh_l1_bound, max_pixel_size)) ? 1 : 0;
}

// Host-side shortcut of a bounded batch: if NO view of the batch could take the polynomial form even for a coefficient bound as
// small as 1/16 (very wide cameras: a tile's half diagonal beyond ~0.1 rad), the launch is the plain exact one -- full
// occupancy, no polynomial kernel that every workgroup would leave, no persistent fallback.  (Exact is always right; forward and
// backward of a batch apply the same rule to the same pixel sizes.)
static bool batch_can_be_polynomial(const std::vector<CompParams> &ps) {
  for (const CompParams &p : ps)
    if (poly_ok(1.0f / 16.0f, fmaxf(fabsf(p.psx), fabsf(p.psy)))) return true;
  return false;
}

size_t gsgen_sh_batch_workspace_bytes(uint32_t n_views) { return 2 * (size_t)n_views * sizeof(CompParams); }
// ... + one flag byte per (view, tile) for the per-tile routing of the *_routed entry points
size_t gsgen_sh_batch_workspace_bytes_routed(uint32_t n_views, uint32_t n_tiles) {
  return gsgen_sh_batch_workspace_bytes(n_views) + (((size_t)n_views * n_tiles + 15u) & ~(size_t)15u);
}

int gsgen_vol_render_sh_batch(uint32_t n_views, const gsgen_sh_view *views, uint32_t N,
                              const float *sh_coeffs, const float *alpha, uint32_t tile_size,
                              uint32_t n_tiles_h, uint32_t n_tiles_w, uint32_t H, uint32_t W, uint32_t C,
                              float thresh, uint32_t n_segments, void *batch_workspace,
                              gsgen_stream_t stream) {
  return gsgen_vol_render_sh_batch_bounded(n_views, views, N, sh_coeffs, alpha, tile_size, n_tiles_h, n_tiles_w, H, W, C,
                                           thresh, n_segments, nullptr, batch_workspace, stream);
}

int gsgen_vol_render_sh_batch_bounded(uint32_t n_views, const gsgen_sh_view *views, uint32_t N,
                                      const float *sh_coeffs, const float *alpha, uint32_t tile_size,
                                      uint32_t n_tiles_h, uint32_t n_tiles_w, uint32_t H, uint32_t W, uint32_t C,
                                      float thresh, uint32_t n_segments, const float *sh_l1_bound, void *batch_workspace,
                                      gsgen_stream_t stream) {
  return gsgen_vol_render_sh_batch_routed(n_views, views, N, sh_coeffs, alpha, tile_size, n_tiles_h, n_tiles_w, H, W, C, thresh,
                                          n_segments, sh_l1_bound, nullptr, batch_workspace, stream);
}

int gsgen_vol_render_sh_batch_routed(uint32_t n_views, const gsgen_sh_view *views, uint32_t N,
                                     const float *sh_coeffs, const float *alpha, uint32_t tile_size,
                                     uint32_t n_tiles_h, uint32_t n_tiles_w, uint32_t H, uint32_t W, uint32_t C,
                                     float thresh, uint32_t n_segments, const float *sh_l1_bound, const float *sh_row_bounds,
                                     void *batch_workspace, gsgen_stream_t stream) {
  if (tile_size != 16) return GSGEN_EUNSUPPORTED;
  if (C < 1 || C > 4) return GSGEN_EUNSUPPORTED;
  if (n_views == 0) return 0;
  if (!views || !batch_workspace) return GSGEN_EINVAL;
  if (n_views > 65535) return GSGEN_EINVAL;
  (void)N;
  std::vector<CompParams> ps;
  // (per-tile routing: the flag bytes sit behind the two parameter tables of gsgen_sh_batch_workspace_bytes_routed)
  uint8_t *flags = reinterpret_cast<uint8_t *>(batch_workspace) + gsgen_sh_batch_workspace_bytes(n_views);
  if (int e = fill_view_params(n_views, views, sh_coeffs, alpha, nullptr, nullptr, n_tiles_w, n_tiles_h, H, W,
                               thresh, n_segments, false, C == 4 ? sh_l1_bound : nullptr, ps, C == 4 ? sh_row_bounds : nullptr, flags))
    return e;
  const bool bounded = C == 4 && (sh_l1_bound != nullptr || sh_row_bounds != nullptr) && batch_can_be_polynomial(ps);
  if (!bounded)
    for (CompParams &p : ps) { p.sh_bound = nullptr; p.sh_rows = nullptr; p.tile_flags = nullptr; }
  hipStream_t s = (hipStream_t)stream;
  return launch_fwd_sh_batch((int)C, ps.data(), n_views, s, bounded);
}

int gsgen_vol_render_backward_sh_batch(uint32_t n_views, const gsgen_sh_view *views, uint32_t N,
                                       const float *sh_coeffs, const float *alpha, float *grad_sh_coeffs,
                                       float *grad_alpha, uint32_t tile_size, uint32_t n_tiles_h,
                                       uint32_t n_tiles_w, uint32_t H, uint32_t W, uint32_t C, float thresh,
                                       uint32_t n_segments, void *batch_workspace, gsgen_stream_t stream) {
  return gsgen_vol_render_backward_sh_batch_bounded(n_views, views, N, sh_coeffs, alpha, grad_sh_coeffs, grad_alpha, tile_size,
                                                    n_tiles_h, n_tiles_w, H, W, C, thresh, n_segments, nullptr, batch_workspace,
                                                    stream);
}

int gsgen_vol_render_backward_sh_batch_bounded(uint32_t n_views, const gsgen_sh_view *views, uint32_t N,
                                               const float *sh_coeffs, const float *alpha, float *grad_sh_coeffs,
                                               float *grad_alpha, uint32_t tile_size, uint32_t n_tiles_h,
                                               uint32_t n_tiles_w, uint32_t H, uint32_t W, uint32_t C, float thresh,
                                               uint32_t n_segments, const float *sh_l1_bound, void *batch_workspace,
                                               gsgen_stream_t stream) {
  return gsgen_vol_render_backward_sh_batch_routed(n_views, views, N, sh_coeffs, alpha, grad_sh_coeffs, grad_alpha, tile_size,
                                                   n_tiles_h, n_tiles_w, H, W, C, thresh, n_segments, sh_l1_bound, nullptr,
                                                   batch_workspace, stream);
}

static int backward_sh_batch_routed(bool moments, uint32_t n_views, const gsgen_sh_view *views, uint32_t N,
                                    const float *sh_coeffs, const float *alpha, float *grad_sh_coeffs,
                                    float *grad_alpha, uint32_t tile_size, uint32_t n_tiles_h,
                                    uint32_t n_tiles_w, uint32_t H, uint32_t W, uint32_t C, float thresh,
                                    uint32_t n_segments, const float *sh_l1_bound, const float *sh_row_bounds,
                                    void *batch_workspace, gsgen_stream_t stream) {
  if (tile_size != 16) return GSGEN_EUNSUPPORTED;
  if (C < 1 || C > 4) return GSGEN_EUNSUPPORTED;
  if (n_views == 0 || N == 0) return 0;
  if (!views || !batch_workspace || !grad_sh_coeffs || !grad_alpha) return GSGEN_EINVAL;
  if (n_views > 65535) return GSGEN_EINVAL;
  std::vector<CompParams> ps;
  uint8_t *flags = reinterpret_cast<uint8_t *>(batch_workspace) + gsgen_sh_batch_workspace_bytes(n_views);  // (the forward's)
  if (int e = fill_view_params(n_views, views, sh_coeffs, alpha, grad_sh_coeffs, grad_alpha, n_tiles_w,
                               n_tiles_h, H, W, thresh, n_segments, true, C == 4 ? sh_l1_bound : nullptr, ps,
                               C == 4 ? sh_row_bounds : nullptr, flags))
    return e;
  const bool bounded = C == 4 && (sh_l1_bound != nullptr || sh_row_bounds != nullptr) && batch_can_be_polynomial(ps);
  if (!bounded)
    for (CompParams &p : ps) { p.sh_bound = nullptr; p.sh_rows = nullptr; p.tile_flags = nullptr; }
  hipStream_t s = (hipStream_t)stream;
  return launch_bwd_sh_batch((int)C, ps.data(), n_views, s, bounded, moments);
}
int gsgen_vol_render_backward_sh_batch_routed(uint32_t n_views, const gsgen_sh_view *views, uint32_t N,
                                              const float *sh_coeffs, const float *alpha, float *grad_sh_coeffs,
                                              float *grad_alpha, uint32_t tile_size, uint32_t n_tiles_h,
                                              uint32_t n_tiles_w, uint32_t H, uint32_t W, uint32_t C, float thresh,
                                              uint32_t n_segments, const float *sh_l1_bound, const float *sh_row_bounds,
                                              void *batch_workspace, gsgen_stream_t stream) {
  return backward_sh_batch_routed(false, n_views, views, N, sh_coeffs, alpha, grad_sh_coeffs, grad_alpha, tile_size, n_tiles_h,
                                  n_tiles_w, H, W, C, thresh, n_segments, sh_l1_bound, sh_row_bounds, batch_workspace, stream);
}
int gsgen_vol_render_backward_sh_batch_routed_moments(uint32_t n_views, const gsgen_sh_view *views, uint32_t N,
                                                      const float *sh_coeffs, const float *alpha, float *grad_sh_coeffs,
                                                      float *grad_alpha, uint32_t tile_size, uint32_t n_tiles_h,
                                                      uint32_t n_tiles_w, uint32_t H, uint32_t W, uint32_t C, float thresh,
                                                      uint32_t n_segments, const float *sh_l1_bound, const float *sh_row_bounds,
                                                      void *batch_workspace, gsgen_stream_t stream) {
  return backward_sh_batch_routed(true, n_views, views, N, sh_coeffs, alpha, grad_sh_coeffs, grad_alpha, tile_size, n_tiles_h,
                                  n_tiles_w, H, W, C, thresh, n_segments, sh_l1_bound, sh_row_bounds, batch_workspace, stream);
}

static int fill_rgbd_params(uint32_t n_views, const gsgen_rgbd_view *views, const float *color, const float *alpha,
                            float *g_alpha, uint32_t ntw, uint32_t nth, uint32_t H, uint32_t W, float thresh,
                            bool backward, std::vector<CompParams> &ps, bool heads = true,
                            float *g_color = nullptr) {
  ps.assign(n_views, CompParams{});
  for (uint32_t b = 0; b < n_views; ++b) {
    const gsgen_rgbd_view &v = views[b];
    const bool planes = heads && v.out_rgb != nullptr;
    if (heads && ((v.out_rgb != nullptr) != (v.out_depth != nullptr) || (v.out_rgb != nullptr) != (v.out_opacity != nullptr) ||
                  (v.out_rgb != nullptr) != (v.out_depth2 != nullptr)))
      return GSGEN_EINVAL;  // the four separate images: all or none
    if (!v.start || !v.end || (!v.out6 && !planes) || (heads && !v.depth)) return GSGEN_EINVAL;
    if (planes && backward && v.grad_out6 != nullptr) return GSGEN_EINVAL;  // separate images go with the four-image gradient form
    if ((v.tile_order == nullptr) != (views[0].tile_order == nullptr)) return GSGEN_EINVAL;
    if (backward && (!v.grad_mean || !v.grad_cov || (heads && !v.grad_chan6))) return GSGEN_EINVAL;
    if (backward && !v.grad_out6 && !heads) return GSGEN_EINVAL;  // the split form exists for the heads only
    CompParams &p = ps[b];
    p.mean = v.mean; p.cov = v.cov; p.col = color; p.depth = v.depth; p.alpha = alpha;
    p.start = v.start; p.end = v.end; p.ids = v.gaussian_ids; p.topleft = v.topleft;
    p.ntw = (int)ntw; p.nth = (int)nth; p.H = (int)H; p.W = (int)W;
    p.psx = v.pixel_size_x; p.psy = v.pixel_size_y; p.thresh = thresh;
    p.ps_dev = v.pixel_size_dev;
    p.tile_order = v.tile_order;
    p.n_hi = 0x7fffffff;
    if (backward) {
      p.final_img = v.out6; p.grad_out = v.grad_out6;
      if (heads && v.grad_out6 == nullptr) {
        p.go_rgb = v.grad_rgb; p.go_d = v.grad_depth; p.go_o = v.grad_opacity; p.go_z2 = v.grad_depth2;
      }
      p.g_mean = v.grad_mean; p.g_cov = v.grad_cov; p.g_col = heads ? v.grad_chan6 : g_color; p.g_alpha = g_alpha;
      if (heads) { p.T = v.T; p.g_bg = v.grad_bg; }  // (T: read by the background gradient only)
    } else {
      p.out = v.out6; p.T = v.T;
      p.fill_empty = 1;  // the batched forward writes every pixel of out6 / T (include/gsgen_hip.h)
    }
    if (heads) {
      p.pl_rgb = v.out_rgb; p.pl_d = v.out_depth; p.pl_o = v.out_opacity; p.pl_z = v.out_depth2;
      p.bg = v.bg_rgb; p.zvar = v.depth_variance ? 1 : 0;
    }
    p.chol = v.chol;
  }
  return 0;
}

int gsgen_vol_render_rgbd_batch(uint32_t n_views, const gsgen_rgbd_view *views, uint32_t N, const float *color,
                                const float *alpha, uint32_t tile_size, uint32_t n_tiles_h, uint32_t n_tiles_w,
                                uint32_t H, uint32_t W, float thresh, void *batch_workspace,
                                gsgen_stream_t stream) {
  if (tile_size != 16) return GSGEN_EUNSUPPORTED;
  if (n_views == 0 || N == 0) return 0;
  if (!views || !batch_workspace) return GSGEN_EINVAL;
  if (n_views > 65535) return GSGEN_EINVAL;
  std::vector<CompParams> ps;
  if (int e = fill_rgbd_params(n_views, views, color, alpha, nullptr, n_tiles_w, n_tiles_h, H, W, thresh, false, ps))
    return e;
  hipStream_t s = (hipStream_t)stream;
  return launch_fwd_rgbd_batch(ps.data(), n_views, s);
}

int gsgen_vol_render_rgbd_backward_batch(uint32_t n_views, const gsgen_rgbd_view *views, uint32_t N,
                                         const float *color, const float *alpha, float *grad_alpha,
                                         uint32_t tile_size, uint32_t n_tiles_h, uint32_t n_tiles_w, uint32_t H,
                                         uint32_t W, float thresh, void *batch_workspace, gsgen_stream_t stream) {
  if (tile_size != 16) return GSGEN_EUNSUPPORTED;
  if (n_views == 0 || N == 0) return 0;
  if (!views || !batch_workspace || !grad_alpha) return GSGEN_EINVAL;
  if (n_views > 65535) return GSGEN_EINVAL;
  std::vector<CompParams> ps;
  if (int e = fill_rgbd_params(n_views, views, color, alpha, grad_alpha, n_tiles_w, n_tiles_h, H, W, thresh, true, ps))
    return e;
  hipStream_t s = (hipStream_t)stream;
  return launch_bwd_rgbd_batch(ps.data(), n_views, s);
}

int gsgen_vol_render_rgbd_backward_batch_moments(uint32_t n_views, const gsgen_rgbd_view *views, uint32_t N,
                                                 const float *color, const float *alpha, float *grad_alpha,
                                                 uint32_t tile_size, uint32_t n_tiles_h, uint32_t n_tiles_w, uint32_t H,
                                                 uint32_t W, float thresh, void *batch_workspace, gsgen_stream_t stream) {
  if (tile_size != 16) return GSGEN_EUNSUPPORTED;
  if (n_views == 0 || N == 0) return 0;
  if (!views || !batch_workspace || !grad_alpha) return GSGEN_EINVAL;
  if (n_views > 65535) return GSGEN_EINVAL;
  std::vector<CompParams> ps;
  if (int e = fill_rgbd_params(n_views, views, color, alpha, grad_alpha, n_tiles_w, n_tiles_h, H, W, thresh, true, ps))
    return e;
  hipStream_t s = (hipStream_t)stream;
  return launch_bwd_rgbd_batch_moments(ps.data(), n_views, s);
}

int gsgen_vol_render_rgb_batch(uint32_t n_views, const gsgen_rgbd_view *views, uint32_t N, const float *color,
                               const float *alpha, uint32_t tile_size, uint32_t n_tiles_h, uint32_t n_tiles_w,
                               uint32_t H, uint32_t W, float thresh, void *batch_workspace, gsgen_stream_t stream) {
  if (tile_size != 16) return GSGEN_EUNSUPPORTED;
  if (n_views == 0 || N == 0) return 0;
  if (!views || !batch_workspace) return GSGEN_EINVAL;
  if (n_views > 65535) return GSGEN_EINVAL;
  std::vector<CompParams> ps;
  if (int e = fill_rgbd_params(n_views, views, color, alpha, nullptr, n_tiles_w, n_tiles_h, H, W, thresh, false, ps,
                               false))
    return e;
  hipStream_t s = (hipStream_t)stream;
  return launch_fwd_rgb_batch(ps.data(), n_views, s);
}

int gsgen_vol_render_rgb_backward_batch(uint32_t n_views, const gsgen_rgbd_view *views, uint32_t N,
                                        const float *color, const float *alpha, float *grad_color,
                                        float *grad_alpha, uint32_t tile_size, uint32_t n_tiles_h,
                                        uint32_t n_tiles_w, uint32_t H, uint32_t W, float thresh,
                                        void *batch_workspace, gsgen_stream_t stream) {
  if (tile_size != 16) return GSGEN_EUNSUPPORTED;
  if (n_views == 0 || N == 0) return 0;
  if (!views || !batch_workspace || !grad_alpha || !grad_color) return GSGEN_EINVAL;
  if (n_views > 65535) return GSGEN_EINVAL;
  std::vector<CompParams> ps;
  if (int e = fill_rgbd_params(n_views, views, color, alpha, grad_alpha, n_tiles_w, n_tiles_h, H, W, thresh, true, ps,
                               false, grad_color))
    return e;
  hipStream_t s = (hipStream_t)stream;
  return launch_bwd_rgb_batch(ps.data(), n_views, s);
}

int gsgen_vol_render_sh(uint32_t N, uint32_t D, const float *mean, const float *cov,
                        const float *sh_coeffs, const float *alpha, const int *start,
                        const int *end, const int *gaussian_ids, float *out, const float *topleft,
                        const float *c2w, uint32_t tile_size, uint32_t n_tiles_h,
                        uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y, uint32_t H,
                        uint32_t W, uint32_t C, float thresh, const float *bg_rgb, float *T,
                        gsgen_stream_t stream) {
  return gsgen_vol_render_sh_ordered(N, D, mean, cov, sh_coeffs, alpha, start, end, gaussian_ids, out, topleft,
                                     c2w, tile_size, n_tiles_h, n_tiles_w, pixel_size_x, pixel_size_y, H, W, C,
                                     thresh, bg_rgb, T, nullptr, stream);
}

}  // extern "C"
