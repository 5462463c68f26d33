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
// reference's one LDS atomic per (pixel, Gaussian, component).
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

namespace gs {

// ---- LDS staging ---------------------------------------------------------------------------
template <int MODE, int CB>
struct Stage {
  using TR = Traits<MODE, CB>;
  float mx[kBatch], my[kBatch], a[kBatch];
  float c0[kBatch], c1[kBatch], c2[kBatch], c3[kBatch];
  float p0[kBatch], p1[kBatch], p2[kBatch];  // RGB/scalar: scaled Cholesky factor; SH: p0 = kk, p1 = 1/det
  int id[kBatch];
  alignas(16) float col[kBatch * TR::NCOLP];
};

template <int MODE, int CB, int NT>
__device__ __forceinline__ void stage_batch(Stage<MODE, CB> &S, const CompParams &p, int list_base,
                                            int nb) {
  using TR = Traits<MODE, CB>;
  const int t = (int)threadIdx.x;
  if (t < kBatch) {
    int id = 0;
    float mx = 0.f, my = 0.f, a = 0.f, c0 = 1.f, c1 = 0.f, c2 = 0.f, c3 = 1.f, p0 = 0.f, p1 = 0.f,
          p2 = 0.f;
    if (t < nb) {
      id = p.ids[list_base + t];
      const float2 m = *reinterpret_cast<const float2 *>(p.mean + 2 * (size_t)id);
      const float4 c = *reinterpret_cast<const float4 *>(p.cov + 4 * (size_t)id);
      mx = m.x; my = m.y; c0 = c.x; c1 = c.y; c2 = c.z; c3 = c.w;
      const GRec r = prep_record<MODE>(mx, my, c0, c1, c2, c3, p.alpha[id]);
      a = r.a; p0 = r.p0; p1 = r.p1; p2 = r.p2;
    }
    S.id[t] = id; S.mx[t] = mx; S.my[t] = my; S.a[t] = a;
    S.c0[t] = c0; S.c1[t] = c1; S.c2[t] = c2; S.c3[t] = c3;
    S.p0[t] = p0; S.p1[t] = p1; S.p2[t] = p2;
  }
  if constexpr (MODE == MODE_SH) __syncthreads();  // S.id is consumed below by other lanes
  // colour / scalar / SH coefficients: NCOL contiguous floats per record in HBM
  if constexpr (MODE == MODE_SH && TR::CCP == TR::CC) {
    constexpr int Q = TR::NCOL / 4;
    for (int e = t; e < nb * Q; e += NT) {
      const int g = e / Q, k = e - g * Q;
      const float4 v = *reinterpret_cast<const float4 *>(p.col + (size_t)S.id[g] * TR::NCOL + 4 * k);
      *reinterpret_cast<float4 *>(&S.col[g * TR::NCOLP + 4 * k]) = v;
    }
  } else if constexpr (MODE == MODE_SH) {
    // channel stride CC in HBM -> CCP in LDS; the pad lanes were zeroed once at kernel start
    for (int e = t; e < nb * TR::NCOL; e += NT) {
      const int g = e / TR::NCOL, k = e - g * TR::NCOL;
      const int c = k / TR::CC, kk = k - c * TR::CC;
      S.col[g * TR::NCOLP + c * TR::CCP + kk] = p.col[(size_t)S.id[g] * TR::NCOL + k];
    }
  } else {
    if (t < nb) {
      const int id = S.id[t];  // own record: written by this same thread above
      load_channels<MODE, TR::NCOL>(p, id, &S.col[t * TR::NCOL]);
    }
  }
}

// per-Gaussian values broadcast from LDS into registers
template <int MODE, int CB>
__device__ __forceinline__ GRec load_rec(const Stage<MODE, CB> &S, int g) {
  GRec r;
  r.mx = S.mx[g]; r.my = S.my[g]; r.a = S.a[g];
  r.c0 = S.c0[g]; r.c1 = S.c1[g]; r.c2 = S.c2[g]; r.c3 = S.c3[g];
  r.p0 = S.p0[g]; r.p1 = S.p1[g]; r.p2 = S.p2[g];
  return r;
}

// ============================================================================================
// forward
// ============================================================================================
template <int MODE, int CB, int PPL>
__global__ void __launch_bounds__(256 / PPL) k_composite_fwd(CompParams p) {
  using TR = Traits<MODE, CB>;
  constexpr int NT = 256 / PPL;
  constexpr int ROWS = NT / 16;
  constexpr int NCH = TR::NCH;
  __shared__ Stage<MODE, CB> S;

  int tx, ty;
  if (!block_tile(p, tx, ty)) return;  // uniform over the workgroup
  const int tile = ty * p.ntw + tx;
  const int st = p.start[tile];
  const int n = (st < 0) ? 0 : (p.end[tile] - st);
  const int t = (int)threadIdx.x;
  const int lx = t & 15, ly0 = t >> 4;
  const int gx = tx * kTile + lx;

  bool valid[PPL];
  int gy[PPL];
#pragma unroll
  for (int j = 0; j < PPL; ++j) {
    gy[j] = ty * kTile + ly0 + j * ROWS;
    valid[j] = (gx < p.W) && (gy[j] < p.H);
  }

  if (n == 0) {  // uniform over the workgroup
    if (p.bg != nullptr) {  // vol_render_bg.h:34-53: empty tiles show the background
#pragma unroll
      for (int j = 0; j < PPL; ++j)
        if (valid[j]) {
          float *o = p.out + 3 * ((size_t)gy[j] * p.W + gx);
          o[0] = p.bg[0]; o[1] = p.bg[1]; o[2] = p.bg[2];
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
      float dx = R[0] * px + R[1] * py[j] + R[2];
      float dy = R[3] * px + R[4] * py[j] + R[5];
      float dz = R[6] * px + R[7] * py[j] + R[8];
      const float len = sqrtf(dx * dx + dy * dy + dz * dz);
      dx /= len; dy /= len; dz /= len;
      float Yf[TR::CCP];
#pragma unroll
      for (int k = 0; k < TR::CCP; ++k) Yf[k] = 0.0f;
      sh_basis<CB>(dx, dy, dz, *reinterpret_cast<float (*)[TR::CC]>(&Yf[0]));
#pragma unroll
      for (int k = 0; k < TR::NPAIR; ++k) Yp[j][k] = v2f{Yf[2 * k], Yf[2 * k + 1]};
    }
  }

  float acc[PPL][NCH];
  float Tr[PPL];
  bool alive[PPL];
#pragma unroll
  for (int j = 0; j < PPL; ++j) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[j][c] = 0.0f;
    Tr[j] = 1.0f;
    alive[j] = valid[j];
  }

  for (int base = 0; base < n; base += kBatch) {
    const int nb = min(kBatch, n - base);
    if (base > 0) __syncthreads();  // everyone is done with the previous batch
    stage_batch<MODE, CB, NT>(S, p, st + base, nb);
    __syncthreads();

    for (int g = 0; g < nb; ++g) {
      bool any_alive = false;
#pragma unroll
      for (int j = 0; j < PPL; ++j) any_alive |= alive[j];
      if (__ballot(any_alive) == 0ull) break;  // this wave's 64*PPL pixels are saturated

      const GRec r = load_rec<MODE, CB>(S, g);
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
      if (__ballot(any_con) == 0ull) continue;  // nobody in the wave sees this Gaussian

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
            v2f s2 = q[0] * Yp[j][0];
#pragma unroll
            for (int k = 1; k < TR::NPAIR; ++k) s2 = fma2(q[k], Yp[j][k], s2);
            acc[j][c] += w[j] * sigmoid_fast(s2[0] + s2[1]);
          }
        }
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
          const float om = con[j] ? (1.0f - r.a * G[j]) : 1.0f;
          Tr[j] *= om;
          alive[j] = alive[j] && !(Tr[j] < p.thresh);
        }
      } else {
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
          if (con[j]) {
            const float ag = r.a * G[j];
            const float coeff = (r.a * Tr[j]) * G[j];
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc[j][c] += cg[c] * coeff;
            Tr[j] *= (1.0f - ag);
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
}

// ============================================================================================
// backward
// ============================================================================================
template <int MODE, int CB, int PPL>
__global__ void __launch_bounds__(256 / PPL) k_composite_bwd_pixel(CompParams p) {
  using TR = Traits<MODE, CB>;
  constexpr int NT = 256 / PPL;
  constexpr int ROWS = NT / 16;
  constexpr int NCH = TR::NCH;
  constexpr int P = TR::P;
  __shared__ Stage<MODE, CB> S;

  int tx, ty;
  if (!block_tile(p, tx, ty)) return;  // uniform over the workgroup
  const int tile = ty * p.ntw + tx;
  const int st = p.start[tile];
  const int n = (st < 0) ? 0 : (p.end[tile] - st);
  if (n == 0 || n < p.n_lo || n >= p.n_hi) return;
  const int t = (int)threadIdx.x;
  const int lane = t & 63;
  const int lx = t & 15, ly0 = t >> 4;
  const int gx = tx * kTile + lx;

  bool valid[PPL];
  int gy[PPL];
  float py[PPL];
  const float px = pixel_coord(p.topleft[0], gx, p.psx);
#pragma unroll
  for (int j = 0; j < PPL; ++j) {
    gy[j] = ty * kTile + ly0 + j * ROWS;
    valid[j] = (gx < p.W) && (gy[j] < p.H);
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
      float dx = R[0] * px + R[1] * py[j] + R[2];
      float dy = R[3] * px + R[4] * py[j] + R[5];
      float dz = R[6] * px + R[7] * py[j] + R[8];
      const float len = sqrtf(dx * dx + dy * dy + dz * dz);
      dx /= len; dy /= len; dz /= len;
      float Yf[TR::CCP];
#pragma unroll
      for (int k = 0; k < TR::CCP; ++k) Yf[k] = 0.0f;
      sh_basis<CB>(dx, dy, dz, *reinterpret_cast<float (*)[TR::CC]>(&Yf[0]));
#pragma unroll
      for (int k = 0; k < TR::NPAIR; ++k) Yp[j][k] = v2f{Yf[2 * k], Yf[2 * k + 1]};
    }
  }

  float go[PPL][NCH], fin[PPL][NCH], pre[PPL][NCH], Tr[PPL];
  bool alive[PPL];
#pragma unroll
  for (int j = 0; j < PPL; ++j) {
    const size_t pix = valid[j] ? ((size_t)gy[j] * p.W + gx) : 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      go[j][c] = valid[j] ? p.grad_out[NCH * pix + c] : 0.0f;
      fin[j][c] = valid[j] ? p.final_img[NCH * pix + c] : 0.0f;
      pre[j][c] = 0.0f;
    }
    Tr[j] = 1.0f;
    alive[j] = valid[j];
  }

  for (int base = 0; base < n; base += kBatch) {
    const int nb = min(kBatch, n - base);
    if (base > 0) __syncthreads();
    stage_batch<MODE, CB, NT>(S, p, st + base, nb);
    __syncthreads();

    for (int g = 0; g < nb; ++g) {
      bool any_alive = false;
#pragma unroll
      for (int j = 0; j < PPL; ++j) any_alive |= alive[j];
      if (__ballot(any_alive) == 0ull) break;

      const GRec r = load_rec<MODE, CB>(S, g);
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
      if (__ballot(any_con) == 0ull) continue;

      // per-lane partial gradient of this Gaussian over the lane's pixels
      // layout: 0,1 mean | 2 c00 3 c01 4 c10 5 c11 | 6 alpha | 7.. colour/scalar/sh
      float gr[P];
#pragma unroll
      for (int i = 0; i < P; ++i) gr[i] = 0.0f;

      float inv_det;
      if constexpr (MODE == MODE_SH) {
        inv_det = r.p1;
      } else {
        inv_det = 1.0f / (r.c0 * r.c3 - r.c1 * r.c2);
      }
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
        const float vx = (x * r.c3 - y * r.c2) * inv_det;
        const float vy = (y * r.c0 - x * r.c1) * inv_det;
        gr[0] += gg * vx;
        gr[1] += gg * vy;
        const float h = 0.5f * gg;
        gr[2] += h * vx * vx;
        gr[3] += h * vx * vy;
        gr[5] += h * vy * vy;
        gr[6] += pa * G[j];
        const float om = con[j] ? (1.0f - ag[j]) : 1.0f;
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

// ---- launch helpers ---------------------------------------------------------------------------
// Pixels per lane: 4 = one wavefront per tile (north_star design), 2 / 1 = two / four
// wavefronts per tile sharing the staged records.  Defaults chosen by measurement on MI355X
// (profiles/r01_notes.md): forward 1 (127 us vs 195 us at 4: one wave per tile leaves 2.4 waves
// per SIMD and the launch ends on the centre tiles' serial chains), backward 4 (the per-Gaussian
// gradient reduction costs the same per wave whatever the number of pixels behind it).
// GSGEN_PPL_FWD / GSGEN_PPL_BWD override them for A/B runs.
template <int MODE, int CB>
static int launch_fwd(const CompParams &p, hipStream_t s) {
  static const int ppl = env_ppl("GSGEN_PPL_FWD", 1);
  const uint32_t nblk = comp_grid(p);
  if (p.ntw * p.nth == 0) return 0;
  if (ppl == 1) hipLaunchKernelGGL((k_composite_fwd<MODE, CB, 1>), dim3(nblk), dim3(256), 0, s, p);
  else if (ppl == 2) hipLaunchKernelGGL((k_composite_fwd<MODE, CB, 2>), dim3(nblk), dim3(128), 0, s, p);
  else hipLaunchKernelGGL((k_composite_fwd<MODE, CB, 4>), dim3(nblk), dim3(64), 0, s, p);
  return (int)hipGetLastError();
}
template <int MODE, int CB>
static int launch_bwd(const CompParams &p_, hipStream_t s) {
  static const int ppl = env_ppl("GSGEN_PPL_BWD", 4);
  CompParams p = p_;
  if (p.n_hi == 0) p.n_hi = 0x7fffffff;
  const uint32_t nblk = comp_grid(p);
  if (p.ntw * p.nth == 0) return 0;
  if (ppl == 1) hipLaunchKernelGGL((k_composite_bwd_pixel<MODE, CB, 1>), dim3(nblk), dim3(256), 0, s, p);
  else if (ppl == 2) hipLaunchKernelGGL((k_composite_bwd_pixel<MODE, CB, 2>), dim3(nblk), dim3(128), 0, s, p);
  else hipLaunchKernelGGL((k_composite_bwd_pixel<MODE, CB, 4>), dim3(nblk), dim3(64), 0, s, p);
  return (int)hipGetLastError();
}

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
  p.psx = pixel_size_x; p.psy = pixel_size_y; p.thresh = thresh;
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
  p.psx = pixel_size_x; p.psy = pixel_size_y; p.thresh = thresh;
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
  p.psx = pixel_size_x; p.psy = pixel_size_y; p.thresh = thresh;
  p.tile_order = tile_order;
  return launch_fwd<MODE_RGBD, 1>(p, (hipStream_t)stream);
}

int gsgen_vol_render_sh_ordered(uint32_t N, uint32_t D, const float *mean, const float *cov,
                                const float *sh_coeffs, const float *alpha, const int *start,
                                const int *end, const int *gaussian_ids, float *out, const float *topleft,
                                const float *c2w, uint32_t tile_size, uint32_t n_tiles_h,
                                uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y, uint32_t H,
                                uint32_t W, uint32_t C, float thresh, const float *bg_rgb, float *T,
                                const uint32_t *tile_order, gsgen_stream_t stream) {
  if (int e = check_common(tile_size, start, end, out)) return e;
  if (C < 1 || C > 4) return GSGEN_EUNSUPPORTED;  // reference dispatches C = 1..4 only (render.cu:507-544)
  if (!c2w) return GSGEN_EINVAL;
  if ((N == 0 || D == 0) && bg_rgb == nullptr) return 0;
  CompParams p{};
  p.mean = mean; p.cov = cov; p.col = sh_coeffs; p.alpha = alpha;
  p.start = start; p.end = end; p.ids = gaussian_ids; p.topleft = topleft; p.rot = c2w;
  p.bg = bg_rgb; p.out = out; p.T = T;
  p.ntw = (int)n_tiles_w; p.nth = (int)n_tiles_h; p.H = (int)H; p.W = (int)W;
  p.psx = pixel_size_x; p.psy = pixel_size_y; p.thresh = thresh;
  p.tile_order = tile_order;
  hipStream_t s = (hipStream_t)stream;
  switch (C) {
    case 1: return launch_fwd<MODE_SH, 1>(p, s);
    case 2: return launch_fwd<MODE_SH, 2>(p, s);
    case 3: return launch_fwd<MODE_SH, 3>(p, s);
    default: return launch_fwd<MODE_SH, 4>(p, s);
  }
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
