// composite_bwd.hip -- backward of the front-to-back compositing, SPLAT-PARALLEL, for gfx950.
//
// Replaces (reference paths relative to /root/reference/gs/src/include):
//   RGB     vol_render.h:866-973 (body :318-418)
//   scalar  vol_render_scalar.h:104-234
//   SH      vol_render_sh.h:268-455 / vol_render_bg.h:131-242
// and the Gaussian backward kernels.h:394-418.
//
// The reference (and the first version of this file) walks the list with one thread per PIXEL
// and therefore has to sum every Gaussian's gradient over the 256 pixels of the tile: one LDS
// atomic per (pixel, Gaussian, component) there, a 55-component cross-lane reduce-scatter per
// (tile, Gaussian) here -- which measured ~1000 SIMD cycles each on MI355X, half of the kernel.
//
// This kernel turns the loop nest inside out.  A workgroup owns one 16x16 tile; the
// depth-sorted list is taken 64 Gaussians at a time and LANE g OWNS GAUSSIAN g of the batch:
// its SH coefficients and its 55 gradient accumulators live in that lane's registers.  The
// four waves of the workgroup each walk a quarter of the tile's pixels; for one pixel
//   * the per-pixel constants (position, SH basis, grad_out, final colour) are wave-uniform
//     broadcast reads of a table built once per tile in LDS,
//   * the transmittance in front of every Gaussian is an exclusive PREFIX PRODUCT over the
//     lanes, the prefix colour an inclusive PREFIX SUM (3 channels): four 7-step DPP scans,
//   * every gradient contribution is lane-local.
// The front-to-back recurrences of the reference become scans, the cross-pixel sums become
// plain register accumulation, and nothing is reduced across lanes.  At the end of a batch the
// four waves' accumulators meet in LDS and leave as one coalesced vector atomic per Gaussian.
//
// Semantics are the reference's: the prefix at a Gaussian only contains the Gaussians in
// front of it; "stop when T < thresh, tested before each splat" is the monotone mask
// (T_before < thresh); skip when a*G < 1/255; alpha clamped at 0.99 but differentiated as if
// it were not (vol_render.h:365,409); `final` includes bg*T.
#include "composite_common.hpp"

namespace gs {

// the pixel-parallel backward of composite.hip (kept for A/B: GSGEN_BWD=pixel)
int launch_bwd_pixel_dispatch(int mode, int C, const CompParams &p, hipStream_t s);

template <int MODE, int CB>
struct BwdCfg {
  using TR = Traits<MODE, CB>;
  static constexpr int NCH = TR::NCH;
  static constexpr int NCOL = TR::NCOL;
  // per-pixel constant block: px, py | gout[NCH] | fin[NCH] | Y[CCP]   (padded to float4s)
  static constexpr int NCONST_RAW = 2 + 2 * NCH + (MODE == MODE_SH ? TR::CCP : 0);
  static constexpr int NCONST = (NCONST_RAW + 3) & ~3;
  static constexpr int OFF_GO = 2, OFF_FIN = 2 + NCH, OFF_Y = 2 + 2 * NCH;
  // staged coefficient record stride in LDS (floats): odd multiple of 4 -> conflict-free
  // ds_read_b128 when lane g reads record g
  static constexpr int RSTRIDE = (MODE == MODE_SH) ? (((TR::NCOLP + 3) & ~3) | 4) : NCOL;
  static constexpr int NCOMP = TR::NCOMP;
  static constexpr int ASTRIDE = NCOMP | 1;  // accumulator row stride (odd -> conflict-free)
};

template <int MODE, int CB>
__global__ void __launch_bounds__(256) k_composite_bwd_splat(CompParams p) {
  using TR = Traits<MODE, CB>;
  using CF = BwdCfg<MODE, CB>;
  constexpr int NCH = CF::NCH;
  constexpr int NCOL = CF::NCOL;
  constexpr int NCOMP = CF::NCOMP;

  __shared__ alignas(16) float s_const[256 * CF::NCONST];
  __shared__ float s_T[256];
  __shared__ float s_pre[NCH][256];
  // staged records of the batch; re-used as the cross-wave accumulator block afterwards
  constexpr int kRecFloats = kBatch * CF::RSTRIDE;
  constexpr int kAccFloats = kBatch * CF::ASTRIDE;
  __shared__ alignas(16) float s_rec[(kRecFloats > kAccFloats ? kRecFloats : kAccFloats) + 4];
  __shared__ int s_id[kBatch];
  __shared__ int s_any[kBatch];

  int tx, ty;
  if (!block_tile(p, tx, ty)) return;  // uniform over the workgroup
  const int tile = ty * p.ntw + tx;
  const int st = p.start[tile];
  const int n = (st < 0) ? 0 : (p.end[tile] - st);
  if (n == 0 || n < p.n_lo || n >= p.n_hi) return;
  const int t = (int)threadIdx.x;
  const int lane = t & 63, wave = t >> 6;

  // ---- per-pixel constant table (thread t <-> pixel t of the tile, row-major 16x16) -------
  {
    const int lx = t & 15, ly = t >> 4;
    const int gx = tx * kTile + lx, gy = ty * kTile + ly;
    const bool valid = (gx < p.W) && (gy < p.H);
    const float px = pixel_coord(p.topleft[0], gx, p.psx);
    const float py = pixel_coord(p.topleft[1], gy, p.psy);
    float *c = &s_const[t * CF::NCONST];
    c[0] = px; c[1] = py;
    const size_t pix = valid ? ((size_t)gy * p.W + gx) : 0;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      c[CF::OFF_GO + k] = valid ? p.grad_out[NCH * pix + k] : 0.0f;
      c[CF::OFF_FIN + k] = valid ? p.final_img[NCH * pix + k] : 0.0f;
      s_pre[k][t] = 0.0f;
    }
    if constexpr (MODE == MODE_SH) {
      float dx = p.rot[0] * px + p.rot[1] * py + p.rot[2];
      float dy = p.rot[3] * px + p.rot[4] * py + p.rot[5];
      float dz = p.rot[6] * px + p.rot[7] * py + p.rot[8];
      const float len = sqrtf(dx * dx + dy * dy + dz * dz);
      dx /= len; dy /= len; dz /= len;
      float Yf[TR::CCP];
#pragma unroll
      for (int k = 0; k < TR::CCP; ++k) Yf[k] = 0.0f;
      sh_basis<CB>(dx, dy, dz, *reinterpret_cast<float (*)[TR::CC]>(&Yf[0]));
#pragma unroll
      for (int k = 0; k < TR::CCP; ++k) c[CF::OFF_Y + k] = Yf[k];
    }
    s_T[t] = valid ? 1.0f : 0.0f;  // an out-of-image pixel is dead from the start
  }
  if constexpr (MODE == MODE_SH && TR::CCP != TR::CC) {
    for (int e = t; e < kRecFloats; e += 256) s_rec[e] = 0.0f;  // pad lanes of the coefficient rows
  }

  for (int base = 0; base < n; base += kBatch) {
    const int nb = min(kBatch, n - base);
    __syncthreads();  // const table ready / previous batch's accumulator block drained

    // ---- stage the batch: ids, then coalesced coefficient rows --------------------------
    if (t < kBatch) s_id[t] = (t < nb) ? p.ids[st + base + t] : 0;
    __syncthreads();
    if constexpr (MODE == MODE_SH) {
      if constexpr (TR::CCP == TR::CC) {
        constexpr int Q = NCOL / 4;
        for (int e = t; e < nb * Q; e += 256) {
          const int g = e / Q, k = e - g * Q;
          const float4 v = *reinterpret_cast<const float4 *>(p.col + (size_t)s_id[g] * NCOL + 4 * k);
          *reinterpret_cast<float4 *>(&s_rec[g * CF::RSTRIDE + 4 * k]) = v;
        }
      } else {
        for (int e = t; e < nb * NCOL; e += 256) {
          const int g = e / NCOL, k = e - g * NCOL;
          const int c = k / TR::CC, kk = k - c * TR::CC;
          s_rec[g * CF::RSTRIDE + c * TR::CCP + kk] = p.col[(size_t)s_id[g] * NCOL + k];
        }
      }
    }
    __syncthreads();

    // ---- lane g takes Gaussian g into registers -----------------------------------------
    const bool have = lane < nb;
    const int id = s_id[lane];
    GRec r;
    {
      float mx = 0.f, my = 0.f, c0 = 1.f, c1 = 0.f, c2 = 0.f, c3 = 1.f, al = 0.f;
      if (have) {
        const float2 m = *reinterpret_cast<const float2 *>(p.mean + 2 * (size_t)id);
        const float4 c = *reinterpret_cast<const float4 *>(p.cov + 4 * (size_t)id);
        mx = m.x; my = m.y; c0 = c.x; c1 = c.y; c2 = c.z; c3 = c.w;
        al = p.alpha[id];
      }
      r = prep_record<MODE>(mx, my, c0, c1, c2, c3, al);
      if (!have) r.a = 0.0f;
    }
    float inv_det;
    if constexpr (MODE == MODE_SH) inv_det = r.p1;
    else inv_det = 1.0f / (r.c0 * r.c3 - r.c1 * r.c2);

    v2f q[MODE == MODE_SH ? 3 : 1][MODE == MODE_SH ? TR::NPAIR : 1];  // SH coefficients, (k,k+1) pairs
    float colv[MODE == MODE_SH ? 1 : NCH];                            // rgb / scalar value
    if constexpr (MODE == MODE_SH) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < TR::NPAIR; ++k)
          q[c][k] = *reinterpret_cast<const v2f *>(&s_rec[lane * CF::RSTRIDE + c * TR::CCP + 2 * k]);
    } else {
#pragma unroll
      for (int c = 0; c < NCH; ++c) colv[c] = 0.0f;
      if (have) load_channels<MODE, NCOL>(p, id, colv);
    }

    // gradient accumulators of this lane's Gaussian over this wave's pixels
    float g_m0 = 0.f, g_m1 = 0.f, g_c00 = 0.f, g_c01 = 0.f, g_c11 = 0.f, g_a = 0.f;
    v2f g_sh[MODE == MODE_SH ? 3 : 1][MODE == MODE_SH ? TR::NPAIR : 1];
    float g_col[MODE == MODE_SH ? 1 : NCH];
    if constexpr (MODE == MODE_SH) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < TR::NPAIR; ++k) g_sh[c][k] = v2f{0.f, 0.f};
    } else {
#pragma unroll
      for (int c = 0; c < NCH; ++c) g_col[c] = 0.f;
    }
    bool touched = false;
    bool wave_alive = false;

    // ---- this wave's 64 pixels ------------------------------------------------------------
    for (int i = 0; i < 64; ++i) {
      const int pidx = wave * 64 + i;
      const float Tin = s_T[pidx];
      if (__builtin_amdgcn_readfirstlane(__float_as_int(Tin < p.thresh ? 1.0f : 0.0f)) != 0) continue;
      const float *pc = &s_const[pidx * CF::NCONST];
      const float px = pc[0], py = pc[1];
      const float x = px - r.mx, y = py - r.my;
      const float G = gauss_eval<MODE>(r, x, y, px, py, true);
      const float ag = r.a * G;
      const bool con = !(ag < kMinAlpha);  // r.a == 0 for padding / degenerate lanes -> never
      if (__ballot(con) == 0ull) { wave_alive = true; continue; }

      // transmittance in front of each Gaussian: exclusive prefix product of (1 - a G)
      const float om = con ? (1.0f - ag) : 1.0f;
      const float Tincl = wave_scan_mul(om);
      const float Tex = Tin * wave_shift_up(1.0f, Tincl);
      const bool dead = Tex < p.thresh;  // monotone: once dead, every later lane is dead
      const bool live = con && !dead;
      const float w = live ? (r.a * Tex) * G : 0.0f;

      // colour of this Gaussian at this pixel, weighted prefix colour
      float yv[NCH], inc[NCH];
      if constexpr (MODE == MODE_SH) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          v2f s2 = q[c][0] * (*reinterpret_cast<const v2f *>(pc + CF::OFF_Y));
#pragma unroll
          for (int k = 1; k < TR::NPAIR; ++k)
            s2 = fma2(q[c][k], *reinterpret_cast<const v2f *>(pc + CF::OFF_Y + 2 * k), s2);
          yv[c] = sigmoid_fast(s2[0] + s2[1]);
        }
      } else {
#pragma unroll
        for (int c = 0; c < NCH; ++c) yv[c] = colv[c];
      }
#pragma unroll
      for (int c = 0; c < NCH; ++c) inc[c] = s_pre[c][pidx] + wave_scan_add(w * yv[c]);

      // pixel state for the next batch: T where the walk stopped, total prefix colour
      {
        const unsigned long long dm = __ballot(dead);
        const float Tlast = read_lane(Tex * om, 63);
        const float Tstop = read_lane(Tex, dm ? (__ffsll((long long)dm) - 1) : 0);
        const float Tout = dm ? Tstop : Tlast;
        if (lane == 63) {
          s_T[pidx] = Tout;
#pragma unroll
          for (int c = 0; c < NCH; ++c) s_pre[c][pidx] = inc[c];
        }
        wave_alive = wave_alive || !(Tout < p.thresh);
      }

      // lane-local gradient contributions
      const float inv1m = __builtin_amdgcn_rcpf(1.0f - ag);
      float pAG = 0.0f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const float go = pc[CF::OFF_GO + c], fin = pc[CF::OFF_FIN + c];
        pAG += go * (yv[c] * Tex - (fin - inc[c]) * inv1m);
        if constexpr (MODE == MODE_SH) {
          const v2f gs2 = splat2(w * (yv[c] * (1.0f - yv[c])) * go);
#pragma unroll
          for (int k = 0; k < TR::NPAIR; ++k)
            g_sh[c][k] = fma2(gs2, *reinterpret_cast<const v2f *>(pc + CF::OFF_Y + 2 * k), g_sh[c][k]);
        } else {
          g_col[c] += w * go;
        }
      }
      const float pa = live ? pAG : 0.0f;
      const float gg = pa * ag;
      const float vx = (x * r.c3 - y * r.c2) * inv_det;
      const float vy = (y * r.c0 - x * r.c1) * inv_det;
      g_m0 += gg * vx;
      g_m1 += gg * vy;
      const float h = 0.5f * gg;
      g_c00 += h * vx * vx;
      g_c01 += h * vx * vy;
      g_c11 += h * vy * vy;
      g_a += pa * G;
      touched = touched || live;
    }

    // ---- the four waves' accumulators meet in LDS ----------------------------------------
    const int any_alive = __syncthreads_or((int)wave_alive);  // also: everyone is done reading s_rec
    float *acc = &s_rec[lane * CF::ASTRIDE];
    for (int wv = 0; wv < 4; ++wv) {
      if (wave == wv) {
        float vals[NCOMP];
        vals[0] = g_m0; vals[1] = g_m1; vals[2] = g_c00; vals[3] = g_c01; vals[4] = g_c01; vals[5] = g_c11;
        vals[6] = g_a;
        if constexpr (MODE == MODE_SH) {
#pragma unroll
          for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int k = 0; k < TR::CC; ++k) vals[7 + c * TR::CC + k] = g_sh[c][k >> 1][k & 1];
        } else {
#pragma unroll
          for (int c = 0; c < NCH; ++c) vals[7 + c] = g_col[c];
        }
#pragma unroll
        for (int k = 0; k < NCOMP; ++k) acc[k] = (wv == 0) ? vals[k] : acc[k] + vals[k];
        if (wv == 0) s_any[lane] = touched ? 1 : 0;
        else if (touched) s_any[lane] = 1;
      }
      __syncthreads();
    }
    // one coalesced vector atomic per Gaussian that received anything
    for (int g = wave; g < nb; g += 4) {
      if (s_any[g] == 0) continue;
      if (lane < NCOMP) {
        const size_t gid = (size_t)s_id[g];
        float *dst;
        if (lane < 2) dst = p.g_mean + 2 * gid + lane;
        else if (lane < 6) dst = p.g_cov + 4 * gid + (lane - 2);
        else if (lane == 6) dst = p.g_alpha + gid;
        else dst = p.g_col + (size_t)NCOL * gid + (lane - 7);
        atomicAdd(dst, s_rec[g * CF::ASTRIDE + lane]);
      }
    }
    if (any_alive == 0) break;  // every pixel of the tile is saturated
    if constexpr (MODE == MODE_SH && TR::CCP != TR::CC) {
      __syncthreads();
      for (int e = t; e < kRecFloats; e += 256) s_rec[e] = 0.0f;  // restore the zero pad lanes
    }
  }
}

template <int MODE, int CB>
static int launch_bwd_splat(const CompParams &p, hipStream_t s) {
  const uint32_t nblk = comp_grid(p);
  if (p.ntw * p.nth == 0) return 0;
  hipLaunchKernelGGL((k_composite_bwd_splat<MODE, CB>), dim3(nblk), dim3(256), 0, s, p);
  return (int)hipGetLastError();
}

static int launch_bwd_splat_dispatch(int mode, int C, const CompParams &p, hipStream_t s) {
  if (mode == MODE_RGB) return launch_bwd_splat<MODE_RGB, 1>(p, s);
  if (mode == MODE_SCALAR) return launch_bwd_splat<MODE_SCALAR, 1>(p, s);
  if (mode == MODE_RGBD) return launch_bwd_splat<MODE_RGBD, 1>(p, s);
  switch (C) {
    case 1: return launch_bwd_splat<MODE_SH, 1>(p, s);
    case 2: return launch_bwd_splat<MODE_SH, 2>(p, s);
    case 3: return launch_bwd_splat<MODE_SH, 3>(p, s);
    default: return launch_bwd_splat<MODE_SH, 4>(p, s);
  }
}

static int launch_bwd(int mode, int C, const CompParams &p_, hipStream_t s) {
  // Two complete implementations (both parity-green).  GSGEN_BWD=pixel|splat forces one;
  // GSGEN_BWD_SPLIT=n sends tiles whose list is >= n entries long to the splat-parallel kernel
  // (four waves per tile, no serial chain per wave) and the rest to the pixel-parallel one.
  static const char *force = getenv("GSGEN_BWD");
  static const int split = getenv("GSGEN_BWD_SPLIT") ? atoi(getenv("GSGEN_BWD_SPLIT")) : 0;
  CompParams p = p_;
  p.n_lo = 0; p.n_hi = 0x7fffffff;
  if (force && force[0] == 's') return launch_bwd_splat_dispatch(mode, C, p, s);
  if (force && force[0] == 'p') return launch_bwd_pixel_dispatch(mode, C, p, s);
  if (split > 0) {
    p.n_lo = split;
    if (int e = launch_bwd_splat_dispatch(mode, C, p, s)) return e;
    p.n_lo = 0; p.n_hi = split;
    return launch_bwd_pixel_dispatch(mode, C, p, s);
  }
  return launch_bwd_pixel_dispatch(mode, C, p, s);
}

}  // namespace gs

using namespace gs;

extern "C" {

int gsgen_vol_render_backward_start_end(uint32_t N, uint32_t D, const float *mean, const float *cov,
                                        const float *color, const float *alpha, const int *start,
                                        const int *end, const int *gaussian_ids, const float *out,
                                        float *grad_mean, float *grad_cov, float *grad_color,
                                        float *grad_alpha, const float *grad_out,
                                        const float *topleft, uint32_t tile_size,
                                        uint32_t n_tiles_h, uint32_t n_tiles_w, float pixel_size_x,
                                        float pixel_size_y, uint32_t H, uint32_t W, float thresh,
                                        gsgen_stream_t stream) {
  if (int e = check_common(tile_size, start, end, out)) return e;
  if (N == 0 || D == 0) return 0;
  CompParams p{};
  p.mean = mean; p.cov = cov; p.col = color; p.alpha = alpha;
  p.start = start; p.end = end; p.ids = gaussian_ids; p.topleft = topleft;
  p.final_img = out; p.grad_out = grad_out;
  p.g_mean = grad_mean; p.g_cov = grad_cov; p.g_col = grad_color; p.g_alpha = grad_alpha;
  p.ntw = (int)n_tiles_w; p.nth = (int)n_tiles_h; p.H = (int)H; p.W = (int)W;
  p.psx = pixel_size_x; p.psy = pixel_size_y; p.thresh = thresh;
  return launch_bwd(MODE_RGB, 1, p, (hipStream_t)stream);
}

int gsgen_vol_render_rgbd_backward(uint32_t N, uint32_t D, const float *mean, const float *cov, const float *color,
                                   const float *depth, const float *alpha, const int *start, const int *end,
                                   const int *gaussian_ids, const float *out6, float *grad_mean, float *grad_cov,
                                   float *grad_chan6, float *grad_alpha, const float *grad_out6,
                                   const float *topleft, uint32_t tile_size, uint32_t n_tiles_h,
                                   uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y, uint32_t H,
                                   uint32_t W, float thresh, const uint32_t *tile_order, gsgen_stream_t stream) {
  if (int e = check_common(tile_size, start, end, out6)) return e;
  if (N == 0 || D == 0) return 0;
  if (!depth) return GSGEN_EINVAL;
  CompParams p{};
  p.mean = mean; p.cov = cov; p.col = color; p.depth = depth; p.alpha = alpha;
  p.start = start; p.end = end; p.ids = gaussian_ids; p.topleft = topleft;
  p.final_img = out6; p.grad_out = grad_out6;
  p.g_mean = grad_mean; p.g_cov = grad_cov; p.g_col = grad_chan6; p.g_alpha = grad_alpha;
  p.ntw = (int)n_tiles_w; p.nth = (int)n_tiles_h; p.H = (int)H; p.W = (int)W;
  p.psx = pixel_size_x; p.psy = pixel_size_y; p.thresh = thresh;
  p.tile_order = tile_order;
  return launch_bwd(MODE_RGBD, 1, p, (hipStream_t)stream);
}

int gsgen_vol_render_scalar_backward(uint32_t N, uint32_t D, const float *mean, const float *cov,
                                     const float *scalar, const float *alpha, const int *start,
                                     const int *end, const int *gaussian_ids, const float *out,
                                     float *grad_mean, float *grad_cov, float *grad_scalar,
                                     float *grad_alpha, const float *grad_out, const float *topleft,
                                     uint32_t tile_size, uint32_t n_tiles_h, uint32_t n_tiles_w,
                                     float pixel_size_x, float pixel_size_y, uint32_t H, uint32_t W,
                                     float thresh, gsgen_stream_t stream) {
  if (int e = check_common(tile_size, start, end, out)) return e;
  if (N == 0 || D == 0) return 0;
  CompParams p{};
  p.mean = mean; p.cov = cov; p.col = scalar; p.alpha = alpha;
  p.start = start; p.end = end; p.ids = gaussian_ids; p.topleft = topleft;
  p.final_img = out; p.grad_out = grad_out;
  p.g_mean = grad_mean; p.g_cov = grad_cov; p.g_col = grad_scalar; p.g_alpha = grad_alpha;
  p.ntw = (int)n_tiles_w; p.nth = (int)n_tiles_h; p.H = (int)H; p.W = (int)W;
  p.psx = pixel_size_x; p.psy = pixel_size_y; p.thresh = thresh;
  return launch_bwd(MODE_SCALAR, 1, p, (hipStream_t)stream);
}

int gsgen_vol_render_backward_sh_ordered(uint32_t N, uint32_t D, const float *mean, const float *cov,
                                         const float *sh_coeffs, const float *alpha, const int *start,
                                         const int *end, const int *gaussian_ids, const float *out,
                                         float *grad_mean, float *grad_cov, float *grad_sh_coeffs,
                                         float *grad_alpha, const float *grad_out, const float *topleft,
                                         const float *c2w, uint32_t tile_size, uint32_t n_tiles_h,
                                         uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y,
                                         uint32_t H, uint32_t W, uint32_t C, float thresh,
                                         const float *bg_rgb, const uint32_t *tile_order,
                                         gsgen_stream_t stream) {
  return gsgen_vol_render_backward_sh_segmented(N, D, mean, cov, sh_coeffs, alpha, start, end, gaussian_ids, out,
                                                grad_mean, grad_cov, grad_sh_coeffs, grad_alpha, grad_out, topleft,
                                                c2w, tile_size, n_tiles_h, n_tiles_w, pixel_size_x, pixel_size_y, H,
                                                W, C, thresh, bg_rgb, tile_order, nullptr, 0, stream);
}

int gsgen_vol_render_backward_sh_segmented(uint32_t N, uint32_t D, const float *mean, const float *cov,
                                           const float *sh_coeffs, const float *alpha, const int *start,
                                           const int *end, const int *gaussian_ids, const float *out,
                                           float *grad_mean, float *grad_cov, float *grad_sh_coeffs,
                                           float *grad_alpha, const float *grad_out, const float *topleft,
                                           const float *c2w, uint32_t tile_size, uint32_t n_tiles_h,
                                           uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y,
                                           uint32_t H, uint32_t W, uint32_t C, float thresh,
                                           const float *bg_rgb, const uint32_t *tile_order,
                                           const void *segment_workspace, uint32_t n_segments,
                                           gsgen_stream_t stream) {
  (void)bg_rgb;  // the background only enters through `out` (= final incl. bg*T)
  if (int e = check_common(tile_size, start, end, out)) return e;
  if (n_segments > 1 && segment_workspace == nullptr) return GSGEN_EINVAL;
  if (C < 1 || C > 4) return GSGEN_EUNSUPPORTED;
  if (!c2w) return GSGEN_EINVAL;
  if (N == 0 || D == 0) return 0;
  CompParams p{};
  p.mean = mean; p.cov = cov; p.col = sh_coeffs; p.alpha = alpha;
  p.start = start; p.end = end; p.ids = gaussian_ids; p.topleft = topleft; p.rot = c2w;
  p.final_img = out; p.grad_out = grad_out;
  p.g_mean = grad_mean; p.g_cov = grad_cov; p.g_col = grad_sh_coeffs; p.g_alpha = grad_alpha;
  p.ntw = (int)n_tiles_w; p.nth = (int)n_tiles_h; p.H = (int)H; p.W = (int)W;
  p.psx = pixel_size_x; p.psy = pixel_size_y; p.thresh = thresh;
  p.tile_order = tile_order;
  if (n_segments > 1) {
    p.nseg = (int)n_segments;
    p.ckpt = reinterpret_cast<float4 *>(const_cast<void *>(segment_workspace));
    p.stop = reinterpret_cast<int *>(p.ckpt + (size_t)n_tiles_h * n_tiles_w * 256 * n_segments);
  }
  return launch_bwd(MODE_SH, (int)C, p, (hipStream_t)stream);
}

int gsgen_vol_render_backward_sh(uint32_t N, uint32_t D, const float *mean, const float *cov,
                                 const float *sh_coeffs, const float *alpha, const int *start,
                                 const int *end, const int *gaussian_ids, const float *out,
                                 float *grad_mean, float *grad_cov, float *grad_sh_coeffs,
                                 float *grad_alpha, const float *grad_out, const float *topleft,
                                 const float *c2w, uint32_t tile_size, uint32_t n_tiles_h,
                                 uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y,
                                 uint32_t H, uint32_t W, uint32_t C, float thresh,
                                 const float *bg_rgb, gsgen_stream_t stream) {
  return gsgen_vol_render_backward_sh_ordered(N, D, mean, cov, sh_coeffs, alpha, start, end, gaussian_ids, out,
                                              grad_mean, grad_cov, grad_sh_coeffs, grad_alpha, grad_out, topleft,
                                              c2w, tile_size, n_tiles_h, n_tiles_w, pixel_size_x, pixel_size_y,
                                              H, W, C, thresh, bg_rgb, nullptr, stream);
}

}  // extern "C"
