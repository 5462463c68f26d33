// composite_common.hpp -- pieces shared by the compositing forward (composite.hip) and the
// splat-parallel backward (composite_bwd.hip): parameter block, the reference-arithmetic
// Gaussian evaluations of the guard path, the SH basis, per-record preparation.
#pragma once
#include "common.hpp"
#include "../../include/gsgen_hip.h"

namespace gs {

// MODE_RGBD: RGB and the three scalar heads of render_one (depth, opacity, depth^2) composited in
// one pass: 6 channels whose per-Gaussian values are (r, g, b, d, 1, d*d).
enum : int { MODE_RGB = 0, MODE_SCALAR = 1, MODE_SH = 2, MODE_RGBD = 3 };
constexpr int kBatch = 64;  // Gaussian records staged per LDS round

struct CompParams {
  const float *mean, *cov, *col, *alpha;
  const float *depth;  // MODE_RGBD only: per-Gaussian depth feeding the (d, 1, d*d) channels
  const int *start, *end, *ids;
  const float *topleft, *rot, *bg;
  float *out, *T;
  const float *final_img, *grad_out;
  float *g_mean, *g_cov, *g_col, *g_alpha;
  int ntw, nth, H, W;
  float psx, psy, thresh;
  const uint32_t *tile_order;  // optional launch order (longest list first); NULL = spatial map
  int n_lo, n_hi;  // this launch only handles tiles with n_lo <= list length < n_hi (0, INT_MAX = all)
  // Segmented backward (SH): the forward leaves, per tile, the state in front of list entry
  // kSegLen * k (k = 1 .. nseg-1) and the entry at which each pixel stopped; the backward then runs
  // one workgroup per (tile, segment) instead of one per tile.  nseg <= 1: unsegmented.
  float4 *ckpt;  // [tile][nseg][256 pixels, row-major in the tile]: T, prefix rgb
  int *stop;     // [tile][256]: first list index the pixel did not process (n if it never saturated)
  int nseg;
  int tile_side;  // host side only (kernel selection): 0 / 16 = this library's tiles; 8, 32: see k_composite_fwd
  // MODE_RGBD backward with grad_out == NULL: the four head gradients as autograd delivers them (any may be NULL = 0)
  const float *go_rgb, *go_d, *go_o, *go_z2;  // [H,W,3], [H,W], [H,W], [H,W]
  // SH degree 3: DEVICE pointer to the coefficient bound S of the scene (gsgen_sh_l1_bound), or NULL.  With a bound the
  // host enqueues the ROUTED kernel (composite.hip, kRouted): every workgroup reads S and its view's pixel size and runs the
  // polynomial form of the per-pixel basis or the exact one (poly_route) -- no host decision, no host sync.
  const float *sh_bound;
  // Per-TILE routing (round 4; the *_routed entry points): sh_rows[i] = the largest sum_{k>=1} |sh| over the three channels of
  // splat i (gsgen_sh_l1_bound_rows), measured on the device per step.  A tile takes the polynomial form as long as every splat
  // it STAGES satisfies the bound for its view's pixel size (poly_row_ok); the first staged batch that holds a splat beyond it
  // sends the tile to the exact kernel: batched launches through tile_flags (one byte per tile of the view, written by the
  // polynomial forward -- 1 = exact -- and read by the exact fallback behind it and by both backward kernels), per-camera
  // launches by scanning the tile's list first (forward and backward scan the same list: the same decision).  One splat with
  // large higher-band coefficients, or a wide camera, then costs the tiles it touches, not the view.  NULL: per-view routing
  // on sh_bound as in round 3.
  const float *sh_rows;
  uint8_t *tile_flags;
  uint32_t vgrid;  // persistent launches (the exact fallback of a bounded batch): size of the virtual grid they stride through
  // post-activation channel modes, batched launches: the forward WRITES the empty tiles too (channels 0, T = 1), so the caller
  // need not pre-initialise [B,H,W,NCH] + [B,H,W] every step (the reference's contract -- "out zeroed, T set to 1 by the
  // caller", vol_render.h:1006-1013 -- stays that of the per-camera `_gs` entry points)
  int fill_empty;
  // RGB + heads, batched launches (round 6; gsgen_rgbd_view's optional fields): the four heads as separate contiguous images
  // (pl_rgb [H,W,3], pl_d / pl_o / pl_z [H,W]; the forward writes them instead of out, the backward reads them as the final image),
  // the background composited in the forward's epilogue (bg: rgb + T bg, gs/renderer.py:1182), its gradient summed by the
  // backward's prologue into 64 slots of g_bg ([64][4], slot = tile % 64: sum nan_to_num(grad_rgb T), gs/renderer.py:1283), and the
  // sixth head delivered as z_var = depth2 - depth^2 (gs/gaussian_splatting.py:1397) with grad_depth2 read as d L / d z_var.
  float *pl_rgb, *pl_d, *pl_o, *pl_z;
  float *g_bg;
  int zvar;
  // RGB / RGB + heads, batched launches: the view's prepared records [N,4] = (p0, p1, p2, ok) (gsgen_geometry_view::chol, written by
  // the projection launch once per (view, Gaussian)) -- staging reads them instead of running chol_prep's fp64 chain per staged
  // (tile, Gaussian) record in both kernels; the raw covariance is then fetched from `cov` only by the rare threshold guard.
  const float *chol;
  // SH degree 3, per-tile routing (gsgen_sh_view::route_report / no_fallback): the host-visible word a crowded tile is reported in,
  // and "no exact fallback is launched behind this kernel: keep every tile, take the per-entry exact tier"
  uint32_t *route_report;
  int no_fallback;
  // RGB / RGB + heads, batched launches (gsgen_rgbd_view::pixel_size_dev): the view's {psx, psy} in device memory -- read at the top of
  // the kernel in place of the two floats above, so that a captured step replays for other intrinsics (view_params below)
  const float *ps_dev;
};
// a batched channel-mode kernel's parameter block: the view's entry of the kernel-argument table, pixel sizes from device memory if given
__device__ __forceinline__ CompParams view_params(const CompParams &src) {
  CompParams p = src;
  if (p.ps_dev != nullptr) {
    p.psx = p.ps_dev[0];
    p.psy = p.ps_dev[1];
  }
  return p;
}
constexpr int kSegLen = 32;

// workgroup -> tile: explicit order if given, else the XCD-balanced spatial map
__device__ __forceinline__ bool block_tile(const CompParams &p, int &tx, int &ty, uint32_t bid) {
  if (p.tile_order != nullptr) {
    const uint32_t t = p.tile_order[bid];
    tx = (int)(t % (uint32_t)p.ntw);
    ty = (int)(t / (uint32_t)p.ntw);
    return true;
  }
  return tile_of_block(bid, p.ntw, p.nth, tx, ty);
}
__device__ __forceinline__ bool block_tile(const CompParams &p, int &tx, int &ty) {
  return block_tile(p, tx, ty, blockIdx.x);
}
__host__ __forceinline__ uint32_t comp_grid(const CompParams &p) {
  return p.tile_order != nullptr ? (uint32_t)(p.ntw * p.nth) : tile_map_blocks(p.ntw, p.nth);
}

// ---- reference-arithmetic Gaussian evaluations (rare path) ------------------------------
// kernels.h:195-224
static __device__ __noinline__ float gauss_ref_f64(float mx, float my, float c0f, float c1f, float c2f,
                                            float c3f, float px, float py) {
  const double c0 = c0f, c1 = c1f, c2 = c2f, c3 = c3f;
  const double det = c0 * c3 - c1 * c2;
  const double x = (double)(px - mx);
  const double y = (double)(py - my);
  const double tx = x * c3 - y * c2;
  const double ty = -x * c1 + y * c0;
  double radial = tx * x + ty * y;
  radial /= det;
  if (radial < 0.0) radial = 1000.0;
  return (float)exp(-0.5 * radial);
}
// kernels.h:172-193, fp32 with every product rounded (the oracle's order)
static __device__ __noinline__ float gauss_ref_f32(float mx, float my, float c0, float c1, float c2,
                                            float c3, float px, float py) {
#pragma clang fp contract(off)
  const float det = c0 * c3 - c1 * c2;
  const float x = px - mx;
  const float y = py - my;
  const float tx = x * c3 - y * c2;
  const float ty = -x * c1 + y * c0;
  float radial = tx * x + ty * y;
  radial = radial / det;
  if (radial < 0.0f) radial = 1000.0f;
  // the correctly rounded fp32 exponential (through fp64): this value decides a discontinuity of the image, and a
  // 1-ulp expf on one side of the comparison against a <0.51-ulp libm expf on the other flips ~0.3 pixels per
  // 800x800 frame (measured); two correctly rounded values agree
  return (float)exp((double)(-0.5f * radial));
}

// real SH basis, bands CB = 1..4 (shencoder.h:13-62)
template <int CB>
__device__ __forceinline__ void sh_basis(float x, float y, float z, float (&Y)[CB * CB]) {
  // every product rounded (as the oracle evaluates shencoder.h): the table is the same in every kernel, whatever
  // the compiler would have liked to contract in the code around it
#pragma clang fp contract(off)
  Y[0] = 0.28209479177387814f;
  if constexpr (CB >= 2) {
    Y[1] = -0.48860251190291987f * y;
    Y[2] = 0.48860251190291987f * z;
    Y[3] = -0.48860251190291987f * x;
  }
  if constexpr (CB >= 3) {
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    Y[4] = 1.0925484305920792f * xy;
    Y[5] = -1.0925484305920792f * yz;
    Y[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
    Y[7] = -1.0925484305920792f * xz;
    Y[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
    if constexpr (CB >= 4) {
      Y[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
      Y[10] = 2.8906114426405538f * xy * z;
      Y[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
      Y[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
      Y[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
      Y[14] = 1.4453057213202769f * z * (x2 - y2);
      Y[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
    }
  }
}

// The SH basis of the pixel whose normalised camera-space position is (qx, qy, 1): dir = normalize(R (qx, qy, 1))
// (vol_render_sh.h:48-65; R = the 9 packed floats the reference reads from c2w), Y zero-padded to NY entries.
// One function with contraction off: forward and backward kernels of every shape hold bit-identical tables, hence
// render a camera to the same bits whether it goes through a per-camera or a batched launch.
template <int CB, int NY>
__device__ __forceinline__ void sh_basis_of_pixel(const float (&R)[9], float qx, float qy, float (&Y)[NY]) {
#pragma clang fp contract(off)
  static_assert(NY >= CB * CB, "room for the basis");
  float dx = R[0] * qx + R[1] * qy + R[2];
  float dy = R[3] * qx + R[4] * qy + R[5];
  float dz = R[6] * qx + R[7] * qy + R[8];
  const float len = sqrtf(dx * dx + dy * dy + dz * dz);
  dx /= len; dy /= len; dz /= len;
#pragma unroll
  for (int k = 0; k < NY; ++k) Y[k] = 0.0f;
  sh_basis<CB>(dx, dy, dz, *reinterpret_cast<float (*)[CB * CB]>(&Y[0]));
}

__device__ __forceinline__ float sigmoid_fast(float s) {
  // 1/(1+exp(-s)) (shencoder.h:4) on v_exp_f32 / v_rcp_f32
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-kLog2e * s));
}

template <int MODE, int CB>
struct Traits {
  static constexpr int CC = CB * CB;
  static constexpr int NCOL = (MODE == MODE_SH) ? 3 * CC : (MODE == MODE_RGB ? 3 : (MODE == MODE_RGBD ? 6 : 1));
  static constexpr int NCH = (MODE == MODE_SCALAR) ? 1 : (MODE == MODE_RGBD ? 6 : 3);
  // gradient components per Gaussian: mean(2) cov(4, the two off-diagonals carry the same
  // value) alpha(1) colour/scalar/sh(NCOL)
  static constexpr int NCOMP = 7 + NCOL;
  static constexpr int P = NCOMP <= 8 ? 8 : (NCOMP <= 16 ? 16 : (NCOMP <= 32 ? 32 : 64));
  // LDS layout of the SH coefficients: each channel padded to a multiple of 4 floats so that
  // a channel is read as aligned float4s and multiplied as (k, k+1) pairs by v_pk_fma_f32
  static constexpr int CCP = (MODE == MODE_SH) ? ((CC + 3) & ~3) : CC;
  static constexpr int NCOLP = (MODE == MODE_SH) ? 3 * CCP : NCOL;
  static constexpr int NPAIR = CCP / 2;
};


// d L / d out[pix][c]: one [H,W,NCH] image, or (RGB + heads only, grad_out == NULL) the four images of the heads
template <int MODE, int NCH>
__device__ __forceinline__ float load_grad_out(const CompParams &p, size_t pix, int c) {
  if constexpr (MODE == MODE_RGBD) {
    if (p.grad_out == nullptr) {  // uniform over the launch
      if (c < 3) return p.go_rgb != nullptr ? p.go_rgb[3 * pix + c] : 0.0f;
      const float *q = (c == 3) ? p.go_d : (c == 4 ? p.go_o : p.go_z2);
      return q != nullptr ? q[pix] : 0.0f;
    }
  }
  return p.grad_out[NCH * pix + c];
}

// the forward's image at pix, channel c: [H,W,NCH], or (RGB + heads with separate images) rgb [H,W,3] + three [H,W]
template <int MODE, int NCH>
__device__ __forceinline__ float load_final(const CompParams &p, size_t pix, int c) {
  if constexpr (MODE == MODE_RGBD) {
    if (p.pl_rgb != nullptr) {  // uniform over the launch
      if (c < 3) return p.pl_rgb[3 * pix + c];
      return (c == 3 ? p.pl_d : (c == 4 ? p.pl_o : p.pl_z))[pix];
    }
  }
  return p.final_img[NCH * pix + c];
}
// torch.nan_to_num with its defaults: NaN -> 0, +-inf -> +-FLT_MAX
__device__ __forceinline__ float nan_to_num_f(float v) {
  if (!(v == v)) return 0.0f;
  return fminf(fmaxf(v, -3.402823466e+38f), 3.402823466e+38f);
}

// the NCOL per-Gaussian channel values of the non-SH modes
template <int MODE, int NCOL>
__device__ __forceinline__ void load_channels(const CompParams &p, int id, float *dst) {
  if constexpr (MODE == MODE_RGBD) {
    const float d = p.depth[id];
    dst[0] = p.col[3 * (size_t)id]; dst[1] = p.col[3 * (size_t)id + 1]; dst[2] = p.col[3 * (size_t)id + 2];
    dst[3] = d; dst[4] = 1.0f; dst[5] = d * d;
  } else {
#pragma unroll
    for (int k = 0; k < NCOL; ++k) dst[k] = p.col[(size_t)id * NCOL + k];
  }
}

// per-Gaussian values used by every (pixel, Gaussian) evaluation
struct GRec {
  float mx, my, a, c0, c1, c2, c3, p0, p1, p2;
};

// Builds the evaluation record of one Gaussian from its raw 2-D mean / covariance / opacity.
// RGB/scalar: p0..p2 = Cholesky factor of the quadratic form scaled so that
// G = exp2(-(u^2+v^2)), u = p0 x + p1 y, v = p2 y (computed in fp64 once per record).
// SH: p0 = -0.5 log2(e)/det, p1 = 1/det with the fp32 determinant of kernels.h:179.
// A degenerate / non-finite record gets a = 0 and never contributes.
template <int MODE>
__device__ __forceinline__ GRec prep_record(float mx, float my, float c0, float c1, float c2, float c3,
                                            float alpha) {
  GRec r;
  r.mx = mx; r.my = my; r.c0 = c0; r.c1 = c1; r.c2 = c2; r.c3 = c3;
  r.p0 = 0.f; r.p1 = 0.f; r.p2 = 0.f;
  float a = fminf(alpha, kAlphaClamp);
  bool ok = finite_f(mx) && finite_f(my) && finite_f(c0) && finite_f(c1) && finite_f(c2) && finite_f(c3) &&
            finite_f(a);
  if constexpr (MODE == MODE_SH) {
    float det;
    {
#pragma clang fp contract(off)
      det = c0 * c3 - c1 * c2;
    }
    ok = ok && (det > 0.0f) && finite_f(det);
    const float inv = 1.0f / (ok ? det : 1.0f);
    r.p0 = -0.5f * kLog2e * inv;
    r.p1 = inv;
  } else {
    const CholRec c = chol_prep(c0, c1, c2, c3);  // (common.hpp: the same bits wherever it is formed)
    ok = ok && c.ok;
    if (ok) { r.p0 = c.p0; r.p1 = c.p1; r.p2 = c.p2; }
  }
  r.a = ok ? a : 0.0f;
  return r;
}

// Gaussian value for one pixel.  x = px - mx, y = py - my.
template <int MODE>
__device__ __forceinline__ float gauss_eval(const GRec &r, float x, float y, float px, float py,
                                            bool alive) {
  float G;
  if constexpr (MODE == MODE_SH) {
    // One fixed operation sequence (explicit fused multiply-adds, every other product rounded), shared with the
    // packed two-pixel form gauss_sh_pair below: forward and backward kernels of any shape get the same bits.
#pragma clang fp contract(off)
    const float yc2 = y * r.c2, xc1 = x * r.c1;
    const float tx = ffma(x, r.c3, -yc2);
    const float ty = ffma(y, r.c0, -xc1);
    const float tyy = ty * y;
    const float q = ffma(tx, x, tyy);
    const float e = r.p0 * q;
    G = __builtin_amdgcn_exp2f(e);
    G = (q < 0.0f) ? 0.0f : G;  // kernels.h:186-188: radial < 0 -> exp(-500) == 0
  } else {
    // one fixed operation sequence here too (shared with gauss_chol_pair): a camera renders to the same bits through
    // the per-camera kernels and the packed batched ones
#pragma clang fp contract(off)
    const float p0x = r.p0 * x;
    const float u = ffma(r.p1, y, p0x);
    const float v = r.p2 * y;
    const float vv = v * v;
    G = __builtin_amdgcn_exp2f(-ffma(u, u, vv));
  }
  const float ag = r.a * G;
  // within rounding of the skip threshold: take the reference's arithmetic.  The wave-uniform test in front keeps the
  // common case (no lane anywhere near: all but a handful of pixels per frame) free of exec-mask juggling
  const bool near = alive && fabsf(ag - kMinAlpha) <= kMinAlpha * kGuardTol;
  if (__ballot(near) != 0ull) {
    if (near) {
      if constexpr (MODE == MODE_SH)
        G = gauss_ref_f32(r.mx, r.my, r.c0, r.c1, r.c2, r.c3, px, py);
      else
        G = gauss_ref_f64(r.mx, r.my, r.c0, r.c1, r.c2, r.c3, px, py);
    }
  }
  return G;
}

// MODE_SH Gaussian values of the two pixels (x, y2[0]) and (x, y2[1]) of one lane: the operation sequence of
// gauss_eval<MODE_SH> element for element (v_pk_mul / v_pk_fma round exactly like their scalar forms), without the
// threshold guard -- the caller applies it per pixel with gauss_ref_f32 as gauss_eval does.
// tx, ty: det * Sigma^-1 d, the offsets the value is formed from (the moment form of the backward sums against them).
__device__ __forceinline__ v2f gauss_sh_pair(float c0, float c1, float c2, float c3, float p0, float x, v2f y2, v2f &tx, v2f &ty) {
#pragma clang fp contract(off)
  const float xc1 = x * c1;
  const v2f yc2 = y2 * splat2(c2);
  tx = ffma2(splat2(x), splat2(c3), -yc2);
  ty = ffma2(y2, splat2(c0), -splat2(xc1));
  const v2f tyy = ty * y2;
  const v2f q2 = ffma2(tx, splat2(x), tyy);
  const v2f e = splat2(p0) * q2;
  v2f G = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
  G[0] = (q2[0] < 0.0f) ? 0.0f : G[0];
  G[1] = (q2[1] < 0.0f) ? 0.0f : G[1];
  return G;
}
__device__ __forceinline__ v2f gauss_sh_pair(float c0, float c1, float c2, float c3, float p0, float x, v2f y2) {
  v2f tx, ty;
  return gauss_sh_pair(c0, c1, c2, c3, p0, x, y2, tx, ty);
}

// The Cholesky-form Gaussian (RGB / scalar / RGB + heads) of the two pixels (x, y2[0]), (x, y2[1]) of one lane: the
// operation sequence of gauss_eval's non-SH branch element for element, without the threshold guard.  p0x = p0 * x.
// u, v: the whitened offsets u = p0 x + p1 y, v = p2 y the value is formed from (the moment form of the backward sums against them).
__device__ __forceinline__ v2f gauss_chol_pair(float p0x, float p1, float p2, v2f y2, v2f &u, v2f &v) {
#pragma clang fp contract(off)
  u = ffma2(splat2(p1), y2, splat2(p0x));
  v = splat2(p2) * y2;
  const v2f vv = v * v;
  const v2f e = -ffma2(u, u, vv);
  return v2f{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
}
__device__ __forceinline__ v2f gauss_chol_pair(float p0x, float p1, float p2, v2f y2) {
  v2f u, v;
  return gauss_chol_pair(p0x, p1, p2, y2, u, v);
}

// ---- tile-local polynomial form of the per-pixel SH basis (SH degree 3, launches that are given the coefficient bound) ----
// The reference evaluates the SH basis per PIXEL, for dir = normalize(R (qx, qy, 1)) (vol_render_sh.h:48-65).  Inside a
// 16x16 tile that direction moves by ~1e-2 rad, and Y[pixel][0..16) is a degree-2 polynomial in the tile-local offsets
// (u, v) in [-1, 1]^2 to within 2e-6 of a basis value at f = image size (2e-5 at f = 0.7 x size: tools/tile_basis_error.py,
// profiles/r02_notes.md): Y ~ U(u, v) V with U = (1, v, u, v^2, uv, u^2) and V[6][16] the least-squares fit through the exact
// basis at the 3 x 3 nodes (u, v) in {-1, 0, 1}^2, set up once per tile.  The staged coefficients become
// w[c][r] = sum_k V[r][k] sh[c][k] (18 values instead of 48, transformed once per (tile, splat)), the per-pixel contractions
// are 6 terms instead of 16, 3 x 6 instead of 3 x 16 gradient components cross the lanes and are expanded by V in front of
// the atomics.  It renders a view only where the bound 0.25 * S * kPolyFitErr * delta^3 (S: the scene's largest per-splat sum of
// |non-constant SH coefficients| of one channel, measured on the device per step; delta: the tile's half diagonal in camera
// space) stays below 1e-5 -- decided on the device by every workgroup (poly_route), which runs the exact form otherwise.
constexpr int kPolyNB = 6;
// A (splat, channel) row of transformed coefficients in LDS: eight floats (w0, w1 | w2, w4 | w5, 0 | w3, 0) for the monomials
// r = 0..5 = (1, v, u, v^2, uv, u^2), so that the kernels form  s = (w0 + u (w2 + u w5)) + v ((w1 + u w4) + v w3)  from whole
// register PAIRS:  (A, B) = (w0, w1) + u ((w2, w4) + u (w5, 0)),  s = A + v (B + v w3)  -- two packed FMAs per row and lane
// instead of three scalar ones plus the moves that put their results into aligned pairs.
// Round 5 -- the Taylor tier: inside a tile the logit of a (splat, channel) moves by d = s(u, v) - s(0, 0) = L + Q, its linear part
// L = w1 v + w2 u, |L| <= l := |w1| + |w2|, and its quadratic part Q, |Q| <= q := |w3| + |w4| + |w5|; dmax := l + q is a few
// hundredths of the scaled unit on every BASELINE workload (median 0.006, largest 0.03 at 800^2 and at cfg4's widest cameras),
// and q is another factor delta ~ 0.01 below l.  The sigmoid  f(z) = 1 / (1 + 2^z)  around the tile centre z0 = w0 is
//     f(z0 + d) = f0 + f1 d + f2 d^2 + R3,   |R3| <= max |f'''| / 6 * |d|^3 <= 0.00694 dmax^3,
// and d^2 = L^2 + (2 L Q + Q^2), the bracket at most q (2 l + q): so the COLOUR ITSELF is the quadratic
//     c(u, v) = f0 + f1 (L + Q) + f2 L^2
//             = f0 + (f1 w1) v + (f1 w2) u + (f1 w3 + f2 w1^2) v^2 + (f1 w4 + 2 f2 w1 w2) u v + (f1 w5 + f2 w2^2) u^2
// to within  0.00694 dmax^3 + max |f2| q (2 l + q),  max |f2| = 0.02312.  Rows whose bound stays below kTaylorErr = 8.7e-7 (the
// cubic term alone reaches it at dmax = 0.05) are the Taylor tier: poly_transform tests them, taylor_convert overwrites the
// six logit coefficients of such a splat IN PLACE with the six colour coefficients -- one exponential per (tile, splat, channel)
// --, and the entry loops of the polynomial kernels evaluate the row ONCE, by the same five packed FMAs per pixel pair and channel
// whatever it holds: a Taylor row delivers the colour (no exponential, no reciprocal per pixel; until round 5: 12 exponentials +
// 4 reciprocals per lane and entry, 80 of the forward's 154 issue slots per (wavefront, entry), 64 of the backward's 328), any
// other row the logit, which takes the exponentials entry by entry as before (a wave-uniform bit per staged record, as the exact
// tier: outliers with large higher bands, very wide cameras).  Forward and backward take the same bit, hence the same values.
// Row layout (8 floats per (splat, channel)): (w0, w1 | w2, w4 | w5, 0 | w3, -) for the monomials (1, v | u, uv | u^2, - | v^2).
constexpr int kPolyStride = 8;
constexpr float kTaylorErr = 8.7e-7f;
__host__ __device__ __forceinline__ bool taylor_row_ok(float l, float q) {  // (NaN: false)
  const float d = l + q;
  return 0.00694f * d * d * d + 0.02312f * q * (2.0f * l + q) <= kTaylorErr;
}
constexpr int kPolyNodes = 9;
// colour error of the degree-2 form <= 0.25 (sigmoid slope) x S x kPolyFitErr delta^3, delta = half diagonal of a tile in
// camera space; used where that stays below kPolyFitTol = 1e-5 - kTaylorErr: with the Taylor tier's 8.7e-7 on top a routed colour
// is within 1e-5 of the exact kernels', a tenth of the 1e-4 image tolerance.  kPolyFitErr = 1.0: the SHIPPED fit's
// largest basis error is 0.93 delta^3 (tests/test_poly_fit_bound.py sweeps kPolyFit over rotations, tile positions and focal
// lengths).  Rounds 2-4 routed on 0.7 -- a calibration of a different interpolation -- which made the promise 1.4e-5.
// S <= 0, NaN or infinite: never.  One function for the host's report and the kernels' routing.
constexpr float kPolyFitErr = 1.0f;
constexpr float kPolyFitTol = 1e-5f - kTaylorErr;
__host__ __device__ __forceinline__ bool poly_ok(float S, float ps_max) {
  if (!(S > 0.0f) || !(S <= 3.0e38f)) return false;
  const float delta = 7.5f * 1.41421356f * ps_max;
  return 0.25f * S * kPolyFitErr * delta * delta * delta <= kPolyFitTol;
}
// per-splat form of the same rule: may a splat whose rows sum to at most S be rendered through the polynomial basis in a view of
// this pixel size?  S = 0 (no higher bands: the fit of the constant term is exact) passes; NaN does not.
__host__ __device__ __forceinline__ bool poly_row_ok(float S, float ps_max) {
  const float delta = 7.5f * 1.41421356f * ps_max;
  return 0.25f * S * kPolyFitErr * delta * delta * delta <= kPolyFitTol;
}
// does the polynomial form render this view?  (uniform over the workgroup: one scalar load)
__device__ __forceinline__ bool poly_route(const float *sh_bound, float psx, float psy) {
  return sh_bound != nullptr && poly_ok(sh_bound[0], fmaxf(fabsf(psx), fabsf(psy)));
}
constexpr float kPolyFit[kPolyNB][kPolyNodes] = {
    {-0.111111111f, 0.222222222f, -0.111111111f, 0.222222222f, 0.555555556f, 0.222222222f, -0.111111111f, 0.222222222f, -0.111111111f},
    {-0.166666667f, -0.166666667f, -0.166666667f, 0.0f, 0.0f, 0.0f, 0.166666667f, 0.166666667f, 0.166666667f},
    {-0.166666667f, 0.0f, 0.166666667f, -0.166666667f, 0.0f, 0.166666667f, -0.166666667f, 0.0f, 0.166666667f},
    {0.166666667f, 0.166666667f, 0.166666667f, -0.333333333f, -0.333333333f, -0.333333333f, 0.166666667f, 0.166666667f, 0.166666667f},
    {0.25f, 0.0f, -0.25f, 0.0f, 0.0f, 0.0f, -0.25f, 0.0f, 0.25f},
    {0.166666667f, -0.333333333f, 0.166666667f, 0.166666667f, -0.333333333f, 0.166666667f, 0.166666667f, -0.333333333f, 0.166666667f}};

// tile-local offset of a pixel column / row index 0..15
__device__ __forceinline__ float poly_offset(int l) { return ((float)l - 7.5f) * (1.0f / 7.5f); }
// V[6][16] of tile (tx, ty) into LDS (Vs), through the exact basis at the nine nodes (Yn: 9 x 16 floats of LDS scratch).
// All NT threads of the workgroup call it; ends with a barrier.  The fit's coefficients are compile-time literals of fully
// unrolled loops (a table in memory, read through a dependent chain of loads, cost tens of microseconds per tile once the
// memory system was busy with the launch's atomics).
template <int NT>
__device__ __forceinline__ void poly_tile_setup(const CompParams &p, int tx, int ty, float *Yn, float *Vs) {
  const int t = (int)threadIdx.x;
  if (t < kPolyNodes) {
    float R[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = p.rot[i];
    const float u = (float)(t % 3 - 1), v = (float)(t / 3 - 1);
    const float qx = p.topleft[0] + ((float)(tx * kTile) + 7.5f + 7.5f * u) * p.psx;
    const float qy = p.topleft[1] + ((float)(ty * kTile) + 7.5f + 7.5f * v) * p.psy;
    float Yf[16];
    sh_basis_of_pixel<4>(R, qx, qy, Yf);
#pragma unroll
    for (int k = 0; k < 16; ++k) Yn[t * 16 + k] = Yf[k];
  }
  __syncthreads();
  if (t < 16) {  // lane k: column k of V, all six rows
    float y[kPolyNodes];
#pragma unroll
    for (int n = 0; n < kPolyNodes; ++n) y[n] = Yn[n * 16 + t];
#pragma unroll
    for (int r = 0; r < kPolyNB; ++r) {
      float acc = 0.0f;
#pragma unroll
      for (int n = 0; n < kPolyNodes; ++n)
        if (kPolyFit[r][n] != 0.0f) acc = fmaf(kPolyFit[r][n], y[n], acc);
      Vs[r * 16 + t] = acc;
    }
  }
  __syncthreads();
}
// the coefficients sh[id][3][16] of the nb staged splats, straight from HBM, -> w[nb][3][6] in LDS:
// w[c][r] = -log2(e) sum_k V[r][k] sh[c][k]  (the scale of stage_batch<SCALE>: the colour evaluation's exp2 needs no multiply).
// One HALF row per lane and pass -- (splat, channel) row e = it / 2, monomials 3 (it & 1) .. +2 -- so that a batch's 6 nb
// items fill whole passes of 64 or 128 lanes (whole rows left the second pass of a 32-record batch half empty), the dot
// products as packed FMAs over (even, odd) coefficient pairs: 8 + 1 vector instructions per monomial instead of 16.
// (Loading a lane's two items together hides one trip to memory per batch and costs 32 more registers -- a wavefront per
// SIMD; not taken.)
template <int NT, int KB>
__device__ __forceinline__ void poly_transform(const float *__restrict__ sh, const int *ids, const float *Vs, float *w, int nb,
                                               float *tay_ok /* [KB * 3]: 1 where the row may take the Taylor tier */) {
  static_assert(NT % 2 == 0, "the two halves of a row sit in neighbouring lanes");
#pragma unroll 1
  for (int it0 = 0; it0 < nb * 6; it0 += NT) {  // (block-uniform trips: the lane exchange below is a wave collective)
    const int it = it0 + (int)threadIdx.x;
    const bool live = it < nb * 6;
    const int e = live ? (it >> 1) : 0, half = it & 1;
    const int g = e / 3, c = e - 3 * g;
    const float4 *q4 = reinterpret_cast<const float4 *>(sh + (size_t)ids[g] * 48 + c * 16);
    const float4 q0 = q4[0], q1 = q4[1], q2 = q4[2], q3 = q4[3];
    const v2f qa = {q0.x, q0.y}, qb = {q0.z, q0.w}, qc = {q1.x, q1.y}, qd = {q1.z, q1.w}, qe = {q2.x, q2.y}, qf = {q2.z, q2.w},
              qg = {q3.x, q3.y}, qh = {q3.z, q3.w};
    float *row = w + e * kPolyStride;
    // monomials (1, v, u | v^2, uv, u^2) -> slots (0, 1, 2 | 6, 3, 4); 5 stays zero (it is multiplied by u), 7 is padding
    const int s0 = half ? 6 : 0, s1 = half ? 3 : 1, s2 = half ? 4 : 2;
#pragma unroll 1  // (rolled: unrolled, the three monomials' 48 values of V stay live -- a wavefront per SIMD)
    for (int k = 0; k < 3; ++k) {
      const float4 *v4 = reinterpret_cast<const float4 *>(Vs + (3 * half + k) * 16);  // two addresses per wavefront: broadcast
      const float4 a = v4[0], b = v4[1], cc = v4[2], d = v4[3];
      v2f acc = qa * v2f{a.x, a.y};
      acc = ffma2(qb, v2f{a.z, a.w}, acc);
      acc = ffma2(qc, v2f{b.x, b.y}, acc);
      acc = ffma2(qd, v2f{b.z, b.w}, acc);
      acc = ffma2(qe, v2f{cc.x, cc.y}, acc);
      acc = ffma2(qf, v2f{cc.z, cc.w}, acc);
      acc = ffma2(qg, v2f{d.x, d.y}, acc);
      acc = ffma2(qh, v2f{d.z, d.w}, acc);
      if (live) row[k == 0 ? s0 : (k == 1 ? s1 : s2)] = -kLog2e * (acc[0] + acc[1]);
    }
    if (live && !half) row[5] = 0.0f;
    // |s| <= sum_r |w_r| on the tile (|u|, |v| <= 1).  The kernels take ONE reciprocal per pixel for the product of the three
    // channels' 1 + exp2(s): rows that could reach |s| > 40 are scaled back to 40 -- their sigmoid is 0 or 1 to 1e-12 either way
    // (and d sigmoid / d s ~ 1e-12: no gradient is lost that the exact kernels would deliver)
    const float mine = live ? fabsf(row[s0]) + fabsf(row[s1]) + fabsf(row[s2]) : 0.0f;  // (its own stores: no barrier)
    const float l1 = xor_add<1>(mine);  // + the row's other half, one lane over
    if (live && l1 > 40.0f) {
      const float sc = 40.0f / l1;
      row[s0] *= sc; row[s1] *= sc; row[s2] *= sc;
    }
    // the Taylor tier's test, by the lane that holds the row's first half (w0, w1, w2): l from its own sum, q from its neighbour's
    // (of the unscaled row; a row scaled back to 40 is far beyond the tier anyway)
    if (live && !half) {
      const float l = mine - fabsf(row[0]), q = l1 - mine;
      tay_ok[e] = (l1 <= 40.0f && taylor_row_ok(l, q)) ? 1.0f : 0.0f;  // (NaN coefficients: false)
    }
  }
}
// ... and the conversion of the Taylor tier's rows (bit g of mask = entry g of the staged batch), behind the barrier that follows
// poly_transform and before the one the entry loop waits for: logit coefficients -> colour coefficients, in place.
// f0 = 1 / (1 + 2^w0) is the very expression the exponential path evaluates -- the two agree to the bit where d = 0 --,
// f1 = df/dz = -ln2 f (1 - f), f2 = half the second derivative = (ln2^2 / 2) f (1 - f) (1 - 2 f).
template <int NT>
__device__ __forceinline__ void taylor_convert(float *w, uint32_t mask, int nb) {
#pragma unroll 1
  for (int e = (int)threadIdx.x; e < nb * 3; e += NT) {
    if (!((mask >> (e / 3)) & 1u)) continue;
    float *row = w + e * kPolyStride;
    const float w1 = row[1], w2 = row[2], w4 = row[3], w5 = row[4], w3 = row[6];
    const float f0 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(row[0]));
    const float ff = f0 * (1.0f - f0);
    const float f1 = -0.6931471805599453f * ff, f2 = 0.2402265069591007f * ff * (1.0f - 2.0f * f0);
    row[0] = f0;
    row[1] = f1 * w1;
    row[2] = f1 * w2;
    row[3] = fmaf(f1, w4, 2.0f * f2 * w1 * w2);
    row[4] = fmaf(f1, w5, f2 * w2 * w2);
    row[6] = fmaf(f1, w3, f2 * w1 * w1);
  }
}
// the staged batch's mask of splats whose three channels may take the Taylor tier (bit g = entry g; KB <= 32): every wavefront forms
// it for itself from the flags poly_transform left in LDS (behind the barrier that follows it)
__device__ __forceinline__ uint32_t taylor_mask(const float *tay_ok, int nb) {
  const int l = lane_id();
  const bool ok = l < nb && tay_ok[3 * l] != 0.0f && tay_ok[3 * l + 1] != 0.0f && tay_ok[3 * l + 2] != 0.0f;
  return (uint32_t)__ballot((int)ok);
}

// tile_size: 16 (every kernel variant) or 8 / 32 (the unpacked vector kernels at one fixed shape; no segments, no batch)
static inline int check_common(uint32_t tile_size, const void *a, const void *b, const void *c) {
  if (tile_size < 1u || tile_size > 32u) return GSGEN_EUNSUPPORTED;  // (the reference launches tile_size^2 <= 1024 threads per tile)
  if (!a || !b || !c) return GSGEN_EINVAL;
  return 0;
}

}  // namespace gs
