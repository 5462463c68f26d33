// geometry.hip -- per-Gaussian stages: frustum cull, EWA projection (fwd/bwd), AABB -> tile
// rectangle.  One thread per Gaussian, ~100 B of HBM traffic each: purely bandwidth-bound.
//
// THIS FILE IS COMPILED WITH -ffp-contract=off.  Every fp32 expression below is written in
// the order the oracle (oracle/gs_oracle.c) executes it and uses only correctly-rounded IEEE
// operations (+ - * / sqrt), so mean2d / cov2d / depth and therefore the integer tile
// rectangles, the pair count D and the per-tile lists are BIT-IDENTICAL to the oracle (tile
// membership is part of the image: SURVEY.md 8a trap 1).  Against the reference's torch ops
// the projection agrees to a few ulp (its einsum/bmm sum the 3-term dot products in BLAS
// order; bit-exact for axis-aligned poses), the AABB -> tile arithmetic given the same
// mean2d/cov2d is exact (tests/test_oracle_golden.py, tests/test_gpu_golden.py).
//
// Replaces (paths relative to /root/reference):
//   cull        gs/src/include/culling.h:10-33, kernels.h:156-170
//   projection  gs/renderer.py:366-421, utils/transforms.py:34-46 (kornia 0.6.0 quaternion)
//   AABB count  gs/culling.py:8-37, utils/camera.py:301-314
#include <string.h>

#include "common.hpp"
#include "../../include/gsgen_hip.h"

namespace gs {

constexpr int kThreads = 256;

// ---- frustum cull -------------------------------------------------------------------------
__device__ __forceinline__ bool sphere_in_frustum(float mx, float my, float mz, float r,
                                                  const float *__restrict__ normal,
                                                  const float *__restrict__ pts) {
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const float d = (mx - pts[3 * k]) * normal[3 * k] + (my - pts[3 * k + 1]) * normal[3 * k + 1] +
                    (mz - pts[3 * k + 2]) * normal[3 * k + 2];
    ok = ok && (d > -r);
  }
  return ok;
}

__global__ void __launch_bounds__(kThreads)
k_cull_bsphere(uint32_t N, const float *__restrict__ mean, const float *__restrict__ svec,
               const float *__restrict__ normal, const float *__restrict__ pts,
               uint8_t *__restrict__ mask, float thresh) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float r = fmaxf(fmaxf(svec[3 * i], svec[3 * i + 1]), svec[3 * i + 2]) * thresh;
  mask[i] = sphere_in_frustum(mean[3 * i], mean[3 * i + 1], mean[3 * i + 2], r, normal, pts) ? 1 : 0;
}

// ---- projection ----------------------------------------------------------------------------
struct Proj {
  float m2x, m2y, c00, c01, c10, c11, depth;
  float A[9];  // JW
};

__device__ __forceinline__ void quat_to_rot(const float *__restrict__ q, float *R) {
  float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  n = fmaxf(n, 1e-12f);  // F.normalize(p=2, eps=1e-12)
  const float w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
  const float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
  const float twx = tx * w, twy = ty * w, twz = tz * w;
  const float txx = tx * x, txy = ty * x, txz = tz * x;
  const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0f - (tyy + tzz); R[1] = txy - twz;          R[2] = txz + twy;
  R[3] = txy + twz;          R[4] = 1.0f - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;          R[7] = tyz + twx;          R[8] = 1.0f - (txx + tyy);
}

__device__ __forceinline__ Proj project_one(const float *__restrict__ p, const float *__restrict__ q,
                                            const float *__restrict__ s, const float *Rc,
                                            const float *t) {
  Proj o;
  const float d0 = p[0] - t[0], d1 = p[1] - t[1], d2 = p[2] - t[2];
  float u[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) u[i] = Rc[i] * d0 + Rc[3 + i] * d1 + Rc[6 + i] * d2;
  float Rq[9];
  quat_to_rot(q, Rq);
  float M[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) M[i * 3 + j] = s[j] * Rq[i * 3 + j];
  float S[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k)
      S[i * 3 + k] = M[i * 3] * M[k * 3] + M[i * 3 + 1] * M[k * 3 + 1] + M[i * 3 + 2] * M[k * 3 + 2];
  const float x = u[0], y = u[1], z = u[2];
  const float l = sqrtf(x * x + y * y + z * z);
  const float J[9] = {1.0f / z, 0.0f, -x / z / z, 0.0f, 1.0f / z, -y / z / z, x / l, y / l, z / l};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k)
      o.A[i * 3 + k] = J[i * 3] * Rc[k * 3] + J[i * 3 + 1] * Rc[k * 3 + 1] + J[i * 3 + 2] * Rc[k * 3 + 2];
  float T1[6];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int k = 0; k < 3; ++k)
      T1[a * 3 + k] = o.A[a * 3] * S[k] + o.A[a * 3 + 1] * S[3 + k] + o.A[a * 3 + 2] * S[6 + k];
  o.c00 = T1[0] * o.A[0] + T1[1] * o.A[1] + T1[2] * o.A[2];
  o.c01 = T1[0] * o.A[3] + T1[1] * o.A[4] + T1[2] * o.A[5];
  o.c10 = T1[3] * o.A[0] + T1[4] * o.A[1] + T1[5] * o.A[2];
  o.c11 = T1[3] * o.A[3] + T1[4] * o.A[4] + T1[5] * o.A[5];
  o.depth = z;
  o.m2x = x / z;
  o.m2y = y / z;
  return o;
}

__device__ __forceinline__ void load_pose(const float *__restrict__ c2w, float *Rc, float *t) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) Rc[i * 3 + j] = c2w[i * 4 + j];
    t[i] = c2w[i * 4 + 3];
  }
}

__global__ void __launch_bounds__(kThreads)
k_project(uint32_t N, const float *__restrict__ mean, const float *__restrict__ qvec,
          const float *__restrict__ svec, const float *__restrict__ c2w, float *__restrict__ mean2d,
          float *__restrict__ cov2d, float *__restrict__ JW, float *__restrict__ depth) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float Rc[9], t[3];
  load_pose(c2w, Rc, t);
  const Proj o = project_one(mean + 3 * (size_t)i, qvec + 4 * (size_t)i, svec + 3 * (size_t)i, Rc, t);
  *reinterpret_cast<float2 *>(mean2d + 2 * (size_t)i) = make_float2(o.m2x, o.m2y);
  *reinterpret_cast<float4 *>(cov2d + 4 * (size_t)i) = make_float4(o.c00, o.c01, o.c10, o.c11);
  depth[i] = o.depth;
  if (JW != nullptr) {
#pragma unroll
    for (int k = 0; k < 9; ++k) JW[9 * (size_t)i + k] = o.A[k];
  }
}

// torch.optim.Adam (no weight decay, no amsgrad) over ONE flat fp32 parameter vector holding all
// fields back to back (gs/gaussian_splatting.py:398-419 builds one param group per field with its
// own scheduled lr): 16 B read + 12 B written per parameter in a single pass instead of the
// ~10 foreach kernels per group.  Update order as torch's single-tensor path:
//   m += (g - m) * (1 - b1);  v = v * b2 + (1 - b2) * g * g
//   p -= step_size * m / (sqrt(v) / sqrt(1 - b2^t) + eps),   step_size = lr / (1 - b1^t)
constexpr int kAdamMaxGroups = 8;
struct AdamArgs {
  uint64_t end[kAdamMaxGroups];
  float step_size[kAdamMaxGroups];
  uint32_t n_groups;
  float bc2_sqrt, beta1, beta2, eps;
  const float *dev;  // or NULL: step_size[0 .. 7] | bc2_sqrt from device memory (gsgen_adam_step_device_scalars)
};

__global__ void __launch_bounds__(kThreads)
k_adam_step(uint64_t n, float *__restrict__ param, const float *__restrict__ grad,
            float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq, AdamArgs a) {
  const uint64_t i0 = 4 * ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x);
  if (i0 >= n) return;
  const float w1 = 1.0f - a.beta1, w2 = 1.0f - a.beta2;
  if (a.dev != nullptr) {  // (uniform: scalar loads)
#pragma unroll
    for (int k = 0; k < kAdamMaxGroups; ++k) a.step_size[k] = a.dev[k];
    a.bc2_sqrt = a.dev[kAdamMaxGroups];
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const uint64_t i = i0 + e;
    if (i >= n) break;
    float ss = a.step_size[0];
#pragma unroll
    for (int k = 1; k < kAdamMaxGroups; ++k)
      if (k < (int)a.n_groups && i >= a.end[k - 1]) ss = a.step_size[k];
    const float g = grad[i];
    float m = exp_avg[i], v = exp_avg_sq[i];
    m = m + (g - m) * w1;
    v = v * a.beta2 + (w2 * g) * g;
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    param[i] = param[i] - ss * (m / denom);
    exp_avg[i] = m;
    exp_avg_sq[i] = v;
  }
}

// Densification statistics of one rendered camera, rows aligned with the frustum mask
// (gs/gaussian_splatting.py:1240-1245 and :464-469): the running maximum of the screen-space
// "radius" m + sqrt(max(m^2 - det, 0)) (no outer sqrt on this path), the running sum of
// |d L / d mean2d| and the visit count.  One pass, 25 B read + 3 atomics per visible Gaussian.
__global__ void __launch_bounds__(kThreads)
k_densify_update(uint32_t N, const float *__restrict__ cov2d, const float *__restrict__ g_mean2d,
                 const uint8_t *__restrict__ mask, float *__restrict__ max_radii2d,
                 float *__restrict__ grad_accum, float *__restrict__ cnt) {
  const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  if (mask != nullptr && mask[n] == 0) return;
  if (max_radii2d != nullptr) {
    const float4 c = *reinterpret_cast<const float4 *>(cov2d + 4 * (size_t)n);
    const float m = (c.x + c.w) / 2.0f;
    const float det = c.x * c.w - c.y * c.z;
    const float r = m + sqrtf(fmaxf(m * m - det, 0.0f));
    // atomics: cameras of a batch run on concurrent streams and share the statistics arrays.
    // Non-negative floats order like their bit patterns; a NaN radius sticks, as torch.max does.
    if (!(r < 0.0f)) atomicMax(reinterpret_cast<unsigned int *>(max_radii2d) + n, __float_as_uint(r));
  }
  if (grad_accum != nullptr) {
    const float2 g = *reinterpret_cast<const float2 *>(g_mean2d + 2 * (size_t)n);
    atomicAdd(grad_accum + n, sqrtf(g.x * g.x + g.y * g.y));
    if (cnt != nullptr) atomicAdd(cnt + n, 1.0f);
  }
}

// The same over the cameras of a batch in one launch: running maximum, gradient-norm sum and visit count of
// a Gaussian are folded over the views in registers, then one atomic each (other batches may be updating the
// same statistics from other streams).  Per-view pointers in the kernel arguments.
constexpr int kStatViews = 16;
struct DensifyViews {
  const float *cov2d[kStatViews], *g_mean2d[kStatViews];
  const uint8_t *mask[kStatViews];
};
__global__ void __launch_bounds__(kThreads)
k_densify_update_views(uint32_t N, DensifyViews dv, int n_views, float *__restrict__ max_radii2d,
                       float *__restrict__ grad_accum, float *__restrict__ cnt) {
  const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float rmax = 0.0f, gsum = 0.0f, visits = 0.0f;
  bool seen = false, rnan = false;
  for (int v = 0; v < n_views; ++v) {
    const uint8_t *m = dv.mask[v];
    if (m != nullptr && m[n] == 0) continue;
    seen = true;
    if (max_radii2d != nullptr) {
      const float4 c = *reinterpret_cast<const float4 *>(dv.cov2d[v] + 4 * (size_t)n);
      const float mm = (c.x + c.w) / 2.0f;
      const float det = c.x * c.w - c.y * c.z;
      const float r = mm + sqrtf(fmaxf(mm * mm - det, 0.0f));
      if (r != r) rnan = true;       // a NaN radius sticks, as torch.max does
      else rmax = fmaxf(rmax, r);    // (negative radii never raise the maximum, as in the per-view kernel)
    }
    if (grad_accum != nullptr) {
      const float2 g = *reinterpret_cast<const float2 *>(dv.g_mean2d[v] + 2 * (size_t)n);
      gsum += sqrtf(g.x * g.x + g.y * g.y);
      visits += 1.0f;
    }
  }
  if (!seen) return;
  if (max_radii2d != nullptr)
    atomicMax(reinterpret_cast<unsigned int *>(max_radii2d) + n,
              rnan ? 0x7fc00000u : __float_as_uint(rmax));
  if (grad_accum != nullptr) {
    atomicAdd(grad_accum + n, gsum);
    if (cnt != nullptr) atomicAdd(cnt + n, visits);
  }
}

// Backward of the projection as autograd differentiates gs/renderer.py:391-421: J is a
// constant (@torch.no_grad), the depth in the perspective divide is detached iff
// detach_depth.  ACC = false: gradients are overwritten (masked-out rows get zeros); ACC = true:
// added atomically into caller-zeroed arrays shared by the cameras of a batch, which may run on
// concurrent streams (masked-out rows are left alone).
struct ProjGrad { float gm[3], gq[4], gs[3]; };
// The Cholesky factor the compositing kernels evaluate an RGB / scalar / RGB + heads Gaussian through (prep_record in
// composite_common.hpp: the same fp64 expressions, rounded to fp32 like the staged record), as k_i = kInvSc2 * p_i -- so that
// Sigma^-1 d = (k0 u, k1 u + k2 v) for the kernels' u = p0 x + p1 y, v = p2 y.  A degenerate record never contributed: zeros.
struct CholK { double k0, k1, k2; };
__device__ __forceinline__ CholK chol_k_of(const float *__restrict__ cov2d_n) {
  const CholRec c = chol_prep(cov2d_n[0], cov2d_n[1], cov2d_n[2], cov2d_n[3]);  // (common.hpp: the staged record's bits)
  CholK k{0.0, 0.0, 0.0};
  if (c.ok) { k.k0 = (double)kInvSc2 * (double)c.p0; k.k1 = (double)kInvSc2 * (double)c.p1; k.k2 = (double)kInvSc2 * (double)c.p2; }
  return k;
}
// gradients of one Gaussian through one view's projection (rows of the reference's autograd graph,
// gs/renderer.py:366-421).  Round 6: evaluated in fp64 from the fp32 inputs -- the oracle's arithmetic.  The chain
// d cov2d -> d Sigma -> d M -> d R -> d q cancels STRUCTURALLY for near-isotropic scales (d q is proportional to differences
// of s_j^2): in fp32 the nine-term sums of d q lost up to 1.2e-3 of the tensor's largest entry on image-sized splats and
// the result moved with the order of the compositing atomics (profiles/r05_notes.md section 11).  One thread per
// (Gaussian, view), ~400 flops: the launch stays bound by its 132 B of traffic per Gaussian.
// gcov: d L / d cov2d (4 entries, row-major), gm0 / gm1: d L / d mean2d.
__device__ __forceinline__ ProjGrad project_bwd_one(uint32_t n, const float *__restrict__ mean,
                                                    const float *__restrict__ qvec, const float *__restrict__ svec,
                                                    const float *__restrict__ c2w, int detach_depth,
                                                    double gm0, double gm1, const double (&gc)[4], float g_depth_n) {
  // (fp64 throughout, fused multiply-adds allowed: the launch is bound by its fp64 issue -- 42 us per 8 x 100 k (view, Gaussian) chains
  // on MI355X with every product and sum a separate instruction and nine divisions; the exact zeros of J are left out, the two
  // normalisations multiply by one reciprocal each)
#pragma clang fp contract(fast)
  ProjGrad o;
  double Rc[9], t[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) Rc[i * 3 + j] = (double)c2w[i * 4 + j];
    t[i] = (double)c2w[i * 4 + 3];
  }
  const float *p = mean + 3 * (size_t)n, *q = qvec + 4 * (size_t)n, *sf = svec + 3 * (size_t)n;
  const double s[3] = {(double)sf[0], (double)sf[1], (double)sf[2]};
  const double d0 = (double)p[0] - t[0], d1 = (double)p[1] - t[1], d2 = (double)p[2] - t[2];
  double u[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) u[i] = Rc[i] * d0 + Rc[3 + i] * d1 + Rc[6 + i] * d2;
  double nq = sqrt((double)q[0] * q[0] + (double)q[1] * q[1] + (double)q[2] * q[2] + (double)q[3] * q[3]);
  nq = nq < 1e-12 ? 1e-12 : nq;
  const double inq = 1.0 / nq;
  const double w = q[0] * inq, x = q[1] * inq, y = q[2] * inq, z = q[3] * inq;
  const double Rq[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
  double M[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) M[i * 3 + j] = s[j] * Rq[i * 3 + j];
  const double ux = u[0], uy = u[1], uz = u[2];
  const double iz = 1.0 / uz;
  // J = [[iz, 0, -ux iz^2], [0, iz, -uy iz^2]]: A = J Rc^T (the first two rows of JW)
  const double j02 = -ux * iz * iz, j12 = -uy * iz * iz;
  double A[6];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    A[k] = iz * Rc[k * 3] + j02 * Rc[k * 3 + 2];
    A[3 + k] = iz * Rc[k * 3 + 1] + j12 * Rc[k * 3 + 2];
  }
  // dSigma = A^T G A through G A (2 x 3); only the symmetrised form enters dM
  double GA[6];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    GA[k] = gc[0] * A[k] + gc[1] * A[3 + k];
    GA[3 + k] = gc[2] * A[k] + gc[3] * A[3 + k];
  }
  double dS[9];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int k = 0; k < 3; ++k) dS[j * 3 + k] = A[j] * GA[k] + A[3 + j] * GA[3 + k];
  double dM[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) acc += (dS[i * 3 + k] + dS[k * 3 + i]) * M[k * 3 + j];
      dM[i * 3 + j] = acc;
    }
  double dR[9];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      acc += dM[i * 3 + j] * Rq[i * 3 + j];
      dR[i * 3 + j] = dM[i * 3 + j] * s[j];
    }
    o.gs[j] = (float)acc;
  }
  double dq[4];
  dq[0] = 2 * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
  dq[1] = 2 * (y * dR[1] + z * dR[2] + y * dR[3] - 2 * x * dR[4] - w * dR[5] + z * dR[6] + w * dR[7] - 2 * x * dR[8]);
  dq[2] = 2 * (-2 * y * dR[0] + x * dR[1] + w * dR[2] + x * dR[3] + z * dR[5] - w * dR[6] + z * dR[7] - 2 * y * dR[8]);
  dq[3] = 2 * (-2 * z * dR[0] - w * dR[1] + x * dR[2] + w * dR[3] - 2 * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
  const double qh[4] = {w, x, y, z};
  const double dot = qh[0] * dq[0] + qh[1] * dq[1] + qh[2] * dq[2] + qh[3] * dq[3];
#pragma unroll
  for (int k = 0; k < 4; ++k) o.gq[k] = (float)((dq[k] - qh[k] * dot) * inq);
  double du[3] = {gm0 * iz, gm1 * iz, (double)g_depth_n};
  if (!detach_depth) du[2] += -(ux * gm0 + uy * gm1) * iz * iz;
#pragma unroll
  for (int j = 0; j < 3; ++j) o.gm[j] = (float)(Rc[j * 3] * du[0] + Rc[j * 3 + 1] * du[1] + Rc[j * 3 + 2] * du[2]);
  return o;
}
__device__ __forceinline__ ProjGrad project_bwd_one(uint32_t n, const float *__restrict__ mean,
                                                    const float *__restrict__ qvec, const float *__restrict__ svec,
                                                    const float *__restrict__ c2w, int detach_depth,
                                                    const float *__restrict__ g_mean2d,
                                                    const float *__restrict__ g_cov2d, float g_depth_n) {
  const float4 g = *reinterpret_cast<const float4 *>(g_cov2d + 4 * (size_t)n);
  const double gc[4] = {(double)g.x, (double)g.y, (double)g.z, (double)g.w};
  return project_bwd_one(n, mean, qvec, svec, c2w, detach_depth, (double)g_mean2d[2 * (size_t)n],
                         (double)g_mean2d[2 * (size_t)n + 1], gc, g_depth_n);
}

template <bool ACC>
__global__ void __launch_bounds__(kThreads)
k_project_bwd(uint32_t N, const float *__restrict__ mean, const float *__restrict__ qvec,
              const float *__restrict__ svec, const float *__restrict__ c2w, int detach_depth,
              const uint8_t *__restrict__ mask, const float *__restrict__ g_mean2d,
              const float *__restrict__ g_cov2d, const float *__restrict__ g_depth,
              float *__restrict__ g_mean, float *__restrict__ g_qvec, float *__restrict__ g_svec) {
  const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float *gm = g_mean + 3 * (size_t)n, *gq = g_qvec + 4 * (size_t)n, *gs_ = g_svec + 3 * (size_t)n;
  if (mask != nullptr && mask[n] == 0) {
    if constexpr (!ACC) {
      gm[0] = gm[1] = gm[2] = 0.f;
      gq[0] = gq[1] = gq[2] = gq[3] = 0.f;
      gs_[0] = gs_[1] = gs_[2] = 0.f;
    }
    return;
  }
  auto put = [](float *dst, float v) {
    if constexpr (ACC) atomicAdd(dst, v);
    else *dst = v;
  };
  const ProjGrad o = project_bwd_one(n, mean, qvec, svec, c2w, detach_depth, g_mean2d, g_cov2d,
                                     g_depth != nullptr ? g_depth[n] : 0.0f);
#pragma unroll
  for (int j = 0; j < 3; ++j) put(gs_ + j, o.gs[j]);
#pragma unroll
  for (int k = 0; k < 4; ++k) put(gq + k, o.gq[k]);
#pragma unroll
  for (int j = 0; j < 3; ++j) put(gm + j, o.gm[j]);
}

// The cameras of a batch in one launch: thread n sums its Gaussian's gradients over the views in
// registers (view order) and writes once -- no atomics, one pass over the parameters.  The per-view
// pointers travel in the kernel arguments.
constexpr int kProjViews = 16;
struct ProjBwdViews {
  const float *cam[kProjViews];
  const uint8_t *mask[kProjViews];
  const float *g_mean2d[kProjViews], *g_cov2d[kProjViews], *g_depth[kProjViews];
  // RGB + heads (gsgen_project_gaussians_backward_batch_heads): the view's channel gradients [N,6] = d L / d (r, g, b, d, 1, d*d)
  // and its depths [N]; d L / d depth = g3 + 2 d g5 is formed here, the colour gradient summed over the views
  const float *g_chan6[kProjViews], *depth[kProjViews];
  // moment form (gsgen_project_gaussians_backward_batch_heads_moments): the view's cov2d [N,2,2]; g_mean2d / g_cov2d then hold
  // the MOMENTS (Mu, Mv) / (Muu, Muv, Mvv, -) of the compositing backward's per-pixel weight against the whitened offsets
  // (u, v) of the Cholesky-form Gaussian, expanded here (moments_to_grads)
  const float *cov2d[kProjViews];
  // ... or (RGB + heads) the records the projection launch prepared from it, [N,4] = (p0, p1, p2, ok) (gsgen_geometry_view::chol): no
  // second fp64 Cholesky per (view, Gaussian)
  const float *chol[kProjViews];
};
// d L / d mean2d and d L / d cov2d from the moments  M_ab = sum_pixels g a b,  g = d L / d (a G) * a G,  (a, b) in (u, v):
// with Sigma^-1 d = (k0 u, k1 u + k2 v) (chol_k_of) the reference's sums (kernels.h:394-418)
//   d mean2d += g Sigma^-1 d,   d cov2d += 0.5 g (Sigma^-1 d)(Sigma^-1 d)^T
// are linear in (Mu, Mv) and (Muu, Muv, Mvv) -- once per (view, Gaussian) here instead of 13 packed operations per pixel
// pair in the compositing backward's entry loop.
__device__ __forceinline__ void moments_to_grads(const CholK &k, const float *__restrict__ mom2, const float *__restrict__ mom4,
                                                 double &gm0, double &gm1, double (&gc)[4]) {
  const double Mu = (double)mom2[0], Mv = (double)mom2[1];
  const double Muu = (double)mom4[0], Muv = (double)mom4[1], Mvv = (double)mom4[2];
  gm0 = k.k0 * Mu;
  gm1 = k.k1 * Mu + k.k2 * Mv;
  gc[0] = 0.5 * k.k0 * k.k0 * Muu;
  gc[1] = gc[2] = 0.5 * k.k0 * (k.k1 * Muu + k.k2 * Muv);  // both off-diagonals receive the same value (kernels.h:414-415)
  gc[3] = 0.5 * (k.k1 * k.k1 * Muu + 2.0 * k.k1 * k.k2 * Muv + k.k2 * k.k2 * Mvv);
}
// ... and from the SH kernels' moments against (tx, ty) = det Sigma^-1 d (gauss_sh_pair; det in fp32 as prep_record<MODE_SH> forms
// it, kernels.h:179):  d mean2d = M1 / det,  d cov2d = 0.5 M2 / det^2.  A degenerate record (det <= 0) never contributed.
__device__ __forceinline__ void moments_to_grads_sh(const float *__restrict__ cov2d_n, const float *__restrict__ mom2,
                                                    const float *__restrict__ mom4, double &gm0, double &gm1, double (&gc)[4]) {
  const float det = cov2d_n[0] * cov2d_n[3] - cov2d_n[1] * cov2d_n[2];  // (this file is compiled with -ffp-contract=off)
  const bool ok = (det > 0.0f) && (fabsf(det) <= 3.402823466e+38f);
  const double inv = ok ? 1.0 / (double)det : 0.0;
  const double h = 0.5 * inv * inv;
  gm0 = inv * (double)mom2[0];
  gm1 = inv * (double)mom2[1];
  gc[0] = h * (double)mom4[0];
  gc[1] = gc[2] = h * (double)mom4[1];
  gc[3] = h * (double)mom4[2];
}
// Round 6: a workgroup is 64 Gaussians x 4 VIEW LANES (one wavefront each): wavefront q takes the views q, q + 4, ... of its 64
// Gaussians, the four partial sums meet in LDS and wavefront 0 writes.  One thread per Gaussian looping over all views (rounds
// 1-5) left a 100 k-Gaussian launch at 1.5 wavefronts per SIMD, each running eight dependent fp64 chains back to back: 43 us per
// 8 views on MI355X once the chain went to fp64 -- four times the wavefronts, a quarter of the serial work each.
constexpr int kPbvThreads = 128;  // (round 6, in flight: 64 / 128 / 256 threads measured -- profiles/r06_notes.md section 20)
constexpr int kPbvGauss = 64, kPbvLanes = kPbvThreads / kPbvGauss;
__global__ void __launch_bounds__(kPbvThreads)
k_project_bwd_views(uint32_t N, const float *__restrict__ mean, const float *__restrict__ qvec,
                    const float *__restrict__ svec, ProjBwdViews pv, int n_views, int detach_depth, int accumulate,
                    int moments, float *__restrict__ g_mean, float *__restrict__ g_qvec, float *__restrict__ g_svec,
                    float *__restrict__ g_color, float *__restrict__ stat_accum, float *__restrict__ stat_cnt) {
  __shared__ float part[kPbvLanes > 1 ? kPbvLanes - 1 : 1][15][kPbvGauss];  // the other view lanes' partial sums, [component][Gaussian]: conflict-free
  const int gl = (int)(threadIdx.x & (kPbvGauss - 1)), vq = (int)(threadIdx.x / kPbvGauss);
  const uint32_t n = blockIdx.x * kPbvGauss + gl;
  const bool live = n < N;
  ProjGrad a;
  float gc[3] = {0.f, 0.f, 0.f};
  float gsum = 0.f, visits = 0.f;  // the densify statistics of the backward (gs/gaussian_splatting.py:464-469): sum |d L / d mean2d|, visits
#pragma unroll
  for (int j = 0; j < 3; ++j) { a.gm[j] = 0.f; a.gs[j] = 0.f; }
#pragma unroll
  for (int k = 0; k < 4; ++k) a.gq[k] = 0.f;
  if (live) {
    for (int v = vq; v < n_views; v += kPbvLanes) {
      const uint8_t *m = pv.mask[v];
      if (m != nullptr && m[n] == 0) continue;
      const float *ch = pv.g_chan6[v];
      float gd = pv.g_depth[v] != nullptr ? pv.g_depth[v][n] : 0.f;
      if (ch != nullptr) {  // the depth head and the depth^2 head both feed the view-space depth (include/gsgen_hip.h)
        const float2 *c2 = reinterpret_cast<const float2 *>(ch + 6 * (size_t)n);
        const float2 c01 = c2[0], c23 = c2[1], c45 = c2[2];
        gc[0] += c01.x; gc[1] += c01.y; gc[2] += c23.x;
        gd = c23.y + 2.0f * pv.depth[v][n] * c45.y;
      }
      ProjGrad o;
      if (moments) {
        double gm0, gm1, gcv[4];
        if (moments == 2)
          moments_to_grads_sh(pv.cov2d[v] + 4 * (size_t)n, pv.g_mean2d[v] + 2 * (size_t)n, pv.g_cov2d[v] + 4 * (size_t)n, gm0, gm1, gcv);
        else {
          CholK ck;
          if (pv.chol[v] != nullptr) {
            const float4 c = *reinterpret_cast<const float4 *>(pv.chol[v] + 4 * (size_t)n);
            const double on = c.w != 0.0f ? (double)kInvSc2 : 0.0;
            ck = CholK{on * (double)c.x, on * (double)c.y, on * (double)c.z};
          } else {
            ck = chol_k_of(pv.cov2d[v] + 4 * (size_t)n);
          }
          moments_to_grads(ck, pv.g_mean2d[v] + 2 * (size_t)n, pv.g_cov2d[v] + 4 * (size_t)n, gm0, gm1, gcv);
        }
        // the view's d L / d mean2d in place of its two first moments: the densify statistics read it (gsgen_densify_update_batch,
        // gs/gaussian_splatting.py:464-469)
        *reinterpret_cast<float2 *>(const_cast<float *>(pv.g_mean2d[v]) + 2 * (size_t)n) = make_float2((float)gm0, (float)gm1);
        if (stat_accum != nullptr) {
          const float gx = (float)gm0, gy = (float)gm1;
          gsum += sqrtf(gx * gx + gy * gy);
          visits += 1.0f;
        }
        o = project_bwd_one(n, mean, qvec, svec, pv.cam[v], detach_depth, gm0, gm1, gcv, gd);
      } else {
        if (stat_accum != nullptr) {
          const float2 g2 = *reinterpret_cast<const float2 *>(pv.g_mean2d[v] + 2 * (size_t)n);
          gsum += sqrtf(g2.x * g2.x + g2.y * g2.y);
          visits += 1.0f;
        }
        o = project_bwd_one(n, mean, qvec, svec, pv.cam[v], detach_depth, pv.g_mean2d[v], pv.g_cov2d[v], gd);
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) { a.gm[j] += o.gm[j]; a.gs[j] += o.gs[j]; }
#pragma unroll
      for (int k = 0; k < 4; ++k) a.gq[k] += o.gq[k];
    }
  }
  if (vq > 0) {
    float (*mine)[kPbvGauss] = part[vq - 1];
#pragma unroll
    for (int j = 0; j < 3; ++j) { mine[j][gl] = a.gm[j]; mine[3 + j][gl] = a.gs[j]; mine[10 + j][gl] = gc[j]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) mine[6 + k][gl] = a.gq[k];
    mine[13][gl] = gsum; mine[14][gl] = visits;
  }
  __syncthreads();
  if (vq > 0 || !live) return;
  float *gm = g_mean + 3 * (size_t)n, *gq = g_qvec + 4 * (size_t)n, *gs_ = g_svec + 3 * (size_t)n;
  float *gc_ = g_color != nullptr ? g_color + 3 * (size_t)n : nullptr;
#pragma unroll
  for (int q = 0; q < kPbvLanes - 1; ++q) {
#pragma unroll
    for (int j = 0; j < 3; ++j) { a.gm[j] += part[q][j][gl]; a.gs[j] += part[q][3 + j][gl]; gc[j] += part[q][10 + j][gl]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) a.gq[k] += part[q][6 + k][gl];
    gsum += part[q][13][gl]; visits += part[q][14][gl];
  }
  if (stat_accum != nullptr && visits > 0.0f) {  // (atomics: other batches may be updating the same statistics from other streams)
    atomicAdd(stat_accum + n, gsum);
    if (stat_cnt != nullptr) atomicAdd(stat_cnt + n, visits);
  }
  if (accumulate) {
#pragma unroll
    for (int j = 0; j < 3; ++j) { a.gm[j] += gm[j]; a.gs[j] += gs_[j]; if (gc_ != nullptr) gc[j] += gc_[j]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) a.gq[k] += gq[k];
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) { gm[j] = a.gm[j]; gs_[j] = a.gs[j]; }
#pragma unroll
  for (int k = 0; k < 4; ++k) gq[k] = a.gq[k];
  if (gc_ != nullptr) { gc_[0] = gc[0]; gc_[1] = gc[1]; gc_[2] = gc[2]; }
}

// ---- AABB -> tile rectangle -------------------------------------------------------------------
__device__ __forceinline__ int to_i32_trunc(float v) {
  // torch .to(int32) truncates toward zero; out-of-range / NaN -> INT_MIN like cvttss2si.
  if (!(v == v) || v >= 2147483648.0f || v <= -2147483904.0f) return (int)0x80000000;
  return (int)v;
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

struct Rect { int x0, y0, x1, y1; };
// shift = log2(tile side): 4 in this library's own pipeline; the `_gs` entry point takes any side 1 .. 32 (side > 0: a division
// by a side that is no power of two -- gs/culling.py:27-30 divides by whatever tile_size the config holds)
__device__ __forceinline__ Rect tile_rect(float m2x, float m2y, float c00, float c11, float Dr,
                                          float fx, float fy, float cx, float cy, int w, int h, int shift = 4, int side = 0) {
  const float ax = sqrtf(Dr * c00), ay = sqrtf(Dr * c11);
  const float tlx = m2x - ax, tly = m2y - ay, brx = m2x + ax, bry = m2y + ay;
  float m;
  m = tlx * fx; int px0 = to_i32_trunc(m + cx);
  m = tly * fy; int py0 = to_i32_trunc(m + cy);
  m = brx * fx; int px1 = to_i32_trunc(m + cx);
  m = bry * fy; int py1 = to_i32_trunc(m + cy);
  px0 = clampi(px0, 0, w - 1); px1 = clampi(px1, 0, w - 1);
  py0 = clampi(py0, 0, h - 1); py1 = clampi(py1, 0, h - 1);
  Rect r;  // clamped pixels are >= 0, so floor division == shift
  if (side > 0) { r.x0 = px0 / side; r.y0 = py0 / side; r.x1 = px1 / side; r.y1 = py1 / side; return r; }
  r.x0 = px0 >> shift; r.y0 = py0 >> shift; r.x1 = px1 >> shift; r.y1 = py1 >> shift;
  return r;
}

__global__ void __launch_bounds__(kThreads)
k_aabb_count(uint32_t N, const float *__restrict__ mean2d, const float *__restrict__ cov2d,
             float fx, float fy, float cx, float cy, int w, int h, float Dr, int shift, int side,
             int *__restrict__ tl, int *__restrict__ br, uint32_t *__restrict__ total) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t cnt = 0;
  if (i < N) {
    const float2 m = *reinterpret_cast<const float2 *>(mean2d + 2 * (size_t)i);
    const float4 c = *reinterpret_cast<const float4 *>(cov2d + 4 * (size_t)i);
    const Rect r = tile_rect(m.x, m.y, c.x, c.w, Dr, fx, fy, cx, cy, w, h, shift, side);
    *reinterpret_cast<int2 *>(tl + 2 * (size_t)i) = make_int2(r.x0, r.y0);
    *reinterpret_cast<int2 *>(br + 2 * (size_t)i) = make_int2(r.x1, r.y1);
    cnt = (uint32_t)((r.x1 - r.x0 + 1) * (r.y1 - r.y0 + 1));
  }
  // one atomic per wave
  float dummy = 0.f; (void)dummy;
  uint32_t s = cnt;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += (uint32_t)__shfl_xor((int)s, m, 64);
  if ((threadIdx.x & 63u) == 0 && s != 0) atomicAdd(total, s);
}

// ---- fused per-frame geometry: cull + project + rectangle + per-tile histogram ---------------
// cam layout documented in include/gsgen_hip.h (gsgen_frame_geometry).
__device__ __forceinline__ void
frame_project_body(uint32_t N, const float *__restrict__ mean, const float *__restrict__ qvec,
                const float *__restrict__ svec, const float *__restrict__ cam, int w, int h, int ntw,
                float *__restrict__ mean2d, float *__restrict__ cov2d, float *__restrict__ depth,
                uint8_t *__restrict__ mask, int *__restrict__ tl, int *__restrict__ br, float *__restrict__ chol = nullptr,
                float *__restrict__ max_radii2d = nullptr) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float Rc[9], t[3];
  load_pose(cam, Rc, t);
  const float fx = cam[12], fy = cam[13], cx = cam[14], cy = cam[15];
  const float frustum_r = cam[16], Dr = cam[17];
  const float *p = mean + 3 * (size_t)i, *s = svec + 3 * (size_t)i;
  bool in = true;
  if (frustum_r > 0.0f) {
    const float r = fmaxf(fmaxf(s[0], s[1]), s[2]) * frustum_r;
    in = sphere_in_frustum(p[0], p[1], p[2], r, cam + 20, cam + 38);
  }
  Rect rc{0, 0, -1, -1};  // empty
  float2 m2 = make_float2(0.f, 0.f);
  float4 c2 = make_float4(1.f, 0.f, 0.f, 1.f);
  float z = 0.f;
  if (in) {
    const Proj o = project_one(p, qvec + 4 * (size_t)i, s, Rc, t);
    m2 = make_float2(o.m2x, o.m2y);
    c2 = make_float4(o.c00, o.c01, o.c10, o.c11);
    z = o.depth;
    rc = tile_rect(o.m2x, o.m2y, o.c00, o.c11, Dr, fx, fy, cx, cy, w, h);
  }
  *reinterpret_cast<float2 *>(mean2d + 2 * (size_t)i) = m2;
  *reinterpret_cast<float4 *>(cov2d + 4 * (size_t)i) = c2;
  if (chol != nullptr) {  // the compositing kernels' evaluation record, once per (view, Gaussian) (gsgen_geometry_view::chol)
    const CholRec c = chol_prep(c2.x, c2.y, c2.z, c2.w);
    *reinterpret_cast<float4 *>(chol + 4 * (size_t)i) = make_float4(c.p0, c.p1, c.p2, (in && c.ok) ? 1.0f : 0.0f);
  }
  if (max_radii2d != nullptr && in) {
    // the densify / prune statistic of the forward (gs/gaussian_splatting.py:1240-1245; k_densify_update_views: the same arithmetic), in
    // the launch that forms cov2d anyway: a NaN radius sticks, a negative one never raises the maximum
    const float mm = (c2.x + c2.w) / 2.0f;
    const float det = c2.x * c2.w - c2.y * c2.z;
    const float r = mm + sqrtf(fmaxf(mm * mm - det, 0.0f));
    if (r != r) atomicMax(reinterpret_cast<unsigned int *>(max_radii2d) + i, 0x7fc00000u);
    else if (!(r < 0.0f)) atomicMax(reinterpret_cast<unsigned int *>(max_radii2d) + i, __float_as_uint(r));
  }
  depth[i] = z;
  mask[i] = in ? 1 : 0;
  *reinterpret_cast<int2 *>(tl + 2 * (size_t)i) = make_int2(rc.x0, rc.y0);
  *reinterpret_cast<int2 *>(br + 2 * (size_t)i) = make_int2(rc.x1, rc.y1);
  (void)ntw;
}

__global__ void __launch_bounds__(kThreads)
k_frame_project(uint32_t N, const float *__restrict__ mean, const float *__restrict__ qvec,
                const float *__restrict__ svec, const float *__restrict__ cam, int w, int h, int ntw,
                float *__restrict__ mean2d, float *__restrict__ cov2d, float *__restrict__ depth,
                uint8_t *__restrict__ mask, int *__restrict__ tl, int *__restrict__ br) {
  frame_project_body(N, mean, qvec, svec, cam, w, h, ntw, mean2d, cov2d, depth, mask, tl, br);
}
// Up to kViewPack views per launch: gridDim.y = views.  The per-view pointer table (GeoView) arrives in the KERNEL ARGUMENTS
// -- copied into the dispatch packet at enqueue: nothing staged or pinned, capture-safe -- and the first workgroup of every view
// also leaves its entry in device memory, where the later launches of the chain (binning.hip) read it.  (The table used to
// be written by a one-workgroup launch of its own in front: one more link in every batch's chain of dependent launches.)
constexpr int kViewPack = 8;
struct GeoViewPack { GeoView v[kViewPack]; };
// z_shared / z_quads: an optional block of float4s every view's backward accumulates into (d L / d alpha, d L / d sh or colour),
// zero-filled by ALL workgroups of the launch together (grid-stride over the launch's threads) -- with the per-view targets of
// GeoView this replaces the caller's fill kernel between forward and backward (38.8 MB per 8-view step at cfg2: a launch of its
// own in every step's chain).
constexpr int kFrameThreads = 256;  // (workgroup of the batched projection: 64 / 128 / 256 measured in flight, profiles/r06_notes.md section 20)
__global__ void __launch_bounds__(kFrameThreads)
k_frame_project_views(uint32_t N, const float *__restrict__ mean, const float *__restrict__ qvec,
                      const float *__restrict__ svec, int w, int h, int ntw, GeoViewPack pack, GeoView *__restrict__ dst,
                      float4 *__restrict__ z_shared, uint32_t z_quads) {
  const GeoView &v = pack.v[blockIdx.y];
  if (blockIdx.x == 0 && threadIdx.x == 0) dst[blockIdx.y] = v;
  if (z_shared != nullptr) {
    const uint32_t nthr = gridDim.x * gridDim.y * blockDim.x;
    for (uint32_t i = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; i < z_quads; i += nthr)
      z_shared[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) {
    if (v.z_mean2d != nullptr) *reinterpret_cast<float2 *>(v.z_mean2d + 2 * (size_t)i) = make_float2(0.f, 0.f);
    if (v.z_cov2d != nullptr) *reinterpret_cast<float4 *>(v.z_cov2d + 4 * (size_t)i) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (v.z_chan6 != nullptr) {
      float2 *z = reinterpret_cast<float2 *>(v.z_chan6 + 6 * (size_t)i);
      z[0] = z[1] = z[2] = make_float2(0.f, 0.f);
    }
  }
  frame_project_body(N, mean, qvec, svec, v.cam, w, h, ntw, v.mean2d, v.cov2d, v.depth, v.mask, v.tl, v.br, v.chol, v.max_r);
}

static inline dim3 grid_for(uint32_t n) { return dim3((n + kThreads - 1) / kThreads); }

}  // namespace gs

using namespace gs;

// ---- the model's parameter activations (utils/activations.py:36-57, applied at gs/gaussian_splatting.py:113-124) ---------------------
// svec = act(svec_before_activation) etc. are three torch kernels forward and three autograd nodes backward in the reference's model
// -- launch-latency-sized work on [N,3] / [N] tensors, but ~120 us of HOST time per step of a small training step.  One launch each
// way here, enqueued by the camera batch's C++ autograd node (csrc/torch_batch.cpp) on raw parameters.
// codes: 0 nothing, 1 exp, 2 sigmoid, 3 abs, 4 relu, 5 softplus, 6 biased_relu (+1e-3), 7 biased_abs (+1e-3)
namespace gs {
constexpr float kMinScale = 1e-3f;  // utils/activations.py:17
__device__ __forceinline__ float act_fwd(int code, float x) {
  switch (code) {
    case 1: return expf(x);
    case 2: return 1.0f / (1.0f + expf(-x));
    case 3: return fabsf(x);
    case 4: return fmaxf(x, 0.0f);
    case 5: return x > 20.0f ? x : log1pf(expf(x));  // torch.nn.functional.softplus (beta 1, threshold 20)
    case 6: return fmaxf(x, 0.0f) + kMinScale;
    case 7: return fabsf(x) + kMinScale;
    default: return x;
  }
}
// d act / d x from the raw value x and the activated value y
__device__ __forceinline__ float act_bwd(int code, float x, float y) {
  switch (code) {
    case 1: return y;
    case 2: return y * (1.0f - y);
    case 3: case 7: return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f);
    case 4: case 6: return x > 0.0f ? 1.0f : 0.0f;
    case 5: return x > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-x));
    default: return 1.0f;
  }
}
__global__ void __launch_bounds__(kThreads)
k_activate_fields(uint32_t N, const float *__restrict__ svec_raw, const float *__restrict__ alpha_raw,
                  const float *__restrict__ color_raw, int sa, int aa, int ca, float *__restrict__ svec, float *__restrict__ alpha,
                  float *__restrict__ color) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3 * N) return;
  svec[i] = act_fwd(sa, svec_raw[i]);
  color[i] = act_fwd(ca, color_raw[i]);
  if (i < N) alpha[i] = act_fwd(aa, alpha_raw[i]);
}
__global__ void __launch_bounds__(kThreads)
k_activate_fields_bwd(uint32_t N, const float *__restrict__ svec_raw, const float *__restrict__ alpha_raw,
                      const float *__restrict__ color_raw, const float *__restrict__ svec, const float *__restrict__ alpha,
                      const float *__restrict__ color, int sa, int aa, int ca, float *__restrict__ g_svec, float *__restrict__ g_alpha,
                      float *__restrict__ g_color) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3 * N) return;
  g_svec[i] *= act_bwd(sa, svec_raw[i], svec[i]);
  g_color[i] *= act_bwd(ca, color_raw[i], color[i]);
  if (i < N) g_alpha[i] *= act_bwd(aa, alpha_raw[i], alpha[i]);
}
}  // namespace gs
extern "C" int gsgen_activate_fields(uint32_t N, const float *svec_raw, const float *alpha_raw, const float *color_raw,
                                     int svec_act, int alpha_act, int color_act, float *svec, float *alpha, float *color,
                                     gsgen_stream_t stream) {
  if (N == 0) return 0;
  if (!svec_raw || !alpha_raw || !color_raw || !svec || !alpha || !color) return GSGEN_EINVAL;
  if (svec_act < 0 || svec_act > 7 || alpha_act < 0 || alpha_act > 7 || color_act < 0 || color_act > 7) return GSGEN_EUNSUPPORTED;
  hipLaunchKernelGGL(gs::k_activate_fields, gs::grid_for(3 * N), dim3(gs::kThreads), 0, (hipStream_t)stream, N, svec_raw, alpha_raw,
                     color_raw, svec_act, alpha_act, color_act, svec, alpha, color);
  return (int)hipGetLastError();
}
extern "C" int gsgen_activate_fields_backward(uint32_t N, const float *svec_raw, const float *alpha_raw, const float *color_raw,
                                              const float *svec, const float *alpha, const float *color, int svec_act,
                                              int alpha_act, int color_act, float *g_svec, float *g_alpha, float *g_color,
                                              gsgen_stream_t stream) {
  if (N == 0) return 0;
  if (!svec_raw || !alpha_raw || !color_raw || !svec || !alpha || !color || !g_svec || !g_alpha || !g_color) return GSGEN_EINVAL;
  if (svec_act < 0 || svec_act > 7 || alpha_act < 0 || alpha_act > 7 || color_act < 0 || color_act > 7) return GSGEN_EUNSUPPORTED;
  hipLaunchKernelGGL(gs::k_activate_fields_bwd, gs::grid_for(3 * N), dim3(gs::kThreads), 0, (hipStream_t)stream, N, svec_raw, alpha_raw,
                     color_raw, svec, alpha, color, svec_act, alpha_act, color_act, g_svec, g_alpha, g_color);
  return (int)hipGetLastError();
}

// ---- small host -> device uploads through kernel arguments ------------------------------------------------------
// Per-render constants (a camera block is 272 bytes) reach the device as the ARGUMENTS of a one-workgroup kernel
// instead of a hipMemcpyAsync: the bytes are copied into the dispatch packet when the launch is enqueued, so the
// host buffer may be reused at once, nothing is staged or pinned, the call never waits for the stream, and it can be
// captured into a hipGraph.  (Measured on the public autograd path: torch's pinned `.to(device, non_blocking=True)`
// of the same block cost 0.7 ms of host time per batch -- it waited for the stream -- and made BatchRenderer
// host-bound below 4 x 512^2.)
namespace gs {
constexpr int kUploadWords = 896;  // 3 584 bytes per launch: below the 4 KB kernel-argument segment
struct UploadPack { uint32_t w[kUploadWords]; };
__global__ void __launch_bounds__(256) k_upload_words(UploadPack pack, uint32_t *dst, int n) {
  for (int i = (int)threadIdx.x; i < n; i += (int)blockDim.x) dst[i] = pack.w[i];
}
}  // namespace gs

extern "C" int gsgen_upload_small(void *dst, const void *host_src, size_t bytes, gsgen_stream_t stream) {
  if (bytes == 0) return 0;
  if (!dst || !host_src || (bytes & 3u) != 0 || (reinterpret_cast<uintptr_t>(dst) & 3u) != 0) return GSGEN_EINVAL;
  const uint32_t *src = static_cast<const uint32_t *>(host_src);
  uint32_t *d = static_cast<uint32_t *>(dst);
  size_t words = bytes / 4;
  while (words > 0) {
    const int n = (int)(words < (size_t)gs::kUploadWords ? words : (size_t)gs::kUploadWords);
    gs::UploadPack pack;
    memcpy(pack.w, src, (size_t)n * 4);
    hipLaunchKernelGGL(gs::k_upload_words, dim3(1), dim3(256), 0, (hipStream_t)stream, pack, d, n);
    src += n; d += n; words -= (size_t)n;
  }
  return (int)hipGetLastError();
}

extern "C" {

int gsgen_culling_gaussian_bsphere(uint32_t N, const float *mean, const float *qvec,
                                   const float *svec, const float *normal, const float *pts,
                                   uint8_t *mask, float thresh, gsgen_stream_t stream) {
  (void)qvec;  // unused by the reference too (culling.h:10-19)
  if (N == 0) return 0;
  if (!mean || !svec || !normal || !pts || !mask) return GSGEN_EINVAL;
  hipLaunchKernelGGL(k_cull_bsphere, grid_for(N), dim3(kThreads), 0, (hipStream_t)stream, N, mean,
                     svec, normal, pts, mask, thresh);
  return (int)hipGetLastError();
}

int gsgen_project_gaussians(uint32_t N, const float *mean, const float *qvec, const float *svec,
                            const float *c2w, float *mean2d, float *cov2d, float *JW, float *depth,
                            gsgen_stream_t stream) {
  if (N == 0) return 0;
  if (!mean || !qvec || !svec || !c2w || !mean2d || !cov2d || !depth) return GSGEN_EINVAL;
  hipLaunchKernelGGL(k_project, grid_for(N), dim3(kThreads), 0, (hipStream_t)stream, N, mean, qvec,
                     svec, c2w, mean2d, cov2d, JW, depth);
  return (int)hipGetLastError();
}

int gsgen_project_gaussians_backward_masked(uint32_t N, const float *mean, const float *qvec,
                                            const float *svec, const float *c2w, int detach_depth,
                                            const uint8_t *mask, const float *g_mean2d,
                                            const float *g_cov2d, const float *g_depth,
                                            float *g_mean, float *g_qvec, float *g_svec,
                                            gsgen_stream_t stream) {
  if (N == 0) return 0;
  if (!mean || !qvec || !svec || !c2w || !g_mean2d || !g_cov2d || !g_mean || !g_qvec || !g_svec)
    return GSGEN_EINVAL;
  hipLaunchKernelGGL(k_project_bwd<false>, grid_for(N), dim3(kThreads), 0, (hipStream_t)stream, N,
                     mean, qvec, svec, c2w, detach_depth, mask, g_mean2d, g_cov2d, g_depth, g_mean,
                     g_qvec, g_svec);
  return (int)hipGetLastError();
}

int gsgen_project_gaussians_backward_accum(uint32_t N, const float *mean, const float *qvec,
                                           const float *svec, const float *c2w, int detach_depth,
                                           const uint8_t *mask, const float *g_mean2d,
                                           const float *g_cov2d, const float *g_depth,
                                           float *g_mean, float *g_qvec, float *g_svec,
                                           gsgen_stream_t stream) {
  if (N == 0) return 0;
  if (!mean || !qvec || !svec || !c2w || !g_mean2d || !g_cov2d || !g_mean || !g_qvec || !g_svec)
    return GSGEN_EINVAL;
  hipLaunchKernelGGL(k_project_bwd<true>, grid_for(N), dim3(kThreads), 0, (hipStream_t)stream, N,
                     mean, qvec, svec, c2w, detach_depth, mask, g_mean2d, g_cov2d, g_depth, g_mean,
                     g_qvec, g_svec);
  return (int)hipGetLastError();
}

static int project_bwd_batch(uint32_t n_views, uint32_t N, const float *mean, const float *qvec, const float *svec,
                             const float *const *c2w, int detach_depth, const uint8_t *const *mask,
                             const float *const *g_mean2d, const float *const *g_cov2d, const float *const *g_depth,
                             const float *const *g_chan6, const float *const *depth, float *g_mean, float *g_qvec,
                             float *g_svec, float *g_color, gsgen_stream_t stream, const float *const *cov2d = nullptr,
                             int moments_form = 1, const float *const *chol = nullptr, float *stat_accum = nullptr,
                             float *stat_cnt = nullptr) {
  if (N == 0) return 0;
  if (!mean || !qvec || !svec || !g_mean || !g_qvec || !g_svec) return GSGEN_EINVAL;
  if (n_views && (!c2w || !g_mean2d || !g_cov2d)) return GSGEN_EINVAL;
  if ((g_chan6 != nullptr) != (depth != nullptr) || (g_chan6 != nullptr) != (g_color != nullptr)) return GSGEN_EINVAL;
  for (uint32_t v = 0; v < n_views; ++v) {
    if (!c2w[v] || !g_mean2d[v] || !g_cov2d[v]) return GSGEN_EINVAL;
    if (g_chan6 && (!g_chan6[v] || !depth[v])) return GSGEN_EINVAL;
    if (cov2d && !cov2d[v]) return GSGEN_EINVAL;
  }
  uint32_t v0 = 0;
  do {  // (an empty batch still zero-fills)
    ProjBwdViews pv{};
    const uint32_t nv = (n_views - v0) < (uint32_t)kProjViews ? (n_views - v0) : (uint32_t)kProjViews;
    for (uint32_t i = 0; i < nv; ++i) {
      pv.cam[i] = c2w[v0 + i];
      pv.mask[i] = mask ? mask[v0 + i] : nullptr;
      pv.g_mean2d[i] = g_mean2d[v0 + i];
      pv.g_cov2d[i] = g_cov2d[v0 + i];
      pv.g_depth[i] = g_depth ? g_depth[v0 + i] : nullptr;
      pv.g_chan6[i] = g_chan6 ? g_chan6[v0 + i] : nullptr;
      pv.depth[i] = depth ? depth[v0 + i] : nullptr;
      pv.cov2d[i] = cov2d ? cov2d[v0 + i] : nullptr;
      pv.chol[i] = chol ? chol[v0 + i] : nullptr;
    }
    hipLaunchKernelGGL(k_project_bwd_views, dim3((N + kPbvGauss - 1) / kPbvGauss), dim3(kPbvThreads), 0, (hipStream_t)stream, N, mean, qvec,
                       svec, pv, (int)nv, detach_depth, v0 ? 1 : 0, cov2d ? moments_form : 0, g_mean, g_qvec, g_svec, g_color,
                       stat_accum, stat_cnt);
    v0 += nv;
  } while (v0 < n_views);
  return (int)hipGetLastError();
}

int gsgen_project_gaussians_backward_batch(uint32_t n_views, uint32_t N, const float *mean, const float *qvec,
                                           const float *svec, const float *const *c2w, int detach_depth,
                                           const uint8_t *const *mask, const float *const *g_mean2d,
                                           const float *const *g_cov2d, const float *const *g_depth,
                                           float *g_mean, float *g_qvec, float *g_svec, gsgen_stream_t stream) {
  return project_bwd_batch(n_views, N, mean, qvec, svec, c2w, detach_depth, mask, g_mean2d, g_cov2d, g_depth, nullptr,
                           nullptr, g_mean, g_qvec, g_svec, nullptr, stream);
}

int gsgen_project_gaussians_backward_batch_heads(uint32_t n_views, uint32_t N, const float *mean, const float *qvec,
                                                 const float *svec, const float *const *c2w, int detach_depth,
                                                 const uint8_t *const *mask, const float *const *g_mean2d,
                                                 const float *const *g_cov2d, const float *const *g_chan6,
                                                 const float *const *depth, float *g_mean, float *g_qvec, float *g_svec,
                                                 float *g_color, gsgen_stream_t stream) {
  if (!g_chan6 || !depth || !g_color) return GSGEN_EINVAL;
  return project_bwd_batch(n_views, N, mean, qvec, svec, c2w, detach_depth, mask, g_mean2d, g_cov2d, nullptr, g_chan6,
                           depth, g_mean, g_qvec, g_svec, g_color, stream);
}

int gsgen_project_gaussians_backward_batch_moments_sh(uint32_t n_views, uint32_t N, const float *mean, const float *qvec,
                                                      const float *svec, const float *const *c2w, int detach_depth,
                                                      const uint8_t *const *mask, float *const *g_mom2,
                                                      const float *const *g_mom4, const float *const *cov2d, float *g_mean,
                                                      float *g_qvec, float *g_svec, float *stat_grad_accum, float *stat_cnt,
                                                      gsgen_stream_t stream) {
  if (!cov2d) return GSGEN_EINVAL;
  return project_bwd_batch(n_views, N, mean, qvec, svec, c2w, detach_depth, mask, g_mom2, g_mom4, nullptr, nullptr, nullptr, g_mean,
                           g_qvec, g_svec, nullptr, stream, cov2d, 2, nullptr, stat_grad_accum, stat_cnt);
}

int gsgen_project_gaussians_backward_batch_heads_moments(uint32_t n_views, uint32_t N, const float *mean, const float *qvec,
                                                         const float *svec, const float *const *c2w, int detach_depth,
                                                         const uint8_t *const *mask, float *const *g_mom2,
                                                         const float *const *g_mom4, const float *const *g_chan6,
                                                         const float *const *depth, const float *const *cov2d,
                                                         const float *const *chol, float *g_mean, float *g_qvec, float *g_svec,
                                                         float *g_color, float *stat_grad_accum, float *stat_cnt,
                                                         gsgen_stream_t stream) {
  if (!g_chan6 || !depth || !g_color || !cov2d) return GSGEN_EINVAL;
  return project_bwd_batch(n_views, N, mean, qvec, svec, c2w, detach_depth, mask, g_mom2, g_mom4, nullptr, g_chan6, depth,
                           g_mean, g_qvec, g_svec, g_color, stream, cov2d, 1, chol, stat_grad_accum, stat_cnt);
}

// Host side: the 56-float camera block of gsgen_frame_geometry.  Frustum planes as
// utils/camera.py:225-226,260-294 builds them: scalars in double (python floats), rounded to
// fp32 where they meet an fp32 tensor, every tensor op rounded to fp32 (this file is compiled
// with -ffp-contract=off).
int gsgen_pack_camera(const float *c2w, float fx, float fy, float cx, float cy, uint32_t w, uint32_t h,
                      double near_plane, double far_plane, float frustum_radius, float tile_radius,
                      float *cam) {
  if (!c2w || !cam || w == 0 || h == 0) return GSGEN_EINVAL;
  for (int i = 0; i < 56; ++i) cam[i] = 0.0f;
  for (int i = 0; i < 12; ++i) cam[i] = c2w[i];
  cam[12] = fx; cam[13] = fy; cam[14] = cx; cam[15] = cy;
  cam[16] = frustum_radius; cam[17] = tile_radius;
  const double yfov = 2.0 * atan((double)h / (2.0 * (double)fy));
  const double aspect = (double)w / (double)h;
  const double half_v_d = far_plane * tan(yfov * 0.5);
  const float half_v = (float)half_v_d, half_h = (float)(half_v_d * aspect);
  float up[3], right[3], look[3], t[3], nearp[3], farp[3];
  for (int i = 0; i < 3; ++i) {
    up[i] = -c2w[4 * i + 1]; right[i] = c2w[4 * i]; look[i] = c2w[4 * i + 2]; t[i] = c2w[4 * i + 3];
    nearp[i] = (float)near_plane * look[i];
    farp[i] = (float)far_plane * look[i];
  }
  auto cross = [](const float *a, const float *b, float *o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
  };
  float nrm[6][3], v[3];
  for (int i = 0; i < 3; ++i) { nrm[0][i] = look[i]; nrm[1][i] = -look[i]; }
  for (int i = 0; i < 3; ++i) v[i] = farp[i] - half_h * right[i];
  cross(v, up, nrm[2]);
  for (int i = 0; i < 3; ++i) v[i] = farp[i] + half_h * right[i];
  cross(up, v, nrm[3]);
  for (int i = 0; i < 3; ++i) v[i] = farp[i] + half_v * up[i];
  cross(v, right, nrm[4]);
  for (int i = 0; i < 3; ++i) v[i] = farp[i] - half_v * up[i];
  cross(right, v, nrm[5]);
  for (int k = 0; k < 6; ++k) {
    const float xx = nrm[k][0] * nrm[k][0], yy = nrm[k][1] * nrm[k][1], zz = nrm[k][2] * nrm[k][2];
    float len = sqrtf(xx + yy + zz);
    len = len > 1e-12f ? len : 1e-12f;
    for (int i = 0; i < 3; ++i) cam[20 + 3 * k + i] = nrm[k][i] / len;
  }
  for (int i = 0; i < 3; ++i) {
    cam[38 + i] = nearp[i] + t[i];
    cam[41 + i] = farp[i] + t[i];
    for (int k = 2; k < 6; ++k) cam[38 + 3 * k + i] = t[i];
  }
  return 0;
}

int gsgen_project_gaussians_backward(uint32_t N, const float *mean, const float *qvec,
                                     const float *svec, const float *c2w, int detach_depth,
                                     const float *g_mean2d, const float *g_cov2d,
                                     const float *g_depth, float *g_mean, float *g_qvec,
                                     float *g_svec, gsgen_stream_t stream) {
  return gsgen_project_gaussians_backward_masked(N, mean, qvec, svec, c2w, detach_depth, nullptr,
                                                 g_mean2d, g_cov2d, g_depth, g_mean, g_qvec, g_svec,
                                                 stream);
}

int gsgen_pack_camera_blocks(uint32_t n, const float *c2w, uint32_t c2w_stride, const double *intr,
                             float frustum_radius, float tile_radius, float *blocks) {
  if (n == 0) return 0;
  if (!c2w || !intr || !blocks || c2w_stride < 12) return GSGEN_EINVAL;
  for (uint32_t b = 0; b < n; ++b) {
    const float *m = c2w + (size_t)b * c2w_stride;
    const double *k = intr + 8 * (size_t)b;  // fx fy cx cy w h near far
    float *o = blocks + 68 * (size_t)b;
    if (int e = gsgen_pack_camera(m, (float)k[0], (float)k[1], (float)k[2], (float)k[3], (uint32_t)k[4], (uint32_t)k[5],
                                  k[6], k[7], frustum_radius, tile_radius, o))
      return e;
    // pixel origin of the image plane (utils/camera.py: -cx/fx, -cy/fy in double, rounded once) and the rotation rows
    o[56] = (float)(-k[2] / k[0]);
    o[57] = (float)(-k[3] / k[1]);
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) o[58 + 3 * r + c] = m[4 * r + c];
    o[67] = 0.0f;
  }
  return 0;
}

int gsgen_tile_culling_aabb_count(uint32_t N, const float *mean2d, const float *cov2d,
                                  uint32_t tile_size, float fx, float fy, float cx, float cy,
                                  uint32_t w, uint32_t h, float D, int *aabb_topleft,
                                  int *aabb_bottomright, uint32_t *total, gsgen_stream_t stream) {
  if (tile_size < 1u || tile_size > 32u) return GSGEN_EUNSUPPORTED;  // (the reference: tile_size^2 <= 1024 threads per tile)
  const bool pow2 = (tile_size & (tile_size - 1u)) == 0u;
  int shift = 0;
  while ((1u << shift) < tile_size) ++shift;
  const int side = pow2 ? 0 : (int)tile_size;
  if (!total) return GSGEN_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (hipError_t e = hipMemsetAsync(total, 0, sizeof(uint32_t), s)) return (int)e;
  if (N == 0) return 0;
  if (!mean2d || !cov2d || !aabb_topleft || !aabb_bottomright) return GSGEN_EINVAL;
  hipLaunchKernelGGL(k_aabb_count, grid_for(N), dim3(kThreads), 0, s, N, mean2d, cov2d, fx, fy, cx, cy,
                     (int)w, (int)h, D, shift, side, aabb_topleft, aabb_bottomright, total);
  return (int)hipGetLastError();
}

int gsgen_adam_step(uint64_t n, float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                    uint32_t n_groups, const uint64_t *group_end, const float *group_lr, float beta1,
                    float beta2, float eps, uint32_t step, gsgen_stream_t stream) {
  if (n == 0) return 0;
  if (!param || !grad || !exp_avg || !exp_avg_sq || !group_end || !group_lr) return GSGEN_EINVAL;
  if (n_groups == 0 || n_groups > kAdamMaxGroups || step == 0) return GSGEN_EINVAL;
  AdamArgs a{};
  uint64_t prev = 0;
  for (uint32_t k = 0; k < n_groups; ++k) {
    if (group_end[k] < prev || group_end[k] > n) return GSGEN_EINVAL;
    // lr / (1 - beta1^step) and sqrt(1 - beta2^step) in double, as torch's python-scalar path does
    a.end[k] = prev = group_end[k];
    a.step_size[k] = (float)((double)group_lr[k] / (1.0 - pow((double)beta1, (double)step)));
  }
  if (prev != n) return GSGEN_EINVAL;
  a.n_groups = n_groups;
  a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  const uint64_t n4 = (n + 3) / 4;
  const uint32_t blocks = (uint32_t)((n4 + kThreads - 1) / kThreads);
  hipLaunchKernelGGL(k_adam_step, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, n, param, grad,
                     exp_avg, exp_avg_sq, a);
  return (int)hipGetLastError();
}

int gsgen_adam_step_scalars(uint32_t n_groups, const float *group_lr, float beta1, float beta2, uint32_t step, float *out9) {
  if (!group_lr || !out9 || n_groups == 0 || n_groups > kAdamMaxGroups || step == 0) return GSGEN_EINVAL;
  for (int k = 0; k < kAdamMaxGroups; ++k)  // (gsgen_adam_step's arithmetic)
    out9[k] = k < (int)n_groups ? (float)((double)group_lr[k] / (1.0 - pow((double)beta1, (double)step))) : 0.0f;
  out9[kAdamMaxGroups] = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  return 0;
}

int gsgen_adam_step_device_scalars(uint64_t n, float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                                   uint32_t n_groups, const uint64_t *group_end, float beta1, float beta2, float eps,
                                   const float *scalars, gsgen_stream_t stream) {
  if (n == 0) return 0;
  if (!param || !grad || !exp_avg || !exp_avg_sq || !group_end || !scalars) return GSGEN_EINVAL;
  if (n_groups == 0 || n_groups > kAdamMaxGroups) return GSGEN_EINVAL;
  AdamArgs a{};
  uint64_t prev = 0;
  for (uint32_t k = 0; k < n_groups; ++k) {
    if (group_end[k] < prev || group_end[k] > n) return GSGEN_EINVAL;
    a.end[k] = prev = group_end[k];
  }
  if (prev != n) return GSGEN_EINVAL;
  a.n_groups = n_groups;
  a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  a.dev = scalars;
  const uint64_t n4 = (n + 3) / 4;
  const uint32_t blocks = (uint32_t)((n4 + kThreads - 1) / kThreads);
  hipLaunchKernelGGL(k_adam_step, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, n, param, grad,
                     exp_avg, exp_avg_sq, a);
  return (int)hipGetLastError();
}

int gsgen_densify_update(uint32_t N, const float *cov2d, const float *grad_mean2d,
                         const uint8_t *mask, float *max_radii2d, float *grad_accum, float *cnt,
                         gsgen_stream_t stream) {
  if (N == 0) return 0;
  if ((cov2d == nullptr) != (max_radii2d == nullptr)) return GSGEN_EINVAL;
  if ((grad_mean2d == nullptr) != (grad_accum == nullptr)) return GSGEN_EINVAL;
  if (cnt != nullptr && grad_accum == nullptr) return GSGEN_EINVAL;
  hipLaunchKernelGGL(k_densify_update, grid_for(N), dim3(kThreads), 0, (hipStream_t)stream, N, cov2d,
                     grad_mean2d, mask, max_radii2d, grad_accum, cnt);
  return (int)hipGetLastError();
}

int gsgen_densify_update_batch(uint32_t n_views, uint32_t N, const float *const *cov2d,
                               const float *const *grad_mean2d, const uint8_t *const *mask, float *max_radii2d,
                               float *grad_accum, float *cnt, gsgen_stream_t stream) {
  if (N == 0 || n_views == 0) return 0;
  if ((cov2d == nullptr) != (max_radii2d == nullptr)) return GSGEN_EINVAL;
  if ((grad_mean2d == nullptr) != (grad_accum == nullptr)) return GSGEN_EINVAL;
  if (cnt != nullptr && grad_accum == nullptr) return GSGEN_EINVAL;
  for (uint32_t v = 0; v < n_views; ++v)
    if ((cov2d && !cov2d[v]) || (grad_mean2d && !grad_mean2d[v])) return GSGEN_EINVAL;
  for (uint32_t v0 = 0; v0 < n_views; v0 += kStatViews) {
    DensifyViews dv{};
    const uint32_t nv = (n_views - v0) < (uint32_t)kStatViews ? (n_views - v0) : (uint32_t)kStatViews;
    for (uint32_t i = 0; i < nv; ++i) {
      dv.cov2d[i] = cov2d ? cov2d[v0 + i] : nullptr;
      dv.g_mean2d[i] = grad_mean2d ? grad_mean2d[v0 + i] : nullptr;
      dv.mask[i] = mask ? mask[v0 + i] : nullptr;
    }
    hipLaunchKernelGGL(k_densify_update_views, grid_for(N), dim3(kThreads), 0, (hipStream_t)stream, N, dv, (int)nv,
                       max_radii2d, grad_accum, cnt);
  }
  return (int)hipGetLastError();
}

// used by binning.hip (gsgen_frame_geometry)
// host table -> device table through kernel arguments (no host-pinned staging, capture-safe), then
// the projection of every view in one launch
int gsgen_internal_frame_project_views(uint32_t N, const float *mean, const float *qvec, const float *svec,
                                       int w, int h, int ntw, const GeoView *host_views, GeoView *dev_views,
                                       uint32_t B, float *zero_shared, size_t zero_shared_floats, gsgen_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  const uint32_t gx = N ? (N + (uint32_t)kFrameThreads - 1u) / (uint32_t)kFrameThreads : 1u;  // (N == 0: one workgroup per view, for the table alone)
  for (uint32_t b0 = 0; b0 < B; b0 += kViewPack) {
    GeoViewPack pack{};
    const uint32_t n = (B - b0) < (uint32_t)kViewPack ? (B - b0) : (uint32_t)kViewPack;
    for (uint32_t i = 0; i < n; ++i) pack.v[i] = host_views[b0 + i];
    // (the shared block is zeroed by the first launch of the batch; zero_shared_floats is a multiple of 4, checked by the caller)
    hipLaunchKernelGGL(k_frame_project_views, dim3(gx, n), dim3(kFrameThreads), 0, s, N, mean, qvec, svec, w, h, ntw, pack,
                       dev_views + b0, b0 == 0 ? reinterpret_cast<float4 *>(zero_shared) : (float4 *)nullptr,
                       (uint32_t)(zero_shared_floats / 4));
  }
  return (int)hipGetLastError();
}

int gsgen_internal_frame_project(uint32_t N, const float *mean, const float *qvec, const float *svec,
                                 const float *cam, int w, int h, int ntw, float *mean2d, float *cov2d,
                                 float *depth, uint8_t *mask, int *tl, int *br, gsgen_stream_t stream) {
  if (N == 0) return 0;
  hipLaunchKernelGGL(k_frame_project, grid_for(N), dim3(kThreads), 0, (hipStream_t)stream, N, mean,
                     qvec, svec, cam, w, h, ntw, mean2d, cov2d, depth, mask, tl, br);
  return (int)hipGetLastError();
}

}  // extern "C"
