/*
 * gs_oracle.c -- CPU ORACLE for the GSGEN Gaussian-splatting rasterizer hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (gsgen_amd/) never
 * links, imports or falls back to anything in oracle/.
 *
 * It restates, in plain C, the algorithm of the reference (paths relative to
 * /root/reference).  Every function cites the lines it follows.  fp64 is used exactly
 * where the reference uses fp64 (the RGB/scalar Gaussian evaluation and every Gaussian
 * backward), fp32 elsewhere.  Build with -ffp-contract=off so the fp32 expression
 * order written here is the order executed.
 *
 * Pinning status: PINNED (see oracle/README.md).  The cull / binning / compositing functions
 * are bit-exact on every forward output (and <= 5e-7 on gradients) against the reference's own
 * CUDA sources executed on the CPU through the SIMT shim in oracle/emu (oracle/_ref/libgs_ref.so)
 * and against golden vectors generated from it (tests/golden, tests/test_oracle_golden.py).
 * The projection / tile-count / frustum functions are checked against the reference's Python
 * (gs/renderer.py project_gaussians etc., imported by tests/golden/make_golden.py): exact for
 * the integer tile arithmetic, a few ulp for the projection (torch's BLAS summation order).
 *
 * Gradients: the reference accumulates fp32 atomics in a hardware-dependent order.  The
 * oracle computes every per-(pixel,Gaussian) contribution with the reference's own
 * arithmetic, then sums the contributions in fp64 and rounds once -- the centre of the
 * distribution any atomic order can produce.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MIN_RENDER_ALPHA 0.00392156862745098f /* gs/src/include/common.h:89 */

/* fp64 accumulation shared between OpenMP threads (tiles run in parallel in the backward) */
static inline void acc_add(double *p, double v) {
#pragma omp atomic
  *p += v;
}

/* ------------------------------------------------------------------------------------ */
/* Frustum  (utils/camera.py:260-294)                                                    */
/* ------------------------------------------------------------------------------------ */
static void cross3(const float *a, const float *b, float *o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
static void normalize3(float *v) { /* F.normalize(dim=-1): v / max(||v||, 1e-12) */
  float n = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (n < 1e-12f) n = 1e-12f;
  v[0] /= n; v[1] /= n; v[2] /= n;
}
/* half_vside/half_hside are computed by the caller in double exactly as the reference's
 * python does (far*np.tan(yfov/2), *aspect) and handed over rounded to fp32, which is what
 * `python_float * fp32_tensor` does in torch. */
void gso_frustum(const float *c2w, float near_plane, float far_plane, float half_vside,
                 float half_hside, float *normals /*[6,3]*/, float *pts /*[6,3]*/) {
  float up[3], right[3], look[3], t[3];
  for (int i = 0; i < 3; ++i) {
    up[i] = -c2w[i * 4 + 1];
    right[i] = c2w[i * 4 + 0];
    look[i] = c2w[i * 4 + 2];
    t[i] = c2w[i * 4 + 3];
  }
  float nearp[3], farp[3], a[3], b[3];
  for (int i = 0; i < 3; ++i) { nearp[i] = near_plane * look[i]; farp[i] = far_plane * look[i]; }
  float *n_near = normals, *n_far = normals + 3, *n_l = normals + 6, *n_r = normals + 9,
        *n_u = normals + 12, *n_d = normals + 15;
  for (int i = 0; i < 3; ++i) { n_near[i] = look[i]; n_far[i] = -look[i]; }
  for (int i = 0; i < 3; ++i) a[i] = farp[i] - half_hside * right[i];
  cross3(a, up, n_l);
  for (int i = 0; i < 3; ++i) a[i] = farp[i] + half_hside * right[i];
  cross3(up, a, n_r);
  for (int i = 0; i < 3; ++i) a[i] = farp[i] + half_vside * up[i];
  cross3(a, right, n_u);
  for (int i = 0; i < 3; ++i) b[i] = farp[i] - half_vside * up[i];
  cross3(right, b, n_d);
  for (int k = 0; k < 6; ++k) normalize3(normals + 3 * k);
  for (int i = 0; i < 3; ++i) {
    pts[0 + i] = nearp[i] + t[i];
    pts[3 + i] = farp[i] + t[i];
    pts[6 + i] = t[i]; pts[9 + i] = t[i]; pts[12 + i] = t[i]; pts[15 + i] = t[i];
  }
}

/* ------------------------------------------------------------------------------------ */
/* Frustum cull  (gs/src/include/culling.h:10-19, kernels.h:156-170)                      */
/* ------------------------------------------------------------------------------------ */
void gso_cull_bsphere(int N, const float *mean, const float *svec, const float *normals,
                      const float *pts, float thresh, uint8_t *mask) {
  for (int i = 0; i < N; ++i) {
    const float *s = svec + 3 * i, *m = mean + 3 * i;
    float r = fmaxf(fmaxf(s[0], s[1]), s[2]) * thresh;
    uint8_t ok = 1;
    for (int k = 0; k < 6; ++k) {
      const float *n = normals + 3 * k, *p = pts + 3 * k;
      /* helper_math dot(): a.x*b.x + a.y*b.y + a.z*b.z */
      float d = (m[0] - p[0]) * n[0] + (m[1] - p[1]) * n[1] + (m[2] - p[2]) * n[2];
      if (!(d > -r)) { ok = 0; break; }
    }
    mask[i] = ok;
  }
}

/* ------------------------------------------------------------------------------------ */
/* EWA projection forward  (gs/renderer.py:366-421, utils/transforms.py:34-46,            */
/* kornia 0.6.0 quaternion_to_rotation_matrix(order=WXYZ) -- not vendored; its published  */
/* formula is restated in quat_to_rot below)                                              */
/* ------------------------------------------------------------------------------------ */
static void quat_to_rot(const float *q, float *R /*3x3*/, float *qn /*normalised*/, float *nrm) {
  float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < 1e-12f) n = 1e-12f; /* F.normalize(p=2, eps=1e-12) */
  *nrm = n;
  float w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
  qn[0] = w; qn[1] = x; qn[2] = y; qn[3] = z;
  float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
  float twx = tx * w, twy = ty * w, twz = tz * w;
  float txx = tx * x, txy = ty * x, txz = tz * x;
  float tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0f - (tyy + tzz); R[1] = txy - twz;          R[2] = txz + twy;
  R[3] = txy + twz;          R[4] = 1.0f - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;          R[7] = tyz + twx;          R[8] = 1.0f - (txx + tyy);
}

void gso_project(int N, const float *mean, const float *qvec, const float *svec,
                 const float *c2w, int detach_depth, float *mean2d /*[N,2]*/,
                 float *cov2d /*[N,4]*/, float *JW /*[N,9] or NULL*/, float *depth /*[N]*/) {
  (void)detach_depth; /* forward value is identical either way */
  float Rc[9], t[3];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) Rc[i * 3 + j] = c2w[i * 4 + j];
    t[i] = c2w[i * 4 + 3];
  }
#pragma omp parallel for schedule(static)
  for (int n = 0; n < N; ++n) {
    const float *p = mean + 3 * n, *s = svec + 3 * n;
    /* project_pts (gs/renderer.py:382-388): u = R^T (p + (-t)) */
    float d0 = p[0] - t[0], d1 = p[1] - t[1], d2 = p[2] - t[2];
    float u[3];
    for (int i = 0; i < 3; ++i) u[i] = Rc[0 * 3 + i] * d0 + Rc[1 * 3 + i] * d1 + Rc[2 * 3 + i] * d2;
    float Rq[9], qn[4], nrm;
    quat_to_rot(qvec + 4 * n, Rq, qn, &nrm);
    float M[9]; /* utils/transforms.py:41: M[i][j] = s[j] * R[i][j] */
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) M[i * 3 + j] = s[j] * Rq[i * 3 + j];
    float S[9]; /* gs/renderer.py:401: Sigma = M M^T */
    for (int i = 0; i < 3; ++i)
      for (int k = 0; k < 3; ++k)
        S[i * 3 + k] = M[i * 3 + 0] * M[k * 3 + 0] + M[i * 3 + 1] * M[k * 3 + 1] + M[i * 3 + 2] * M[k * 3 + 2];
    /* jacobian (gs/renderer.py:366-378) */
    float x = u[0], y = u[1], z = u[2];
    float l = sqrtf(x * x + y * y + z * z);
    float J[9] = {1.0f / z, 0.0f, -x / z / z, 0.0f, 1.0f / z, -y / z / z, x / l, y / l, z / l};
    float A[9]; /* JW = J W, W = Rc^T  (gs/renderer.py:402-404) */
    for (int i = 0; i < 3; ++i)
      for (int k = 0; k < 3; ++k)
        A[i * 3 + k] = J[i * 3 + 0] * Rc[k * 3 + 0] + J[i * 3 + 1] * Rc[k * 3 + 1] + J[i * 3 + 2] * Rc[k * 3 + 2];
    float T1[6]; /* rows 0..1 of A*S */
    for (int a = 0; a < 2; ++a)
      for (int k = 0; k < 3; ++k)
        T1[a * 3 + k] = A[a * 3 + 0] * S[0 * 3 + k] + A[a * 3 + 1] * S[1 * 3 + k] + A[a * 3 + 2] * S[2 * 3 + k];
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b)
        cov2d[4 * n + a * 2 + b] = T1[a * 3 + 0] * A[b * 3 + 0] + T1[a * 3 + 1] * A[b * 3 + 1] + T1[a * 3 + 2] * A[b * 3 + 2];
    if (JW) memcpy(JW + 9 * n, A, sizeof(A));
    depth[n] = z;
    mean2d[2 * n + 0] = x / z; /* gs/renderer.py:416-419 */
    mean2d[2 * n + 1] = y / z;
  }
}

/* Backward of gso_project as torch autograd would compute it through
 * gs/renderer.py:391-421 (J is @no_grad; depth in the divide detached iff detach_depth). */
void gso_project_bwd(int N, const float *mean, const float *qvec, const float *svec,
                     const float *c2w, int detach_depth, const float *g_mean2d,
                     const float *g_cov2d, const float *g_depth /*may be NULL*/,
                     float *g_mean, float *g_qvec, float *g_svec) {
  double Rc[9], t[3];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) Rc[i * 3 + j] = c2w[i * 4 + j];
    t[i] = c2w[i * 4 + 3];
  }
#pragma omp parallel for schedule(static)
  for (int n = 0; n < N; ++n) {
    const float *p = mean + 3 * n, *s = svec + 3 * n, *q = qvec + 4 * n;
    double d[3] = {p[0] - t[0], p[1] - t[1], p[2] - t[2]}, u[3];
    for (int i = 0; i < 3; ++i) u[i] = Rc[i] * d[0] + Rc[3 + i] * d[1] + Rc[6 + i] * d[2];
    double nq = sqrt((double)q[0] * q[0] + (double)q[1] * q[1] + (double)q[2] * q[2] + (double)q[3] * q[3]);
    if (nq < 1e-12) nq = 1e-12;
    double w = q[0] / nq, x = q[1] / nq, y = q[2] / nq, z = q[3] / nq;
    double Rq[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                    2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                    2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
    double M[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) M[i * 3 + j] = s[j] * Rq[i * 3 + j];
    double ux = u[0], uy = u[1], uz = u[2];
    double J[6] = {1.0 / uz, 0.0, -ux / uz / uz, 0.0, 1.0 / uz, -uy / uz / uz};
    double A[6]; /* first two rows of JW */
    for (int a = 0; a < 2; ++a)
      for (int k = 0; k < 3; ++k)
        A[a * 3 + k] = J[a * 3 + 0] * Rc[k * 3 + 0] + J[a * 3 + 1] * Rc[k * 3 + 1] + J[a * 3 + 2] * Rc[k * 3 + 2];
    const float *g = g_cov2d + 4 * n;
    double dS[9];
    for (int j = 0; j < 3; ++j)
      for (int k = 0; k < 3; ++k) {
        double acc = 0;
        for (int a = 0; a < 2; ++a)
          for (int b = 0; b < 2; ++b) acc += A[a * 3 + j] * (double)g[a * 2 + b] * A[b * 3 + k];
        dS[j * 3 + k] = acc;
      }
    double dM[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double acc = 0;
        for (int k = 0; k < 3; ++k) acc += (dS[i * 3 + k] + dS[k * 3 + i]) * M[k * 3 + j];
        dM[i * 3 + j] = acc;
      }
    double dR[9];
    for (int j = 0; j < 3; ++j) {
      double acc = 0;
      for (int i = 0; i < 3; ++i) { acc += dM[i * 3 + j] * Rq[i * 3 + j]; dR[i * 3 + j] = dM[i * 3 + j] * s[j]; }
      g_svec[3 * n + j] = (float)acc;
    }
    double dq[4];
    dq[0] = 2 * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
    dq[1] = 2 * (y * dR[1] + z * dR[2] + y * dR[3] - 2 * x * dR[4] - w * dR[5] + z * dR[6] + w * dR[7] - 2 * x * dR[8]);
    dq[2] = 2 * (-2 * y * dR[0] + x * dR[1] + w * dR[2] + x * dR[3] + z * dR[5] - w * dR[6] + z * dR[7] - 2 * y * dR[8]);
    dq[3] = 2 * (-2 * z * dR[0] - w * dR[1] + x * dR[2] + w * dR[3] - 2 * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
    double qh[4] = {w, x, y, z};
    double dot = qh[0] * dq[0] + qh[1] * dq[1] + qh[2] * dq[2] + qh[3] * dq[3];
    for (int k = 0; k < 4; ++k) g_qvec[4 * n + k] = (float)((dq[k] - qh[k] * dot) / nq);
    double gm0 = g_mean2d[2 * n], gm1 = g_mean2d[2 * n + 1];
    double du[3] = {gm0 / uz, gm1 / uz, g_depth ? (double)g_depth[n] : 0.0};
    if (!detach_depth) du[2] += -(ux * gm0 + uy * gm1) / (uz * uz);
    for (int j = 0; j < 3; ++j)
      g_mean[3 * n + j] = (float)(Rc[j * 3 + 0] * du[0] + Rc[j * 3 + 1] * du[1] + Rc[j * 3 + 2] * du[2]);
  }
}

/* ------------------------------------------------------------------------------------ */
/* AABB -> tile rectangle + pair count (gs/culling.py:8-37, utils/camera.py:301-314)      */
/* ------------------------------------------------------------------------------------ */
static int to_i32_trunc(float v) { /* torch .to(int32): truncation toward zero */
  if (!(v == v)) return (int)0x80000000;
  if (v >= 2147483648.0f || v <= -2147483904.0f) return (int)0x80000000; /* x86/torch-CPU cvttss2si convention */
  return (int)v;
}
static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static int floordiv(int a, int b) { int q = a / b; if ((a % b != 0) && ((a < 0) != (b < 0))) --q; return q; }

long long gso_aabb_count(int N, const float *mean2d, const float *cov2d, int tile_size, float fx,
                         float fy, float cx, float cy, int w, int h, float Dr, int *tl /*[N,2]*/,
                         int *br /*[N,2]*/) {
  long long total = 0;
  for (int n = 0; n < N; ++n) {
    float ax = sqrtf(Dr * cov2d[4 * n + 0]);
    float ay = sqrtf(Dr * cov2d[4 * n + 3]);
    float tlx = mean2d[2 * n] - ax, tly = mean2d[2 * n + 1] - ay;
    float brx = mean2d[2 * n] + ax, bry = mean2d[2 * n + 1] + ay;
    /* camera_space_to_pixel_space: p*f (rounded) then + c (rounded), then trunc */
    float m;
    m = tlx * fx; int px0 = to_i32_trunc(m + cx);
    m = tly * fy; int py0 = to_i32_trunc(m + cy);
    m = brx * fx; int px1 = to_i32_trunc(m + cx);
    m = bry * fy; int py1 = to_i32_trunc(m + cy);
    px0 = clampi(px0, 0, w - 1); px1 = clampi(px1, 0, w - 1);
    py0 = clampi(py0, 0, h - 1); py1 = clampi(py1, 0, h - 1);
    tl[2 * n] = floordiv(px0, tile_size); tl[2 * n + 1] = floordiv(py0, tile_size);
    br[2 * n] = floordiv(px1, tile_size); br[2 * n + 1] = floordiv(py1, tile_size);
    total += (long long)(br[2 * n] - tl[2 * n] + 1) * (long long)(br[2 * n + 1] - tl[2 * n + 1] + 1);
  }
  return total;
}

/* Densification statistics (gs/gaussian_splatting.py:1240-1245 radii, :464-469 grad norm). */
/* det restated as c00*c11 - c01*c10 (torch.det goes through an LU whose rounding is library  */
/* specific; tests compare against torch with a tolerance).                                  */
void gso_densify_update(int N, const float *cov2d, const float *g_mean2d, const unsigned char *mask,
                        float *max_radii2d, float *grad_accum, float *cnt) {
  for (int n = 0; n < N; ++n) {
    if (mask && !mask[n]) continue;
    if (max_radii2d) {
      const float *c = cov2d + 4 * n;
      float m = (c[0] + c[3]) / 2.0f;
      float a = c[0] * c[3], b = c[1] * c[2];
      float det = a - b;
      float mm = m * m;
      float d = mm - det;
      float r = m + sqrtf(d > 0.0f ? d : 0.0f);
      if (r > max_radii2d[n]) max_radii2d[n] = r;
    }
    if (grad_accum) {
      float gx = g_mean2d[2 * n], gy = g_mean2d[2 * n + 1];
      float xx = gx * gx, yy = gy * gy;
      grad_accum[n] += sqrtf(xx + yy);
      if (cnt) cnt[n] += 1.0f;
    }
  }
}

/* ------------------------------------------------------------------------------------ */
/* Binning + sort (gs/src/include/aabb_culling.h:15-41, 70-103, 192-260)                  */
/* key = int64{hi = tile id, lo = float bits of depth}, stable ascending sort on all 64   */
/* bits.  Emission order (the tie-break of the reference's atomics) is not defined by the */
/* reference; the oracle emits by ascending Gaussian id, x-major then y as the kernel's    */
/* loop nest does.                                                                        */
/* ------------------------------------------------------------------------------------ */
typedef struct { int64_t key; int id; int seq; } gso_pair;
static int cmp_pair(const void *a, const void *b) {
  const gso_pair *x = (const gso_pair *)a, *y = (const gso_pair *)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->seq < y->seq ? -1 : (x->seq > y->seq ? 1 : 0); /* stability */
}
int gso_bin_sort(int N, long long D, int n_tiles_h, int n_tiles_w, const int *tl, const int *br,
                 const float *depth, int *gaussian_ids /*[D]*/, int *start /*[T]*/, int *end /*[T]*/) {
  int T = n_tiles_h * n_tiles_w;
  for (int t = 0; t < T; ++t) { start[t] = -1; end[t] = -1; }
  if (D <= 0) return 0;
  gso_pair *pairs = (gso_pair *)malloc(sizeof(gso_pair) * (size_t)D);
  long long pos = 0;
  for (int n = 0; n < N; ++n) {
    uint32_t bits; memcpy(&bits, depth + n, 4);
    for (int i = tl[2 * n]; i <= br[2 * n]; ++i)
      for (int j = tl[2 * n + 1]; j <= br[2 * n + 1]; ++j) {
        if (pos >= D) { free(pairs); return -1; }
        int tile = j * n_tiles_w + i;
        pairs[pos].key = (int64_t)(((uint64_t)(uint32_t)tile << 32) | bits);
        pairs[pos].id = n; pairs[pos].seq = (int)pos; ++pos;
      }
  }
  if (pos != D) { free(pairs); return -1; } /* aabb_culling.h:228 assert(size_h == N_with_dub) */
  qsort(pairs, (size_t)D, sizeof(gso_pair), cmp_pair);
  for (long long k = 0; k < D; ++k) {
    gaussian_ids[k] = pairs[k].id;
    int tile = (int)(pairs[k].key >> 32);
    if (k == 0 || (int)(pairs[k - 1].key >> 32) != tile) start[tile] = (int)k;
    if (k == D - 1 || (int)(pairs[k + 1].key >> 32) != tile) end[tile] = (int)(k + 1);
  }
  free(pairs);
  return 0;
}

/* ------------------------------------------------------------------------------------ */
/* 2-D Gaussian evaluation (gs/src/include/kernels.h:172-193 fp32, :195-224 fp64)          */
/* and its backward (kernels.h:394-418): contributions returned, not atomically added.    */
/* ------------------------------------------------------------------------------------ */
static float gauss2d_f64(const float *mean, const float *cov, const float *q) {
  double c0 = cov[0], c1 = cov[1], c2 = cov[2], c3 = cov[3];
  double det = c0 * c3 - c1 * c2;
  double x = q[0] - mean[0]; /* float subtraction, then widened */
  double y = q[1] - mean[1];
  double tx = x * c3 - y * c2;
  double ty = -x * c1 + y * c0;
  double radial = tx * x + ty * y;
  radial /= det;
  if (radial < 0.0) radial = 1000.0;
  return (float)exp(-0.5 * radial);
}
static float gauss2d_f32(const float *mean, const float *cov, const float *q) {
  float c0 = cov[0], c1 = cov[1], c2 = cov[2], c3 = cov[3];
  float det = c0 * c3 - c1 * c2;
  float x = q[0] - mean[0];
  float y = q[1] - mean[1];
  float tx = x * c3 - y * c2;
  float ty = -x * c1 + y * c0;
  float radial = tx * x + ty * y;
  radial /= det;
  if (radial < 0.0) radial = 1000.0f;
  return (float)expf((float)(-0.5 * radial)); /* -0.5*radial is a double product of an fp32 value: exact */
}
/* adds the six contributions (each rounded to fp32 as the reference's atomicAdd operand is) into fp64 accumulators */
static void gauss2d_bwd(const float *mean, const float *cov, const float *q, float grad,
                        double *acc_mean /*2*/, double *acc_cov /*4*/) {
  double dg = (double)grad;
  double c0 = cov[0], c1 = cov[1], c2 = cov[2], c3 = cov[3];
  double det = c0 * c3 - c1 * c2;
  double x = q[0] - mean[0];
  double y = q[1] - mean[1];
  double tx = (x * c3 - y * c2) / det;
  double ty = (-x * c1 + y * c0) / det;
  acc_add(acc_mean + 0, (double)(float)(dg * tx));
  acc_add(acc_mean + 1, (double)(float)(dg * ty));
  acc_add(acc_cov + 0, (double)(float)(0.5 * (float)(dg * tx * tx)));
  acc_add(acc_cov + 1, (double)(float)(0.5 * (float)(dg * tx * ty)));
  acc_add(acc_cov + 2, (double)(float)(0.5 * (float)(dg * ty * tx)));
  acc_add(acc_cov + 3, (double)(float)(0.5 * (float)(dg * ty * ty)));
}

/* pixel position: topleft + g*pixel_size, product rounded before the add
 * (vol_render.h:186-187; nvcc may contract this into an fma -- the two differ by <=1ulp of
 * the position and the oracle keeps the uncontracted form) */
static void pixel_pos(const float *topleft, int gx, int gy, float psx, float psy, float *pos) {
  float mx = gx * psx, my = gy * psy;
  pos[0] = topleft[0] + mx;
  pos[1] = topleft[1] + my;
}

/* ------------------------------------------------------------------------------------ */
/* RGB compositing forward                                                               */
/*  variant 0: tile_based_vol_rendering_start_end_with_T (vol_render.h:994-1062) whose     */
/*             body is vol_render_one_batch_v1 (:169-265, NaN guards + T clamp)           */
/*  variant 1: tile_based_vol_rendering_start_end (:782-847), body identical to v1 on     */
/*             the live path                                                              */
/* out must be pre-zeroed, T pre-set to 1 by the caller (empty tiles are not written).    */
/* ------------------------------------------------------------------------------------ */
void gso_render_rgb_fwd(const float *mean, const float *cov, const float *color, const float *alpha,
                        const int *start, const int *end, const int *ids, const float *topleft,
                        int tile_size, int n_tiles_h, int n_tiles_w, float psx, float psy, int H,
                        int W, float thresh, float *out /*[H,W,3]*/, float *T /*[H,W] or NULL*/) {
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
  for (int ty = 0; ty < n_tiles_h; ++ty)
    for (int tx = 0; tx < n_tiles_w; ++tx) {
      int tile = ty * n_tiles_w + tx;
      if (start[tile] == -1) continue;
      int n = end[tile] - start[tile];
      if (n == 0) continue;
      const int *lst = ids + start[tile];
      for (int ly = 0; ly < tile_size; ++ly)
        for (int lx = 0; lx < tile_size; ++lx) {
          int gy = ty * tile_size + ly, gx = tx * tile_size + lx;
          if (gy >= H || gx >= W) continue;
          float pos[2];
          pixel_pos(topleft, gx, gy, psx, psy, pos);
          float o[3] = {0.f, 0.f, 0.f}, cum = 1.0f;
          for (int k = 0; k < n; ++k) {
            if (cum < thresh) break;
            int g = lst[k];
            float a = fminf(alpha[g], 0.99f);
            float coeff = a * cum;
            float val = gauss2d_f64(mean + 2 * g, cov + 4 * g, pos);
            coeff *= val;
            if (isnan(coeff)) coeff = 0.0f;
            if (a * val < MIN_RENDER_ALPHA) continue;
            for (int c = 0; c < 3; ++c) { o[c] += color[3 * g + c] * coeff; if (isnan(o[c])) o[c] = 0.0f; }
            cum *= (1 - a * val);
            if (isnan(cum) || cum < 0.0 || cum > 1.0) cum = 0.0f;
          }
          for (int c = 0; c < 3; ++c) out[3 * (gy * W + gx) + c] = o[c];
          if (T) T[gy * W + gx] = cum;
        }
    }
}

/* RGB compositing backward (vol_render.h:866-973, body :318-418).  `final` is the saved
 * forward image INCLUDING T*bg (gs/renderer.py:1182,1239). grads are overwritten (not
 * accumulated) with the fp64 sum of the per-pair contributions. */
void gso_render_rgb_bwd(int N, const float *mean, const float *cov, const float *color,
                        const float *alpha, const int *start, const int *end, const int *ids,
                        const float *final, const float *grad_out, const float *topleft,
                        int tile_size, int n_tiles_h, int n_tiles_w, float psx, float psy, int H,
                        int W, float thresh, float *g_mean, float *g_cov, float *g_color,
                        float *g_alpha) {
  double *acc = (double *)calloc((size_t)N * 10, sizeof(double));
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
  for (int ty = 0; ty < n_tiles_h; ++ty)
    for (int tx = 0; tx < n_tiles_w; ++tx) {
      int tile = ty * n_tiles_w + tx;
      if (start[tile] == -1) continue;
      int n = end[tile] - start[tile];
      if (n == 0) continue;
      const int *lst = ids + start[tile];
      for (int ly = 0; ly < tile_size; ++ly)
        for (int lx = 0; lx < tile_size; ++lx) {
          int gy = ty * tile_size + ly, gx = tx * tile_size + lx;
          if (gy >= H || gx >= W) continue;
          float pos[2];
          pixel_pos(topleft, gx, gy, psx, psy, pos);
          const float *go = grad_out + 3 * (gy * W + gx), *fin = final + 3 * (gy * W + gx);
          float pre[3] = {0.f, 0.f, 0.f}, cum = 1.0f;
          for (int k = 0; k < n; ++k) {
            if (cum < thresh) break;
            int g = lst[k];
            float a = fminf(alpha[g], 0.99f);
            float G = gauss2d_f64(mean + 2 * g, cov + 4 * g, pos);
            if (a * G < MIN_RENDER_ALPHA) continue;
            float coeff = a * cum * G;
            double *A = acc + (size_t)g * 10;
            for (int c = 0; c < 3; ++c) {
              pre[c] += color[3 * g + c] * coeff;
              acc_add(A + 6 + c, (double)(coeff * go[c]));
            }
            double pAG = 0.0;
            for (int c = 0; c < 3; ++c)
              pAG += (color[3 * g + c] * cum - (fin[c] - pre[c]) / (1 - a * G)) * go[c];
            gauss2d_bwd(mean + 2 * g, cov + 4 * g, pos, (float)(pAG * a * G), A, A + 2);
            acc_add(A + 9, (double)(float)(pAG * G));
            cum *= (1 - a * G);
          }
        }
    }
  for (int g = 0; g < N; ++g) {
    const double *A = acc + (size_t)g * 10;
    g_mean[2 * g] = (float)A[0]; g_mean[2 * g + 1] = (float)A[1];
    for (int c = 0; c < 4; ++c) g_cov[4 * g + c] = (float)A[2 + c];
    for (int c = 0; c < 3; ++c) g_color[3 * g + c] = (float)A[6 + c];
    g_alpha[g] = (float)A[9];
  }
  free(acc);
}

/* ------------------------------------------------------------------------------------ */
/* Scalar compositing (vol_render_scalar.h:14-102 fwd, :104-234 bwd)                      */
/* ------------------------------------------------------------------------------------ */
void gso_render_scalar_fwd(const float *mean, const float *cov, const float *scalar,
                           const float *alpha, const int *start, const int *end, const int *ids,
                           const float *topleft, int tile_size, int n_tiles_h, int n_tiles_w,
                           float psx, float psy, int H, int W, float thresh, float *out /*[H,W]*/,
                           float *T /*[H,W]*/) {
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
  for (int ty = 0; ty < n_tiles_h; ++ty)
    for (int tx = 0; tx < n_tiles_w; ++tx) {
      int tile = ty * n_tiles_w + tx;
      if (start[tile] == -1) continue;
      int n = end[tile] - start[tile];
      if (n == 0) continue;
      const int *lst = ids + start[tile];
      for (int ly = 0; ly < tile_size; ++ly)
        for (int lx = 0; lx < tile_size; ++lx) {
          int gy = ty * tile_size + ly, gx = tx * tile_size + lx;
          if (gy >= H || gx >= W) continue;
          float pos[2];
          pixel_pos(topleft, gx, gy, psx, psy, pos);
          float o = 0.f, cum = 1.0f;
          for (int k = 0; k < n; ++k) {
            if (cum < thresh) break;
            int g = lst[k];
            float a = fminf(alpha[g], 0.99f);
            float coeff = a * cum;
            float val = gauss2d_f64(mean + 2 * g, cov + 4 * g, pos);
            coeff *= val;
            if (a * val < MIN_RENDER_ALPHA) continue;
            o += coeff * scalar[g];
            cum *= (1 - a * val);
          }
          out[gy * W + gx] = o;
          T[gy * W + gx] = cum;
        }
    }
}

void gso_render_scalar_bwd(int N, const float *mean, const float *cov, const float *scalar,
                           const float *alpha, const int *start, const int *end, const int *ids,
                           const float *final, const float *grad_out, const float *topleft,
                           int tile_size, int n_tiles_h, int n_tiles_w, float psx, float psy, int H,
                           int W, float thresh, float *g_mean, float *g_cov, float *g_scalar,
                           float *g_alpha) {
  double *acc = (double *)calloc((size_t)N * 8, sizeof(double));
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
  for (int ty = 0; ty < n_tiles_h; ++ty)
    for (int tx = 0; tx < n_tiles_w; ++tx) {
      int tile = ty * n_tiles_w + tx;
      if (start[tile] == -1) continue;
      int n = end[tile] - start[tile];
      if (n == 0) continue;
      const int *lst = ids + start[tile];
      for (int ly = 0; ly < tile_size; ++ly)
        for (int lx = 0; lx < tile_size; ++lx) {
          int gy = ty * tile_size + ly, gx = tx * tile_size + lx;
          if (gy >= H || gx >= W) continue;
          float pos[2];
          pixel_pos(topleft, gx, gy, psx, psy, pos);
          float go = grad_out[gy * W + gx], fin = final[gy * W + gx];
          float o = 0.f, cum = 1.0f;
          for (int k = 0; k < n; ++k) {
            if (cum < thresh) break;
            int g = lst[k];
            float a = fminf(alpha[g], 0.99f);
            float G = gauss2d_f64(mean + 2 * g, cov + 4 * g, pos);
            if (a * G < MIN_RENDER_ALPHA) continue;
            float coeff = a * cum * G;
            double *A = acc + (size_t)g * 8;
            o += scalar[g] * coeff;
            acc_add(A + 6, (double)(coeff * go));
            float pAG = 0.0f;
            pAG += go * (scalar[g] * cum - (fin - o) / (1 - a * G));
            gauss2d_bwd(mean + 2 * g, cov + 4 * g, pos, pAG * a * G, A, A + 2);
            acc_add(A + 7, (double)(pAG * G));
            cum *= (1 - a * G);
          }
        }
    }
  for (int g = 0; g < N; ++g) {
    const double *A = acc + (size_t)g * 8;
    g_mean[2 * g] = (float)A[0]; g_mean[2 * g + 1] = (float)A[1];
    for (int c = 0; c < 4; ++c) g_cov[4 * g + c] = (float)A[2 + c];
    g_scalar[g] = (float)A[6];
    g_alpha[g] = (float)A[7];
  }
  free(acc);
}

/* ------------------------------------------------------------------------------------ */
/* Real spherical-harmonic basis, bands C = 1..4 (gs/src/include/shencoder.h:13-62).       */
/* The constants are the closed forms noted beside each line there.                        */
/* ------------------------------------------------------------------------------------ */
void gso_sh_basis(const float *dir, int C, float *Y) {
  float x = dir[0], y = dir[1], z = dir[2];
  float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
  Y[0] = 0.28209479177387814f;
  if (C <= 1) return;
  Y[1] = -0.48860251190291987f * y;
  Y[2] = 0.48860251190291987f * z;
  Y[3] = -0.48860251190291987f * x;
  if (C <= 2) return;
  Y[4] = 1.0925484305920792f * xy;
  Y[5] = -1.0925484305920792f * yz;
  Y[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
  Y[7] = -1.0925484305920792f * xz;
  Y[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
  if (C <= 3) return;
  Y[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
  Y[10] = 2.8906114426405538f * xy * z;
  Y[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
  Y[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
  Y[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
  Y[14] = 1.4453057213202769f * z * (x2 - y2);
  Y[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
}
static float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); } /* shencoder.h:4 */

/* ray direction of a pixel (vol_render_sh.h:48-65): rows are read as three PACKED float3
 * from `rot9` -- the reference reinterprets whatever it is given that way. */
static void pixel_dir(const float *rot9, const float *pos3, float *dir) {
  for (int i = 0; i < 3; ++i) dir[i] = rot9[3 * i] * pos3[0] + rot9[3 * i + 1] * pos3[1] + rot9[3 * i + 2] * pos3[2];
  float len = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
  dir[0] /= len; dir[1] /= len; dir[2] /= len;
}

/* SH compositing forward: vol_render_sh.h:97-248 (bg == NULL) and vol_render_bg.h:12-110
 * (bg != NULL: adds bg*T in-kernel and writes bg to empty tiles). */
void gso_render_sh_fwd(const float *mean, const float *cov, const float *sh /*[N,3,C*C]*/,
                       const float *alpha, const int *start, const int *end, const int *ids,
                       const float *topleft, const float *rot9, int C, const float *bg /*[3]|NULL*/,
                       int tile_size, int n_tiles_h, int n_tiles_w, float psx, float psy, int H,
                       int W, float thresh, float *out /*[H,W,3]*/, float *T /*[H,W]|NULL (extra)*/) {
  int CC = C * C;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
  for (int ty = 0; ty < n_tiles_h; ++ty)
    for (int tx = 0; tx < n_tiles_w; ++tx) {
      int tile = ty * n_tiles_w + tx;
      int n = (start[tile] == -1) ? 0 : end[tile] - start[tile];
      const int *lst = ids + (n ? start[tile] : 0);
      for (int ly = 0; ly < tile_size; ++ly)
        for (int lx = 0; lx < tile_size; ++lx) {
          int gy = ty * tile_size + ly, gx = tx * tile_size + lx;
          if (gy >= H || gx >= W) continue;
          if (n == 0) {
            if (bg) for (int c = 0; c < 3; ++c) out[3 * (gy * W + gx) + c] = bg[c];
            continue;
          }
          float pos[3], dir[3], Y[16];
          pixel_pos(topleft, gx, gy, psx, psy, pos);
          pos[2] = 1.0f;
          pixel_dir(rot9, pos, dir);
          gso_sh_basis(dir, C, Y);
          float o[3] = {0.f, 0.f, 0.f}, cum = 1.0f;
          for (int k = 0; k < n; ++k) {
            if (cum < thresh) break;
            int g = lst[k];
            float a = fminf(alpha[g], 0.99f);
            float coeff = a * cum;
            float val = gauss2d_f32(mean + 2 * g, cov + 4 * g, pos);
            coeff *= val;
            if (a * val < MIN_RENDER_ALPHA) continue;
            if (isnan(coeff)) coeff = 0.0f;
            for (int c = 0; c < 3; ++c) {
              const float *co = sh + ((size_t)3 * g + c) * CC;
              float s = 0.0f;
              for (int i = 0; i < CC; ++i) s += co[i] * Y[i];
              float yv = sigmoidf_(s);
              if (isnan(yv * coeff)) yv = 0.0f;
              o[c] += coeff * yv;
            }
            cum *= (1 - a * val);
          }
          for (int c = 0; c < 3; ++c) out[3 * (gy * W + gx) + c] = bg ? o[c] + bg[c] * cum : o[c];
          if (T) T[gy * W + gx] = cum;
        }
    }
}

/* Test aid (no counterpart in the reference): how close every pixel of the SH forward above came to its two
 * discontinuous decisions -- margin[gy*W+gx][0] = min over the splats it evaluated of |a*G - 1/255| / (1/255) (the
 * skip test of line "a * val < MIN_RENDER_ALPHA"), [1] = min over its stop tests of |T - thresh| / thresh.  A GPU
 * pixel that differs from this oracle by more than the 1e-4 tolerance is only acceptable if its margin is a few
 * ulps: two correct fp32 evaluations of G (different exp implementations) may then decide differently. */
void gso_sh_decision_margin(const float *mean, const float *cov, const float *alpha, const int *start,
                            const int *end, const int *ids, const float *topleft, int tile_size, int n_tiles_h,
                            int n_tiles_w, float psx, float psy, int H, int W, float thresh, float *margin /*[H,W,2]*/) {
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
  for (int ty = 0; ty < n_tiles_h; ++ty)
    for (int tx = 0; tx < n_tiles_w; ++tx) {
      int tile = ty * n_tiles_w + tx;
      int n = (start[tile] == -1) ? 0 : end[tile] - start[tile];
      const int *lst = ids + (n ? start[tile] : 0);
      for (int ly = 0; ly < tile_size; ++ly)
        for (int lx = 0; lx < tile_size; ++lx) {
          int gy = ty * tile_size + ly, gx = tx * tile_size + lx;
          if (gy >= H || gx >= W) continue;
          float m_skip = 1e30f, m_stop = 1e30f;
          float pos[3];
          pixel_pos(topleft, gx, gy, psx, psy, pos);
          float cum = 1.0f;
          for (int k = 0; k < n; ++k) {
            m_stop = fminf(m_stop, fabsf(cum - thresh) / thresh);
            if (cum < thresh) break;
            int g = lst[k];
            float a = fminf(alpha[g], 0.99f);
            float val = gauss2d_f32(mean + 2 * g, cov + 4 * g, pos);
            m_skip = fminf(m_skip, fabsf(a * val - MIN_RENDER_ALPHA) / MIN_RENDER_ALPHA);
            if (a * val < MIN_RENDER_ALPHA) continue;
            cum *= (1 - a * val);
          }
          margin[2 * (gy * W + gx)] = m_skip;
          margin[2 * (gy * W + gx) + 1] = m_stop;
        }
    }
}

/* SH compositing backward: vol_render_sh.h:268-455 / vol_render_bg.h:131-242.  `final` is
 * the saved forward output (including bg*T for the bg variant). */
void gso_render_sh_bwd(int N, const float *mean, const float *cov, const float *sh,
                       const float *alpha, const int *start, const int *end, const int *ids,
                       const float *final, const float *grad_out, const float *topleft,
                       const float *rot9, int C, int tile_size, int n_tiles_h, int n_tiles_w,
                       float psx, float psy, int H, int W, float thresh, float *g_mean,
                       float *g_cov, float *g_sh, float *g_alpha) {
  int CC = C * C;
  int F = 7 + 3 * CC;
  double *acc = (double *)calloc((size_t)N * F, sizeof(double));
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
  for (int ty = 0; ty < n_tiles_h; ++ty)
    for (int tx = 0; tx < n_tiles_w; ++tx) {
      int tile = ty * n_tiles_w + tx;
      if (start[tile] == -1) continue;
      int n = end[tile] - start[tile];
      if (n == 0) continue;
      const int *lst = ids + start[tile];
      for (int ly = 0; ly < tile_size; ++ly)
        for (int lx = 0; lx < tile_size; ++lx) {
          int gy = ty * tile_size + ly, gx = tx * tile_size + lx;
          if (gy >= H || gx >= W) continue;
          float pos[3], dir[3], Y[16];
          pixel_pos(topleft, gx, gy, psx, psy, pos);
          pos[2] = 1.0f;
          pixel_dir(rot9, pos, dir);
          gso_sh_basis(dir, C, Y);
          const float *go = grad_out + 3 * (gy * W + gx), *fin = final + 3 * (gy * W + gx);
          float o[3] = {0.f, 0.f, 0.f}, cum = 1.0f;
          for (int k = 0; k < n; ++k) {
            if (cum < thresh) break;
            int g = lst[k];
            float a = fminf(alpha[g], 0.99f);
            float G = gauss2d_f32(mean + 2 * g, cov + 4 * g, pos);
            if (a * G < MIN_RENDER_ALPHA) continue;
            float coeff = a * cum * G;
            if (isnan(coeff)) coeff = 0.0f;
            double *A = acc + (size_t)g * F;
            float yv[3];
            for (int c = 0; c < 3; ++c) {
              const float *co = sh + ((size_t)3 * g + c) * CC;
              float s = 0.0f;
              for (int i = 0; i < CC; ++i) s += co[i] * Y[i];
              yv[c] = sigmoidf_(s);
              if (isnan(yv[c] * coeff)) yv[c] = 0.0f;
              o[c] += coeff * yv[c];
              float gs = coeff * (yv[c] * (1.0f - yv[c])) * go[c];
              for (int i = 0; i < CC; ++i) acc_add(A + 7 + c * CC + i, (double)(gs * Y[i]));
            }
            float pAG = 0.0f;
            for (int c = 0; c < 3; ++c) pAG += go[c] * (yv[c] * cum - (fin[c] - o[c]) / (1 - a * G));
            gauss2d_bwd(mean + 2 * g, cov + 4 * g, pos, pAG * a * G, A, A + 2);
            acc_add(A + 6, (double)(pAG * G));
            cum *= (1 - a * G);
          }
        }
    }
  for (int g = 0; g < N; ++g) {
    const double *A = acc + (size_t)g * F;
    g_mean[2 * g] = (float)A[0]; g_mean[2 * g + 1] = (float)A[1];
    for (int c = 0; c < 4; ++c) g_cov[4 * g + c] = (float)A[2 + c];
    g_alpha[g] = (float)A[6];
    for (int i = 0; i < 3 * CC; ++i) g_sh[(size_t)g * 3 * CC + i] = (float)A[7 + i];
  }
  free(acc);
}

/* ------------------------------------------------------------------------------------ */
/* Work statistics of the SH forward (not part of the reference): per pixel, the list     */
/* index at which it stopped (n if it never saturated) and how many entries contributed.  */
/* Used by tools/workstats.py to size the GPU kernels' work; never by the product.        */
/* ------------------------------------------------------------------------------------ */
void gso_sh_workstats(const float *mean, const float *cov, const float *alpha, const int *start,
                      const int *end, const int *ids, const float *topleft, int tile_size,
                      int n_tiles_h, int n_tiles_w, float psx, float psy, int H, int W, float thresh,
                      int *stop_idx /*[H,W]*/, int *n_contrib /*[H,W]*/, int *last_contrib /*[H,W]*/) {
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
  for (int ty = 0; ty < n_tiles_h; ++ty)
    for (int tx = 0; tx < n_tiles_w; ++tx) {
      int tile = ty * n_tiles_w + tx;
      int n = (start[tile] == -1) ? 0 : end[tile] - start[tile];
      const int *lst = ids + (n ? start[tile] : 0);
      for (int ly = 0; ly < tile_size; ++ly)
        for (int lx = 0; lx < tile_size; ++lx) {
          int gy = ty * tile_size + ly, gx = tx * tile_size + lx;
          if (gy >= H || gx >= W) continue;
          float pos[2];
          pixel_pos(topleft, gx, gy, psx, psy, pos);
          float cum = 1.0f;
          int k, nc = 0, last = -1;
          for (k = 0; k < n; ++k) {
            if (cum < thresh) break;
            int g = lst[k];
            float a = fminf(alpha[g], 0.99f);
            float val = gauss2d_f32(mean + 2 * g, cov + 4 * g, pos);
            if (a * val < MIN_RENDER_ALPHA) continue;
            ++nc; last = k;
            cum *= (1 - a * val);
          }
          stop_idx[gy * W + gx] = k;
          n_contrib[gy * W + gx] = nc;
          last_contrib[gy * W + gx] = last;
        }
    }
}

/* Per-wavefront work statistics of the compositing kernels' pixel partitions (tools only):
 * rows_of_part[16] maps a tile row to its wavefront (4 strips of 4 rows for the forward; the
 * backward's two interleaved halves).  Counts (wavefront, entry) pairs walked (some pixel of
 * the part still alive) and contributing (some pixel alive and a*G >= 1/255). */
void gso_part_workstats(const float *mean, const float *cov, const float *alpha, const int *start,
                        const int *end, const int *ids, const float *topleft, int n_tiles_h,
                        int n_tiles_w, float psx, float psy, int H, int W, float thresh,
                        const int *rows_of_part, int n_parts, long long *walked, long long *contrib,
                        long long *pix_pairs) {
  long long w_tot = 0, c_tot = 0, p_tot = 0;
#pragma omp parallel for schedule(dynamic, 1) collapse(2) reduction(+ : w_tot, c_tot, p_tot)
  for (int ty = 0; ty < n_tiles_h; ++ty)
    for (int tx = 0; tx < n_tiles_w; ++tx) {
      int tile = ty * n_tiles_w + tx;
      int n = (start[tile] == -1) ? 0 : end[tile] - start[tile];
      const int *lst = ids + (n ? start[tile] : 0);
      float cum[256];
      for (int i = 0; i < 256; ++i) cum[i] = 1.0f;
      for (int k = 0; k < n; ++k) {
        int g = lst[k];
        float a = fminf(alpha[g], 0.99f);
        int part_alive[8] = {0}, part_con[8] = {0};
        for (int ly = 0; ly < 16; ++ly)
          for (int lx = 0; lx < 16; ++lx) {
            int gy = ty * 16 + ly, gx = tx * 16 + lx;
            if (gy >= H || gx >= W) continue;
            float *c = &cum[ly * 16 + lx];
            if (*c < thresh) continue;
            int part = rows_of_part[ly];
            part_alive[part] = 1;
            float pos[2];
            pixel_pos(topleft, gx, gy, psx, psy, pos);
            float val = gauss2d_f32(mean + 2 * g, cov + 4 * g, pos);
            if (a * val < MIN_RENDER_ALPHA) continue;
            part_con[part] = 1;
            ++p_tot;
            *c *= (1 - a * val);
          }
        for (int q = 0; q < n_parts; ++q) { w_tot += part_alive[q]; c_tot += part_con[q]; }
      }
    }
  *walked = w_tot; *contrib = c_tot; *pix_pairs = p_tot;
}


/* ------------------------------------------------------------------------------------ */
/* Legacy binning (gs/src/include/tile_ops.h, culling.h:48-130, kernels.h:253-350): one   */
/* membership test per (tile, Gaussian) -- the Gaussian's value at the four tile corners  */
/* above `thresh` (mode 0), or the bounding circle reaching the tile (mode 1).  Used by   */
/* the reference's older GaussianRenderer / gs/debug.py / gs/benchmarks.py only.         */
/* ------------------------------------------------------------------------------------ */
static int legacy_hit_corners(float tlx, float tly, unsigned tile_size, float psx, float psy,
                              const float *mean, const float *cov, float thresh) {
  /* kernels.h:253-272; corner coordinates in fp32, Gaussian in fp64 (kernels.h:226-251) */
  float ex = (float)tile_size * psx, ey = (float)tile_size * psy;
  float xr = tlx + ex, yb = tly + ey;
  float q[2], m = 0.0f, v;
  q[0] = tlx; q[1] = tly; v = gauss2d_f64(mean, cov, q); m = v > m ? v : m;
  q[0] = xr; q[1] = tly; v = gauss2d_f64(mean, cov, q); m = v > m ? v : m;
  q[0] = tlx; q[1] = yb; v = gauss2d_f64(mean, cov, q); m = v > m ? v : m;
  q[0] = xr; q[1] = yb; v = gauss2d_f64(mean, cov, q); m = v > m ? v : m;
  return m > thresh;
}
static float legacy_dist_seg(float x, float y, float x1, float x2, float y1, float y2) {
  /* kernels.h:274-304 */
  float A = x - x1, B = y - y1, C = x2 - x1, D = y2 - y1;
  float ac = A * C, bd = B * D, dot = ac + bd;
  float cc = C * C, dd = D * D, len_sq = cc + dd;
  float param = -1.0f, xx, yy;
  if (len_sq != 0) param = dot / len_sq;
  if (param < 0) { xx = x1; yy = y1; }
  else if (param > 1) { xx = x2; yy = y2; }
  else { float pc = param * C, pd = param * D; xx = x1 + pc; yy = y1 + pd; }
  float dx = x - xx, dy = y - yy;
  float dx2 = dx * dx, dy2 = dy * dy;
  return sqrtf(dx2 + dy2);
}
static int legacy_hit_bcircle(float tlx, float tly, unsigned tile_size, float psx, float psy,
                              const float *mean, float radius) {
  /* kernels.h:306-350 */
  float rx = mean[0] - tlx, ry = mean[1] - tly;
  float px = psx * (float)tile_size, py = psy * (float)tile_size;
  if (rx >= 0 && rx <= px && ry >= 0 && ry <= py) return 1;
  float d1 = legacy_dist_seg(rx, ry, 0.0f, px, 0.0f, 0.0f);
  float d2 = legacy_dist_seg(rx, ry, 0.0f, px, py, py);
  float d3 = legacy_dist_seg(rx, ry, 0.0f, 0.0f, 0.0f, py);
  float d4 = legacy_dist_seg(rx, ry, px, px, 0.0f, py);
  float d = fminf(fminf(d1, d2), fminf(d3, d4));
  return d < radius;
}
static void legacy_tile_topleft(const float *topleft, int tx, int ty, unsigned tile_size, float psx, float psy,
                                float *tlx, float *tly) {
  /* topleft[0] + pixel_size_x * tile_x * tile_size, left to right in fp32 (tile_ops.h:51-53) */
  float ax = psx * (float)tx, ay = psy * (float)ty;
  ax = ax * (float)tile_size; ay = ay * (float)tile_size;
  *tlx = topleft[0] + ax; *tly = topleft[1] + ay;
}
static int legacy_hit(int mode, const float *topleft, int tx, int ty, unsigned tile_size, float psx, float psy,
                      const float *mean, const float *shape, int i, float thresh) {
  float tlx, tly;
  legacy_tile_topleft(topleft, tx, ty, tile_size, psx, psy, &tlx, &tly);
  return mode == 0 ? legacy_hit_corners(tlx, tly, tile_size, psx, psy, mean + 2 * i, shape + 4 * i, thresh)
                   : legacy_hit_bcircle(tlx, tly, tile_size, psx, psy, mean + 2 * i, shape[i]);
}

/* count_num_gaussians_each_tile{,_bcircle}: num_gaussians[tile] += hits (render.cu:46-97) */
void gso_legacy_count(int mode, int N, const float *mean, const float *shape, const float *topleft,
                      unsigned tile_size, int nth, int ntw, float psx, float psy, float thresh,
                      int *num_gaussians) {
#pragma omp parallel for schedule(dynamic, 4)
  for (int t = 0; t < nth * ntw; ++t) {
    int cnt = 0;
    for (int i = 0; i < N; ++i)
      cnt += legacy_hit(mode, topleft, t % ntw, t / ntw, tile_size, psx, psy, mean, shape, i, thresh);
    num_gaussians[t] += cnt;
  }
}

/* image_sort (mode 0, tile_ops.h:457-506) / prepare_image_sort (mode 1, :365-455): offset =
 * exclusive scan of the incoming tile_n_gaussians; keys {lo = depth bits, hi = tile} and ids
 * filled per tile in Gaussian-index order (tiledepth keeps them UNSORTED); gaussian_ids = ids
 * stably sorted by the signed 64-bit key; mode 0 also recounts tile_n_gaussians.  Returns the
 * number of pairs written (== N_with_dub when the caller's counts were right). */
long long gso_legacy_image_sort(int mode, int N, long long N_with_dub, int *gaussian_ids,
                                unsigned long long *tiledepth, const float *depth, int *tile_n_gaussians,
                                int *offset, const float *mean, const float *shape, const float *topleft,
                                unsigned tile_size, int nth, int ntw, float psx, float psy, float thresh) {
  int T = nth * ntw;
  long long run = 0;
  for (int t = 0; t < T; ++t) { offset[t] = (int)run; run += tile_n_gaussians[t]; }
  int *uns = (int *)calloc((size_t)(N_with_dub > 0 ? N_with_dub : 1), sizeof(int));
  long long total = 0;
  for (int t = 0; t < T; ++t) {
    long long off = offset[t];
    int cnt = 0;
    for (int i = 0; i < N; ++i)
      if (legacy_hit(mode, topleft, t % ntw, t / ntw, tile_size, psx, psy, mean, shape, i, thresh)) {
        if (off < N_with_dub) {
          unsigned bits; memcpy(&bits, depth + i, 4);
          tiledepth[off] = ((unsigned long long)(unsigned)t << 32) | bits;
          uns[off] = i;
        }
        ++off; ++cnt;
      }
    if (mode == 0) tile_n_gaussians[t] = cnt;
    total += cnt;
  }
  /* stable sort by the signed key: tiles ascending, then depth bits ascending as unsigned */
  long long n = N_with_dub;
  long long *order = (long long *)malloc(sizeof(long long) * (size_t)(n > 0 ? n : 1));
  for (long long k = 0; k < n; ++k) order[k] = k;
  /* insertion-free: per-tile segments are contiguous already, sort each by (depth bits, position) */
  for (int t = 0; t < T; ++t) {
    long long b = offset[t], e = (t + 1 < T) ? offset[t + 1] : n;
    if (e > n) e = n;
    for (long long a = b + 1; a < e; ++a) {  /* stable insertion sort (test sizes are small) */
      long long cur = order[a];
      unsigned long long key = tiledepth[cur];
      long long j = a - 1;
      while (j >= b && (long long)tiledepth[order[j]] > (long long)key) { order[j + 1] = order[j]; --j; }
      order[j + 1] = cur;
    }
  }
  for (long long k = 0; k < n; ++k) gaussian_ids[k] = uns[order[k]];
  free(order); free(uns);
  return total;
}
