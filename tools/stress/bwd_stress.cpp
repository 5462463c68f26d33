// bwd_stress -- fresh-process stress of the SH compositing backward variants (no Python, no torch).
//
//   bwd_stress [--variant vec|mfma1|mfma2|mfma4] [--ppl 1|2|4] [--batch B] [--C 1..4] [--N n] [--size px]
//              [--launches R] [--poison] [--test-first] [--seed s] [--segments k] [--quiet]
//
// Builds a seeded synthetic scene, runs the library's own geometry + SH forward through the C ABI
// (include/gsgen_hip.h), then launches the SH backward R times with the chosen kernel variant and compares every
// launch with a reference made IN THE SAME PROCESS by the vector-ALU kernel (gsgen_debug_set_variant switches the
// variant table).  Identical inputs must give gradients that agree to fp32 atomics-order noise (plus the
// split-bf16 rounding of the matrix-core contraction): anything above --tol is reported with the output it was
// in and the first offending indices.  --test-first runs the R test launches BEFORE the reference launches, so
// that the very first kernels this process executes are the ones under test (the "first launches of a process"
// form of the matrix-core hazard, profiles/r01_notes.md).  --poison fills the LDS of every CU with NaN patterns
// before each launch: a kernel that reads LDS it never wrote then produces NaNs instead of plausible numbers.
// Exit code: 0 all launches agree, 1 some launch disagreed, 2 setup error.  One line of JSON on stdout.
//
// tools/stress/run_matrix.sh starts it in many fresh processes per variant.  TEST TOOL: not part of the product.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/gsgen_hip.h"

#define HIPCHECK(x)                                                                      \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                           \
    }                                                                                    \
  } while (0)
#define GSCHECK(x)                                                                        \
  do {                                                                                    \
    int e_ = (x);                                                                         \
    if (e_ != 0) {                                                                        \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, gsgen_error_string(e_)); \
      exit(2);                                                                            \
    }                                                                                     \
  } while (0)

struct Rng {  // splitmix64 -> uniform / normal
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 1) {}
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  double normal() {
    const double u = uni() + 1e-300, v = uni();
    return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v);
  }
};

template <typename T>
static T *dalloc(size_t n) {
  void *p = nullptr;
  HIPCHECK(hipMalloc(&p, (n ? n : 1) * sizeof(T)));
  return (T *)p;
}
template <typename T>
static T *dupload(const std::vector<T> &h) {
  T *d = dalloc<T>(h.size());
  HIPCHECK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

#include "poison_regs.inc"
// every SIMD, every vector / accumulator register: NaN patterns (a 512-register wavefront: one per SIMD at a time)
__global__ void __launch_bounds__(64) k_poison_regs(int *sink) {
  const uint32_t pat = 0x7fc0beef;
  GSGEN_POISON_REGS_ASM(pat);
  if (threadIdx.x == 1000) *sink = 2;
}
// every CU, all of its LDS: NaN patterns (quiet NaN with a recognisable payload)
__global__ void __launch_bounds__(1024) k_poison_lds(int *sink) {
  extern __shared__ uint32_t lds[];
  const int n = 160 * 1024 / 4;
  for (int i = (int)threadIdx.x; i < n; i += (int)blockDim.x) lds[i] = 0x7fc0dead;
  __syncthreads();
  if (lds[(threadIdx.x * 37u) % (unsigned)n] == 0u) *sink = 1;  // keeps the stores alive
}

static void look_at(const double pos[3], float c2w[12]) {  // tests/scenes.py look_at (up = +z, at = origin)
  double z[3] = {-pos[0], -pos[1], -pos[2]};
  const double zn = sqrt(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]);
  for (double &v : z) v /= zn;
  const double y0[3] = {0, 0, -1};
  double x[3] = {y0[1] * z[2] - y0[2] * z[1], y0[2] * z[0] - y0[0] * z[2], y0[0] * z[1] - y0[1] * z[0]};
  const double xn = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  for (double &v : x) v /= xn;
  const double y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]};
  for (int r = 0; r < 3; ++r) {
    c2w[4 * r + 0] = (float)x[r]; c2w[4 * r + 1] = (float)y[r]; c2w[4 * r + 2] = (float)z[r]; c2w[4 * r + 3] = (float)pos[r];
  }
}

struct View {
  float *cam, *topleft, *rot;
  float *mean2d, *cov2d, *depth;
  uint8_t *mask;
  int *ids, *start, *end;
  uint32_t *total;
  void *ws;
  size_t ws_bytes;
  float *out, *go;
  void *seg_ws;
  float fx;
};

int main(int argc, char **argv) {
  std::string variant = "vec";
  int ppl = 0, B = 0, C = 4, N = 30000, size = 512, R = 8, seed = 1, nseg = 0;
  bool poison = false, poison_regs = false, test_first = false, quiet = false;
  int dump_bad = 0;
  double tol = 3e-5;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto val = [&]() -> const char * { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2); } return argv[++i]; };
    if (a == "--variant") variant = val();
    else if (a == "--ppl") ppl = atoi(val());
    else if (a == "--batch") B = atoi(val());
    else if (a == "--C") C = atoi(val());
    else if (a == "--N") N = atoi(val());
    else if (a == "--size") size = atoi(val());
    else if (a == "--launches") R = atoi(val());
    else if (a == "--seed") seed = atoi(val());
    else if (a == "--segments") nseg = atoi(val());
    else if (a == "--tol") tol = atof(val());
    else if (a == "--poison") poison = true;
    else if (a == "--poison-regs") poison_regs = true;
    else if (a == "--dump-bad") dump_bad = atoi(val());
    else if (a == "--test-first") test_first = true;
    else if (a == "--quiet") quiet = true;
    else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
  }
  const int mfma = variant == "vec" ? 0 : (variant == "mfma1" ? 1 : (variant == "mfma2" ? 2 : (variant == "mfma4" ? 4 : -1)));
  if (mfma < 0 || C < 1 || C > 4 || N < 1 || R < 1) { fprintf(stderr, "bad arguments\n"); return 2; }
  const int nv = B > 0 ? B : 1;  // views
  const int W = size, H = size, nth = (H + 15) / 16, ntw = (W + 15) / 16, T = nth * ntw, CC3 = 3 * C * C;

  // ---- scene (post-activation parameters, as tests/scenes.py random_scene) ----
  Rng rng((uint64_t)seed);
  std::vector<float> mean(3 * (size_t)N), qvec(4 * (size_t)N), svec(3 * (size_t)N), alpha(N), sh((size_t)N * CC3);
  for (int i = 0; i < N; ++i) {
    for (int k = 0; k < 3; ++k) mean[3 * i + k] = (float)(0.6 * rng.normal());
    double q[4], qn = 0;
    for (double &v : q) { v = rng.normal(); qn += v * v; }
    for (int k = 0; k < 4; ++k) qvec[4 * i + k] = (float)(q[k] / sqrt(qn));
    for (int k = 0; k < 3; ++k) svec[3 * i + k] = (float)(0.03 * exp(0.3 * rng.normal()));
    alpha[i] = (float)(0.1 + 0.89 * rng.uni());
    for (int k = 0; k < CC3; ++k) sh[(size_t)i * CC3 + k] = (float)(0.3 * rng.normal());
    for (int c = 0; c < 3; ++c) sh[(size_t)i * CC3 + c * C * C] = (float)(1.5 * rng.normal());
  }
  float *d_mean = dupload(mean), *d_qvec = dupload(qvec), *d_svec = dupload(svec), *d_alpha = dupload(alpha), *d_sh = dupload(sh);

  hipStream_t s;
  HIPCHECK(hipStreamCreate(&s));
  const uint32_t D_cap = (uint32_t)(24 * (size_t)N > (1u << 16) ? 24 * (size_t)N : (1u << 16));
  std::vector<View> views(nv);
  for (int v = 0; v < nv; ++v) {
    View &V = views[v];
    const double az = 0.7 * v + 0.3, el = 0.25 + 0.1 * v, dist = 2.4 + 0.05 * v;
    const double pos[3] = {dist * cos(el) * cos(az), dist * cos(el) * sin(az), dist * sin(el)};
    float c2w[12], cam[56];
    look_at(pos, c2w);
    V.fx = (float)size * (1.0f + 0.05f * v);
    GSCHECK(gsgen_pack_camera(c2w, V.fx, V.fx, size / 2.0f, size / 2.0f, W, H, 0.01, 100.0, 6.0f, 6.0f, cam));
    V.cam = dupload(std::vector<float>(cam, cam + 56));
    V.topleft = dupload(std::vector<float>{-(size / 2.0f) / V.fx, -(size / 2.0f) / V.fx});
    V.rot = dupload(std::vector<float>{c2w[0], c2w[1], c2w[2], c2w[4], c2w[5], c2w[6], c2w[8], c2w[9], c2w[10]});
    V.mean2d = dalloc<float>(2 * (size_t)N); V.cov2d = dalloc<float>(4 * (size_t)N); V.depth = dalloc<float>(N);
    V.mask = dalloc<uint8_t>(N); V.ids = dalloc<int>(D_cap); V.start = dalloc<int>(T); V.end = dalloc<int>(T);
    V.total = dalloc<uint32_t>(1);
    V.ws_bytes = gsgen_frame_workspace_bytes(N, D_cap, T);
    V.ws = dalloc<uint8_t>(V.ws_bytes);
    V.out = dalloc<float>(3 * (size_t)W * H);
    std::vector<float> go(3 * (size_t)W * H);
    for (float &g : go) g = (float)rng.normal();
    V.go = dupload(go);
    V.seg_ws = nseg > 1 ? (void *)dalloc<uint8_t>(gsgen_segment_workspace_bytes(T, nseg)) : nullptr;
    GSCHECK(gsgen_frame_geometry(N, d_mean, d_qvec, d_svec, V.cam, W, H, D_cap, V.mean2d, V.cov2d, V.depth, V.mask, V.ids,
                                 V.start, V.end, V.total, V.ws, V.ws_bytes, s));
    HIPCHECK(hipMemsetAsync(V.out, 0, 3 * (size_t)W * H * sizeof(float), s));
    GSCHECK(gsgen_vol_render_sh_segmented(N, D_cap, V.mean2d, V.cov2d, d_sh, d_alpha, V.start, V.end, V.ids, V.out, V.topleft,
                                          V.rot, 16, nth, ntw, 1.0f / V.fx, 1.0f / V.fx, H, W, C, 1e-4f, nullptr, nullptr,
                                          gsgen_frame_tile_order(V.ws, N, D_cap, T), V.seg_ws, nseg, s));
  }
  HIPCHECK(hipStreamSynchronize(s));
  uint32_t total0 = 0;
  HIPCHECK(hipMemcpy(&total0, views[0].total, 4, hipMemcpyDeviceToHost));
  if (total0 > D_cap || total0 == 0) { fprintf(stderr, "pair list: %u of %u\n", total0, D_cap); return 2; }

  // gradient block of one launch: per view mean2d(2N) cov2d(4N) | shared sh(CC3 N) alpha(N)
  const size_t per_view = 6 * (size_t)N, shared = (size_t)N * (CC3 + 1), G = nv * per_view + shared;
  float *d_g = dalloc<float>(G * (size_t)(R + 2));  // R test launches + 2 reference launches
  void *bws = dalloc<uint8_t>(gsgen_sh_batch_workspace_bytes(nv));
  int *d_sink = dalloc<int>(1);
  if (poison) HIPCHECK(hipFuncSetAttribute((const void *)k_poison_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));

  auto launch = [&](float *g) {
    HIPCHECK(hipMemsetAsync(g, 0, G * sizeof(float), s));
    if (poison) {
      hipLaunchKernelGGL(k_poison_lds, dim3(1024), dim3(1024), 160 * 1024, s, d_sink);
      HIPCHECK(hipGetLastError());
    }
    if (poison_regs) {
      hipLaunchKernelGGL(k_poison_regs, dim3(8192), dim3(64), 0, s, d_sink);
      HIPCHECK(hipGetLastError());
    }
    float *g_sh = g + nv * per_view, *g_alpha = g_sh + (size_t)N * CC3;
    if (B > 0) {
      std::vector<gsgen_sh_view> sv(nv);
      for (int v = 0; v < nv; ++v) {
        const View &V = views[v];
        gsgen_sh_view &x = sv[v];
        memset(&x, 0, sizeof x);
        x.mean = V.mean2d; x.cov = V.cov2d; x.start = V.start; x.end = V.end; x.gaussian_ids = V.ids;
        x.tile_order = gsgen_frame_tile_order(V.ws, N, D_cap, T);
        x.topleft = V.topleft; x.c2w = V.rot; x.bg_rgb = nullptr;
        x.pixel_size_x = x.pixel_size_y = 1.0f / V.fx;
        x.out = V.out; x.T = nullptr; x.segment_workspace = V.seg_ws; x.grad_out = V.go;
        x.grad_mean = g + v * per_view; x.grad_cov = g + v * per_view + 2 * (size_t)N;
      }
      GSCHECK(gsgen_vol_render_backward_sh_batch(nv, sv.data(), N, d_sh, d_alpha, g_sh, g_alpha, 16, nth, ntw, H, W, C, 1e-4f,
                                                 nseg, bws, s));
    } else {
      const View &V = views[0];
      GSCHECK(gsgen_vol_render_backward_sh_segmented(N, D_cap, V.mean2d, V.cov2d, d_sh, d_alpha, V.start, V.end, V.ids, V.out, g,
                                                     g + 2 * (size_t)N, g_sh, g_alpha, V.go, V.topleft, V.rot, 16, nth, ntw,
                                                     1.0f / V.fx, 1.0f / V.fx, H, W, C, 1e-4f, nullptr,
                                                     gsgen_frame_tile_order(V.ws, N, D_cap, T), V.seg_ws, nseg, s));
    }
  };
  auto set_variant = [&](bool test) {
    GSCHECK(gsgen_debug_set_variant(B > 0 ? "mfma_batch" : "mfma", test ? mfma : 0));
    // the reference is always the vector kernel with one wavefront per tile
    GSCHECK(gsgen_debug_set_variant(B > 0 ? "ppl_bwd_sh_batch" : "ppl_bwd", (test && ppl) ? ppl : 4));
  };
  auto run_refs = [&]() { set_variant(false); launch(d_g + G * R); launch(d_g + G * (R + 1)); };
  auto run_tests = [&]() { set_variant(true); for (int r = 0; r < R; ++r) launch(d_g + G * r); };
  char kname[192] = "";
  set_variant(true);
  gsgen_kernel_variant(B > 0 ? "sh_bwd_batch" : "sh_bwd", C, nseg, kname, sizeof kname);
  if (test_first) { run_tests(); run_refs(); } else { run_refs(); run_tests(); }
  HIPCHECK(hipStreamSynchronize(s));

  std::vector<float> h(G * (size_t)(R + 2));
  HIPCHECK(hipMemcpy(h.data(), d_g, h.size() * sizeof(float), hipMemcpyDeviceToHost));
  const float *ref = h.data() + G * R, *ref2 = h.data() + G * (R + 1);
  // output groups: mean2d, cov2d (all views), sh, alpha
  struct Grp { const char *name; size_t lo, hi; } grp[4] = {{"mean2d", 0, 0}, {"cov2d", 0, 0}, {"sh", nv * per_view, nv * per_view + (size_t)N * CC3}, {"alpha", nv * per_view + (size_t)N * CC3, G}};
  auto grp_of = [&](size_t i) { if (i >= nv * per_view) return i < grp[2].hi ? 2 : 3; return (i % per_view) < 2 * (size_t)N ? 0 : 1; };
  double scale[4] = {0, 0, 0, 0};
  size_t nonfinite_ref = 0;
  for (size_t i = 0; i < G; ++i) {
    if (!isfinite(ref[i])) { ++nonfinite_ref; continue; }
    const int g = grp_of(i);
    if (fabs((double)ref[i]) > scale[g]) scale[g] = fabs((double)ref[i]);
  }
  double noise = 0;  // reference vs reference: atomics-order noise of the vector kernel itself
  for (size_t i = 0; i < G; ++i) {
    const double d = fabs((double)ref[i] - (double)ref2[i]) / (scale[grp_of(i)] + 1e-30);
    if (d > noise) noise = d;
  }
  std::vector<float> h_m2;
  if (dump_bad > 0) {
    h_m2.resize(2 * (size_t)N);
    HIPCHECK(hipMemcpy(h_m2.data(), views[0].mean2d, h_m2.size() * sizeof(float), hipMemcpyDeviceToHost));
  }
  int bad_launches = 0;
  std::string detail;
  double worst = 0;
  for (int r = 0; r < R; ++r) {
    const float *x = h.data() + G * r;
    double w[4] = {0, 0, 0, 0};
    size_t nbad = 0, first_bad = 0, nonfinite = 0;
    for (size_t i = 0; i < G; ++i) {
      if (!isfinite(x[i])) { ++nonfinite; if (!nbad++) first_bad = i; continue; }
      const int g = grp_of(i);
      const double d = fabs((double)x[i] - (double)ref[i]) / (scale[g] + 1e-30);
      if (d > w[g]) w[g] = d;
      if (d > tol && !nbad++) first_bad = i;
    }
    const double wl = fmax(fmax(w[0], w[1]), fmax(w[2], w[3]));
    if (wl > worst) worst = wl;
    std::string dump;
    if (nbad && dump_bad > 0) {  // Gaussians of view 0 whose mean2d gradient is off: id @ tile (x, y) : relative error
      int shown = 0;
      for (size_t gi = 0; gi < (size_t)N && shown < dump_bad; ++gi) {
        double e = 0;
        for (int k = 0; k < 2; ++k) e = fmax(e, fabs((double)x[2 * gi + k] - (double)ref[2 * gi + k]) / (scale[0] + 1e-30));
        if (!(e > tol)) continue;
        const float fx = views[0].fx;
        const int tx = (int)floorf((h_m2[2 * gi] * fx + size / 2.0f) / 16.0f), ty = (int)floorf((h_m2[2 * gi + 1] * fx + size / 2.0f) / 16.0f);
        char b2[96];
        snprintf(b2, sizeof b2, "%s\"%zu@%d,%d:%.1e(%.3g vs %.3g)\"", shown ? "," : "", gi, tx, ty, e, x[2 * gi], ref[2 * gi]);
        dump += b2;
        ++shown;
      }
    }
    if (nbad) {
      ++bad_launches;
      char buf[256];
      const int g = grp_of(first_bad);
      const size_t gi = g < 2 ? ((first_bad % per_view) - (g == 1 ? 2 * (size_t)N : 0)) / (g == 0 ? 2 : 4)
                              : (g == 2 ? (first_bad - grp[2].lo) / CC3 : first_bad - grp[3].lo);
      snprintf(buf, sizeof buf, "%s{\"launch\":%d,\"bad\":%zu,\"nonfinite\":%zu,\"first\":\"%s[%zu]\",\"rel\":[%.2e,%.2e,%.2e,%.2e]}",
               detail.empty() ? "" : ",", r, nbad, nonfinite, grp[g].name, gi, w[0], w[1], w[2], w[3]);
      detail += buf;
      if (!dump.empty()) { detail.pop_back(); detail += ",\"gaussians\":[" + dump + "]}"; }
    }
  }
  printf("{\"kernel\":\"%s\",\"views\":%d,\"C\":%d,\"N\":%d,\"size\":%d,\"pairs\":%u,\"launches\":%d,\"test_first\":%s,\"poison\":\"%s\","
         "\"ref_noise\":%.2e,\"ref_nonfinite\":%zu,\"worst\":%.2e,\"bad_launches\":%d,\"detail\":[%s]}\n",
         kname, nv, C, N, size, total0, R, test_first ? "true" : "false", poison ? (poison_regs ? "lds+regs" : "lds") : (poison_regs ? "regs" : "none"), noise, nonfinite_ref, worst,
         bad_launches, quiet ? "" : detail.c_str());
  return (bad_launches || nonfinite_ref) ? 1 : 0;
}
