// torch_gs.cpp -- the `_gs` CPython extension module: torch::Tensor in, C ABI of libgsgen_hip.so underneath.
//
// Replaces the reference's pybind module (gs/src/bindings.cpp:5-82, prototypes gs/src/render.h:3-155): the same 23
// names, argument orders, in-place output semantics and precondition errors (the CHECK_* macros of
// gs/src/include/common.h:29-54 -> c10::Error -> Python RuntimeError).  Every wrapper unpacks its tensors to raw
// device pointers and calls the matching gsgen_* entry point of include/gsgen_hip.h on torch's CURRENT stream of the
// tensors' device (the reference uses the legacy default stream for most entry points); temporaries come from
// torch's caching allocator; a HIP failure raises instead of exit()ing.  No device code in this file (g++ builds
// it); no CPU path: tensors must live on the GPU.
//
// Built in-tree by gsgen_amd.build.build_torch_ext() as gsgen_amd/ext/_gs.<abi>.so, so that `import _gs` (what
// gs/renderer.py:20-24 does) finds it with gsgen_amd/ext on sys.path, or through gsgen_amd.install_as_gs().
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/extension.h>

#include "../../include/gsgen_hip.h"

using torch::Tensor;

namespace {

// ---- the reference's CHECK_* macros (gs/src/include/common.h:29-54) --------------------------------------------
void check_dc(const Tensor &x, const char *name, c10::ScalarType dtype, const char *what) {
  TORCH_CHECK(x.is_cuda(), name, " must be a CUDA tensor");
  TORCH_CHECK(x.is_contiguous(), name, " must be a contiguous tensor");
  TORCH_CHECK(x.scalar_type() == dtype, name, " must be ", what, " tensor");
}
#define CHECK_F(x) check_dc(x, #x, at::kFloat, "a floating")
#define CHECK_I(x) check_dc(x, #x, at::kInt, "an int")
#define CHECK_B(x) check_dc(x, #x, at::kBool, "an bool")
#define CHECK_D(x) check_dc(x, #x, at::kDouble, "a double")

void check_status(int rc, const char *fn) {
  TORCH_CHECK(rc == 0, fn, " failed: ", gsgen_error_string(rc), " (code ", rc, ")");
}
#define GS(call) check_status((call), #call)

struct Ctx {  // device guard + current stream of the call (the reference installs no guard: SURVEY.md 8b)
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard;
  gsgen_stream_t stream;
  explicit Ctx(const Tensor &t) : guard(t.device()) {
    stream = (gsgen_stream_t)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream();
  }
};

inline const float *F(const Tensor &t) { return t.data_ptr<float>(); }
inline float *Fm(Tensor &t) { return t.data_ptr<float>(); }
inline const int *I(const Tensor &t) { return t.data_ptr<int>(); }
inline int *Im(Tensor &t) { return t.data_ptr<int>(); }

// ---- live entry points --------------------------------------------------------------------------------------------
// render.h:3 / render.cu:16-44
void culling_gaussian_bsphere(Tensor mean, Tensor qvec, Tensor svec, Tensor normal, Tensor pts, Tensor mask, float thresh) {
  CHECK_F(mean); CHECK_F(qvec); CHECK_F(svec); CHECK_F(normal); CHECK_F(pts); CHECK_B(mask);
  Ctx c(mean);
  GS(gsgen_culling_gaussian_bsphere((uint32_t)mean.size(0), F(mean), F(qvec), F(svec), F(normal), F(pts),
                                    reinterpret_cast<uint8_t *>(mask.data_ptr<bool>()), thresh, c.stream));
}

// render.h:61 / render.cu:381-398
void tile_culling_aabb_start_end(Tensor aabb_topleft, Tensor aabb_bottomright, Tensor gaussian_ids, Tensor start,
                                 Tensor end, Tensor depth, uint32_t n_tiles_h, uint32_t n_tiles_w) {
  CHECK_I(aabb_topleft); CHECK_I(aabb_bottomright); CHECK_I(gaussian_ids); CHECK_I(start); CHECK_I(end); CHECK_F(depth);
  const uint32_t N = (uint32_t)aabb_topleft.size(0), D = (uint32_t)gaussian_ids.size(0);
  Ctx c(depth);
  const size_t nbytes = gsgen_tile_culling_workspace_bytes(N, D, n_tiles_h * n_tiles_w);
  Tensor ws = torch::empty({(int64_t)nbytes}, depth.options().dtype(at::kByte));
  GS(gsgen_tile_culling_aabb_start_end(N, D, n_tiles_h, n_tiles_w, I(aabb_topleft), I(aabb_bottomright), F(depth),
                                       Im(gaussian_ids), Im(start), Im(end), ws.data_ptr(), nbytes, c.stream));
}

void rgb_forward(const Tensor &mean, const Tensor &cov, const Tensor &color, const Tensor &alpha, const Tensor &start,
                 const Tensor &end, const Tensor &gaussian_ids, Tensor &out, const Tensor &topleft, uint32_t tile_size,
                 uint32_t n_tiles_h, uint32_t n_tiles_w, float psx, float psy, uint32_t H, uint32_t W, float thresh,
                 float *T) {
  CHECK_F(mean); CHECK_F(cov); CHECK_F(color); CHECK_F(alpha); CHECK_I(start); CHECK_I(end); CHECK_I(gaussian_ids);
  CHECK_F(out); CHECK_F(topleft);
  Ctx c(mean);
  GS(gsgen_vol_render_start_end_with_T((uint32_t)mean.size(0), (uint32_t)gaussian_ids.size(0), F(mean), F(cov), F(color),
                                       F(alpha), I(start), I(end), I(gaussian_ids), Fm(out), F(topleft), tile_size,
                                       n_tiles_h, n_tiles_w, psx, psy, H, W, thresh, T, c.stream));
}
// render.h:149 / render.cu:989-1012
void tile_based_vol_rendering_start_end_with_T(Tensor mean, Tensor cov, Tensor color, Tensor alpha, Tensor start, Tensor end,
                                               Tensor gaussian_ids, Tensor out, Tensor topleft, uint32_t tile_size,
                                               uint32_t n_tiles_h, uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y,
                                               uint32_t H, uint32_t W, float thresh, Tensor T) {
  CHECK_F(T);
  rgb_forward(mean, cov, color, alpha, start, end, gaussian_ids, out, topleft, tile_size, n_tiles_h, n_tiles_w, pixel_size_x,
              pixel_size_y, H, W, thresh, Fm(T));
}
// render.h:65 / render.cu:400-424
void tile_based_vol_rendering_start_end(Tensor mean, Tensor cov, Tensor color, Tensor alpha, Tensor start, Tensor end,
                                        Tensor gaussian_ids, Tensor out, Tensor topleft, uint32_t tile_size,
                                        uint32_t n_tiles_h, uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y,
                                        uint32_t H, uint32_t W, float thresh) {
  rgb_forward(mean, cov, color, alpha, start, end, gaussian_ids, out, topleft, tile_size, n_tiles_h, n_tiles_w, pixel_size_x,
              pixel_size_y, H, W, thresh, nullptr);
}
// render.h:73 / render.cu:426-482
void tile_based_vol_rendering_backward_start_end(Tensor mean, Tensor cov, Tensor color, Tensor alpha, Tensor start, Tensor end,
                                                 Tensor gaussian_ids, Tensor out, Tensor grad_mean, Tensor grad_cov,
                                                 Tensor grad_color, Tensor grad_alpha, Tensor grad_out, Tensor topleft,
                                                 uint32_t tile_size, uint32_t n_tiles_h, uint32_t n_tiles_w,
                                                 float pixel_size_x, float pixel_size_y, uint32_t H, uint32_t W,
                                                 float thresh) {
  CHECK_F(mean); CHECK_F(cov); CHECK_F(color); CHECK_F(alpha); CHECK_I(start); CHECK_I(end); CHECK_I(gaussian_ids);
  CHECK_F(out); CHECK_F(topleft); CHECK_F(grad_mean); CHECK_F(grad_cov); CHECK_F(grad_color); CHECK_F(grad_alpha);
  CHECK_F(grad_out);
  Ctx c(mean);
  GS(gsgen_vol_render_backward_start_end((uint32_t)mean.size(0), (uint32_t)gaussian_ids.size(0), F(mean), F(cov), F(color),
                                         F(alpha), I(start), I(end), I(gaussian_ids), F(out), Fm(grad_mean), Fm(grad_cov),
                                         Fm(grad_color), Fm(grad_alpha), F(grad_out), F(topleft), tile_size, n_tiles_h,
                                         n_tiles_w, pixel_size_x, pixel_size_y, H, W, thresh, c.stream));
}

// render.h:131 / render.cu:928-956
void tile_based_vol_rendering_scalar(Tensor mean, Tensor cov, Tensor scalar, Tensor alpha, Tensor start, Tensor end,
                                     Tensor gaussian_ids, Tensor out, Tensor topleft, uint32_t tile_size, uint32_t n_tiles_h,
                                     uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y, uint32_t H, uint32_t W,
                                     float thresh, Tensor T) {
  CHECK_F(mean); CHECK_F(cov); CHECK_F(scalar); CHECK_F(alpha); CHECK_I(start); CHECK_I(end); CHECK_I(gaussian_ids);
  CHECK_F(out); CHECK_F(topleft); CHECK_F(T);
  Ctx c(mean);
  GS(gsgen_vol_render_scalar((uint32_t)mean.size(0), (uint32_t)gaussian_ids.size(0), F(mean), F(cov), F(scalar), F(alpha),
                             I(start), I(end), I(gaussian_ids), Fm(out), F(topleft), tile_size, n_tiles_h, n_tiles_w,
                             pixel_size_x, pixel_size_y, H, W, thresh, Fm(T), c.stream));
}
// render.h:139 / render.cu:958-987
void tile_based_vol_rendering_scalar_backward(Tensor mean, Tensor cov, Tensor scalar, Tensor alpha, Tensor start, Tensor end,
                                              Tensor gaussian_ids, Tensor out, Tensor grad_mean, Tensor grad_cov,
                                              Tensor grad_scalar, Tensor grad_alpha, Tensor grad_out, Tensor topleft,
                                              uint32_t tile_size, uint32_t n_tiles_h, uint32_t n_tiles_w, float pixel_size_x,
                                              float pixel_size_y, uint32_t H, uint32_t W, float thresh) {
  CHECK_F(mean); CHECK_F(cov); CHECK_F(scalar); CHECK_F(alpha); CHECK_I(start); CHECK_I(end); CHECK_I(gaussian_ids);
  CHECK_F(out); CHECK_F(topleft); CHECK_F(grad_mean); CHECK_F(grad_cov); CHECK_F(grad_scalar); CHECK_F(grad_alpha);
  CHECK_F(grad_out);
  Ctx c(mean);
  GS(gsgen_vol_render_scalar_backward((uint32_t)mean.size(0), (uint32_t)gaussian_ids.size(0), F(mean), F(cov), F(scalar),
                                      F(alpha), I(start), I(end), I(gaussian_ids), F(out), Fm(grad_mean), Fm(grad_cov),
                                      Fm(grad_scalar), Fm(grad_alpha), F(grad_out), F(topleft), tile_size, n_tiles_h,
                                      n_tiles_w, pixel_size_x, pixel_size_y, H, W, thresh, c.stream));
}

// The reference's SH entry points cannot say which form of the per-pixel SH basis to use: a module-level switch
// (set_sh_basis("auto" | "exact"), INTEGRATION.md "SH basis").  exact = the reference's per-pixel basis, always.
bool g_sh_exact = false;
// The per-splat bounds S_i = max_c sum_{k >= 1} |sh[i][c][k]| of the call's coefficients into a scratch tensor [N + 1] (their
// maximum behind them) from torch's caching allocator (stream-ordered: the block is reused only behind the launches that read
// it); the SH kernels route PER TILE on them (include/gsgen_hip.h "per-TILE routing").  Undefined tensor = exact kernels (other
// degrees / tile sizes, or the switch).  Forward AND backward measure (one ~10-us pass each): round 4 kept the forward's tensor in
// a process-global cache keyed on storage and version for the backward -- unsynchronised, blind to `.data` edits, destroyed after
// the allocator at exit; the same coefficients give the same bounds, so nothing is lost but the pass.
template <class C_>
Tensor sh_bound(const Tensor &sh_coeffs, uint32_t C, uint32_t tile_size, C_ &c) {
  if (g_sh_exact || C != 4 || tile_size != 16 || sh_coeffs.numel() == 0) return Tensor();
  const int64_t n = sh_coeffs.numel() / 48;
  Tensor b = at::empty({n + 1}, sh_coeffs.options());
  GS(gsgen_sh_l1_bound_rows((uint32_t)n, F(sh_coeffs), C, Fm(b) + n, Fm(b), c.stream));
  return b;
}

void sh_forward(const Tensor &mean, const Tensor &cov, const Tensor &sh_coeffs, const Tensor &alpha, const Tensor &start,
                const Tensor &end, const Tensor &gaussian_ids, Tensor &out, const Tensor &topleft, const Tensor &c2w,
                uint32_t tile_size, uint32_t n_tiles_h, uint32_t n_tiles_w, float psx, float psy, uint32_t H, uint32_t W,
                uint32_t C, float thresh, const float *bg) {
  CHECK_F(mean); CHECK_F(cov); CHECK_F(sh_coeffs); CHECK_F(alpha); CHECK_I(start); CHECK_I(end); CHECK_I(gaussian_ids);
  CHECK_F(out); CHECK_F(topleft); CHECK_F(c2w);
  if (C < 1 || C > 4) return;  // the reference dispatches C = 1..4 and silently does nothing otherwise (render.cu:507-544)
  Ctx c(mean);
  // SH degree 3: the coefficient bound of THESE coefficients, measured on the device in front of the launch (one 5-us pass,
  // no sync); the kernels route on it -- polynomial form of the per-pixel basis where its error bound holds, else exact
  const Tensor bound_t = sh_bound(sh_coeffs, C, tile_size, c);  // (held until the launch is enqueued)
  const float *bound = bound_t.defined() ? F(bound_t) : nullptr;
  GS(gsgen_vol_render_sh_routed((uint32_t)mean.size(0), (uint32_t)gaussian_ids.size(0), F(mean), F(cov), F(sh_coeffs), F(alpha),
                                I(start), I(end), I(gaussian_ids), Fm(out), F(topleft), F(c2w), tile_size, n_tiles_h, n_tiles_w,
                                psx, psy, H, W, C, thresh, bg, nullptr, nullptr, nullptr, 0, bound ? bound + sh_coeffs.numel() / 48 : nullptr, bound,
                                c.stream));
}
void sh_backward(const Tensor &mean, const Tensor &cov, const Tensor &sh_coeffs, const Tensor &alpha, const Tensor &start,
                 const Tensor &end, const Tensor &gaussian_ids, const Tensor &out, Tensor &grad_mean, Tensor &grad_cov,
                 Tensor &grad_sh_coeffs, Tensor &grad_alpha, const Tensor &grad_out, const Tensor &topleft, const Tensor &c2w,
                 uint32_t tile_size, uint32_t n_tiles_h, uint32_t n_tiles_w, float psx, float psy, uint32_t H, uint32_t W,
                 uint32_t C, float thresh, const float *bg) {
  CHECK_F(mean); CHECK_F(cov); CHECK_F(sh_coeffs); CHECK_F(alpha); CHECK_I(start); CHECK_I(end); CHECK_I(gaussian_ids);
  CHECK_F(out); CHECK_F(topleft); CHECK_F(c2w); CHECK_F(grad_mean); CHECK_F(grad_cov); CHECK_F(grad_sh_coeffs);
  CHECK_F(grad_alpha); CHECK_F(grad_out);
  if (C < 1 || C > 4) return;
  Ctx c(mean);
  // the same coefficients, the same bounds, the same routing as this frame's forward
  const Tensor bound_t = sh_bound(sh_coeffs, C, tile_size, c);
  const float *bound = bound_t.defined() ? F(bound_t) : nullptr;
  GS(gsgen_vol_render_backward_sh_routed((uint32_t)mean.size(0), (uint32_t)gaussian_ids.size(0), F(mean), F(cov), F(sh_coeffs),
                                         F(alpha), I(start), I(end), I(gaussian_ids), F(out), Fm(grad_mean), Fm(grad_cov),
                                         Fm(grad_sh_coeffs), Fm(grad_alpha), F(grad_out), F(topleft), F(c2w), tile_size,
                                         n_tiles_h, n_tiles_w, psx, psy, H, W, C, thresh, bg, nullptr, nullptr, 0,
                                         bound ? bound + sh_coeffs.numel() / 48 : nullptr, bound, c.stream));
}
// render.h:83 / render.cu:484-545
void tile_based_vol_rendering_sh(Tensor mean, Tensor cov, Tensor sh_coeffs, Tensor alpha, Tensor start, Tensor end,
                                 Tensor gaussian_ids, Tensor out, Tensor topleft, Tensor c2w, uint32_t tile_size,
                                 uint32_t n_tiles_h, uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y, uint32_t H,
                                 uint32_t W, uint32_t C, float thresh) {
  sh_forward(mean, cov, sh_coeffs, alpha, start, end, gaussian_ids, out, topleft, c2w, tile_size, n_tiles_h, n_tiles_w,
             pixel_size_x, pixel_size_y, H, W, C, thresh, nullptr);
}
// render.h:91 / render.cu:547-625 (and the experimental variants render.h:99, :106: same result)
void tile_based_vol_rendering_backward_sh(Tensor mean, Tensor cov, Tensor sh_coeffs, Tensor alpha, Tensor start, Tensor end,
                                          Tensor gaussian_ids, Tensor out, Tensor grad_mean, Tensor grad_cov,
                                          Tensor grad_sh_coeffs, Tensor grad_alpha, Tensor grad_out, Tensor topleft,
                                          Tensor c2w, uint32_t tile_size, uint32_t n_tiles_h, uint32_t n_tiles_w,
                                          float pixel_size_x, float pixel_size_y, uint32_t H, uint32_t W, uint32_t C,
                                          float thresh) {
  sh_backward(mean, cov, sh_coeffs, alpha, start, end, gaussian_ids, out, grad_mean, grad_cov, grad_sh_coeffs, grad_alpha,
              grad_out, topleft, c2w, tile_size, n_tiles_h, n_tiles_w, pixel_size_x, pixel_size_y, H, W, C, thresh, nullptr);
}
// render.h:113 / render.cu:781-845
void tile_based_vol_rendering_sh_with_bg(Tensor mean, Tensor cov, Tensor sh_coeffs, Tensor alpha, Tensor start, Tensor end,
                                         Tensor gaussian_ids, Tensor out, Tensor topleft, Tensor c2w, uint32_t tile_size,
                                         uint32_t n_tiles_h, uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y,
                                         uint32_t H, uint32_t W, uint32_t C, float thresh, Tensor bg_rgb) {
  CHECK_F(bg_rgb);
  sh_forward(mean, cov, sh_coeffs, alpha, start, end, gaussian_ids, out, topleft, c2w, tile_size, n_tiles_h, n_tiles_w,
             pixel_size_x, pixel_size_y, H, W, C, thresh, F(bg_rgb));
}
// render.h:121 / render.cu:847-926
void tile_based_vol_rendering_backward_sh_with_bg(Tensor mean, Tensor cov, Tensor sh_coeffs, Tensor alpha, Tensor start,
                                                  Tensor end, Tensor gaussian_ids, Tensor out, Tensor grad_mean,
                                                  Tensor grad_cov, Tensor grad_sh_coeffs, Tensor grad_alpha, Tensor grad_out,
                                                  Tensor topleft, Tensor c2w, uint32_t tile_size, uint32_t n_tiles_h,
                                                  uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y, uint32_t H,
                                                  uint32_t W, uint32_t C, float thresh, Tensor bg_rgb) {
  CHECK_F(bg_rgb);
  sh_backward(mean, cov, sh_coeffs, alpha, start, end, gaussian_ids, out, grad_mean, grad_cov, grad_sh_coeffs, grad_alpha,
              grad_out, topleft, c2w, tile_size, n_tiles_h, n_tiles_w, pixel_size_x, pixel_size_y, H, W, C, thresh, F(bg_rgb));
}

// ---- compat entry points (gs/renderer.py:GaussianRenderer, gs/debug.py, gs/benchmarks.py) ---------------------------
// CSR `offset[T+1]` forms of the RGB forward / backward (render.h:24-54; v1 / v2 keep different things on chip: same result)
void tile_based_vol_rendering(Tensor mean, Tensor cov, Tensor color, Tensor alpha, Tensor offset, Tensor gaussian_ids,
                              Tensor out, Tensor topleft, uint32_t tile_size, uint32_t n_tiles_h, uint32_t n_tiles_w,
                              float pixel_size_x, float pixel_size_y, uint32_t H, uint32_t W, float thresh) {
  CHECK_I(offset);
  Tensor start = offset.slice(0, 0, offset.size(0) - 1).contiguous(), end = offset.slice(0, 1).contiguous();
  rgb_forward(mean, cov, color, alpha, start, end, gaussian_ids, out, topleft, tile_size, n_tiles_h, n_tiles_w, pixel_size_x,
              pixel_size_y, H, W, thresh, nullptr);
}
void tile_based_vol_rendering_backward(Tensor mean, Tensor cov, Tensor color, Tensor alpha, Tensor offset, Tensor gaussian_ids,
                                       Tensor out, Tensor grad_mean, Tensor grad_cov, Tensor grad_color, Tensor grad_alpha,
                                       Tensor grad_out, Tensor topleft, uint32_t tile_size, uint32_t n_tiles_h,
                                       uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y, uint32_t H, uint32_t W,
                                       float thresh) {
  CHECK_I(offset);
  Tensor start = offset.slice(0, 0, offset.size(0) - 1).contiguous(), end = offset.slice(0, 1).contiguous();
  tile_based_vol_rendering_backward_start_end(mean, cov, color, alpha, start, end, gaussian_ids, out, grad_mean, grad_cov,
                                              grad_color, grad_alpha, grad_out, topleft, tile_size, n_tiles_h, n_tiles_w,
                                              pixel_size_x, pixel_size_y, H, W, thresh);
}
// render.h:56 / render.cu:363-379: as tile_culling_aabb_start_end with a CSR offset; empty tiles get the next tile's
// offset (a valid CSR; the reference leaves them at -1, aabb_culling.h:180-184)
void tile_culling_aabb(Tensor aabb_topleft, Tensor aabb_bottomright, Tensor gaussian_ids, Tensor offset, Tensor depth,
                       uint32_t n_tiles_h, uint32_t n_tiles_w) {
  CHECK_I(offset);
  const int64_t T = (int64_t)n_tiles_h * n_tiles_w;
  TORCH_CHECK(offset.numel() == T + 1, "offset must have n_tiles_h * n_tiles_w + 1 entries");
  Tensor start = torch::empty({T}, offset.options()), end = torch::empty({T}, offset.options());
  tile_culling_aabb_start_end(aabb_topleft, aabb_bottomright, gaussian_ids, start, end, depth, n_tiles_h, n_tiles_w);
  Tensor counts = torch::where(start >= 0, end - start, torch::zeros_like(start));
  offset.slice(0, 0, 1).zero_();
  offset.slice(0, 1).copy_(torch::cumsum(counts, 0).to(at::kInt));
}
// debug.h: (DEBUG) every tile's keys must be sorted by depth
void debug_check_tiledepth(Tensor offset, Tensor tiledepth) {
  CHECK_I(offset); CHECK_D(tiledepth);
  Tensor off = offset.to(at::kCPU), td = tiledepth.to(at::kCPU);
  const int *o = off.data_ptr<int>();
  const uint64_t *k = reinterpret_cast<const uint64_t *>(td.data_ptr<double>());
  for (int64_t t = 0; t + 1 < off.numel(); ++t)
    for (int i = o[t] + 1; i < o[t + 1]; ++i) {
      const float a = *reinterpret_cast<const float *>(&k[i - 1]), b = *reinterpret_cast<const float *>(&k[i]);
      TORCH_CHECK(!(b < a), "tile ", t, ": depth keys out of order at ", i);
    }
}
// older binning pipeline (render.h:7-22 / tile_ops.h): count, then fill + sort
void legacy_count(uint32_t mode, const Tensor &mean, const Tensor &shape, const Tensor &topleft, uint32_t tile_size,
                  uint32_t n_tiles_h, uint32_t n_tiles_w, float psx, float psy, float thresh, Tensor &num_gaussians) {
  CHECK_F(mean); CHECK_F(shape); CHECK_F(topleft); CHECK_I(num_gaussians);
  Ctx c(mean);
  GS(gsgen_legacy_count_tiles(mode, (uint32_t)mean.size(0), F(mean), F(shape), F(topleft), tile_size, n_tiles_h, n_tiles_w,
                              psx, psy, thresh, Im(num_gaussians), c.stream));
}
void count_num_gaussians_each_tile(Tensor mean, Tensor cov_inv, Tensor topleft, uint32_t tile_size, uint32_t n_tiles_h,
                                   uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y, Tensor num_gaussians,
                                   float thresh) {
  legacy_count(0, mean, cov_inv, topleft, tile_size, n_tiles_h, n_tiles_w, pixel_size_x, pixel_size_y, thresh, num_gaussians);
}
void count_num_gaussians_each_tile_bcircle(Tensor mean, Tensor radius, Tensor topleft, uint32_t tile_size, uint32_t n_tiles_h,
                                           uint32_t n_tiles_w, float pixel_size_x, float pixel_size_y, Tensor num_gaussians) {
  legacy_count(1, mean, radius, topleft, tile_size, n_tiles_h, n_tiles_w, pixel_size_x, pixel_size_y, 0.0f, num_gaussians);
}
void legacy_sort(uint32_t mode, Tensor &gaussian_ids, Tensor &tiledepth, const Tensor &depth, Tensor &tile_n_gaussians,
                 Tensor &offset, const Tensor &mean, const Tensor &shape, const Tensor &topleft, uint32_t tile_size,
                 uint32_t n_tiles_h, uint32_t n_tiles_w, float psx, float psy, float thresh) {
  CHECK_I(gaussian_ids); CHECK_D(tiledepth); CHECK_F(depth); CHECK_I(tile_n_gaussians); CHECK_I(offset); CHECK_F(mean);
  CHECK_F(shape); CHECK_F(topleft);
  const uint32_t N = (uint32_t)mean.size(0), D = (uint32_t)tiledepth.size(0), T = n_tiles_h * n_tiles_w;
  Ctx c(mean);
  const size_t nbytes = gsgen_legacy_sort_workspace_bytes(D, T);
  Tensor ws = torch::empty({(int64_t)nbytes}, mean.options().dtype(at::kByte));
  GS(gsgen_legacy_image_sort(mode, N, D, Im(gaussian_ids), reinterpret_cast<unsigned long long *>(tiledepth.data_ptr<double>()),
                             F(depth), Im(tile_n_gaussians), Im(offset), F(mean), F(shape), F(topleft), tile_size, n_tiles_h,
                             n_tiles_w, psx, psy, thresh, ws.data_ptr(), nbytes, c.stream));
}
void prepare_image_sort(Tensor gaussian_ids, Tensor tiledepth, Tensor depth, Tensor tile_n_gaussians, Tensor offset, Tensor mean,
                        Tensor radius, Tensor topleft, uint32_t tile_size, uint32_t n_tiles_h, uint32_t n_tiles_w,
                        float pixel_size_x, float pixel_size_y) {
  legacy_sort(1, gaussian_ids, tiledepth, depth, tile_n_gaussians, offset, mean, radius, topleft, tile_size, n_tiles_h,
              n_tiles_w, pixel_size_x, pixel_size_y, 0.0f);
}
void image_sort(Tensor gaussian_ids, Tensor tiledepth, Tensor depth, Tensor tile_n_gaussians, Tensor offset, Tensor mean,
                Tensor cov, Tensor topleft, uint32_t tile_size, uint32_t n_tiles_h, uint32_t n_tiles_w, float pixel_size_x,
                float pixel_size_y, float thresh) {
  legacy_sort(0, gaussian_ids, tiledepth, depth, tile_n_gaussians, offset, mean, cov, topleft, tile_size, n_tiles_h, n_tiles_w,
              pixel_size_x, pixel_size_y, thresh);
}

}  // namespace

PYBIND11_MODULE(_gs, m) {
  m.doc() = "gsgen_amd: MI355X rasterizer behind the reference's `_gs` interface (gs/src/bindings.cpp)";
  m.def("culling_gaussian_bsphere", &culling_gaussian_bsphere, "Cull Gaussian with Bounding Sphere");
  m.def("count_num_gaussians_each_tile", &count_num_gaussians_each_tile, "Count number of gaussians in each tile");
  m.def("count_num_gaussians_each_tile_bcircle", &count_num_gaussians_each_tile_bcircle,
        "Count number of gaussians in each tile with bounding circle");
  m.def("prepare_image_sort", &prepare_image_sort, "Prepare image for sorting");
  m.def("image_sort", &image_sort, "Image sort");
  m.def("tile_based_vol_rendering", &tile_based_vol_rendering, "Tile based volume rendering");
  m.def("tile_based_vol_rendering_backward", &tile_based_vol_rendering_backward, "Tile based volume rendering backward");
  m.def("debug_check_tiledepth", &debug_check_tiledepth, "(DEBUG) check tile and depth");
  m.def("tile_culling_aabb", &tile_culling_aabb, "Tile culling with AABB");
  m.def("tile_based_vol_rendering_v1", &tile_based_vol_rendering, "Tile based volume rendering (v1: same result)");
  m.def("tile_based_vol_rendering_v2", &tile_based_vol_rendering, "Tile based volume rendering (v2: same result)");
  m.def("tile_culling_aabb_start_end", &tile_culling_aabb_start_end, "Tile culling with aabb, start and end per tile");
  m.def("tile_based_vol_rendering_start_end", &tile_based_vol_rendering_start_end,
        "Tile based volume rendering with start and end array");
  m.def("tile_based_vol_rendering_backward_start_end", &tile_based_vol_rendering_backward_start_end,
        "Tile based volume rendering backward with start and end array");
  m.def("tile_based_vol_rendering_sh", &tile_based_vol_rendering_sh, "Tile based volume rendering with spherical harmonics");
  m.def("tile_based_vol_rendering_backward_sh", &tile_based_vol_rendering_backward_sh,
        "Tile based volume rendering backward with spherical harmonics");
  m.def("tile_based_vol_rendering_backward_sh_v1", &tile_based_vol_rendering_backward_sh, "(variant: same result)");
  m.def("tile_based_vol_rendering_backward_sh_warp_reduce", &tile_based_vol_rendering_backward_sh, "(variant: same result)");
  m.def("tile_based_vol_rendering_sh_with_bg", &tile_based_vol_rendering_sh_with_bg,
        "Tile based volume rendering with spherical harmonics and background");
  m.def("tile_based_vol_rendering_backward_sh_with_bg", &tile_based_vol_rendering_backward_sh_with_bg,
        "Tile based volume rendering backward with spherical harmonics and background");
  m.def("tile_based_vol_rendering_scalar", &tile_based_vol_rendering_scalar, "Tile based volume rendering of a scalar");
  m.def("tile_based_vol_rendering_scalar_backward", &tile_based_vol_rendering_scalar_backward,
        "Tile based volume rendering backward of a scalar");
  m.def("tile_based_vol_rendering_start_end_with_T", &tile_based_vol_rendering_start_end_with_T,
        "Tile based volume rendering with start and end array, returning the transmittance");
  m.def("gsgen_version", []() { return std::string(gsgen_version()); }, "version of the HIP library underneath");
  m.def("set_sh_basis", [](const std::string &mode) {
    TORCH_CHECK(mode == "auto" || mode == "exact", "SH basis: 'auto' or 'exact'");
    g_sh_exact = mode == "exact";
  }, "SH degree 3: 'auto' (default: device-routed polynomial / exact per-pixel basis) or 'exact' (the reference's basis, always)");
  m.def("get_sh_basis", []() { return std::string(g_sh_exact ? "exact" : "auto"); });
}
