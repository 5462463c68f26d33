// torch_batch.cpp -- the camera batch as ONE C++ autograd node (module `_gsbatch`, gsgen_amd/ext/_gsbatch.<abi>.so).
//
// Replaces, on the host side, what gsgen_amd/batch.py's torch.autograd.Functions do per call: the reference's camera loop
// (gs/gaussian_splatting.py:1423-1466 over render_one :1198-1421) is one enqueue per stage here, and with 8 cameras those
// enqueues are ~50 us of C-ABI calls -- the remaining 250 us of a step's host time were Python: the Function's forward and
// backward, per-view ctypes field stores, tensor allocations through the Python API, the autograd engine re-acquiring the
// GIL for a Python node (profiles/r05_host_profile_*.txt).  Here forward and backward are
// torch::autograd::Function<> members: tensors are allocated with at::empty, the per-view pointers are written straight into
// the view tables, the C-ABI entry points of include/gsgen_hip.h are called directly, and the engine runs the backward
// without the GIL.
//
// State stays where it was: gsgen_amd.batch.BatchRenderer owns the slots, lists, workspaces, camera upload, overflow
// handling and the one-forward-one-backward generation counter; it hands this file a Plan -- the host addresses of its
// (ctypes) view tables and pointer tables, the device addresses of its workspaces, the shape -- built once per (kind, batch
// size, list capacity).  BatchRenderer takes this path when the lists are sized, `strict` and `pipeline` are off and the module
// is built; otherwise the Python Functions (same launches, same results).  No device code in this file (g++ builds it).
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/extension.h>

#include "../../include/gsgen_hip.h"

using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

namespace {

void check_status(int rc, const char *fn) {
  TORCH_CHECK(rc == 0, fn, " failed: ", gsgen_error_string(rc), " (code ", rc, ")");
}
#define GS(call) check_status((call), #call)

gsgen_stream_t current_stream(const Tensor &t) {
  return (gsgen_stream_t)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream();
}

enum Kind { kRgbd = 0, kRgb = 1, kSh = 2 };

// What BatchRenderer built for one (kind, batch size, list capacity).  Addresses are plain integers from Python: the host
// tables are ctypes arrays the renderer keeps alive (BatchRenderer._table_cache / _ptr_tabs), the device blocks are its tensors.
// What a pending backward needs alive besides the Plan's own tensors: the renderer's host tables, its generation cell, the
// slots' device buffers -- Python objects, released under the GIL wherever the last reference happens to drop (the autograd
// engine's thread included).  The renderer itself is NOT among them (no reference cycle through the plan it caches).
struct Keep {
  pybind11::object o;
  explicit Keep(pybind11::object x) : o(std::move(x)) {}
  ~Keep() {
    pybind11::gil_scoped_acquire gil;
    o = pybind11::object();
  }
};

struct Plan : std::enable_shared_from_this<Plan> {
  int kind = kRgbd;
  int64_t B = 0, N = 0, Np = 0, W = 0, H = 0, nth = 0, ntw = 0, segments = 1;
  uintptr_t geo = 0, views = 0;  // gsgen_geometry_view[B], gsgen_rgbd_view[B] | gsgen_sh_view[B]  (host)
  uintptr_t cam_tab = 0, mask_tab = 0, gmean_tab = 0, gcov_tab = 0, gchan_tab = 0, depth_tab = 0, cov2d_tab = 0, chol_tab = 0;  // void*[B] (host)
  uintptr_t gws = 0, bws = 0;    // device: the geometry launch's view table, the compositing launches' batch workspace
  uintptr_t generation = 0;      // host int64: BatchRenderer's generation counter (a later render invalidates this batch's lists)
  Tensor g2d, gch;               // the renderer's per-view gradient accumulators (zeroed again for a second backward)
  std::shared_ptr<Keep> keep;
};
// a graph node's reference to its plan (kept in the node's saved_data as a capsule): a backward may run after the renderer
// that built the plan is gone -- it then fails the generation check or runs on buffers this reference kept alive, never on
// freed memory
struct PlanRef : torch::CustomClassHolder {
  std::shared_ptr<Plan> p;
  explicit PlanRef(std::shared_ptr<Plan> x) : p(std::move(x)) {}
};
c10::IValue plan_ref(const Plan &p) {
  return c10::IValue::make_capsule(c10::make_intrusive<PlanRef>(const_cast<Plan &>(p).shared_from_this()));
}

const char *kStale =
    "gsgen_amd.BatchRenderer: another render() / render_heads() (or a regrown slot) came between this batch's forward and "
    "its backward -- the lists the backward needs are gone. Call backward before the next render, or use one BatchRenderer "
    "per batch in flight (e.g. for gradient accumulation or an evaluation render in between).";

template <class T>
T *tab(uintptr_t a) { return reinterpret_cast<T *>(a); }

// optional tensor arguments travel through Function::apply as std::optional (an UNDEFINED at::Tensor argument would be taken for
// a variable input and asked for its device)
using OptT = c10::optional<Tensor>;
Tensor opt(const OptT &t) { return (t.has_value() && t->defined()) ? *t : Tensor(); }

// The kernels take raw pointers and the Plan's N: a renderer reused after densify / prune, or a wrongly shaped statistics buffer,
// must be an error here, not an out-of-bounds access on the device.
void check_param(const Tensor &t, const char *name, int64_t N, int64_t per, const Tensor &like) {
  TORCH_CHECK(t.defined() && t.is_cuda() && t.scalar_type() == at::kFloat, "gsgen_amd: ", name, " must be a float32 CUDA tensor");
  TORCH_CHECK(t.device() == like.device(), "gsgen_amd: ", name, " is on ", t.device(), ", mean on ", like.device());
  TORCH_CHECK(t.numel() == N * per && (t.dim() == 0 || t.size(0) == N || per == 1),
              "gsgen_amd: ", name, " has ", t.numel(), " elements, the renderer was built for N = ", N, " (", N * per,
              "); build a new BatchRenderer after densify / prune");
}
void check_stats(const Tensor &t, const char *name, int64_t N, const Tensor &like) {
  if (!t.defined()) return;
  TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kFloat && t.is_contiguous() && t.device() == like.device() && t.numel() >= N,
              "gsgen_amd: ", name, " must be a contiguous float32 CUDA tensor of at least N = ", N, " elements on ", like.device());
}

Tensor bg_grad(const Tensor &g_rgb, const Tensor &T, const Tensor &bg) {
  // d / d bg of rgb = ... + T * bg (gs/renderer.py:1283: nan_to_num(grad * T)), reduced to bg's shape
  return at::nan_to_num(g_rgb * T).sum_to_size(bg.sizes());
}

// ---- rgb + depth + opacity + depth^2 (the trainer's default outputs) ---------------------------------------------------
// Round 6: the four heads are SEPARATE contiguous images (gsgen_rgbd_view::out_rgb ...: coalesced stores in the kernel, vectorised
// torch kernels for whatever the caller does with them -- [B,H,W,6] slices ran torch's strided loops at half speed), the
// background is composited by the forward's epilogue and its gradient summed by the backward's prologue (no T * bg / add_ / mul /
// nan_to_num / sum launches), and with z_var = true the sixth head is the depth variance the reference's model returns
// (gs/gaussian_splatting.py:1397) with its chain rule inside the backward launch (no mul / sub forward, no four-kernel backward).
const float *bg_pointer(const Tensor &bg, int64_t B, int64_t i, Tensor &keep) {
  if (!bg.defined()) return nullptr;
  if (bg.numel() == 3 && bg.is_contiguous()) return bg.data_ptr<float>();
  if (bg.numel() == 3 * B && bg.is_contiguous()) return bg.data_ptr<float>() + 3 * i;
  if (bg.dim() >= 1 && bg.size(-1) == 3 && bg.stride(-1) == 1 && bg.storage().nbytes() >= 12 && bg.numel() == 3 * B) {
    bool expanded = true;  // one colour expanded over the batch: every other stride 0 or over a dimension of size 1
    for (int64_t d = 0; d + 1 < bg.dim(); ++d) expanded = expanded && (bg.size(d) == 1 || bg.stride(d) == 0);
    if (expanded) return bg.data_ptr<float>();
  }
  if (!keep.defined()) keep = bg.expand({B, 1, 1, 3}).contiguous();
  return keep.data_ptr<float>() + 3 * i;
}

struct HeadsFn : public torch::autograd::Function<HeadsFn> {
  static variable_list forward(AutogradContext *ctx, Tensor mean, Tensor qvec, Tensor svec, Tensor alpha, Tensor color,
                               OptT bg_, int64_t plan_addr, double thresh, bool detach_depth, OptT max_radii2d_,
                               OptT grad_accum_, OptT cnt_, bool z_var, int64_t svec_act, int64_t alpha_act, int64_t color_act) {
    const Plan &p = *reinterpret_cast<const Plan *>(plan_addr);
    const Tensor bg = opt(bg_), max_radii2d = opt(max_radii2d_), grad_accum = opt(grad_accum_), cnt = opt(cnt_);
    check_param(mean, "mean", p.N, 3, mean); check_param(qvec, "qvec", p.N, 4, mean); check_param(svec, "svec", p.N, 3, mean);
    check_param(alpha, "alpha", p.N, 1, mean); check_param(color, "color", p.N, 3, mean);
    check_stats(max_radii2d, "max_radii2d", p.N, mean); check_stats(grad_accum, "grad_accum", p.N, mean); check_stats(cnt, "cnt", p.N, mean);
    TORCH_CHECK(p.kind == kRgbd && p.gch.defined(), "plan / call mismatch");
    TORCH_CHECK(!bg.defined() || (bg.is_cuda() && bg.scalar_type() == at::kFloat && bg.device() == mean.device()),
                "gsgen_amd: bg_rgb must be a float32 tensor on ", mean.device());
    mean = mean.contiguous(); qvec = qvec.contiguous(); svec = svec.contiguous();
    alpha = alpha.contiguous(); color = color.contiguous();
    c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(mean.device());
    const gsgen_stream_t s = current_stream(mean);
    const int64_t B = p.B, H = p.H, W = p.W, N = p.N;
    // raw parameters (svec_act >= 0: the model's svec_before_activation / alpha_before_activation / color_before_activation and its
    // activation codes, include/gsgen_hip.h gsgen_activate_fields): one launch here instead of three torch kernels and, backward,
    // three autograd nodes -- host time of a small training step, not device time
    const bool raw = svec_act >= 0;
    Tensor svec_raw, alpha_raw, color_raw;
    if (raw) {
      svec_raw = svec; alpha_raw = alpha; color_raw = color;
      Tensor act = at::empty({7 * N}, mean.options());
      svec = act.narrow(0, 0, 3 * N).view({N, 3}); color = act.narrow(0, 3 * N, 3 * N).view({N, 3}); alpha = act.narrow(0, 6 * N, N);
      GS(gsgen_activate_fields((uint32_t)N, svec_raw.data_ptr<float>(), alpha_raw.data_ptr<float>(), color_raw.data_ptr<float>(),
                               (int)svec_act, (int)alpha_act, (int)color_act, svec.data_ptr<float>(), alpha.data_ptr<float>(),
                               color.data_ptr<float>(), s));
    }
    Tensor rgb = at::empty({B, H, W, 3}, mean.options()), dep = at::empty({B, H, W, 1}, mean.options());
    Tensor opa = at::empty({B, H, W, 1}, mean.options()), zz = at::empty({B, H, W, 1}, mean.options());
    Tensor T = at::empty({B, H, W, 1}, mean.options());
    // d L / d alpha [Np], shared by the views | d L / d bg, 64 partial rows of 4 floats per view: zeroed by the projection launch
    const bool bg_grad_wanted = bg.defined() && bg.requires_grad();
    Tensor gsh = at::empty({p.Np + (bg_grad_wanted ? 256 * B : 0)}, mean.options());
    gsgen_rgbd_view *v = tab<gsgen_rgbd_view>(p.views);
    Tensor bg_keep;
    for (int64_t i = 0; i < B; ++i) {
      v[i].out6 = nullptr;
      v[i].out_rgb = rgb.data_ptr<float>() + 3 * H * W * i;
      v[i].out_depth = dep.data_ptr<float>() + H * W * i;
      v[i].out_opacity = opa.data_ptr<float>() + H * W * i;
      v[i].out_depth2 = zz.data_ptr<float>() + H * W * i;
      v[i].T = T.data_ptr<float>() + H * W * i;
      v[i].bg_rgb = bg_pointer(bg, B, i, bg_keep);
      v[i].grad_bg = bg_grad_wanted ? gsh.data_ptr<float>() + p.Np + 256 * i : nullptr;
      v[i].depth_variance = z_var ? 1u : 0u;
    }
    {  // the forward's densify statistic (max_radii2d) is raised by the projection launch itself (gsgen_geometry_view::max_radii2d)
      gsgen_geometry_view *gv = tab<gsgen_geometry_view>(p.geo);
      float *mr = max_radii2d.defined() ? max_radii2d.data_ptr<float>() : nullptr;
      for (int64_t i = 0; i < B; ++i) gv[i].max_radii2d = mr;
    }
    GS(gsgen_frame_geometry_batch_zero((uint32_t)B, tab<gsgen_geometry_view>(p.geo), (uint32_t)N, mean.data_ptr<float>(),
                                       qvec.data_ptr<float>(), svec.data_ptr<float>(), (uint32_t)W, (uint32_t)H,
                                       gsh.data_ptr<float>(), (size_t)gsh.numel(), tab<void>(p.gws), s));
    GS(gsgen_vol_render_rgbd_batch((uint32_t)B, v, (uint32_t)N, color.data_ptr<float>(), alpha.data_ptr<float>(), 16,
                                   (uint32_t)p.nth, (uint32_t)p.ntw, (uint32_t)H, (uint32_t)W, (float)thresh, tab<void>(p.bws), s));
    ctx->save_for_backward({mean, qvec, svec, alpha, color, rgb, dep, opa, zz, T, bg, bg_keep, svec_raw, alpha_raw, color_raw});
    ctx->saved_data["acts"] = std::vector<int64_t>{svec_act, alpha_act, color_act};
    ctx->saved_data["plan"] = plan_addr;
    ctx->saved_data["plan_ref"] = plan_ref(p);
    ctx->saved_data["gen"] = *tab<const int64_t>(p.generation);
    ctx->saved_data["thresh"] = thresh;
    ctx->saved_data["detach"] = detach_depth;
    ctx->saved_data["gsh"] = gsh;
    ctx->saved_data["grad_accum"] = grad_accum;
    ctx->saved_data["cnt"] = cnt;
    ctx->saved_data["bg_grad"] = bg_grad_wanted;
    ctx->mark_non_differentiable({T});
    return {rgb, dep, opa, zz, T};
  }

  static variable_list backward(AutogradContext *ctx, variable_list g) {
    const int64_t plan_addr = ctx->saved_data["plan"].toInt();
    const Plan &p = *reinterpret_cast<const Plan *>(plan_addr);
    TORCH_CHECK(*tab<const int64_t>(p.generation) == ctx->saved_data["gen"].toInt(), kStale);
    auto saved = ctx->get_saved_variables();
    const Tensor &mean = saved[0], &qvec = saved[1], &svec = saved[2], &alpha = saved[3], &color = saved[4], &bg = saved[10];
    c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(mean.device());
    const gsgen_stream_t s = current_stream(mean);
    const int64_t B = p.B, H = p.H, W = p.W, N = p.N;
    const bool bg_grad_wanted = ctx->saved_data["bg_grad"].toBool();
    Tensor parts[4];
    for (int k = 0; k < 4; ++k)
      if (g[k].defined()) parts[k] = g[k].contiguous();
    Tensor gsh = ctx->saved_data["gsh"].toTensor();
    if (ctx->saved_data.count("used")) {  // a second backward through the same graph (retain_graph): fresh accumulators
      gsh = at::zeros({gsh.numel()}, mean.options());
      p.g2d.narrow(0, 0, B).zero_();
      p.gch.narrow(0, 0, B).zero_();
    }
    ctx->saved_data["used"] = true;
    Tensor g3d = at::empty({13 * N}, mean.options());  // mean | qvec | svec | colour: overwritten
    Tensor g_mean = g3d.narrow(0, 0, 3 * N).view({N, 3}), g_qvec = g3d.narrow(0, 3 * N, 4 * N).view({N, 4});
    Tensor g_svec = g3d.narrow(0, 7 * N, 3 * N).view({N, 3}), g_col = g3d.narrow(0, 10 * N, 3 * N).view({N, 3});
    gsgen_rgbd_view *v = tab<gsgen_rgbd_view>(p.views);
    const float *pp[4];
    for (int k = 0; k < 4; ++k) pp[k] = parts[k].defined() ? parts[k].data_ptr<float>() : nullptr;
    for (int64_t i = 0; i < B; ++i) {
      v[i].grad_out6 = nullptr;
      v[i].grad_rgb = pp[0] ? pp[0] + 3 * H * W * i : nullptr;
      v[i].grad_depth = pp[1] ? pp[1] + H * W * i : nullptr;
      v[i].grad_opacity = pp[2] ? pp[2] + H * W * i : nullptr;
      v[i].grad_depth2 = pp[3] ? pp[3] + H * W * i : nullptr;
      v[i].grad_bg = bg_grad_wanted ? gsh.data_ptr<float>() + p.Np + 256 * i : nullptr;
    }
    // the backward's densify statistics (sum |d L / d mean2d|, visits: gs/gaussian_splatting.py:464-469) are summed by the projection
    // backward itself, which holds every view's d L / d mean2d in registers
    float *stat_acc = nullptr, *stat_cnt = nullptr;
    {
      const auto &ga = ctx->saved_data["grad_accum"];
      if (ga.isTensor() && ga.toTensor().defined()) {
        stat_acc = ga.toTensor().data_ptr<float>();
        const Tensor cnt = ctx->saved_data["cnt"].toTensor();
        stat_cnt = cnt.defined() ? cnt.data_ptr<float>() : nullptr;
      }
    }
    // the moment form (round 6): ten components per (tile, Gaussian) cross the lanes instead of thirteen; the projection backward
    // expands the moments per (view, Gaussian)
    GS(gsgen_vol_render_rgbd_backward_batch_moments((uint32_t)B, v, (uint32_t)N, color.data_ptr<float>(), alpha.data_ptr<float>(),
                                                    gsh.data_ptr<float>(), 16, (uint32_t)p.nth, (uint32_t)p.ntw, (uint32_t)H,
                                                    (uint32_t)W, (float)ctx->saved_data["thresh"].toDouble(), tab<void>(p.bws), s));
    GS(gsgen_project_gaussians_backward_batch_heads_moments(
        (uint32_t)B, (uint32_t)N, mean.data_ptr<float>(), qvec.data_ptr<float>(), svec.data_ptr<float>(),
        tab<const float *const>(p.cam_tab), ctx->saved_data["detach"].toBool() ? 1 : 0, tab<const uint8_t *const>(p.mask_tab),
        tab<float *const>(p.gmean_tab), tab<const float *const>(p.gcov_tab), tab<const float *const>(p.gchan_tab),
        tab<const float *const>(p.depth_tab), tab<const float *const>(p.cov2d_tab),
        p.chol_tab ? tab<const float *const>(p.chol_tab) : nullptr, g_mean.data_ptr<float>(), g_qvec.data_ptr<float>(),
        g_svec.data_ptr<float>(), g_col.data_ptr<float>(), stat_acc, stat_cnt, s));
    Tensor g_bg;
    if (bg_grad_wanted && ctx->needs_input_grad(5) && g[0].defined())  // the 64 partial rows of every view -> [B,1,1,3] -> bg's shape
      g_bg = gsh.narrow(0, p.Np, 256 * B).view({B, 64, 4}).slice(-1, 0, 3).sum(1).view({B, 1, 1, 3}).sum_to_size(bg.sizes());
    Tensor g_alpha = gsh.narrow(0, 0, N);
    const auto acts = ctx->saved_data["acts"].toIntVector();
    if (acts[0] >= 0) {  // raw parameters went in: d L / d raw = d L / d activated * act'(raw), in place, one launch
      const Tensor &svec_raw = saved[12], &alpha_raw = saved[13], &color_raw = saved[14];
      GS(gsgen_activate_fields_backward((uint32_t)N, svec_raw.data_ptr<float>(), alpha_raw.data_ptr<float>(), color_raw.data_ptr<float>(),
                                        svec.data_ptr<float>(), alpha.data_ptr<float>(), color.data_ptr<float>(), (int)acts[0],
                                        (int)acts[1], (int)acts[2], g_svec.data_ptr<float>(), g_alpha.data_ptr<float>(),
                                        g_col.data_ptr<float>(), s));
    }
    return {g_mean, g_qvec, g_svec, g_alpha, g_col, g_bg, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(),
            Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

// ---- rgb (+ T) from SH coefficients (C = 1..4) or post-activation colours (C = 0) ----------------------------------------
struct RenderFn : public torch::autograd::Function<RenderFn> {
  static variable_list forward(AutogradContext *ctx, Tensor mean, Tensor qvec, Tensor svec, Tensor alpha, Tensor col, OptT bg_,
                               int64_t plan_addr, int64_t C, double thresh, bool detach_depth, OptT sh_bound_, OptT sh_rows_,
                               OptT max_radii2d_, OptT grad_accum_, OptT cnt_) {
    const Plan &p = *reinterpret_cast<const Plan *>(plan_addr);
    const Tensor bg = opt(bg_), sh_bound = opt(sh_bound_), sh_rows = opt(sh_rows_), max_radii2d = opt(max_radii2d_),
                 grad_accum = opt(grad_accum_), cnt = opt(cnt_);
    TORCH_CHECK((C > 0) == (p.kind == kSh) && C >= 0 && C <= 4 && p.kind != kRgbd, "plan / C mismatch");
    check_param(mean, "mean", p.N, 3, mean); check_param(qvec, "qvec", p.N, 4, mean); check_param(svec, "svec", p.N, 3, mean);
    check_param(alpha, "alpha", p.N, 1, mean); check_param(col, C > 0 ? "sh_coeffs" : "color", p.N, C > 0 ? 3 * C * C : 3, mean);
    check_stats(max_radii2d, "max_radii2d", p.N, mean); check_stats(grad_accum, "grad_accum", p.N, mean); check_stats(cnt, "cnt", p.N, mean);
    check_stats(sh_rows, "sh_row_bounds", p.N, mean);
    TORCH_CHECK(!sh_bound.defined() || (sh_bound.is_cuda() && sh_bound.scalar_type() == at::kFloat && sh_bound.numel() >= 1),
                "gsgen_amd: sh_l1_bound must be a float32 CUDA tensor");
    mean = mean.contiguous(); qvec = qvec.contiguous(); svec = svec.contiguous();
    alpha = alpha.contiguous(); col = col.contiguous();
    c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(mean.device());
    const gsgen_stream_t s = current_stream(mean);
    const int64_t B = p.B, H = p.H, W = p.W, N = p.N;
    Tensor out = at::empty({B, H, W, 3}, mean.options());
    Tensor T = at::empty({B, H, W, 1}, mean.options());
    const int64_t ncol4 = (col.numel() + 3) / 4 * 4;
    Tensor gsh = at::empty({p.Np + ncol4}, mean.options());  // d L / d alpha [N] | d L / d col: zeroed by the projection launch
    float *o = out.data_ptr<float>(), *t = T.data_ptr<float>();
    if (p.kind == kSh) {
      gsgen_sh_view *v = tab<gsgen_sh_view>(p.views);
      const float *bgp = bg.defined() ? bg.data_ptr<float>() : nullptr;
      for (int64_t i = 0; i < B; ++i) { v[i].out = o + 3 * H * W * i; v[i].T = t + H * W * i; v[i].bg_rgb = bgp; }
    } else {
      gsgen_rgbd_view *v = tab<gsgen_rgbd_view>(p.views);
      for (int64_t i = 0; i < B; ++i) { v[i].out6 = o + 3 * H * W * i; v[i].T = t + H * W * i; }
    }
    {  // the forward's densify statistic (max_radii2d) is raised by the projection launch itself (gsgen_geometry_view::max_radii2d)
      gsgen_geometry_view *gv = tab<gsgen_geometry_view>(p.geo);
      float *mr = max_radii2d.defined() ? max_radii2d.data_ptr<float>() : nullptr;
      for (int64_t i = 0; i < B; ++i) gv[i].max_radii2d = mr;
    }
    GS(gsgen_frame_geometry_batch_zero((uint32_t)B, tab<gsgen_geometry_view>(p.geo), (uint32_t)N, mean.data_ptr<float>(),
                                       qvec.data_ptr<float>(), svec.data_ptr<float>(), (uint32_t)W, (uint32_t)H,
                                       gsh.data_ptr<float>(), (size_t)gsh.numel(), tab<void>(p.gws), s));
    if (p.kind == kSh) {
      GS(gsgen_vol_render_sh_batch_routed((uint32_t)B, tab<gsgen_sh_view>(p.views), (uint32_t)N, col.data_ptr<float>(),
                                          alpha.data_ptr<float>(), 16, (uint32_t)p.nth, (uint32_t)p.ntw, (uint32_t)H, (uint32_t)W,
                                          (uint32_t)C, (float)thresh, (uint32_t)p.segments,
                                          sh_bound.defined() ? sh_bound.data_ptr<float>() : nullptr,
                                          sh_rows.defined() ? sh_rows.data_ptr<float>() : nullptr, tab<void>(p.bws), s));
    } else {
      GS(gsgen_vol_render_rgb_batch((uint32_t)B, tab<gsgen_rgbd_view>(p.views), (uint32_t)N, col.data_ptr<float>(),
                                    alpha.data_ptr<float>(), 16, (uint32_t)p.nth, (uint32_t)p.ntw, (uint32_t)H, (uint32_t)W,
                                    (float)thresh, tab<void>(p.bws), s));
      if (bg.defined()) {
        out = out + T * bg;  // gs/renderer.py:1182; `out` (saved below) is what the backward reads as final
        gsgen_rgbd_view *v = tab<gsgen_rgbd_view>(p.views);
        float *o2 = out.data_ptr<float>();
        for (int64_t i = 0; i < B; ++i) v[i].out6 = o2 + 3 * H * W * i;
      }
    }
    ctx->save_for_backward({mean, qvec, svec, alpha, col, out, T, bg, sh_bound, sh_rows});
    ctx->saved_data["plan"] = plan_addr;
    ctx->saved_data["plan_ref"] = plan_ref(p);
    ctx->saved_data["gen"] = *tab<const int64_t>(p.generation);
    ctx->saved_data["C"] = C;
    ctx->saved_data["thresh"] = thresh;
    ctx->saved_data["detach"] = detach_depth;
    ctx->saved_data["gsh"] = gsh;
    ctx->saved_data["grad_accum"] = grad_accum;
    ctx->saved_data["cnt"] = cnt;
    ctx->mark_non_differentiable({T});
    return {out, T};
  }

  static variable_list backward(AutogradContext *ctx, variable_list g) {
    const int64_t plan_addr = ctx->saved_data["plan"].toInt();
    const Plan &p = *reinterpret_cast<const Plan *>(plan_addr);
    TORCH_CHECK(*tab<const int64_t>(p.generation) == ctx->saved_data["gen"].toInt(), kStale);
    auto saved = ctx->get_saved_variables();
    const Tensor &mean = saved[0], &qvec = saved[1], &svec = saved[2], &alpha = saved[3], &col = saved[4], &T = saved[6],
                 &bg = saved[7], &sh_bound = saved[8], &sh_rows = saved[9];
    c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(mean.device());
    const gsgen_stream_t s = current_stream(mean);
    const int64_t B = p.B, H = p.H, W = p.W, N = p.N, C = ctx->saved_data["C"].toInt();
    const float thresh = (float)ctx->saved_data["thresh"].toDouble();
    Tensor grad = g[0].contiguous();
    const int64_t ncol4 = (col.numel() + 3) / 4 * 4;
    Tensor gsh = ctx->saved_data["gsh"].toTensor();
    if (ctx->saved_data.count("used")) {  // a second backward through the same graph (retain_graph): fresh accumulators
      gsh = at::zeros({p.Np + ncol4}, mean.options());
      p.g2d.narrow(0, 0, B).zero_();
    }
    ctx->saved_data["used"] = true;
    Tensor g_alpha = gsh.narrow(0, 0, N), g_col = gsh.narrow(0, p.Np, col.numel()).view(col.sizes());
    Tensor g3d = at::empty({10 * N}, mean.options());  // mean | qvec | svec: overwritten
    Tensor g_mean = g3d.narrow(0, 0, 3 * N).view({N, 3}), g_qvec = g3d.narrow(0, 3 * N, 4 * N).view({N, 4});
    Tensor g_svec = g3d.narrow(0, 7 * N, 3 * N).view({N, 3});
    const float *gp = grad.data_ptr<float>();
    float *stat_acc = nullptr, *stat_cnt = nullptr;  // (the backward's densify statistics: summed by the SH path's projection backward)
    {
      const auto &ga = ctx->saved_data["grad_accum"];
      if (ga.isTensor() && ga.toTensor().defined()) {
        stat_acc = ga.toTensor().data_ptr<float>();
        const Tensor cnt = ctx->saved_data["cnt"].toTensor();
        stat_cnt = cnt.defined() ? cnt.data_ptr<float>() : nullptr;
      }
    }
    if (p.kind == kSh) {
      gsgen_sh_view *v = tab<gsgen_sh_view>(p.views);
      for (int64_t i = 0; i < B; ++i) v[i].grad_out = gp + 3 * H * W * i;
      GS(gsgen_vol_render_backward_sh_batch_routed_moments(  // (the moment form: expanded by the projection backward below)
          (uint32_t)B, v, (uint32_t)N, col.data_ptr<float>(), alpha.data_ptr<float>(), g_col.data_ptr<float>(),
          g_alpha.data_ptr<float>(), 16, (uint32_t)p.nth, (uint32_t)p.ntw, (uint32_t)H, (uint32_t)W, (uint32_t)C, thresh,
          (uint32_t)p.segments, sh_bound.defined() ? sh_bound.data_ptr<float>() : nullptr,
          sh_rows.defined() ? sh_rows.data_ptr<float>() : nullptr, tab<void>(p.bws), s));
    } else {
      gsgen_rgbd_view *v = tab<gsgen_rgbd_view>(p.views);
      for (int64_t i = 0; i < B; ++i) v[i].grad_out6 = gp + 3 * H * W * i;
      GS(gsgen_vol_render_rgb_backward_batch((uint32_t)B, v, (uint32_t)N, col.data_ptr<float>(), alpha.data_ptr<float>(),
                                             g_col.data_ptr<float>(), g_alpha.data_ptr<float>(), 16, (uint32_t)p.nth,
                                             (uint32_t)p.ntw, (uint32_t)H, (uint32_t)W, thresh, tab<void>(p.bws), s));
    }
    if (p.kind == kSh)
      GS(gsgen_project_gaussians_backward_batch_moments_sh(
          (uint32_t)B, (uint32_t)N, mean.data_ptr<float>(), qvec.data_ptr<float>(), svec.data_ptr<float>(),
          tab<const float *const>(p.cam_tab), ctx->saved_data["detach"].toBool() ? 1 : 0, tab<const uint8_t *const>(p.mask_tab),
          tab<float *const>(p.gmean_tab), tab<const float *const>(p.gcov_tab), tab<const float *const>(p.cov2d_tab),
          g_mean.data_ptr<float>(), g_qvec.data_ptr<float>(), g_svec.data_ptr<float>(), stat_acc, stat_cnt, s));
    else
      GS(gsgen_project_gaussians_backward_batch(
          (uint32_t)B, (uint32_t)N, mean.data_ptr<float>(), qvec.data_ptr<float>(), svec.data_ptr<float>(),
          tab<const float *const>(p.cam_tab), ctx->saved_data["detach"].toBool() ? 1 : 0, tab<const uint8_t *const>(p.mask_tab),
          tab<const float *const>(p.gmean_tab), tab<const float *const>(p.gcov_tab), nullptr, g_mean.data_ptr<float>(),
          g_qvec.data_ptr<float>(), g_svec.data_ptr<float>(), s));
    if (p.kind != kSh && stat_acc != nullptr)  // (post-activation colours: the plain projection backward, the statistics in a launch of their own)
      GS(gsgen_densify_update_batch((uint32_t)B, (uint32_t)N, nullptr, tab<const float *const>(p.gmean_tab),
                                    tab<const uint8_t *const>(p.mask_tab), nullptr, stat_acc, stat_cnt, s));
    Tensor g_bg;
    if (bg.defined() && ctx->needs_input_grad(5)) g_bg = bg_grad(grad, T, bg);
    return {g_mean, g_qvec, g_svec, g_alpha, g_col, g_bg, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(),
            Tensor(), Tensor()};
  }
};

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "gsgen_amd: the camera batch as one C++ autograd node over the C ABI of libgsgen_hip.so (see gsgen_amd/batch.py)";
  pybind11::class_<Plan, std::shared_ptr<Plan>>(m, "Plan")
      .def(pybind11::init([](int kind, int64_t B, int64_t N, int64_t Np, int64_t W, int64_t H, int64_t nth, int64_t ntw,
                             int64_t segments, uintptr_t geo, uintptr_t views, uintptr_t cam_tab, uintptr_t mask_tab,
                             uintptr_t gmean_tab, uintptr_t gcov_tab, uintptr_t gchan_tab, uintptr_t depth_tab, uintptr_t cov2d_tab,
                             uintptr_t chol_tab, uintptr_t gws, uintptr_t bws, uintptr_t generation, Tensor g2d, c10::optional<Tensor> gch,
                             pybind11::object keep) {
        auto sp = std::make_shared<Plan>();
        Plan &p = *sp;
        p.kind = kind; p.B = B; p.N = N; p.Np = Np; p.W = W; p.H = H; p.nth = nth; p.ntw = ntw; p.segments = segments;
        p.geo = geo; p.views = views; p.cam_tab = cam_tab; p.mask_tab = mask_tab; p.gmean_tab = gmean_tab; p.gcov_tab = gcov_tab;
        p.gchan_tab = gchan_tab; p.depth_tab = depth_tab; p.cov2d_tab = cov2d_tab; p.chol_tab = chol_tab; p.gws = gws; p.bws = bws;
        p.generation = generation; p.g2d = g2d; p.gch = opt(gch);
        p.keep = std::make_shared<Keep>(std::move(keep));
        return sp;
      }))
      .def("address", [](const Plan &p) { return (int64_t) reinterpret_cast<uintptr_t>(&p); });
  m.def("render_heads", [](int64_t plan, Tensor mean, Tensor qvec, Tensor svec, Tensor alpha, Tensor color, c10::optional<Tensor> bg,
                           double thresh, bool detach_depth, c10::optional<Tensor> max_radii2d, c10::optional<Tensor> grad_accum,
                           c10::optional<Tensor> cnt, bool z_var, int64_t svec_act, int64_t alpha_act, int64_t color_act) {
    return HeadsFn::apply(mean, qvec, svec, alpha, color, bg, plan, thresh, detach_depth, max_radii2d, grad_accum, cnt, z_var, svec_act,
                          alpha_act, color_act);
  }, "-> [rgb, depth, opacity, depth2 | z_var, T]; svec_act >= 0: svec / alpha / color are RAW parameters, activated inside");
  m.def("render", [](int64_t plan, Tensor mean, Tensor qvec, Tensor svec, Tensor alpha, Tensor col, c10::optional<Tensor> bg, int64_t C,
                     double thresh, bool detach_depth, c10::optional<Tensor> sh_bound, c10::optional<Tensor> sh_rows,
                     c10::optional<Tensor> max_radii2d, c10::optional<Tensor> grad_accum, c10::optional<Tensor> cnt) {
    return RenderFn::apply(mean, qvec, svec, alpha, col, bg, plan, C, thresh, detach_depth, sh_bound, sh_rows, max_radii2d, grad_accum,
                           cnt);
  }, "-> [rgb, T]");
}
