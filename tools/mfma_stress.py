"""Identical full-size SH backward launches must agree to atomics noise (the guard of tests/test_gpu_stress.py as
a tool, compared on the device): python tools/mfma_stress.py [reps] [C ...] -> worst relative deviation and the number
of launches off by more than 5e-6, per SH degree.  Kernel variant through the usual environment switches."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from gsgen_amd import _capi
from gsgen_amd.renderer import _p
if os.environ.get("STRESS_LIB"):  # an alternative build of the library (experiments)
    _capi._lib = _capi.Lib(os.environ["STRESS_LIB"])
import test_gpu_stress as S
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
Cs = [int(c) for c in sys.argv[2:]] or [1, 2, 3, 4]
N, W, H = 60_000, 800, 800
NB = int(os.environ.get("STRESS_BATCH", "0"))  # > 0: that many cameras per launch through the batched entry point
for C in Cs:
    dev, lib, P, cams = S._setup(C, N, W, H, max(NB, 1))
    buf, out, topleft, rot, ci, go = cams[0]
    if NB:
        views = (_capi.ShView * NB)()
        for v, (b_, o_, tl_, r_, ci_, go_) in zip(views, cams):
            v.mean, v.cov, v.start, v.end, v.gaussian_ids = _p(b_.mean2d), _p(b_.cov2d), _p(b_.start), _p(b_.end), _p(b_.ids)
            v.tile_order, v.topleft, v.c2w, v.bg_rgb = b_.tile_order(), _p(tl_), _p(r_), None
            v.pixel_size_x, v.pixel_size_y, v.out, v.T, v.grad_out = 1 / ci_.fx, 1 / ci_.fy, _p(o_), None, _p(go_)
        bws = torch.empty(lib.sh_batch_workspace_bytes(NB), device=dev, dtype=torch.uint8)

    def run_batch():
        per = 6 * N
        g = torch.zeros(NB * per + N * (3 * C * C + 1), device=dev)
        for i, v in enumerate(views):
            v.grad_mean, v.grad_cov = _p(g) + 4 * per * i, _p(g) + 4 * per * i + 8 * N
        gsh, ga = g[NB * per:NB * per + 3 * C * C * N], g[NB * per + 3 * C * C * N:]
        lib.vol_render_backward_sh_batch(NB, views, N, _p(P["sh"]), _p(P["alpha"]), _p(gsh), _p(ga), 16, buf.nth, buf.ntw, H,
                                         W, C, 1e-4, 0, _p(bws), None)
        g2 = g[:NB * per].view(NB, per)
        return [g2[:, :2 * N].contiguous(), g2[:, 2 * N:].contiguous(), gsh, ga]

    def run():
        if NB:
            return run_batch()
        g = torch.zeros(N * (2 + 4 + 3 * C * C + 1), device=dev)
        gm, gc, gsh, ga = g[:2 * N], g[2 * N:6 * N], g[6 * N:6 * N + 3 * C * C * N], g[6 * N + 3 * C * C * N:]
        lib.vol_render_backward_sh_ordered(N, buf.D_cap, _p(buf.mean2d), _p(buf.cov2d), _p(P["sh"]), _p(P["alpha"]),
                                           _p(buf.start), _p(buf.end), _p(buf.ids), _p(out), _p(gm), _p(gc), _p(gsh), _p(ga),
                                           _p(go), _p(topleft), _p(rot), 16, buf.nth, buf.ntw, 1 / ci.fx, 1 / ci.fy, H, W, C,
                                           1e-4, None, buf.tile_order(), None)
        return [gm, gc, gsh, ga]
    ref = run()
    scale = [float(r.abs().max()) + 1e-30 for r in ref]
    devs = []
    idle = float(os.environ.get("STRESS_IDLE_S", "0"))   # > 0: sleep that long before every 3rd launch (clock ramps)
    import time
    for i in range(reps):
        if idle and i % 3 == 0:
            torch.cuda.synchronize(); time.sleep(idle)
        again = run()
        devs.append(torch.stack([(a - r).abs().max() / s for a, r, s in zip(again, ref, scale)]))
    devs = torch.stack(devs).cpu().numpy()  # [reps, 4]: mean2d, cov2d, sh, alpha
    bad = devs.max(1) > 5e-6
    for i in np.nonzero(bad)[0][:12]:
        print(f"   launch {i}: " + " ".join(f"{x:.1e}" for x in devs[i]), flush=True)
    print(f"C={C} worst {devs.max():.2e} bad {bad.sum()}/{reps}  per output (mean2d cov2d sh alpha) worst "
          + " ".join(f"{x:.1e}" for x in devs.max(0)) + "  bad " + " ".join(str(int(x)) for x in (devs > 5e-6).sum(0)), flush=True)
