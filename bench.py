#!/usr/bin/env python
"""bench.py -- fwd+bwd renders/sec of the rasterizer hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg3|cfg4|cfg1] [--batch B]

One "step" = one pass of the hot path over one BATCH of `--batch` cameras (default 8; the reference renders a
camera batch in a Python loop, gs/gaussian_splatting.py:1423-1466): frustum cull + EWA projection + 16x16 tile
binning with per-tile depth sort + SH (degree 3) front-to-back compositing for every camera of the batch, then the
backward pass to mean[N,3], qvec[N,4], svec[N,3], alpha[N], sh[N,3,16] for dense random grad_out images
 -- one enqueue per stage for the whole batch (gsgen_*_batch entry points of the C ABI).  `value` is
renders (cameras) per second = batch x steps / time, aggregated over all GPUs.  Workload at N=1: BASELINE.json
configs[1] -- 100k Gaussians ("Point-E init" cloud), 800x800, SH degree 3.  Inputs are resident in HBM before the
timed region.  With --gpus N each rank renders its own cameras (camera sharding, weak scaling) and the rendered
images of every step are all-gathered over RCCL (north_star: "RCCL only to gather rendered images"); when started
without torchrun, --gpus N > 1 launches its own ranks (torch.distributed.run on 127.0.0.1).  Prints ONE JSON line
on rank 0.

The timed region is exactly K steps between barrier + synchronize pairs; every ctypes table, event and buffer it
touches is built and used once before it.  The region is repeated (--repeats, default: until 0.5 s have been timed)
and the median repeat is reported, with the spread next to it.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec

WORKLOADS = {
    "cfg2": "BASELINE configs[1]: 100k-Gaussian Point-E-init cloud, 800x800, SH degree 3, fwd+bwd",
    "cfg3": "BASELINE configs[2]: 500k post-densify Gaussians, 1024x1024, SH degree 3, fwd+bwd",
    "cfg4": "BASELINE configs[3]: 100k Gaussians, 64 random-pose cameras at 512x512, camera-sharded",
    "cfg1": "BASELINE configs[0]: 1k random Gaussians, 256x256, SH degree 0",
    "dry": "dry run of the multi-rank step loop: 60 random Gaussians, 40x24, SH degree 3 (no measurement)",
    "dry4": "dry run of the multi-rank step loop on cfg4's camera split: 64 random poses sharded over the ranks, 60 random "
            "Gaussians, 40x24, SH degree 3 (no measurement)",
}


def make_workload(name):
    import scenes
    if name == "cfg2":
        return scenes.pointe_scene(100_000, seed=0, svec=0.02, C=4), 800, 800
    if name == "cfg3":
        return scenes.densified_scene(500_000, seed=0, C=4), 1024, 1024
    if name == "cfg4":
        return scenes.pointe_scene(100_000, seed=0, svec=0.02, C=4), 512, 512
    if name == "cfg1":
        return scenes.random_scene(1000, seed=0, C=1), 256, 256
    if name in ("dry", "dry4"):  # the multi-rank dry run (tests/test_dist_gloo.py): a few dozen splats, SH degree 3
        sc = scenes.random_scene(60, seed=0, svec=0.006, spread=0.03, C=4)
        sc["sh"][:, :, 1:] *= 0.17  # (every splat within every dry-run view's coefficient bound)
        return sc, 40, 24
    raise SystemExit(f"unknown config {name}")


def random_pose_cameras(n_total, rank, world, W, H, seed=0, zoom=1.0):
    """cfg4: poses sampled like CameraPoseProvider.sample_one (data/__init__.py:151-205): distance
    U(2, 2.5), elevation arcsin-uniform in [-20, 90] deg, azimuth U(-180, 180), focal U(0.7, 1.35) x reso;
    the 64-camera batch is split contiguously over the ranks (gsgen_amd.dist.shard_bounds)."""
    import scenes
    from gsgen_amd.dist import shard_bounds
    rng = np.random.default_rng(seed)
    dist_ = rng.uniform(2.0, 2.5, n_total)
    lo, hi = np.sin(np.deg2rad(-20.0)), np.sin(np.deg2rad(90.0))
    elev = np.rad2deg(np.arcsin(rng.uniform(lo, hi, n_total)))
    azim = rng.uniform(-180.0, 180.0, n_total)
    focal = rng.uniform(0.7, 1.35, n_total) * W * zoom
    a, b = shard_bounds(n_total, rank, world)
    return [scenes.Camera(W, H, fx=float(focal[i]), c2w=scenes.orbit(float(dist_[i]), float(min(elev[i], 89.0)), float(azim[i])))
            for i in range(a, b)]


def camera_poses(n, rank, W, H, zoom=1.0):
    import scenes
    return [scenes.Camera(W, H, fx=float(W) * zoom, c2w=scenes.orbit(2.5, 15.0, 30.0 + 45.0 * i + 7.0 * rank)) for i in range(n)]


def b_alg_bytes(N, D, P, T, F):
    """SURVEY.md 8(d) algorithmic bytes per fwd+bwd render, and the per-kernel split."""
    parts = {
        "project_fwd": 88 * N,
        "bin_sort": 36 * D + 8 * T,
        "composite_fwd": (4 + 4 * F) * D + 16 * P,
        "composite_bwd": (4 + 4 * F) * D + 28 * P + 4 * F * D,
        "project_bwd": 108 * N,
    }
    return sum(parts.values()), parts


def choose_batch_and_slots(pairs_per_view, batch=0, slots=0):
    """cameras per launch and launches in flight for a workload of `pairs_per_view` (tile, Gaussian) pairs per camera:
    about 9 M pairs per launch (2 .. 8 cameras; 6 M until round 4, when cfg3 measured +2.5 % at four cameras per launch instead
    of two: profiles/r04_notes.md 11), three launches in flight (light launches and the polynomial-basis kernels gain from the
    third, the exact kernels of a heavy launch neither gain nor lose: profiles/r02_notes.md).  Explicit --batch / --slots win."""
    d = max(1, int(pairs_per_view))
    B = batch if batch > 0 else int(min(8, max(2, round(9.0e6 / d))))
    return B, (slots if slots > 0 else 3)


def heads_alg_bytes(N, D, P, T):
    """SURVEY.md 8(d)'s formula for the fused RGB + heads pass: F = 7 + 6 = 13 floats per record (mean2d 2, cov2d 4, alpha 1,
    r g b depth 1 depth^2), the forward writes 6 channels + T per pixel (28 B), the backward reads grad_out6 + final (48 B)
    + 4 B of pixel state; the projection backward also reads the view's channel gradients (24 B per Gaussian)."""
    F = 13
    parts = {
        "project_fwd": 88 * N,
        "bin_sort": 36 * D + 8 * T,
        "composite_fwd": (4 + 4 * F) * D + 28 * P,
        "composite_bwd": (4 + 4 * F) * D + 52 * P + 4 * F * D,
        "project_bwd": (108 + 24) * N,
    }
    return sum(parts.values()), parts


def committed_traffic(config, kernel, views):
    """HBM bytes per launch of `kernel` from committed PMC passes (profiles/rNN_traffic.json), if they are of THIS kernel,
    workload and launch shape: (bytes, valu_floor_ms, source) or (None, None, None)"""
    for fn_ in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json"):
        try:
            ent = json.load(open(os.path.join(ROOT, "profiles", fn_))).get(f"{config}|{kernel}|views={views}")
        except Exception:
            continue
        if ent:
            return (ent["traffic_bytes_per_launch"], ent.get("valu_floor_ms_per_launch"),
                    f"profiles/{fn_} (separate rocprofv3 --pmc passes of this kernel on this workload and launch shape: "
                    "2 x FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md's gfx950 correction)")
    return None, None, None


def committed_trace_ms(config, fragment, suffix=""):
    """the kernel's average duration in the committed rocprofv3 kernel trace of this command (--only-timed)"""
    import csv
    for rnd in ("r06", "r05", "r04", "r03"):
        fn_ = os.path.join(ROOT, "profiles", f"{rnd}_bench_{config}{suffix}_kernel_stats.csv")
        try:
            for row in csv.DictReader(open(fn_)):
                if fragment in row["Name"]:
                    return float(row["AverageNs"]) * 1e-6, f"profiles/{rnd}_bench_{config}{suffix}_kernel_stats.csv (rocprofv3 --kernel-trace --stats of bench.py --only-timed)"
        except Exception:
            continue
    return None, None


def cpu_baseline(sc, cams, C, budget_s=20.0):
    """The CPU oracle (a port: the reference has no CPU rasteriser) timed on this box's cores
    on a bounded sample of the same workload: whole renders of the bench cameras until
    ~budget_s is spent (at least one)."""
    import scenes
    from oracle import oracle as O
    go = None
    n, t0 = 0, time.perf_counter()
    while True:
        cam = cams[n % len(cams)]
        g = scenes.oracle_geometry(sc, cam)
        m = g["mask"]
        rot = cam.c2w[:3, :3].reshape(-1)
        bg = np.array([0.1, 0.2, 0.3], np.float32)
        out = O.render_sh_fwd(g["mean2d"], g["cov2d"], sc["sh"][m], sc["alpha"][m], g["start"], g["end"],
                              g["ids"], cam.topleft, rot, C, 1 / cam.fx, 1 / cam.fy, cam.h, cam.w, bg=bg)
        if go is None:
            go = np.random.default_rng(0).normal(size=out.shape).astype(np.float32)
        gm2, gc2, _, _ = O.render_sh_bwd(g["mean2d"], g["cov2d"], sc["sh"][m], sc["alpha"][m], g["start"],
                                         g["end"], g["ids"], out, go, cam.topleft, rot, C, 1 / cam.fx,
                                         1 / cam.fy, cam.h, cam.w)
        O.project_bwd(sc["mean"][m], sc["qvec"][m], sc["svec"][m], cam.c2w, gm2, gc2, None, True)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s * 0.6 or n >= 8:
            break
    res = {"value": n / el, "unit": "renders/s", "cores": os.cpu_count(), "kind": "port",
           "sample": f"{n} whole fwd+bwd renders of the bench workload through oracle/gs_oracle.c "
                     f"(OpenMP, {os.cpu_count()} threads) in {el:.1f} s"}
    # the contributing-pair fraction of the run, counted by the checker on the first bench camera: (tile, list
    # entry) pairs a one-wavefront-per-tile kernel walks (some pixel of the tile still alive), those in which some pixel's a*G
    # reaches 1/255, and the contributing (pixel, entry) pairs behind them
    cam = cams[0]
    g = scenes.oracle_geometry(sc, cam)
    m = g["mask"]
    walked, contrib, pix = O.part_workstats(g["mean2d"], g["cov2d"], sc["alpha"][m], g["start"], g["end"], g["ids"], cam.topleft,
                                            1 / cam.fx, 1 / cam.fy, cam.h, cam.w)
    res["workstats"] = {"tile_pairs_D": int(g["D"]), "walked_tile_entries": walked, "walked_fraction_of_D": walked / max(1, g["D"]),
                        "contributing_tile_entries": contrib, "contributing_fraction_of_walked": contrib / max(1, walked),
                        "contributing_pixel_pairs": pix, "contributing_pixels_per_contributing_entry": pix / max(1, contrib),
                        "evaluated_pixel_pairs_without_early_out": 256 * int(g["D"]),
                        "E_actual_over_E": 256 * walked / max(1, 256 * int(g["D"]))}
    # SURVEY 8(d) CPU baseline (1): the two stages the reference itself runs in PyTorch (gs/renderer.py:391-421,
    # gs/culling.py:8-37), restated in oracle/torch_port.py (the reference tree does not exist on this box), on this box's cores
    try:
        import torch
        from oracle import torch_port as TP
        mean, qvec, svec = (torch.from_numpy(sc[k][m]) for k in ("mean", "qvec", "svec"))
        c2w = torch.from_numpy(cam.c2w)

        def once():
            t1 = time.perf_counter()
            m2, c2, _, _ = TP.project_gaussians(mean, qvec, svec, c2w, True)
            TP.tile_culling_aabb_count(m2, c2, 16, cam.fx, cam.fy, cam.cx, cam.cy, cam.w, cam.h, 6.0)
            return time.perf_counter() - t1
        # all cores as SURVEY asks, and 16 threads: 80 k batched 3x3 products over 256 threads are slower than over 16 (13 s
        # against a fraction of a second per camera on the GPU box) -- the better of the two is the baseline; bounded to a few seconds
        best = None
        ncpu = os.cpu_count() or 1
        for nthr in sorted({ncpu if ncpu <= 64 else 16, min(16, ncpu)}):  # (256 threads: 13 s per camera, session r3c: not repeated)
            torch.set_num_threads(nthr)
            ts = [once()]
            while len(ts) < 5 and sum(ts) < 3.0:
                ts.append(once())
            if best is None or float(np.median(ts)) < best[0]:
                best = (float(np.median(ts)), nthr, len(ts))
        res["torch_cpu_projection_and_aabb_count"] = {
            "ms_per_camera": best[0] * 1e3, "gaussians": int(m.sum()), "torch_threads": best[1], "runs": best[2], "kind": "port",
            "what": "project_gaussians + tile_culling_aabb_count in PyTorch on the CPU (oracle/torch_port.py), forward only, "
                    "median; thread count = the faster of all cores and 16"}
    except Exception as e:  # the baseline is a report, never a reason for the bench line to be missing
        res["torch_cpu_projection_and_aabb_count"] = {"error": str(e)[:200]}
    return res


def self_launch(n_gpus):
    """`python bench.py --gpus N` without torchrun: start the N ranks ourselves (one process per GPU, RCCL
    rendezvous on 127.0.0.1) and hand their output through."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


class DryGpu:
    """Stand-in for torch.cuda in the multi-rank DRY RUN (--dry-run-lib: a CPU build of the library's kernels, gloo instead of
    RCCL): the step loop, its streams / events, the broadcasts, the per-step all_gather and the reductions of the report run
    exactly as on GPUs, in order and in shape, with nothing asynchronous underneath.  Test infrastructure
    (tests/test_dist_gloo.py starts two such ranks before the driver ever starts eight real ones); it measures nothing."""

    class Stream:
        cuda_stream = None

        def __init__(self, *a, **k):
            pass

        def wait_event(self, e):
            pass

        def wait_stream(self, s):
            pass

    class Event:
        def __init__(self, *a, **k):
            pass

        def record(self, stream=None):
            pass

        def elapsed_time(self, other):
            return 1.0

        def query(self):
            return True

    @staticmethod
    def stream(s):
        import contextlib
        return contextlib.nullcontext()

    @staticmethod
    def synchronize():
        pass


class _Cfg(dict):
    """the reference reads its OmegaConf node through attribute access, .get and hasattr"""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None


def other_configs(steps=10, warmup=3):
    """The other BASELINE workloads and the routing stress, measured by the driver's own command (until round 5 only the
    builder ever ran them): cfg3 (500 k Gaussians, 1024^2), cfg4 (64 random poses at 512^2), cfg2 with 1 % outlier splats and with
    0.7 x focal length -- each as `bench.py --config ... --only-timed` in a process of its own (nothing but warm-up + timed regions:
    the same timed core as `value`), a few seconds each; value, step time and the dominant kernel's roofline fraction per line."""
    import subprocess
    out = {}
    me = os.path.abspath(__file__)
    for name, extra in (("cfg3", ["--config", "cfg3"]), ("cfg4", ["--config", "cfg4"]),
                        ("cfg2_outliers_0.01", ["--config", "cfg2", "--outlier-fraction", "0.01"]),
                        ("cfg2_focal_0.7", ["--config", "cfg2", "--focal-scale", "0.7"])):
        t0 = time.perf_counter()
        try:
            r = subprocess.run([sys.executable, me, "--gpus", "1", "--steps", str(steps), "--warmup", str(warmup), "--only-timed", "--repeats", "3",
                                *extra], capture_output=True, text=True, timeout=180)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                out[name] = {"error": (r.stderr or r.stdout)[-300:]}
                continue
            j = json.loads(line[-1])
            ro = j.get("roofline") or {}
            out[name] = {"value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "cameras_per_step": (j.get("config") or {}).get("cameras_per_step"),
                         "workload": (j.get("config") or {}).get("workload"), "dominant_kernel": ro.get("kernel"),
                         "roofline_frac": ro.get("frac"), "avg_launch_ms": ro.get("avg_launch_ms"), "whole_render_hbm_frac": ro.get("whole_render_hbm_frac"),
                         "tiles_exact_of_nonempty": (j.get("config") or {}).get("tiles_exact_of_nonempty"), "wall_s": round(time.perf_counter() - t0, 1)}
        except Exception as e:  # a line that cannot be produced is reported as such, never dropped silently
            out[name] = {"error": repr(e)[:300]}
    return out


def trainer_step():
    """BASELINE configs[4] without the diffusion model (tools/bench_step.py: 100 k Gaussians, 4 views at 512^2 drawn afresh per step,
    rgb + depth + opacity + z_var, densify statistics, Adam on the five raw fields) -- eager, and as ONE hipGraph replayed for each
    step's fresh cameras (gsgen_amd.graph.CapturedStep) -- each in a process of its own: iterations/s and host time per iteration."""
    import subprocess
    out = {}
    tool = os.path.join(ROOT, "tools", "bench_step.py")
    for name, extra in (("eager", []), ("captured_hipgraph", ["--graph"])):
        try:
            r = subprocess.run([sys.executable, tool, "--steps", "200", "--warmup", "20", *extra], capture_output=True, text=True, timeout=180)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                out[name] = {"error": (r.stderr or r.stdout)[-300:]}
                continue
            j = json.loads(line[-1])
            out[name] = {"value": j["value"], "unit": j["unit"], "ms_per_iter": j["ms_per_iter"], "host_ms_per_iter": j["host_ms_per_iter"],
                         "hipgraph": j["hipgraph"], "hipgraph_error": j.get("hipgraph_error"), "graph_captures": j.get("graph_captures")}
            out["workload"] = j["config"]["workload"] + "; " + j.get("cameras", "")
        except Exception as e:
            out[name] = {"error": repr(e)[:300]}
    return out


def model_surfaces(sc, cams, dev, B, K, H, W):
    """views/s of the model-level training call on the bench workload (B cameras per step, one step in flight, dense random
    gradients into all four outputs, gradients to the five raw parameter fields, densify statistics updated):
    model_surface = gsgen_amd.model.GaussianSplattingRenderer; dropin_gs_surface = the reference's own class (imported from
    tests/_refpy.zip, unmodified) with `_backend` = this library's compiled `_gs`."""
    import torch
    import gsgen_amd
    from gsgen_amd import renderer as R
    from gsgen_amd.model import GaussianSplattingRenderer
    devs = str(dev)
    cfg = _Cfg(device=devs, svec_act="exp", alpha_act="sigmoid", color_act="sigmoid", tile_size=16, frustum_culling_radius=6.0,
               tile_culling_type="aabb", tile_culling_thresh=0.01, tile_culling_radius=6.0, T_thresh=1e-4,
               skip_frustum_culling=False, normal_as_rgb=False, debug=False, depth_detach=True,
               background=_Cfg(type="fixed", device=devs, color=[0.1, 0.2, 0.3], random_aug=False, random_aug_prob=0.0),
               densify=_Cfg(enabled=True), prune=_Cfg(enabled=False))
    init = lambda: {k: torch.tensor(np.ascontiguousarray(sc[k])) for k in ("mean", "qvec", "svec", "color", "alpha")}  # noqa: E731
    init_ = init()
    init_["alpha"] = init_["alpha"].clamp(1e-4, 1 - 1e-4)
    go = {k: torch.randn(B, H, W, c, device=dev) for k, c in (("rgb", 3), ("depth", 1), ("opacity", 1), ("z_var", 1))}
    c2w = torch.tensor(np.stack([c.c2w for c in cams[:B]]))

    def measure(model, infos, label, path):
        model.train()
        batch = {"c2w": c2w, "camera_info": infos}

        def step(given=False):
            out = model(batch)
            if given:  # the guidance hands over d L / d image (SDS: trainer.py:305-331 turns it into a loss of that gradient): no loss kernels
                torch.autograd.backward([out[k] for k in out], [go[k] for k in out])
            else:
                sum((out[k] * go[k]).sum() for k in out).backward()
            model.post_backward()
            for q in model.parameters():
                q.grad = None

        def timed(given):
            for _ in range(3):
                step(given)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(K):
                step(given)
            host = time.perf_counter() - t0
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            return {"value": B * K / el, "unit": "views/s", "ms_per_step": el / K * 1e3, "host_ms_per_step": host / K * 1e3}

        res = timed(False)
        res.update({"cameras_per_step": B, "steps": K, "class": label, "path": path,
                    # the same step with the image gradients handed to autograd.backward directly: what the harness' own four-term
                    # loss (4 products, 4 sums and their backward over 123 MB of images: torch kernels, not this library's) costs
                    "image_gradients_given": timed(True)})
        return res

    res = {"model_surface": measure(
        GaussianSplattingRenderer(cfg, dict(init_)), [R.CameraInfo(*c.intr) for c in cams[:B]], "gsgen_amd.model.GaussianSplattingRenderer",
        "forward(batch) -> BatchRenderer.render_heads (one enqueue per stage) -> dict; "
        "loss.backward(); post_backward()")}
    try:
        import refshim
        if refshim.staged_available() or os.environ.get("GSGEN_TEST_REFPY") == "staged":
            os.environ["GSGEN_TEST_REFPY"] = "staged"  # never /root/reference at run time: the archive only
            backend = gsgen_amd.compiled_gs() or gsgen_amd.install_as_gs(compiled=False)
            refshim.install()
            sys.modules["_gs"] = backend
            import types
            import refpy_cases as RC
            import gs.renderer as GR
            GR._backend = backend
            stub = types.SimpleNamespace(cudaProfilerStart=lambda: 0, cudaProfilerStop=lambda: 0)
            torch.cuda.profiler.cudart = lambda: stub
            M = RC.import_reference_model(backend)
            from utils.camera import CameraInfo as RefCameraInfo
            rcfg = _Cfg(cfg)
            res["dropin_gs_surface"] = measure(
                M.GaussianSplattingRenderer(rcfg, dict(init_)), [RefCameraInfo(*c.intr) for c in cams[:B]],
                "the reference's gs.gaussian_splatting.GaussianSplattingRenderer, unmodified (tests/_refpy.zip)",
                "render_one per camera on the compiled `_gs` drop-in: culling_gaussian_bsphere, five mask gathers, torch projection, "
                "tile_culling_aabb_count with its .item() sync, tile_culling_aabb_start_end, render_with_T + 3 x render_scalar")
            res["dropin_gs_surface"]["model_surface_speedup"] = res["model_surface"]["value"] / res["dropin_gs_surface"]["value"]
        else:
            res["dropin_gs_surface"] = {"skipped": "tests/_refpy.zip is not present (staged by tests/stage_refpy.py in the authoring container)"}
    except Exception as e:
        res["dropin_gs_surface"] = {"error": repr(e)[:300]}
    return res


class HostClock:
    """host time spent inside each kind of enqueue call of the timed region"""

    def __init__(self):
        self.acc = {}

    def call(self, kind, fn, *a):
        t = time.perf_counter()
        fn(*a)
        self.acc[kind] = self.acc.get(kind, 0.0) + time.perf_counter() - t

    def reset(self):
        self.acc = {}


def main():
    """AUDIT GUIDE.  Everything `value` is made of sits in three places:
      * run_step(j)            -- ONE step: the enqueues of one camera batch (coefficient bounds, geometry, compositing forward,
                                  compositing backward, projection backward) on slot j's stream.  run_heads_step: the RGB + heads step.
      * region(first, ...)     -- THE TIMED REGION: barrier + synchronize, exactly K steps, barrier + synchronize, max over ranks; HIP
                                  events on the launch stream bracket every step's compositing launches.  Nothing else is timed.
      * the block under "timed region" -- W warm-up steps, then region() repeated until 0.5 s are timed; the MEDIAN repeat is `value`.
    What follows are secondary views, one function each, none of which touches `value`: exact_basis_view (the same region, exact SH
    basis), the no-gather region (multi-GPU), one_in_flight (one step at a time), heads_report (the RGB + heads step measured like
    `value`), alone_pass (one launch in flight), latency_view (one camera at a time), autograd_surface_view (BatchRenderer + autograd),
    model_surfaces (the model-level call; module level), other_configs (cfg3 / cfg4 / stress lines; module level), trainer_step (the 4 x 512^2 optimisation step, eager and captured; module level), cpu_baseline
    (the oracle; module level).  `--only-timed` runs warm-up + timed regions and none of them."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50, help="timed steps; a step renders --batch cameras fwd+bwd")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--batch", type=int, default=0,
                    help="cameras per step = cameras per launch of every stage (gsgen_*_batch entry points); 0 = chosen "
                         "from the workload: about 6 M (tile, Gaussian) pairs per launch, between 2 and 8 cameras")
    ap.add_argument("--slots", type=int, default=0,
                    help="steps in flight: own HIP stream and buffers each, so one batch's geometry overlaps the "
                         "other's compositing; 0 = 2, or 3 when the batch is smaller than 4 cameras")
    ap.add_argument("--segments", type=int, default=1,
                    help="backward workgroups per tile (segments of 32 list entries); 1 = one workgroup per tile")
    ap.add_argument("--latency-segments", type=int, default=8,
                    help="the same for the one-render-in-flight pass: uniform work units shorten a lone launch's tail")
    ap.add_argument("--repeats", type=int, default=0, help="repeats of the timed region (0: until 0.5 s are timed, <= 25)")
    ap.add_argument("--geo-priority", type=int, default=0,
                    help="1: a slot's geometry stage runs on its own HIGH-priority HIP stream, ahead of the other slot's compositing launch "
                         "instead of in its shadow (measured: 3 333 vs 3 361 renders/s -- the chip is busy either way, "
                         "profiles/r02_notes.md); 0 (default): everything of a slot on one stream")
    ap.add_argument("--sh-basis", choices=["auto", "exact"], default="auto",
                    help="auto: EVERY step measures the coefficient bound (max over splats and channels of sum_{k>=1} |sh|) on the "
                         "device (gsgen_sh_l1_bound, inside the timed region, no host sync) and hands its device address to the "
                         "SH launches, which route per view on the device: the tile-local polynomial form of the per-pixel SH "
                         "basis where the error bound allows (images within 1e-5 of the exact kernels; include/gsgen_hip.h, "
                         "\"the coefficient bound\"), the exact kernel elsewhere; exact: the exact kernels only.  With auto the "
                         "exact kernels are timed too and reported as `exact_basis`")
    ap.add_argument("--no-surface", action="store_true", help="skip the autograd-surface pass (BatchRenderer.render + backward)")
    ap.add_argument("--only-timed", action="store_true",
                    help="profiling runs: nothing but the warm-up and the timed regions launches kernels (no exact-basis region, no "
                         "secondary views, no CPU baseline), so that a rocprofv3 kernel trace of the process averages exactly the "
                         "launches `value` and `roofline.avg_launch_ms` are made of")
    ap.add_argument("--dry-run-lib", default=None, metavar="PATH",
                    help="DRY RUN of the (multi-rank) step loop on the CPU: PATH is a host build of this library's kernels (the "
                         "tests pass one), the process group is gloo, streams and events are stand-ins.  Checks the protocol -- "
                         "shapes, broadcasts, collectives in the same order on every rank -- not speed; implies --only-timed")
    ap.add_argument("--path", choices=["sh", "heads"], default="sh",
                    help="sh (default): the metric of BASELINE.json -- SH degree 3 compositing; `heads_path` (the trainer's default "
                         "outputs: rgb + depth + opacity + depth^2 from post-activation colours, gs/gaussian_splatting.py:1304-1416) "
                         "is measured the same way afterwards and reported in the same line.  heads: the timed region IS the "
                         "RGB + heads step (profiling: with --only-timed a kernel trace holds exactly its launches)")
    ap.add_argument("--no-heads", action="store_true", help="skip the RGB + heads pass")
    ap.add_argument("--always-fallback", action="store_true",
                    help="SH degree 3: enqueue the persistent exact fallback kernels with every batch (rounds 3-5) instead of only while the "
                         "polynomial forward reports crowded tiles (gsgen_sh_view::route_report / no_fallback, round 6): same-box A/Bs")
    ap.add_argument("--no-heads-chol", action="store_true",
                    help="RGB + heads: stage the records with the fp64 preparation per staged (tile, Gaussian) record (rounds 1-5) instead of "
                         "reading what the projection launch prepared per (view, Gaussian) (gsgen_geometry_view::chol, round 6): same-box A/Bs")
    ap.add_argument("--sh-grad-form", choices=["moments", "plain"], default="moments",
                    help="SH backward: the moment form BatchRenderer.render runs (round 6) or the plain one, for same-box A/Bs")
    ap.add_argument("--heads-grad-form", choices=["moments", "plain"], default="moments",
                    help="RGB + heads backward: the moment form BatchRenderer.render_heads runs (gsgen_vol_render_rgbd_backward_batch_"
                         "moments, round 6) or the plain thirteen-component form, for same-box A/Bs")
    ap.add_argument("--torch-fill", action="store_true",
                    help="A/B: zero the step's gradient accumulators with a torch fill kernel between forward and backward (rounds "
                         "1-3) instead of inside the projection launch (gsgen_frame_geometry_batch_zero)")
    ap.add_argument("--gather", choices=["images", "none"], default="images",
                    help="multi-GPU: images (default) = one all_gather of the step's rendered images per step, as north_star "
                         "asks; the same timed region WITHOUT the gather is reported next to it as `value_no_gather`, so that "
                         "compute scaling and xGMI cost separate.  none: no gather in the headline either")
    ap.add_argument("--focal-scale", type=float, default=1.0,
                    help="stress test of the SH routing: every camera's focal length times this (0.7 = the widest camera of BASELINE "
                         "configs[3]: under round 3's per-view rule such views fell back to the exact kernels)")
    ap.add_argument("--outlier-fraction", type=float, default=0.0,
                    help="stress test of the SH routing: this fraction of the splats gets its higher-band SH coefficients multiplied by "
                         "60 (beyond any view's bound): their tiles go to the exact kernel, the others stay polynomial")
    ap.add_argument("--join-every", type=int, default=0, metavar="J",
                    help="experiment: after every J steps all slot streams wait for each other -- with --batch B/2 --slots 2 "
                         "--join-every 2 a strictly sequential optimiser whose step is two half-batches on two streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the cfg3 / cfg4 / routing-stress lines (`other_configs`) and the trainer-shaped step (`trainer_step`)")
    ap.add_argument("--no-latency", action="store_true", help="skip the one-render-in-flight and hipGraph passes")
    args = ap.parse_args()
    dry = args.dry_run_lib is not None
    if dry:  # (--config cfg4: the 64-pose camera split of BASELINE configs[3] over the ranks, 8 cameras per step)
        cfg4_split = args.config == "cfg4"
        args.only_timed, args.config = True, ("dry4" if cfg4_split else "dry")
        args.batch, args.slots = args.batch or (8 if cfg4_split else 2), args.slots or 2
    if args.only_timed:
        args.no_surface = args.no_latency = args.no_cpu_baseline = args.no_heads = args.no_other_configs = True

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    import ctypes
    import torch
    from gsgen_amd import _capi, renderer as R
    from gsgen_amd.batch import _sub

    # stdout carries the ONE JSON line and nothing else: whatever libraries print there (RCCL's version banner at
    # process-group start-up, flushed at exit) is sent to stderr
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if dry:
        gpu, dev = DryGpu, torch.device("cpu")
        _capi._lib = _capi.Lib(args.dry_run_lib)  # every `_capi.load()` of this process now drives the host build
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
        gpu = torch.cuda
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    dist = None
    # GSGEN_BENCH_FORCE_DIST=1: run the multi-rank code path (process group, broadcasts, the per-step all_gather on the
    # communication stream) with ONE rank -- the only way to exercise it on a single-GPU box
    if world > 1 or os.environ.get("GSGEN_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        if dry:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    lib = _capi.load()
    sc, W, H = make_workload(args.config)
    if args.outlier_fraction > 0.0:
        rng_o = np.random.default_rng(99)
        pick = rng_o.choice(sc["sh"].shape[0], max(1, int(round(args.outlier_fraction * sc["sh"].shape[0]))), replace=False)
        sc["sh"][pick, :, 1:] *= 60.0
    C = sc["C"]
    N = sc["mean"].shape[0]
    K, nseg = max(1, args.steps), max(1, args.segments)
    # batch size and launches in flight from the workload (one untimed probe render): a launch should carry enough tiles
    # to fill the chip and hide its tail, but the cameras of a launch share the gradient accumulators -- with 500 k
    # Gaussians and 2.5 M pairs per view (cfg3) two cameras per launch beat eight (profiles/r02_notes.md); light launches
    # (< 5.2 M pairs: the 512^2 views of cfg4, cfg3's pairs of cameras) overlap better three deep than two
    # (cfg4: 6 988 vs 6 661 renders/s; cfg2, 5.7 M pairs per launch: 3 372 vs 3 368)
    zoom = (12.0 if dry else 1.0) * args.focal_scale  # (the dry run's few splats sit in a narrow view: the routed kernels take their polynomial form)
    pose_split = args.config in ("cfg4", "dry4")
    probe_cam = (random_pose_cameras(64, rank, world, W, H, zoom=zoom) if pose_split else camera_poses(1, rank, W, H, zoom))[0]
    pb = R.FrameBuffers(N, W, H, dev)
    tp = {k: torch.tensor(sc[k], device=dev) for k in ("mean", "qvec", "svec")}
    probe_block = torch.from_numpy(R.CameraInfo(*probe_cam.intr).pack(probe_cam.c2w)).to(dev)
    lib.frame_geometry(N, tp["mean"].data_ptr(), tp["qvec"].data_ptr(), tp["svec"].data_ptr(), probe_block.data_ptr(), W, H, pb.D_cap,
                       pb.mean2d.data_ptr(), pb.cov2d.data_ptr(), pb.depth.data_ptr(), pb.mask.data_ptr(), pb.ids.data_ptr(),
                       pb.start.data_ptr(), pb.end.data_ptr(), pb.total.data_ptr(), pb.ws.data_ptr(), pb.ws.numel(), None)
    gpu.synchronize()
    d_probe = max(1, int(pb.total.item()))
    del pb, tp
    B, auto_slots = choose_batch_and_slots(d_probe, args.batch, args.slots)
    # the pair lists start at 1.5 x the probe camera's count (the sizing pass below still grows them where a camera needs more):
    # no launch of the run, sizing pass included, works on an overflowed -- empty -- frame, so per-launch averages of a
    # profiler run over this process are not diluted by launches that do nothing
    cap0 = int(1.5 * d_probe) + 4096
    if dist is not None:  # one shape for the job (the gathered tensor is [world, B, H, W, 3])
        bt = torch.tensor([B, auto_slots], device=dev)
        dist.broadcast(bt, 0)
        B, auto_slots = int(bt[0].item()), int(bt[1].item())
    cams = random_pose_cameras(64, rank, world, W, H, zoom=zoom) if pose_split else camera_poses(B if dry else max(8, B), rank, W, H, zoom)
    ncam = len(cams)
    cis = [R.CameraInfo(*c.intr) for c in cams]
    nth, ntw = R.n_tiles(H, W)
    CC3 = 3 * C * C
    t = {k: torch.tensor(sc[k], device=dev) for k in ("mean", "qvec", "svec", "alpha", "sh")}
    cam_dev = [torch.from_numpy(ci_.pack(c.c2w)).to(dev) for ci_, c in zip(cis, cams)]
    rot_dev = [torch.from_numpy(np.ascontiguousarray(c.c2w[:3, :3]).reshape(-1).copy()).to(dev) for c in cams]
    topleft_dev = [torch.from_numpy(c.topleft).to(dev) for c in cams]
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    grad_out = torch.randn(H, W, 3, device=dev)
    # SH basis: with "auto" (SH degree 3) every step measures the coefficient bound on the device and the launches route on it
    # per view -- nothing about it is decided, or known, on the host while the steps run
    ps_cam = [max(1.0 / ci_.fx, 1.0 / ci_.fy) for ci_ in cis]
    state = {"bounded": bool(args.sh_basis == "auto" and C == 4)}
    p = lambda x: x.data_ptr()  # noqa: E731
    vtab = lambda vals: (ctypes.c_void_p * len(vals))(*vals)  # noqa: E731
    clock = HostClock()

    # ---- one step = one batch of B cameras, one enqueue per stage, on its slot's stream --------------------------
    Np = (N + 3) // 4 * 4                      # row counts padded so that every gradient block starts 16-byte aligned
    n_sh4 = (N * CC3 + 3) // 4 * 4
    fused_fill = not args.torch_fill           # gradient accumulators zeroed inside the projection launch
    heads_moments = args.heads_grad_form == "moments"
    sh_moments = args.sh_grad_form == "moments"
    heads_chol = not args.no_heads_chol
    route_hint = not args.always_fallback and not dry
    want_heads = (args.path == "heads" or not args.no_heads) and "color" in sc and not dry
    if want_heads:
        t["color"] = torch.tensor(sc["color"], device=dev)
        grad_out6 = torch.randn(H, W, 6, device=dev)

    class Slot:
        def __init__(self, stream):
            self.stream, self.s = stream, stream.cuda_stream
            # geometry on a high-priority stream of its own: the hardware dispatches its workgroups ahead of the other slot's
            # 20 000-workgroup compositing launch instead of behind it
            self.geo_stream = gpu.Stream(dev, priority=-1) if args.geo_priority else stream
            self.e_geo, self.e_done, self.started = gpu.Event(), gpu.Event(), False
            with gpu.stream(stream):
                self.bufs = [R.FrameBuffers(N, W, H, dev, D_cap=cap0) for _ in range(B)]
                self.out = torch.empty(B, H, W, 3, device=dev)
                # per view: mean2d(2) | cov2d(4); shared: alpha(1) | sh -- zeroed once per step (by the projection launch
                # of the step's geometry, or by a torch fill with --torch-fill); the projection backward overwrites
                # mean(3) | qvec(4) | svec(3)
                self.gflat = torch.empty(B * 6 * Np + Np + n_sh4, device=dev)
                self.g3d = torch.empty(N * 10, device=dev)
                self.seg_ws = [torch.empty(max(1, lib.segment_workspace_bytes(nth * ntw, nseg)), device=dev, dtype=torch.uint8)
                               for _ in range(B)]
                self.bws = torch.zeros(lib.sh_batch_workspace_bytes_routed(B, nth * ntw), device=dev, dtype=torch.uint8)
                # a step as two half-batches on two streams (one step in flight): the second half's stream, routing bytes and events
                self.bws2 = torch.zeros(lib.sh_batch_workspace_bytes_routed(B, nth * ntw), device=dev, dtype=torch.uint8)
                self.side = gpu.Stream(dev)
                self.e_fork, self.e_join = gpu.Event(), gpu.Event()
                # the step's own measurement of its coefficients (gsgen_sh_l1_bound_rows): per-splat bounds the launches route on
                # PER TILE, and their maximum (reported; the per-view rule of round 3 routed on it)
                self.bound = torch.zeros(1, device=dev)
                self.rows = torch.zeros(N, device=dev)
                self.gws = torch.empty(lib.frame_batch_workspace_bytes(B), device=dev, dtype=torch.uint8)
                self.gathered = torch.empty(world, B, H, W, 3, device=dev) if dist is not None else None
                if want_heads:
                    # RGB + heads: out6 / T are written whole by the batched forward (empty tiles included); per view
                    # mean2d | cov2d | chan6 gradients, shared alpha; the projection backward writes colour(3) too
                    self.out6 = torch.empty(B, H, W, 6, device=dev)
                    self.T6 = torch.empty(B, H, W, device=dev)
                    self.hflat = torch.empty(B * 12 * Np + Np, device=dev)
                    self.chol = torch.empty(B, Np, 4, device=dev)
                    self.h_color = torch.empty(N * 3, device=dev)
            # the images of a step are complete after its forward: they are gathered on the communication stream while
            # the step's backward runs; the slot's next forward waits for that gather before it overwrites `out`
            self.e_fwd, self.e_gathered, self.gather_pending = gpu.Event(), gpu.Event(), False
            o = B * 6 * Np
            self.g_alpha, self.g_sh = self.gflat[o:o + N], self.gflat[o + Np:o + Np + N * CC3]
            self.g_shared, self.n_shared = self.gflat[o:], Np + n_sh4
            self.g_mean, self.g_qvec, self.g_svec = self.g3d[:3 * N], self.g3d[3 * N:7 * N], self.g3d[7 * N:]
            self.tables, self.htables = {}, {}
            self.route, self.route_clean, self.route_nf, self.route_set = (R.PairCountReport(1) if not dry else None), 0, 0, {}

        def _geo(self, k0, g0, stride, chan6):
            """gsgen_geometry_view table of the step whose first camera is k0; g0: the slot's per-view gradient blocks
            (stride floats apart: mean2d 2 Np | cov2d 4 Np [| chan6 6 Np]), zero-filled by the projection launch"""
            geo = (_capi.GeometryView * B)()
            for i in range(B):
                k, b_, g = (k0 + i) % ncam, self.bufs[i], geo[i]
                g.cam, g.mean2d, g.cov2d, g.depth, g.mask = p(cam_dev[k]), p(b_.mean2d), p(b_.cov2d), p(b_.depth), p(b_.mask)
                g.gaussian_ids, g.start, g.end, g.total = p(b_.ids), p(b_.start), p(b_.end), p(b_.total)
                g.workspace, g.workspace_bytes, g.D_cap = p(b_.ws), b_.ws.numel(), b_.D_cap
                if fused_fill:
                    g.zero_grad_mean2d = g0 + 4 * stride * i
                    g.zero_grad_cov2d = g0 + 4 * stride * i + 4 * 2 * Np
                    if chan6:
                        g.zero_grad_chan6 = g0 + 4 * stride * i + 4 * 6 * Np
            return geo

        def prepared(self, k0):
            """ctypes tables of the step whose first camera is k0 (built once per D_cap of the buffers)"""
            key = (k0, tuple(b_.D_cap for b_ in self.bufs))
            if key not in self.tables:
                views = (_capi.ShView * B)()
                g0 = p(self.gflat)
                geo = self._geo(k0, g0, 6 * Np, False)
                for i in range(B):
                    k, b_, v = (k0 + i) % ncam, self.bufs[i], views[i]
                    v.mean, v.cov, v.start, v.end, v.gaussian_ids = p(b_.mean2d), p(b_.cov2d), p(b_.start), p(b_.end), p(b_.ids)
                    v.tile_order = b_.tile_order()
                    v.topleft, v.c2w, v.bg_rgb = p(topleft_dev[k]), p(rot_dev[k]), p(bg)
                    v.pixel_size_x, v.pixel_size_y = 1.0 / cis[k].fx, 1.0 / cis[k].fy
                    v.out, v.T = p(self.out[i]), None
                    v.segment_workspace = p(self.seg_ws[i]) if nseg > 1 else None
                    v.grad_out = p(grad_out)
                    v.grad_mean = g0 + 4 * 6 * Np * i
                    v.grad_cov = v.grad_mean + 4 * 2 * Np
                proj = (vtab([p(cam_dev[(k0 + i) % ncam]) for i in range(B)]), 1, vtab([p(b_.mask) for b_ in self.bufs]),
                        vtab([g0 + 4 * 6 * Np * i for i in range(B)]), vtab([g0 + 4 * 6 * Np * i + 4 * 2 * Np for i in range(B)]),
                        vtab([p(b_.cov2d) for b_ in self.bufs]) if sh_moments else None)  # (moments: the views' cov2d; plain: no d L / d depth)
                self.tables[key] = (geo, views, proj)
            return self.tables[key]

        def prepared_heads(self, k0):
            """the same for the RGB + heads step: gsgen_rgbd_view table, projection-backward tables with the views' channel
            gradients and depths"""
            key = (k0, tuple(b_.D_cap for b_ in self.bufs))
            if key not in self.htables:
                views = (_capi.RgbdView * B)()
                g0 = p(self.hflat)
                geo = self._geo(k0, g0, 12 * Np, True)
                for i in range(B):
                    k, b_, v = (k0 + i) % ncam, self.bufs[i], views[i]
                    v.mean, v.cov, v.depth = p(b_.mean2d), p(b_.cov2d), p(b_.depth)
                    v.start, v.end, v.gaussian_ids, v.tile_order = p(b_.start), p(b_.end), p(b_.ids), b_.tile_order()
                    v.topleft = p(topleft_dev[k])
                    v.pixel_size_x, v.pixel_size_y = 1.0 / cis[k].fx, 1.0 / cis[k].fy
                    v.out6, v.T = p(self.out6[i]), p(self.T6[i])
                    v.grad_out6 = p(grad_out6)
                    v.grad_mean = g0 + 4 * 12 * Np * i
                    v.grad_cov = v.grad_mean + 4 * 2 * Np
                    v.grad_chan6 = v.grad_mean + 4 * 6 * Np
                    if heads_chol:  # the evaluation records prepared by the projection launch (gsgen_geometry_view::chol, round 6)
                        geo[i].chol = v.chol = p(self.chol[i])
                blk = [g0 + 4 * 12 * Np * i for i in range(B)]
                proj = (vtab([p(cam_dev[(k0 + i) % ncam]) for i in range(B)]), 1, vtab([p(b_.mask) for b_ in self.bufs]),
                        vtab(blk), vtab([a + 4 * 2 * Np for a in blk]), vtab([a + 4 * 6 * Np for a in blk]),
                        vtab([p(b_.depth) for b_ in self.bufs]))
                if heads_moments:  # + the views' cov2d (and prepared records): the projection backward expands the moments (include/gsgen_hip.h)
                    proj = proj + (vtab([p(b_.cov2d) for b_ in self.bufs]), vtab([p(self.chol[i]) for i in range(B)]) if heads_chol else None)
                self.htables[key] = (geo, views, proj)
            return self.htables[key]

    slots = [Slot(gpu.Stream(dev)) for _ in range(max(1, auto_slots))]
    # the communication stream at HIGH priority: the gather's copy kernels must find compute units on a chip the compositing
    # launches fill (its bytes cross xGMI while the step's backward runs)
    comm_stream = gpu.Stream(dev, priority=-1)
    seg_arg = nseg if nseg > 1 else 0

    gv_bytes = lib.frame_batch_workspace_bytes(1)
    half = (B + 1) // 2

    def parts_of(sl, halves):
        """the step's views as one launch per stage on the slot's stream, or as two half-batches (the second on the slot's
        side stream): [(first view, views, raw stream, batch workspace)]"""
        if not halves or B < 2:
            return [(0, B, sl.s, sl.bws)]
        return [(0, half, sl.s, sl.bws), (half, B - half, sl.side.cuda_stream, sl.bws2)]

    def fork(sl, parts):
        if len(parts) > 1:
            sl.e_fork.record(sl.stream)
            sl.side.wait_event(sl.e_fork)

    def join(sl, parts):
        if len(parts) > 1:
            sl.e_join.record(sl.side)
            sl.stream.wait_event(sl.e_join)

    def geometry(sl, geo, zero_ptr, zero_floats, parts=None):
        if parts is not None and len(parts) > 1:  # two half-batches: each half's chain on its own stream
            for k_, (lo, n_, s_, _) in enumerate(parts):
                clock.call("geometry", lib.frame_geometry_batch_zero, n_, _sub(geo, lo, n_), N, p(t["mean"]), p(t["qvec"]), p(t["svec"]), W, H,
                           zero_ptr if k_ == 0 else None, zero_floats if k_ == 0 else 0, p(sl.gws) + lo * gv_bytes, s_)
            return
        if sl.geo_stream is not sl.stream:
            t0 = time.perf_counter()
            if sl.started:
                sl.geo_stream.wait_event(sl.e_done)  # the slot's previous step has finished with the lists
            clock.acc["events"] = clock.acc.get("events", 0.0) + time.perf_counter() - t0
        if fused_fill:
            clock.call("geometry", lib.frame_geometry_batch_zero, B, geo, N, p(t["mean"]), p(t["qvec"]), p(t["svec"]), W, H,
                       zero_ptr, zero_floats, p(sl.gws), sl.geo_stream.cuda_stream)
        else:
            clock.call("geometry", lib.frame_geometry_batch, B, geo, N, p(t["mean"]), p(t["qvec"]), p(t["svec"]), W, H, p(sl.gws),
                       sl.geo_stream.cuda_stream)
        if sl.geo_stream is not sl.stream:
            t0 = time.perf_counter()
            sl.e_geo.record(sl.geo_stream)
            sl.stream.wait_event(sl.e_geo)
            clock.acc["events"] = clock.acc.get("events", 0.0) + time.perf_counter() - t0

    def run_step(j, ev=None, gather=True, halves=False):
        """step j: cameras (j*B .. j*B+B-1) mod the rank's camera set, on slot j mod slots.  halves: the step as two half-batches
        on two streams, forked and joined around its forward and around its backward (what BatchRenderer does for a
        sequential caller: the images of a step are complete, on the step's stream, before its backward starts)"""
        sl = slots[j % len(slots)]
        s, stream = sl.s, sl.stream
        geo, views, proj = sl.prepared((j * B) % ncam)
        if sl.gather_pending:
            stream.wait_event(sl.e_gathered)
            sl.gather_pending = False
        parts = parts_of(sl, halves)
        rows_p = None
        if state["bounded"]:  # the step's own measurement of its coefficients: one pass, on the step's stream, no sync.  The
            # maximum is a running one (no 4-byte fill launch per step; the slot zeroed it once: always an upper bound)
            clock.call("sh_bound", lib.sh_l1_bound_rows_running, N, p(t["sh"]), C, p(sl.bound), p(sl.rows), s)
            rows_p = p(sl.rows)
            if route_hint:  # the exact fallbacks are enqueued only while crowded tiles are being reported (gsgen_sh_view::no_fallback)
                t0 = time.perf_counter()
                seen = int(sl.route._np[0, 0])
                sl.route._np[0, 0] = 0
                sl.route_clean = 0 if seen else sl.route_clean + 1
                nf = 1 if sl.route_clean >= 3 else 0
                if nf != sl.route_nf or not sl.route_set.get(id(views)):
                    for i in range(B):
                        views[i].route_report, views[i].no_fallback = sl.route.ptr(0), nf
                    sl.route_nf, sl.route_set[id(views)] = nf, True
                state["no_fallback_steps"] = state.get("no_fallback_steps", 0) + nf
                clock.acc["route_hint"] = clock.acc.get("route_hint", 0.0) + time.perf_counter() - t0
        fork(sl, parts)
        geometry(sl, geo, p(sl.g_shared), sl.n_shared, parts)
        if ev is not None:
            clock.call("events", ev[0].record, stream)
        for lo, n_, s_, bws_ in parts:
            clock.call("composite_fwd", lib.vol_render_sh_batch_routed, n_, _sub(views, lo, n_), N, p(t["sh"]), p(t["alpha"]), 16, nth, ntw, H, W, C,
                       1e-4, seg_arg, (p(sl.bound) if rows_p else None), rows_p, p(bws_), s_)
        if ev is not None:
            clock.call("events", ev[1].record, stream)
        join(sl, parts)
        if sl.gathered is not None and gather and state["gather"]:
            t0 = time.perf_counter()
            sl.e_fwd.record(stream)
            comm_stream.wait_event(sl.e_fwd)
            with gpu.stream(comm_stream):
                dist.all_gather_into_tensor(sl.gathered.view(world * B, H, W, 3), sl.out)  # (the concatenated form: every backend takes it)
                sl.e_gathered.record(comm_stream)
            sl.gather_pending = True
            clock.acc["gather"] = clock.acc.get("gather", 0.0) + time.perf_counter() - t0
        if not fused_fill:
            t0 = time.perf_counter()
            with gpu.stream(stream):
                sl.gflat.zero_()
            clock.acc["zero_grads"] = clock.acc.get("zero_grads", 0.0) + time.perf_counter() - t0
        fork(sl, parts)
        if ev is not None:
            clock.call("events", ev[2].record, stream)
        for lo, n_, s_, bws_ in parts:
            clock.call("composite_bwd", sh_bwd, n_, _sub(views, lo, n_), N, p(t["sh"]), p(t["alpha"]), p(sl.g_sh),
                       p(sl.g_alpha), 16, nth, ntw, H, W, C, 1e-4, seg_arg, (p(sl.bound) if rows_p else None), rows_p, p(bws_), s_)
        if ev is not None:
            clock.call("events", ev[3].record, stream)
        join(sl, parts)
        clock.call("project_bwd", sh_proj_bwd, B, N, p(t["mean"]), p(t["qvec"]), p(t["svec"]), *proj,
                   p(sl.g_mean), p(sl.g_qvec), p(sl.g_svec), *((None, None) if sh_moments else ()), s)
        if sl.geo_stream is not stream:
            sl.e_done.record(stream)
            sl.started = True

    # the SH backward likewise (gsgen_vol_render_backward_sh_batch_routed_moments + gsgen_project_gaussians_backward_batch_moments_sh)
    sh_bwd = lib.vol_render_backward_sh_batch_routed_moments if sh_moments else lib.vol_render_backward_sh_batch_routed
    sh_proj_bwd = lib.project_gaussians_backward_batch_moments_sh if sh_moments else lib.project_gaussians_backward_batch
    # the RGB + heads backward in its moment form (round 6: what BatchRenderer.render_heads runs) or, for same-box A/Bs, the plain one
    heads_bwd = lib.vol_render_rgbd_backward_batch_moments if heads_moments else lib.vol_render_rgbd_backward_batch
    heads_proj_bwd = lib.project_gaussians_backward_batch_heads_moments if heads_moments else lib.project_gaussians_backward_batch_heads

    def run_heads_step(j, ev=None, gather=True, halves=False):
        """the trainer's default outputs for the same cameras: geometry, fused rgb + depth + opacity + depth^2 compositing
        forward, its backward for dense random gradients of all four heads, projection backward with the depth heads'
        gradients folded in (gs/gaussian_splatting.py:1304-1416: four compositing passes in the reference, one here)"""
        sl = slots[j % len(slots)]
        s, stream = sl.s, sl.stream
        geo, views, proj = sl.prepared_heads((j * B) % ncam)
        o = B * 12 * Np
        parts = parts_of(sl, halves)
        fork(sl, parts)
        geometry(sl, geo, p(sl.hflat) + 4 * o, Np, parts)
        if ev is not None:
            clock.call("events", ev[0].record, stream)
        for lo, n_, s_, bws_ in parts:
            clock.call("composite_fwd", lib.vol_render_rgbd_batch, n_, _sub(views, lo, n_), N, p(t["color"]), p(t["alpha"]), 16, nth, ntw, H, W,
                       1e-4, p(bws_), s_)
        if ev is not None:
            clock.call("events", ev[1].record, stream)
        join(sl, parts)
        if not fused_fill:
            t0 = time.perf_counter()
            with gpu.stream(stream):
                sl.hflat.zero_()
            clock.acc["zero_grads"] = clock.acc.get("zero_grads", 0.0) + time.perf_counter() - t0
        fork(sl, parts)
        if ev is not None:
            clock.call("events", ev[2].record, stream)
        for lo, n_, s_, bws_ in parts:
            clock.call("composite_bwd", heads_bwd, n_, _sub(views, lo, n_), N, p(t["color"]), p(t["alpha"]),
                       p(sl.hflat) + 4 * o, 16, nth, ntw, H, W, 1e-4, p(bws_), s_)
        if ev is not None:
            clock.call("events", ev[3].record, stream)
        join(sl, parts)
        clock.call("project_bwd", heads_proj_bwd, B, N, p(t["mean"]), p(t["qvec"]), p(t["svec"]), *proj,
                   p(sl.g_mean), p(sl.g_qvec), p(sl.g_svec), p(sl.h_color), *((None, None) if heads_moments else ()), s)
        if sl.geo_stream is not stream:
            sl.e_done.record(stream)
            sl.started = True

    state["gather"] = args.gather == "images"
    main_step = run_heads_step if args.path == "heads" else run_step
    if args.path == "heads":
        state["bounded"] = False

    def barrier():
        if dist is not None:
            dist.barrier()
        gpu.synchronize()

    # size the pair buffers once, outside the timed region: step j runs on slot j % slots with camera offset
    # (j * B) % ncam, so one period of that pair covers every (slot, cameras) combination the timed steps will see
    period = int(np.lcm(len(slots), ncam // np.gcd(ncam, B)))
    Ds = np.zeros(ncam)
    for j in range(period):
        for _ in range(3):
            main_step(j, None, False)
            gpu.synchronize()
            if all([b_.ensure_capacity() for b_ in slots[j % len(slots)].bufs]):
                break
        else:
            raise AssertionError("pair buffers still too small after growing")
        for i, b_ in enumerate(slots[j % len(slots)].bufs):
            Ds[((j * B) % ncam + i) % ncam] = int(b_.total.item())
    n_vis = int(slots[0].bufs[0].mask.sum().item())
    # list-length histogram of the run (tiles by the length of their depth-sorted list, slot 0's cameras)
    edges = [0, 1, 16, 32, 64, 128, 256, 512, 1024, 2048, 1 << 30]
    lens = torch.cat([(b_.end - b_.start).clamp(min=0).view(-1) for b_ in slots[0].bufs]).to(torch.float32)
    hist = torch.histogram(lens.cpu(), bins=torch.tensor(edges, dtype=torch.float32)).hist.to(torch.int64).tolist()
    list_hist = {"bin_edges": edges[:-1] + ["inf"], "tiles": hist, "views": B, "mean_length": float(lens.mean().item()),
                 "max_length": int(lens.max().item()), "median_length": float(lens.median().item())}

    # warm-up: W untimed steps exactly as the timed ones (events included, so every event exists before the region)
    evs = [[gpu.Event(enable_timing=True) for _ in range(4)] for _ in range(K)]
    for i in range(args.warmup):
        main_step(i, evs[i % K])
    for i in range(K):  # every event of the timed region has been recorded once
        for e in evs[i]:
            e.record(slots[i % len(slots)].stream)
    barrier()

    # ---- timed region: exactly K steps, repeated ------------------------------------------------------------------
    def region(first, step=None, in_flight=0, halves=False):
        """exactly K steps between barrier + synchronize pairs; in_flight = 1: the steps on ONE slot (one stream), i.e. one
        step in flight -- a strictly sequential optimiser's view; halves: each step as two half-batches on two streams"""
        step = step or main_step
        clock.reset()
        barrier()
        t0 = time.perf_counter()
        for i in range(K):
            step((first + i) * (len(slots) if in_flight == 1 else 1), evs[i], True, halves)
            if args.join_every > 0 and in_flight != 1 and (i + 1) % args.join_every == 0:
                jev = [gpu.Event() for _ in slots]
                for e_, sl_ in zip(jev, slots):
                    e_.record(sl_.stream)
                for sl_ in slots:
                    for e_ in jev:
                        sl_.stream.wait_event(e_)
        host = time.perf_counter() - t0
        barrier()
        el = el_local = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        fwd = float(np.mean([e[0].elapsed_time(e[1]) for e in evs]))
        bwd = float(np.mean([e[2].elapsed_time(e[3]) for e in evs]))
        return {"el": el, "el_local": el_local, "host": host, "fwd_ms": fwd, "bwd_ms": bwd, "host_by_call": dict(clock.acc)}

    regions = [region(args.warmup)]
    n_rep = args.repeats if args.repeats > 0 else int(min(25, max(1, np.ceil(0.5 / regions[0]["el"]))))
    if dist is not None:  # every rank repeats the same number of times
        nr = torch.tensor([n_rep], device=dev)
        dist.broadcast(nr, 0)
        n_rep = int(nr.item())
    for r in range(1, n_rep):
        regions.append(region(args.warmup + r * K))
    order = np.argsort([r_["el"] for r_ in regions])
    med = regions[int(order[len(order) // 2])]
    el, fwd_ms, bwd_ms = med["el"], med["fwd_ms"], med["bwd_ms"]
    value = world * B * K / el
    # Which kernel rendered which camera is the DEVICE's decision, taken per step from the bound the step measured; it is read
    # back here, after the timed regions, for the report only (gsgen_sh_poly_applies is the same rule on the host).
    S_dev = float(slots[0].bound.item()) if state["bounded"] else 0.0
    poly_cams = [bool(state["bounded"] and lib.sh_poly_applies(S_dev, ps_, C)) for ps_ in ps_cam]
    n_poly = int(sum(poly_cams))
    if dist is not None:
        pc = torch.tensor([n_poly, ncam], device=dev)
        dist.all_reduce(pc)
        n_poly_job, ncam_job = int(pc[0].item()), int(pc[1].item())
    else:
        n_poly_job, ncam_job = n_poly, ncam
    # ... and per TILE (round 4): the flag bytes the polynomial forward of each slot's last step left (1 = the tile went to the
    # exact kernel because a splat it staged exceeds the bound for its view's pixel size)
    tiles_exact = tiles_nonempty = 0
    if state["bounded"]:
        for sl_ in slots:
            o_ = lib.sh_batch_workspace_bytes(B)
            fl_ = sl_.bws[o_:o_ + B * nth * ntw].view(B, nth * ntw)
            ne_ = torch.stack([(b_.end > b_.start) & (b_.start >= 0) for b_ in sl_.bufs]).view(B, -1)
            tiles_exact += int((fl_.bool() & ne_).sum().item())
            tiles_nonempty += int(ne_.sum().item())
        if dist is not None:
            tc = torch.tensor([tiles_exact, tiles_nonempty], device=dev)
            dist.all_reduce(tc)
            tiles_exact, tiles_nonempty = int(tc[0].item()), int(tc[1].item())
    poly_applies = state["bounded"] and tiles_exact < max(1, tiles_nonempty)
    # per-rank throughput of the reported region (the driver's scaling record can see that N ranks took part)
    per_rank = [value / world]
    if dist is not None:
        pr = torch.zeros(world, device=dev, dtype=torch.float64)
        pr[rank] = B * K / med["el_local"]
        dist.all_reduce(pr)
        per_rank = [float(x) for x in pr.tolist()]

    # every rank's own view of the job (the driver's SCALE record must show that N processes on N devices took part)
    me = {"rank": rank, "local_rank": local_rank, "world_size_seen": (dist.get_world_size() if dist is not None else 1),
          "device": (torch.cuda.get_device_name(dev) if not dry else "cpu (dry run)"),
          "device_index": (dev.index if not dry else None),
          "HIP_VISIBLE_DEVICES": os.environ.get("HIP_VISIBLE_DEVICES"), "ROCR_VISIBLE_DEVICES": os.environ.get("ROCR_VISIBLE_DEVICES"),
          "pid": os.getpid(), "cameras": ncam, "renders_per_s": B * K / med["el_local"]}
    if not dry:
        try:
            pr_ = torch.cuda.get_device_properties(dev)
            me["arch"], me["compute_units"] = getattr(pr_, "gcnArchName", None), getattr(pr_, "multi_processor_count", None)
            me["pci_bus_id"] = pr_.pci_bus_id
        except Exception:
            pass
    ranks_view = [me]
    if dist is not None:
        ranks_view = [None] * world
        dist.all_gather_object(ranks_view, me)

    # ---- the same timed region with the exact per-pixel SH basis (when the headline used the polynomial form) ----------------
    def exact_basis_view():
        """-> `exact_basis`: the timed region again with the exact per-pixel SH basis in every tile"""
        state["bounded"] = False
        for i in range(max(2, len(slots))):
            run_step(i, evs[i % K])
        ex = [region(args.warmup + r * K) for r in range(min(3, n_rep))]
        exm = sorted(ex, key=lambda r_: r_["el"])[len(ex) // 2]
        exact_basis = {"value": world * B * K / exm["el"], "ms_per_step": exm["el"] / K * 1e3, "bwd_launch_ms": exm["bwd_ms"],
                       "fwd_launch_ms": exm["fwd_ms"], "bwd_kernel": lib.kernel_variant("sh_bwd_batch", C, nseg),
                       "fwd_kernel": lib.kernel_variant("sh_fwd_batch", C, nseg)}
        state["bounded"] = True
        for i in range(max(2, len(slots))):  # back to the headline's kernels for the secondary views
            run_step(i, evs[i % K])
        barrier()
        return exact_basis

    exact_basis = exact_basis_view() if (state["bounded"] and not args.only_timed) else None

    # ---- multi-GPU: the same timed region WITHOUT the per-step all_gather (compute scaling and xGMI cost separate) ------------
    no_gather = None
    if dist is not None and state["gather"]:
        state["gather"] = False
        barrier()
        ng = [region(args.warmup + r * K) for r in range(min(3, n_rep))]
        ngm = sorted(ng, key=lambda r_: r_["el"])[len(ng) // 2]
        no_gather = {"value": world * B * K / ngm["el"], "ms_per_step": ngm["el"] / K * 1e3}
        state["gather"] = True

    # ---- one STEP in flight (a strictly sequential optimiser: every step waits for the previous one's gradients) ------------
    one_step = None

    def one_in_flight(step_fn):
        """one step in flight, in the two shapes a sequential caller can give it: each stage ONE launch for the whole batch, or
        the step as two half-batches on two streams (forked and joined around the forward and around the backward: one half's
        geometry chain hides behind the other's compositing).  The better one is `value`; BatchRenderer(pipeline=True) is the
        second shape (off by default: it does not pay under that join, profiles/r05_notes.md)."""
        res_ = {}
        for nm, hv in (("one_launch_per_stage", False), ("two_half_batches", True)):
            if hv and B < 4:
                continue
            for i in range(2):
                step_fn(i * len(slots), evs[i % K], False, hv)
            o1 = [region(args.warmup + r * K, step_fn, in_flight=1, halves=hv) for r in range(min(3, n_rep))]
            o1m = sorted(o1, key=lambda r_: r_["el"])[len(o1) // 2]
            res_[nm] = {"value": world * B * K / o1m["el"], "ms_per_step": o1m["el"] / K * 1e3, "bwd_launch_ms": o1m["bwd_ms"],
                        "fwd_launch_ms": o1m["fwd_ms"]}
        best = max(res_, key=lambda k_: res_[k_]["value"])
        return dict(res_[best], shape=best, shapes=res_)

    if not args.only_timed:
        one_step = one_in_flight(run_step)

    # ---- the trainer's default outputs (rgb + depth + opacity + depth^2), measured exactly like `value` -------------------------
    def heads_report(m, alone_, one_):
        Dm = float(np.mean(Ds))
        tot_h, parts_h = heads_alg_bytes(n_vis, Dm, W * H, nth * ntw)
        bname, fname = lib.kernel_variant("rgbd_bwd_batch_moments" if heads_moments else "rgbd_bwd_batch", 1, 1), lib.kernel_variant("rgbd_fwd_batch", 1, 1)
        val = world * B * K / m["el"]
        ach_ = B * parts_h["composite_bwd"] / (m["bwd_ms"] * 1e-3) / 1e9
        tr, vf, tr_src = committed_traffic(args.config, bname, B)
        tms, tms_src = committed_trace_ms(args.config, "k_composite_bwd_chan_vec<3, true, true>" if heads_moments else "k_composite_bwd_chan_vec<3, true, false>", "_heads")
        rf = {"bound": "hbm", "kernel": bname, "achieved": ach_, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_ / HBM_PEAK_GBS,
              "traffic": tr, "traffic_source": tr_src, "alg_bytes_per_launch": B * parts_h["composite_bwd"],
              "alg_bytes_formula": "SURVEY 8(d) with F = 13: (4 + 4F) D + 52 P + 4F D per view (bench.heads_alg_bytes)",
              "views_per_launch": B, "avg_launch_ms": m["bwd_ms"], "kernel_trace_avg_launch_ms": tms, "kernel_trace_source": tms_src,
              "fwd_kernel": fname, "fwd_launch_ms": m["fwd_ms"],
              "fwd_GBs": B * parts_h["composite_fwd"] / (m["fwd_ms"] * 1e-3) / 1e9,
              "whole_render_alg_bytes": tot_h, "whole_render_hbm_frac": tot_h * (val / world) / (HBM_PEAK_GBS * 1e9)}
        if alone_ is not None:
            rf["alone_launch_ms"], rf["alone_fwd_launch_ms"] = alone_["bwd_launch_ms"], alone_["fwd_launch_ms"]
            rf["alone_frac"] = B * parts_h["composite_bwd"] / (alone_["bwd_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        if vf is not None:
            rf["valu_floor_ms"] = vf
        out_ = {"metric": "fwd+bwd views/sec of the trainer's default outputs: rgb + depth + opacity + depth^2 in ONE fused "
                          "compositing pass each way (gs/gaussian_splatting.py:1304-1416: four passes in the reference), dense "
                          "random gradients into all four heads, backward to mean, qvec, svec, alpha, colour",
                "value": val, "unit": "views/s", "ms_per_step": m["el"] / K * 1e3, "cameras_per_step": B,
                "steps_in_flight": len(slots), "host_enqueue_ms_per_step": m["host"] / K * 1e3,
                "host_enqueue_us_per_step_by_call": {k: v / K * 1e6 for k, v in m["host_by_call"].items()},
                "path": "C ABI: gsgen_frame_geometry_batch_zero -> gsgen_vol_render_rgbd_batch -> gsgen_vol_render_rgbd_backward_batch -> "
                        "gsgen_project_gaussians_backward_batch_heads", "roofline": rf}
        if one_ is not None:
            out_["one_step_in_flight"] = one_
        return out_

    def alone_pass(step):
        eva_ = [[gpu.Event(enable_timing=True) for _ in range(4)] for _ in range(min(K, 8))]
        barrier()
        for j in range(len(eva_)):
            step(j * len(slots), eva_[j], False)  # slot 0 every time: one stream
        barrier()
        return {"fwd_launch_ms": float(np.mean([e[0].elapsed_time(e[1]) for e in eva_])),
                "bwd_launch_ms": float(np.mean([e[2].elapsed_time(e[3]) for e in eva_]))}

    if args.path == "heads":  # profiling / A-B mode: the timed region was the heads step; its line and nothing else
        res = heads_report(med, None if args.only_timed else alone_pass(run_heads_step), one_step)
        els_ = [r_["el"] for r_ in regions]
        res.update({"n_gpus": world, "steps": K, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                    "dtype": "f32", "data": "synthetic",
                    "config": {"workload": WORKLOADS[args.config] + " -- RGB + heads path", "gaussians": N, "image": [H, W],
                               "tile_pairs_D": float(np.mean(Ds)), "zero_fill": "projection launch" if fused_fill else "torch fill"},
                    "timing": {"repeats": len(regions), "renders_per_s_min": world * B * K / max(els_),
                               "renders_per_s_max": world * B * K / min(els_), "timed_region_s": el}})
        if rank == 0:
            print(json.dumps(res), file=json_out, flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    heads = None
    if want_heads:
        for i in range(max(2 * len(slots), min(args.warmup, 6))):
            run_heads_step(i, evs[i % K])
        barrier()
        hr = [region(args.warmup + r * K, run_heads_step) for r in range(min(5, n_rep))]
        hm = sorted(hr, key=lambda r_: r_["el"])[len(hr) // 2]
        heads = heads_report(hm, alone_pass(run_heads_step), one_in_flight(run_heads_step))
        for i in range(max(2, len(slots))):  # back to the headline's kernels for the secondary views
            run_step(i, evs[i % K])
        barrier()

    # ---- secondary views ---------------------------------------------------------------------------------------------
    # (a) one batch in flight: the duration of a launch that has the chip to itself
    if args.only_timed:
        alone = {"fwd_launch_ms": fwd_ms, "bwd_launch_ms": bwd_ms}  # not measured in this mode
    else:
        alone = alone_pass(run_step)

    # (b) strictly one render (one camera) at a time on one stream, per-camera entry points: the latency view
    def latency_view():
        """-> `one_render_in_flight`: one camera at a time through the per-camera entry points (+ the same from a hipGraph)"""
        sl0 = slots[0]
        b0 = sl0.bufs[0]
        lseg = max(1, args.latency_segments)
        lseg_arg = lseg if lseg > 1 else 0
        lseg_ws = torch.empty(max(1, lib.segment_workspace_bytes(nth * ntw, lseg)), device=dev, dtype=torch.uint8)
        g1 = torch.empty(N * (7 + CC3), device=dev)
        g1_mean2d, g1_cov2d, g1_alpha, g1_sh = g1[:2 * N], g1[2 * N:6 * N], g1[6 * N:7 * N], g1[7 * N:]

        def one_render(k, ev=None):
            s, stream = sl0.s, sl0.stream
            order_ = b0.tile_order()
            lib.frame_geometry(N, p(t["mean"]), p(t["qvec"]), p(t["svec"]), p(cam_dev[k]), W, H, b0.D_cap, p(b0.mean2d),
                               p(b0.cov2d), p(b0.depth), p(b0.mask), p(b0.ids), p(b0.start), p(b0.end), p(b0.total), p(b0.ws),
                               b0.ws.numel(), s)
            rows_p = None
            if state["bounded"]:
                lib.sh_l1_bound_rows(N, p(t["sh"]), C, p(sl0.bound), p(sl0.rows), s)
                rows_p = p(sl0.rows)
            if ev is not None:
                ev[0].record(stream)
            lib.vol_render_sh_routed(N, b0.D_cap, p(b0.mean2d), p(b0.cov2d), p(t["sh"]), p(t["alpha"]), p(b0.start), p(b0.end),
                                     p(b0.ids), p(sl0.out[0]), p(topleft_dev[k]), p(rot_dev[k]), 16, nth, ntw, 1.0 / cis[k].fx,
                                     1.0 / cis[k].fy, H, W, C, 1e-4, p(bg), None, order_, p(lseg_ws), lseg_arg,
                                     p(sl0.bound) if rows_p else None, rows_p, s)  # (the view's bound first, then the lists)
            if ev is not None:
                ev[1].record(stream)
            with gpu.stream(stream):
                g1.zero_()
            if ev is not None:
                ev[2].record(stream)
            lib.vol_render_backward_sh_routed(N, b0.D_cap, p(b0.mean2d), p(b0.cov2d), p(t["sh"]), p(t["alpha"]), p(b0.start),
                                              p(b0.end), p(b0.ids), p(sl0.out[0]), p(g1_mean2d), p(g1_cov2d), p(g1_sh),
                                              p(g1_alpha), p(grad_out), p(topleft_dev[k]), p(rot_dev[k]), 16, nth, ntw,
                                              1.0 / cis[k].fx, 1.0 / cis[k].fy, H, W, C, 1e-4, p(bg), order_,
                                              p(lseg_ws), lseg_arg, p(sl0.bound) if rows_p else None, rows_p, s)
            if ev is not None:
                ev[3].record(stream)
            lib.project_gaussians_backward_masked(N, p(t["mean"]), p(t["qvec"]), p(t["svec"]), p(cam_dev[k]), 1, p(b0.mask),
                                                  p(g1_mean2d), p(g1_cov2d), None, p(sl0.g_mean), p(sl0.g_qvec), p(sl0.g_svec), s)

        for k in range(ncam):  # this buffer now meets every camera: size its pair list (one sync each, untimed)
            for _ in range(3):
                one_render(k)
                gpu.synchronize()
                if b0.ensure_capacity():
                    break
            else:
                raise AssertionError("pair buffer still too small after growing")
        n1 = int(min(B * K, 200))
        ev1 = [[gpu.Event(enable_timing=True) for _ in range(4)] for _ in range(n1)]
        for i in range(min(n1, 16)):
            one_render(i % ncam, ev1[i])
        barrier()
        t1 = time.perf_counter()
        for i in range(n1):
            one_render(i % ncam, ev1[i])
        barrier()
        el1 = time.perf_counter() - t1
        one = {"value": world * n1 / el1, "ms_per_render": el1 / n1 * 1e3, "renders": n1,
               "backward_segments_per_tile": lseg,
               "fwd_kernel": lib.kernel_variant("sh_fwd_poly" if poly_applies else "sh_fwd", C, lseg),
               "bwd_kernel": lib.kernel_variant("sh_bwd_poly" if poly_applies else "sh_bwd", C, lseg),
               "fwd_kernel_ms": float(np.mean([e[0].elapsed_time(e[1]) for e in ev1])),
               "bwd_kernel_ms": float(np.mean([e[2].elapsed_time(e[3]) for e in ev1]))}
        if lseg > 1:
            # list entries the compositing actually walks: the segmented forward leaves, per pixel, the first entry it did
            # not process; a tile's workgroup walks up to the largest of its pixels' (early termination, T < thresh)
            walked = []
            off = nth * ntw * 256 * lseg * 16
            for k in range(min(ncam, 8)):
                one_render(k)
                gpu.synchronize()
                stop_ = lseg_ws[off:off + nth * ntw * 256 * 4].view(torch.int32).view(nth * ntw, 256)
                n_tile = (b0.end - b0.start).clamp(min=0).view(-1)
                walked.append((float(stop_.max(dim=1).values.clamp(min=0).minimum(n_tile).sum().item()), float(n_tile.sum().item())))
            one["walked_pairs_per_view"] = float(np.mean([w_[0] for w_ in walked]))
            one["walked_fraction_of_D"] = float(np.sum([w_[0] for w_ in walked]) / max(1.0, np.sum([w_[1] for w_ in walked])))
        try:  # ... and replayed from one captured hipGraph per camera: same kernels, no launch gaps
            graphs = []
            with gpu.stream(sl0.stream):
                for k in range(min(ncam, 8)):
                    gk = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gk, stream=sl0.stream, capture_error_mode="thread_local"):  # RCCL's watchdog thread may query events meanwhile
                        one_render(k)
                    graphs.append(gk)
                for i in range(8):
                    graphs[i % len(graphs)].replay()
            barrier()
            t2 = time.perf_counter()
            with gpu.stream(sl0.stream):
                for i in range(n1):
                    graphs[i % len(graphs)].replay()
            barrier()
            el2 = time.perf_counter() - t2
            one["hipgraph_replay"] = {"value": world * n1 / el2, "ms_per_render": el2 / n1 * 1e3}
        except Exception as e:  # capture is an optimisation of the latency view only
            one["hipgraph_replay"] = {"error": str(e)[:200]}
        return one

    one = latency_view() if not args.no_latency else None

    # (c) the autograd surface: the same steps through gsgen_amd.BatchRenderer.render(...) + torch.autograd -- the path a
    # training loop takes (one autograd node per camera batch; gradients to mean, qvec, svec, alpha, sh), `surface_slots`
    # independent steps in flight on their own streams (e.g. the micro-batches of a gradient-accumulation step)
    def autograd_surface_view():
        """-> `autograd_surface`: the timed steps again through BatchRenderer.render + torch.autograd.grad (the C++ autograd node)"""
        from gsgen_amd.batch import BatchRenderer
        n_sf = len(slots)
        leaf = {k: t[k].clone().requires_grad_(True) for k in ("mean", "qvec", "svec", "alpha", "sh")}
        names = ("mean", "qvec", "svec", "alpha", "sh")
        sf_streams = [gpu.Stream(dev) for _ in range(n_sf)]
        d_cap = int(max(b_.D_cap for sl_ in slots for b_ in sl_.bufs))
        brs = []
        for st_ in sf_streams:
            with gpu.stream(st_):
                brs.append(BatchRenderer(N, W, H, dev, max_batch=B, D_cap=d_cap, pipeline=False))  # (several steps in flight: one launch per stage)
        go_b = grad_out.unsqueeze(0).expand(B, H, W, 3).contiguous()
        c2w_np = [c.c2w for c in cams]

        def surface_step(j):
            i = j % n_sf
            k0 = (j * B) % ncam
            idx = [(k0 + q) % ncam for q in range(B)]
            with gpu.stream(sf_streams[i]):
                rgb, _ = brs[i].render(leaf["mean"], leaf["qvec"], leaf["svec"], leaf["alpha"], leaf["sh"], [cis[q] for q in idx],
                                       [c2w_np[q] for q in idx], C=C, bg_rgb=bg, sh_basis=args.sh_basis)
                return torch.autograd.grad([rgb], [leaf[n_] for n_ in names], [go_b])

        gpu.synchronize()
        for j in range(max(args.warmup, 2 * n_sf)):
            surface_step(j)
        gpu.synchronize()
        assert all(br_.ensure_capacity(B) for br_ in brs), "surface pass: pair buffers overflowed"
        sf_el = []
        for r in range(min(3, n_rep)):
            barrier()
            t0 = time.perf_counter()
            for j in range(K):
                surface_step(args.warmup + r * K + j)
            sf_host = time.perf_counter() - t0
            barrier()
            sf_el.append((time.perf_counter() - t0, sf_host))
        sf_t, sf_h = sorted(sf_el)[len(sf_el) // 2]
        if dist is not None:
            tt = torch.tensor([sf_t], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            sf_t = float(tt.item())
        surface = {"value": world * B * K / sf_t, "ms_per_step": sf_t / K * 1e3, "host_enqueue_ms_per_step": sf_h / K * 1e3,
                   "steps_in_flight": n_sf, "fraction_of_c_abi_value": (world * B * K / sf_t) / value,
                   "path": "gsgen_amd.BatchRenderer.render -> torch.autograd.grad (mean, qvec, svec, alpha, sh); one autograd "
                           "node per camera batch, the coefficient bound measured inside its forward"}
        del brs
        return surface

    surface = autograd_surface_view() if (not args.no_surface and C > 0) else None

    # (d) the MODEL-LEVEL call a trainer makes -- forward(batch) -> {rgb, depth, opacity, z_var}, loss.backward(), post_backward()
    # (trainer.py:291-422 without guidance and optimiser) -- through gsgen_amd.model.GaussianSplattingRenderer, and, where
    # tests/_refpy.zip travels with the tree, through the reference's OWN unmodified class on the compiled `_gs` drop-in
    # (render_one per camera: mask gathers, torch projection, `.item()` sync, four compositing passes): the price of swapping
    # only `_gs` instead of the class
    model_views = None
    if not args.no_surface and want_heads and world == 1:
        try:
            model_views = model_surfaces(sc, cams, dev, B, min(K, 12), H, W)
        except Exception as e:  # a secondary view never costs the bench line
            model_views = {"error": repr(e)[:300]}

    # ---- report ------------------------------------------------------------------------------------------------------
    D = float(np.mean(Ds))
    P, T = W * H, nth * ntw
    F = 7 + CC3
    total_b, parts = b_alg_bytes(n_vis, D, P, T, F)
    bwd_name = lib.kernel_variant("sh_bwd_batch_poly" if poly_applies else "sh_bwd_batch", C, nseg)
    fwd_name = lib.kernel_variant("sh_fwd_batch_poly" if poly_applies else "sh_fwd_batch", C, nseg)
    traffic, valu_floor, traffic_src = committed_traffic(args.config, bwd_name, B)
    mom_ = "true" if sh_moments else "false"  # (round 6: <..., MOM>; committed traces of round 5 hold the shorter names)
    frag = f"k_composite_bwd_sh_vec<4, 4, true, 6, {mom_}>" if poly_applies else f"k_composite_bwd_sh_vec<4, 4, true, 0, {mom_}>"
    trace_ms, trace_src = committed_trace_ms(args.config, frag) if C == 4 else (None, None)
    ach = B * parts["composite_bwd"] / (bwd_ms * 1e-3) / 1e9
    els = [r_["el"] for r_ in regions]
    res = {
        "metric": "fwd+bwd renders/sec (800x800, 100k Gaussians)" if args.config == "cfg2" else f"fwd+bwd renders/sec ({args.config})",
        "value": value, "unit": "renders/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
        "ms_per_step": el / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": WORKLOADS[args.config], "gaussians": N, "visible_after_cull": n_vis, "image": [H, W],
                   "sh_degree": C - 1, "tile_pairs_D": D, "list_length_histogram": list_hist, "cameras_per_step": B,
                   "steps_in_flight": len(slots),
                   "backward_segments_per_tile": nseg,
                   "sh_basis": (f"routed on the device, per ENTRY and per TILE, every step: per-splat coefficient bounds are measured by every "
                                f"step inside the timed region (gsgen_sh_l1_bound_rows; read back afterwards: largest S = {S_dev:.3f}); a splat "
                                f"within the bound for its view's pixel size takes the tile-local degree-2 polynomial fit of the per-pixel "
                                f"basis, one beyond it is evaluated exactly inside the same kernel, a tile whose staged batch holds more than "
                                f"a quarter of such splats goes to the exact kernel: {tiles_exact} of {tiles_nonempty} non-empty tiles "
                                f"of the slots' last steps went exact.  (Round 3's per-view rule on the global S: {n_poly_job} of "
                                f"{ncam_job} cameras polynomial.)") if state["bounded"] else "exact per-pixel basis",
                   "tiles_exact_of_nonempty": [tiles_exact, tiles_nonempty],
                   "exact_fallback_launches": ("only while the polynomial forward reports crowded tiles (gsgen_sh_view::route_report, a host-"
                                               "visible word read without a sync): three clean reports in a row and they are no longer "
                                               f"enqueued -- {state.get('no_fallback_steps', 0)} steps of this process ran without them"
                                               if (route_hint and state["bounded"]) else "with every batch"),
                   "stress": {"focal_scale": args.focal_scale, "outlier_fraction": args.outlier_fraction},
                   "geometry_stream": "high priority, per slot" if args.geo_priority else "the slot's stream",
                   "parallelism": f"camera-sharded x{world}", "rccl_world_size": (dist.get_world_size() if dist is not None else 1),
                   "renders_per_s_per_rank": per_rank, "ranks": ranks_view,
                   "gather": ("one rccl all_gather of the step's rendered images, on its own HIGH-priority stream behind the step's "
                              "forward" if (dist is not None and state["gather"]) else "none"),
                   "gather_bytes_per_step_per_rank": B * H * W * 3 * 4,
                   "gather_ingest_bytes_per_step_per_rank": (world - 1) * B * H * W * 3 * 4,
                   "gradient_zero_fill": ("inside the projection launch (gsgen_frame_geometry_batch_zero)" if fused_fill
                                          else "torch fill kernel between forward and backward")},
        "timing": {"repeats": len(regions), "reported": "median repeat", "renders_per_s_min": world * B * K / max(els),
                   "renders_per_s_max": world * B * K / min(els), "timed_region_s": el,
                   "host_enqueue_ms_per_step": med["host"] / K * 1e3,
                   "host_enqueue_us_per_step_by_call": {k: v / K * 1e6 for k, v in med["host_by_call"].items()}},
        "roofline": {"bound": "hbm", "kernel": bwd_name, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "alg_bytes_per_launch": B * parts["composite_bwd"], "views_per_launch": B, "avg_launch_ms": bwd_ms,
                     "launches_in_flight": len(slots),
                     "avg_launch_ms_is": ("the interval between two events of the step's stream around the backward launch(es): with "
                                          f"{len(slots)} steps in flight it contains the launch's wait for a chip the other steps' kernels fill; "
                                          "the kernel's own duration in flight is what the kernel trace averages, alone_launch_ms what it "
                                          "takes with the chip to itself"),
                     "kernel_trace_avg_launch_ms": trace_ms, "kernel_trace_source": trace_src,
                     "alone_launch_ms": alone["bwd_launch_ms"],
                     "alone_achieved": B * parts["composite_bwd"] / (alone["bwd_launch_ms"] * 1e-3) / 1e9,
                     "alone_frac": B * parts["composite_bwd"] / (alone["bwd_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "fwd_kernel": fwd_name, "fwd_launch_ms": fwd_ms, "alone_fwd_launch_ms": alone["fwd_launch_ms"],
                     "fwd_GBs": B * parts["composite_fwd"] / (fwd_ms * 1e-3) / 1e9,
                     "whole_render_alg_bytes": total_b,
                     "whole_render_hbm_frac": total_b * (value / world) / (HBM_PEAK_GBS * 1e9)},
    }
    if one is not None and "walked_fraction_of_D" in one:
        # "honest bytes": the list entries the kernels really walk (early termination), not the D of the formula
        wf = one["walked_fraction_of_D"]
        wb = (4 + 4 * F) * D * wf + 28 * P + 4 * F * D * wf
        res["roofline"]["walked_fraction_of_D"] = wf
        res["roofline"]["walked_bytes_per_launch"] = B * wb
        res["roofline"]["walked_achieved_GBs"] = B * wb / (bwd_ms * 1e-3) / 1e9
    if valu_floor is not None:
        # the kernel is bound by vector-ALU issue, not HBM: the time it would take if every
        # SIMD issued its share of the measured vector instructions back to back
        res["roofline"]["valu_floor_ms"] = valu_floor
        res["roofline"]["alone_valu_frac"] = valu_floor / alone["bwd_launch_ms"]
    f_traffic, f_floor, _ = committed_traffic(args.config, fwd_name, B)  # ... and the same for the forward
    if f_floor is not None:
        res["roofline"]["fwd_traffic"] = f_traffic
        res["roofline"]["fwd_valu_floor_ms"] = f_floor
        res["roofline"]["alone_fwd_valu_frac"] = f_floor / alone["fwd_launch_ms"]
    if no_gather is not None:
        res["value_no_gather"] = no_gather["value"]
        res["no_gather"] = dict(no_gather, gather_cost_fraction=1.0 - value / no_gather["value"],
                                what="the same timed region without the per-step all_gather: compute scaling alone")
    # what N GPUs should deliver if camera sharding scales perfectly and the gather is NOT hidden -- labelled projected;
    # the driver's SCALE record is the measurement.  Per GPU and step the gather ingests (N - 1) x the rank's image bytes over
    # (N - 1) of its 7 xGMI links at once (direct all-gather; MI355X_MICROARCH.md: ~153 GB/s per link and direction, ~75 % of
    # which RCCL usually sustains)
    per_gpu = (no_gather["value"] if no_gather is not None else value) / world
    link = 153e9 * 0.75
    proj = {}
    for n_ in (1, 2, 4, 8):
        t_step = B / per_gpu
        t_gather = 0.0 if n_ == 1 else (B * H * W * 12) / link  # (n_ - 1) peers, one link each, concurrently: one image set per link
        proj[str(n_)] = {"perfect_scaling": n_ * per_gpu, "gather_hidden_behind_backward": n_ * per_gpu,
                         "gather_not_hidden_at_all": n_ * B / (t_step + t_gather),
                         "gather_ms_per_step_at_link_rate": t_gather * 1e3,
                         "xgmi_ingest_GBs_per_gpu_needed_to_hide": (n_ - 1) * B * H * W * 12 / t_step / 1e9}
    res["projected"] = {"label": "PROJECTED from this run's per-GPU rate (not measured): renders/s at N GPUs", "per_gpu_renders_per_s": per_gpu,
                        "assumed_link_GBs": link / 1e9, "by_n_gpus": proj}
    if one_step is not None:
        res["one_step_in_flight"] = dict(one_step, what="the same steps on ONE stream: a strictly sequential optimiser's view "
                                                        "(every step waits for the previous one's gradients)")
    if heads is not None:
        res["heads_path"] = heads
    if exact_basis is not None:
        res["exact_basis"] = exact_basis
    if surface is not None:
        res["autograd_surface"] = surface
    if model_views is not None:
        res["model_surface"] = model_views.get("model_surface", model_views)
        if "dropin_gs_surface" in model_views:
            res["dropin_gs_surface"] = model_views["dropin_gs_surface"]
    if one is not None:
        res["one_render_in_flight"] = one
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(sc, cams, C)
        if world == 1 and args.config == "cfg2" and not args.no_other_configs and args.focal_scale == 1.0 and args.outlier_fraction == 0.0:
            res["other_configs"] = other_configs()
            res["trainer_step"] = trainer_step()
        print(json.dumps(res), file=json_out, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
