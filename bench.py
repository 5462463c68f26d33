#!/usr/bin/env python
"""bench.py -- fwd+bwd renders/sec of the rasterizer hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg3|cfg1] [--no-cpu-baseline]

One "step" = one fused render of one camera: frustum cull + EWA projection + 16x16 tile
binning with per-tile depth sort + SH(degree 3) front-to-back compositing, then the backward
pass to mean[N,3], qvec[N,4], svec[N,3], alpha[N], sh[N,3,16] for a dense random grad_out
(SURVEY.md 8d).  Workload at N=1: BASELINE.json configs[1] -- 100k Gaussians ("Point-E init"
cloud), 800x800, SH degree 3.  Inputs are resident in HBM before the timed region.  With
--gpus N each rank renders its own cameras (camera sharding, weak scaling) and the rendered
images are all-gathered over RCCL each step (north_star: "RCCL only to gather rendered
images").  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def make_workload(name):
    import scenes
    if name == "cfg2":
        sc = scenes.pointe_scene(100_000, seed=0, svec=0.02, C=4)
        W = H = 800
    elif name == "cfg3":
        sc = scenes.densified_scene(500_000, seed=0, C=4)
        W = H = 1024
    elif name == "cfg4":
        sc = scenes.pointe_scene(100_000, seed=0, svec=0.02, C=4)
        W = H = 512
    elif name == "cfg1":
        sc = scenes.random_scene(1000, seed=0, C=1)
        W = H = 256
    else:
        raise SystemExit(f"unknown config {name}")
    return sc, W, H


def random_pose_cameras(n_total, rank, world, W, H, seed=0):
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
    focal = rng.uniform(0.7, 1.35, n_total) * W
    a, b = shard_bounds(n_total, rank, world)
    return [scenes.Camera(W, H, fx=float(focal[i]), c2w=scenes.orbit(float(dist_[i]), float(min(elev[i], 89.0)), float(azim[i])))
            for i in range(a, b)]


def camera_poses(n, rank, W, H):
    import scenes
    cams = []
    for i in range(n):
        az = 30.0 + 45.0 * i + 7.0 * rank
        cams.append(scenes.Camera(W, H, fx=float(W), c2w=scenes.orbit(2.5, 15.0, az)))
    return cams


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
    return {"value": n / el, "unit": "renders/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"{n} whole fwd+bwd renders of the bench workload through oracle/gs_oracle.c "
                      f"(OpenMP, {os.cpu_count()} threads) in {el:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)   # 50 batched launches: the pipeline's fill / drain is < 2 % of it
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="also time every stage separately")
    ap.add_argument("--segments", type=int, default=int(os.environ.get("GSGEN_SEGMENTS", "1")),
                    help="backward workgroups per tile in the timed (throughput) region: segments of 32 list entries; "
                         "1 = one workgroup per tile (best with several renders in flight)")
    ap.add_argument("--latency-segments", type=int, default=8,
                    help="same for the one-render-in-flight pass (uniform work units shorten a lone launch's tail)")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("GSGEN_STREAMS", "3")),
                    help="independent renders in flight (HIP streams, own buffers each)")
    ap.add_argument("--batch", type=int, default=int(os.environ.get("GSGEN_BATCH", "8")),
                    help="cameras per compositing launch (gsgen_vol_render_sh_batch: gridDim.y = cameras); 1 = one "
                         "launch per camera.  A step is still one render: K steps run as ceil(K / batch) launches")
    ap.add_argument("--batch-slots", type=int, default=2, help="batches in flight (own stream and buffers each)")
    args = ap.parse_args()

    import torch
    from gsgen_amd import _capi, renderer as R

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    lib = _capi.load()
    sc, W, H = make_workload(args.config)
    C = sc["C"]
    N = sc["mean"].shape[0]
    cams = random_pose_cameras(64, rank, world, W, H) if args.config == "cfg4" else camera_poses(8, rank, W, H)
    cis = [R.CameraInfo(*c.intr) for c in cams]
    ci = cis[0]
    nth, ntw = R.n_tiles(H, W)
    t = {k: torch.tensor(sc[k], device=dev) for k in ("mean", "qvec", "svec", "alpha", "sh")}
    cam_dev = [torch.from_numpy(ci_.pack(c.c2w)).to(dev) for ci_, c in zip(cis, cams)]
    rot_dev = [torch.from_numpy(np.ascontiguousarray(c.c2w[:3, :3]).reshape(-1).copy()).to(dev) for c in cams]
    topleft_dev = [torch.from_numpy(c.topleft).to(dev) for c in cams]
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    grad_out = torch.randn(H, W, 3, device=dev)
    CC3 = 3 * C * C
    p = lambda x: x.data_ptr()  # noqa: E731
    gathered = torch.empty(world, H, W, 3, device=dev) if world > 1 else None

    # Independent renders (different cameras of a batch) are issued round-robin on `--streams`
    # HIP streams, each with its own frame buffers, so that the tail of one render's
    # compositing launch (the image-centre tiles) overlaps the next render's kernels.  Every
    # render still runs its complete cull->...->backward chain in order on its stream.
    class Slot:
        def __init__(self, stream):
            self.stream = stream
            self.s = stream.cuda_stream
            with torch.cuda.stream(stream):
                self.buf = R.FrameBuffers(N, W, H, dev)
                self.out = torch.empty(H, W, 3, device=dev)
                self.gflat = torch.empty(N * (7 + CC3), device=dev)  # mean2d(2) | cov2d(4) | alpha(1) | sh
                self.seg_ws = torch.empty(lib.segment_workspace_bytes(nth * ntw, max(args.segments, args.latency_segments)), device=dev, dtype=torch.uint8)
                self.g_mean = torch.empty(N, 3, device=dev)
                self.g_qvec = torch.empty(N, 4, device=dev)
                self.g_svec = torch.empty(N, 3, device=dev)
            g = self.gflat
            self.g_mean2d, self.g_cov2d, self.g_alpha, self.g_sh = g[:2 * N], g[2 * N:6 * N], g[6 * N:7 * N], g[7 * N:]

    n_streams = max(1, args.streams)
    slots = [Slot(torch.cuda.current_stream(dev) if n_streams == 1 else torch.cuda.Stream(dev)) for _ in range(n_streams)]
    torch.cuda.synchronize()
    buf = slots[0].buf

    def step(i, timed=None, slot=None, gather=True, nseg=None):
        nseg = args.segments if nseg is None else nseg
        k = i % len(cams)
        sl = slots[(i % n_streams) if slot is None else slot]
        b_, s, stream = sl.buf, sl.s, sl.stream
        order = None if os.environ.get("GSGEN_NO_ORDER") else b_.tile_order()
        topleft, psx, psy = topleft_dev[k], 1.0 / cis[k].fx, 1.0 / cis[k].fy
        lib.frame_geometry(N, p(t["mean"]), p(t["qvec"]), p(t["svec"]), p(cam_dev[k]), W, H, b_.D_cap,
                           p(b_.mean2d), p(b_.cov2d), p(b_.depth), p(b_.mask), p(b_.ids), p(b_.start),
                           p(b_.end), p(b_.total), p(b_.ws), b_.ws.numel(), s)
        if timed is not None:
            timed[0].record(stream)
        lib.vol_render_sh_segmented(N, b_.D_cap, p(b_.mean2d), p(b_.cov2d), p(t["sh"]), p(t["alpha"]), p(b_.start),
                                    p(b_.end), p(b_.ids), p(sl.out), p(topleft), p(rot_dev[k]), 16, nth, ntw, psx, psy,
                                    H, W, C, 1e-4, p(bg), None, order, p(sl.seg_ws), nseg, s)
        if timed is not None:
            timed[1].record(stream)
        with torch.cuda.stream(stream):
            sl.gflat.zero_()
        if timed is not None:
            timed[2].record(stream)
        lib.vol_render_backward_sh_segmented(N, b_.D_cap, p(b_.mean2d), p(b_.cov2d), p(t["sh"]), p(t["alpha"]),
                                             p(b_.start), p(b_.end), p(b_.ids), p(sl.out), p(sl.g_mean2d), p(sl.g_cov2d),
                                             p(sl.g_sh), p(sl.g_alpha), p(grad_out), p(topleft), p(rot_dev[k]), 16, nth,
                                             ntw, psx, psy, H, W, C, 1e-4, p(bg), order, p(sl.seg_ws), nseg, s)
        if timed is not None:
            timed[3].record(stream)
        lib.project_gaussians_backward_masked(N, p(t["mean"]), p(t["qvec"]), p(t["svec"]), p(cam_dev[k]), 1,
                                              p(b_.mask), p(sl.g_mean2d), p(sl.g_cov2d), None, p(sl.g_mean),
                                              p(sl.g_qvec), p(sl.g_svec), s)
        if gathered is not None and gather:
            with torch.cuda.stream(stream):
                dist.all_gather_into_tensor(gathered, sl.out)

    # --batch B > 1: the B cameras of a batch share ONE forward and ONE backward compositing launch
    # (gridDim.y = B) and one set of parameter gradients -- what a training step over a camera batch
    # needs (gs/gaussian_splatting.py:1423-1466 loops the cameras, autograd sums their gradients).
    # Geometry / binning and the projection backward stay per camera.  Batches alternate over
    # --batch-slots streams so one batch's geometry overlaps the other's compositing.
    B = max(1, args.batch)

    class BatchSlot:
        def __init__(self, stream):
            self.stream, self.s = stream, stream.cuda_stream
            with torch.cuda.stream(stream):
                self.bufs = [R.FrameBuffers(N, W, H, dev) for _ in range(B)]
                self.out = torch.empty(B, H, W, 3, device=dev)
                # per view: mean2d(2) | cov2d(4); shared: alpha(1) | sh -- zeroed once per batch; the
                # projection backward overwrites mean(3) | qvec(4) | svec(3)
                self.gflat = torch.empty(B * 6 * N + N * (1 + CC3), device=dev)
                self.g3d = torch.empty(N * 10, device=dev)
                self.seg_ws = [torch.empty(max(1, lib.segment_workspace_bytes(nth * ntw, args.segments)), device=dev,
                                           dtype=torch.uint8) for _ in range(B)]
                self.bws = torch.empty(lib.sh_batch_workspace_bytes(B), device=dev, dtype=torch.uint8)
                self.gws = torch.empty(lib.frame_batch_workspace_bytes(B), device=dev, dtype=torch.uint8)
            o = B * 6 * N
            g = self.gflat
            self.g_alpha, self.g_sh = g[o:o + N], g[o + N:o + N * (1 + CC3)]
            g = self.g3d
            self.g_mean, self.g_qvec, self.g_svec = g[:3 * N], g[3 * N:7 * N], g[7 * N:]
            self.views = {}

        def geometry_array(self, k0, nb):
            key = ("geo", k0, nb, tuple(b_.D_cap for b_ in self.bufs[:nb]))
            if key not in self.views:
                arr = (_capi.GeometryView * nb)()
                for i in range(nb):
                    k, b_, a = (k0 + i) % len(cams), self.bufs[i], arr[i]
                    a.cam, a.mean2d, a.cov2d, a.depth, a.mask = p(cam_dev[k]), p(b_.mean2d), p(b_.cov2d), p(b_.depth), p(b_.mask)
                    a.gaussian_ids, a.start, a.end, a.total = p(b_.ids), p(b_.start), p(b_.end), p(b_.total)
                    a.workspace, a.workspace_bytes, a.D_cap = p(b_.ws), b_.ws.numel(), b_.D_cap
                self.views[key] = arr
            return self.views[key]

        def projection_tables(self, k0, nb):
            """(c2w[], detach_depth, mask[], g_mean2d[], g_cov2d[], g_depth[]) of gsgen_project_gaussians_backward_batch"""
            key = ("proj", k0, nb)
            if key not in self.views:
                import ctypes
                tab = lambda vals: (ctypes.c_void_p * nb)(*vals)  # noqa: E731
                g0 = p(self.gflat)
                self.views[key] = (tab([p(cam_dev[(k0 + i) % len(cams)]) for i in range(nb)]), 1,
                                   tab([p(self.bufs[i].mask) for i in range(nb)]),
                                   tab([g0 + 4 * 6 * N * i for i in range(nb)]),
                                   tab([g0 + 4 * 6 * N * i + 4 * 2 * N for i in range(nb)]), None)
            return self.views[key]

        def view_array(self, k0, nb):
            key = (k0, nb, tuple(b_.D_cap for b_ in self.bufs[:nb]))
            if key not in self.views:
                arr = (_capi.ShView * nb)()
                for i in range(nb):
                    k, b_, a = (k0 + i) % len(cams), self.bufs[i], arr[i]
                    a.mean, a.cov, a.start, a.end, a.gaussian_ids = p(b_.mean2d), p(b_.cov2d), p(b_.start), p(b_.end), p(b_.ids)
                    a.tile_order = None if os.environ.get("GSGEN_NO_ORDER") else b_.tile_order()
                    a.topleft, a.c2w, a.bg_rgb = p(topleft_dev[k]), p(rot_dev[k]), p(bg)
                    a.pixel_size_x, a.pixel_size_y = 1.0 / cis[k].fx, 1.0 / cis[k].fy
                    a.out, a.T = p(self.out[i]), None
                    a.segment_workspace = p(self.seg_ws[i]) if args.segments > 1 else None
                    a.grad_out = p(grad_out)
                    a.grad_mean = p(self.gflat) + 4 * 6 * N * i
                    a.grad_cov = a.grad_mean + 4 * 2 * N
                self.views[key] = arr
            return self.views[key]

    bslots = [BatchSlot(torch.cuda.Stream(dev)) for _ in range(max(1, args.batch_slots))] if B > 1 else []
    gathered_b = torch.empty(world, B, H, W, 3, device=dev) if (world > 1 and B > 1) else None

    def batch_step(j, k0, nb, timed=None, gather=True):
        """renders cameras k0 .. k0+nb-1 (mod the camera set) as batch j"""
        sl = bslots[j % len(bslots)]
        s, stream = sl.s, sl.stream
        if os.environ.get("GSGEN_GEO_PER_VIEW"):
            for i in range(nb):
                k, b_ = (k0 + i) % len(cams), sl.bufs[i]
                lib.frame_geometry(N, p(t["mean"]), p(t["qvec"]), p(t["svec"]), p(cam_dev[k]), W, H, b_.D_cap,
                                   p(b_.mean2d), p(b_.cov2d), p(b_.depth), p(b_.mask), p(b_.ids), p(b_.start),
                                   p(b_.end), p(b_.total), p(b_.ws), b_.ws.numel(), s)
        else:
            lib.frame_geometry_batch(nb, sl.geometry_array(k0 % len(cams), nb), N, p(t["mean"]), p(t["qvec"]),
                                     p(t["svec"]), W, H, p(sl.gws), s)
        arr = sl.view_array(k0 % len(cams), nb)
        if timed is not None:
            timed[0].record(stream)
        lib.vol_render_sh_batch(nb, arr, N, p(t["sh"]), p(t["alpha"]), 16, nth, ntw, H, W, C, 1e-4, args.segments,
                                p(sl.bws), s)
        if timed is not None:
            timed[1].record(stream)
        with torch.cuda.stream(stream):
            sl.gflat.zero_()
        if timed is not None:
            timed[2].record(stream)
        lib.vol_render_backward_sh_batch(nb, arr, N, p(t["sh"]), p(t["alpha"]), p(sl.g_sh), p(sl.g_alpha), 16, nth, ntw,
                                         H, W, C, 1e-4, args.segments, p(sl.bws), s)
        if timed is not None:
            timed[3].record(stream)
        lib.project_gaussians_backward_batch(nb, N, p(t["mean"]), p(t["qvec"]), p(t["svec"]),
                                             *sl.projection_tables(k0 % len(cams), nb), p(sl.g_mean), p(sl.g_qvec),
                                             p(sl.g_svec), s)
        if gathered_b is not None and gather:
            with torch.cuda.stream(stream):
                dist.all_gather_into_tensor(gathered_b, sl.out)

    def run_steps(first, count, evs_):
        """`count` renders starting at render index `first`: one launch chain per render (B == 1) or per
        batch of B consecutive cameras"""
        used = []
        if B == 1:
            for i in range(count):
                step(first + i, evs_[i] if evs_ is not None else None)
                used.append(i)
            return used
        done, j = 0, first // B
        while done < count:
            nb = min(B, count - done)
            batch_step(j, first + done, nb, evs_[done] if evs_ is not None else None)
            used.append(done)
            done += nb
            j += 1
        return used

    # size the pair buffers once, outside the timed region (one sync)
    for sl in bslots:
        for k0 in range(len(cams)):  # every buffer meets every camera
            for _ in range(2):
                batch_step(bslots.index(sl), k0, B, gather=False)
                torch.cuda.synchronize()
                if all([b_.ensure_capacity() for b_ in sl.bufs]):
                    break
            else:
                raise AssertionError("pair buffers still too small after growing")
    Ds = []
    for sidx in range(n_streams):
        for k in range(len(cams)):
            step(k, slot=sidx, gather=False)  # (data-dependent retries: no collectives in here)
            torch.cuda.synchronize()
            if not slots[sidx].buf.ensure_capacity():
                step(k, slot=sidx, gather=False)
                torch.cuda.synchronize()
                assert slots[sidx].buf.ensure_capacity()
            if sidx == 0:
                Ds.append(int(slots[0].buf.total.item()))
    n_vis = int(buf.mask.sum().item())

    run_steps(0, args.warmup, None)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # HIP events around the dominant kernel (composite backward) and the forward, every step
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    launched = run_steps(args.warmup, args.steps, evs)
    host_el = time.perf_counter() - t0  # host time to enqueue everything (launch-bound if ~= el)
    barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())

    # the same K steps again, strictly one render at a time on one stream (latency view)
    one = None
    if n_streams > 1:
        ev1 = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
        barrier()
        t1 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i, ev1[i], slot=0, nseg=args.latency_segments)
        barrier()
        el1 = time.perf_counter() - t1
        # ... and once more replayed from one captured hipGraph per camera: same kernels, no launch gaps
        graph = None
        try:
            sl0 = slots[0]
            graphs = []
            with torch.cuda.stream(sl0.stream):
                for k in range(len(cams)):
                    gk = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gk, stream=sl0.stream):
                        step(k, slot=0, gather=False, nseg=args.latency_segments)
                    graphs.append(gk)
                for i in range(args.warmup):
                    graphs[i % len(cams)].replay()
            barrier()
            t2 = time.perf_counter()
            with torch.cuda.stream(sl0.stream):
                for i in range(args.steps):
                    graphs[(args.warmup + i) % len(cams)].replay()
            barrier()
            el2 = time.perf_counter() - t2
            graph = {"value": world * args.steps / el2, "ms_per_step": el2 / args.steps * 1e3,
                     "note": "one captured hipGraph per camera, replayed back to back on one stream"
                             + (" (no image gather inside the graph)" if world > 1 else "")}
        except Exception as e:  # capture is an optimisation of the latency view only
            graph = {"error": str(e)[:200]}
        one = {"value": world * args.steps / el1, "ms_per_step": el1 / args.steps * 1e3, "hipgraph_replay": graph,
               "backward_segments_per_tile": args.latency_segments,
               "fwd_kernel_ms": float(np.mean([e[0].elapsed_time(e[1]) for e in ev1])),
               "bwd_kernel_ms": float(np.mean([e[2].elapsed_time(e[3]) for e in ev1]))}

    # batched launches once more with ONE batch in flight: the duration of a launch that has the chip to
    # itself (in the timed region above two batches share it, so each launch there takes about twice as long)
    alone = None
    if B > 1:
        nb_alone = max(2, min(8, args.steps // B))
        eva = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(nb_alone)]
        barrier()
        for j in range(nb_alone):
            batch_step(0, args.warmup + j * B, B, eva[j], gather=False)  # slot 0 every time: one stream
        barrier()
        alone = {"fwd_launch_ms": float(np.mean([e[0].elapsed_time(e[1]) for e in eva])),
                 "bwd_launch_ms": float(np.mean([e[2].elapsed_time(e[3]) for e in eva])), "views_per_launch": B}

    fwd_ms = float(np.mean([evs[i][0].elapsed_time(evs[i][1]) for i in launched]))
    bwd_ms = float(np.mean([evs[i][2].elapsed_time(evs[i][3]) for i in launched]))
    vpl = args.steps / len(launched)  # views per compositing launch
    D = float(np.mean([Ds[(args.warmup + i) % len(cams)] for i in range(args.steps)]))
    P, T = W * H, nth * ntw
    F = 7 + CC3
    total_b, parts = b_alg_bytes(n_vis, D, P, T, F)
    value = world * args.steps / el
    traffic, valu_floor = None, None
    try:  # per launch of the dominant kernel, from the committed PMC passes (cfg2 only)
        if args.config == "cfg2":
            pmc_all = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            if B > 1 and vpl == 8 and "k_composite_bwd_batch8" in pmc_all:  # measured on the 8-camera launch itself
                pmc = pmc_all["k_composite_bwd_batch8"]
                traffic, valu_floor = pmc["traffic_bytes"] / vpl, pmc.get("valu_floor_ms") / vpl
            else:
                pmc = pmc_all["k_composite_bwd"]
                traffic, valu_floor = pmc["traffic_bytes"], pmc.get("valu_floor_ms")
    except Exception:
        traffic, valu_floor = None, None
    # dominant kernel = composite backward
    ach = vpl * parts["composite_bwd"] / (bwd_ms * 1e-3) / 1e9
    res = {
        "metric": "fwd+bwd renders/sec (800x800, 100k Gaussians)" if args.config == "cfg2" else f"fwd+bwd renders/sec ({args.config})",
        "value": value, "unit": "renders/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": el / args.steps * 1e3, "host_enqueue_ms_per_step": host_el / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": {"cfg2": "BASELINE configs[1]: 100k-Gaussian Point-E-init cloud, 800x800, SH degree 3, fwd+bwd",
                                "cfg3": "BASELINE configs[2]: 500k post-densify Gaussians, 1024x1024, SH degree 3, fwd+bwd",
                                "cfg4": "BASELINE configs[3]: 100k Gaussians, 64 random-pose cameras at 512x512, camera-sharded",
                                "cfg1": "BASELINE configs[0]: 1k random Gaussians, 256x256, SH degree 0"}[args.config],
                   "gaussians": N, "visible_after_cull": n_vis, "image": [H, W], "sh_degree": C - 1,
                   "tile_pairs_D": D, "cameras_per_gpu_per_step": 1, "cameras_per_launch": B, "renders_in_flight": n_streams if B == 1 else B * len(bslots), "backward_segments_per_tile": args.segments, "parallelism": f"camera-sharded x{world}",
                   "gather": "rccl all_gather of rendered images" if world > 1 else "none"},
        "roofline": {"bound": "hbm", "kernel": f"k_composite_bwd_sh_mfma<C={C},2{',batched' if B > 1 else ''}> (compositing backward, matrix-core grad_sh"
                               + (f", {vpl:g} cameras per launch)" if B > 1 else ")"), "achieved": ach, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None if traffic is None else traffic * vpl,
                     "alg_bytes_per_launch": vpl * parts["composite_bwd"], "views_per_launch": vpl, "avg_launch_ms": bwd_ms,
                     "fwd_kernel_ms": fwd_ms,
                     "fwd_kernel_GBs": vpl * parts["composite_fwd"] / (fwd_ms * 1e-3) / 1e9,
                     "whole_render_alg_bytes": total_b,
                     "whole_render_hbm_frac": total_b * (value / world) / (HBM_PEAK_GBS * 1e9)},
    }
    if one is not None:
        res["one_render_in_flight"] = one
        res["roofline"]["isolated_launch_ms"] = one["bwd_kernel_ms"]
        res["roofline"]["isolated_achieved"] = parts["composite_bwd"] / (one["bwd_kernel_ms"] * 1e-3) / 1e9
    if alone is not None:
        res["roofline"]["batches_in_flight"] = len(bslots)
        res["roofline"]["alone_launch_ms"] = alone["bwd_launch_ms"]
        res["roofline"]["alone_achieved"] = B * parts["composite_bwd"] / (alone["bwd_launch_ms"] * 1e-3) / 1e9
        res["roofline"]["alone_frac"] = res["roofline"]["alone_achieved"] / HBM_PEAK_GBS
        res["roofline"]["alone_fwd_launch_ms"] = alone["fwd_launch_ms"]
        if valu_floor is not None:
            res["roofline"]["alone_valu_frac"] = valu_floor * B / alone["bwd_launch_ms"]
    if valu_floor is not None:
        # the kernel is bound by vector-ALU issue, not HBM: time it would take if
        # every SIMD issued its share of the measured vector instructions back to back
        res["roofline"]["valu_floor_ms"] = valu_floor * vpl
        res["roofline"]["valu_frac"] = valu_floor * vpl / bwd_ms
    if args.breakdown and rank == 0:
        names = ["geometry+bin+sort", "composite_fwd", "zero_grads", "composite_bwd"]
        stage_ev = [[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in range(20)]
        for i in range(20):
            e = stage_ev[i]
            e[4].record(slots[0].stream)
            step(i, e, slot=0)
            e[5].record(slots[0].stream)
        torch.cuda.synchronize()
        bd = {"geometry+bin+sort": np.mean([e[4].elapsed_time(e[0]) for e in stage_ev]),
              "composite_fwd": np.mean([e[0].elapsed_time(e[1]) for e in stage_ev]),
              "zero_grads": np.mean([e[1].elapsed_time(e[2]) for e in stage_ev]),
              "composite_bwd": np.mean([e[2].elapsed_time(e[3]) for e in stage_ev]),
              "project_bwd(+gather)": np.mean([e[3].elapsed_time(e[5]) for e in stage_ev])}
        res["breakdown_ms"] = {k: float(v) for k, v in bd.items()}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(sc, cams, C)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
