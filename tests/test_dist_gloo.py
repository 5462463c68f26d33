"""world_size-2 (and 3) CPU test of the camera-sharded path over gloo: every rank renders its
contiguous shard (the CPU oracle stands in for the GPU rasterizer -- tests may use it), one
all_gather assembles the batch, and the result equals rendering all cameras in one process."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import scenes
from oracle import oracle as O
from gsgen_amd import dist as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _render(sc, cam):
    g = scenes.oracle_geometry(sc, cam)
    m = g["mask"]
    img, _ = O.render_rgb_fwd(g["mean2d"], g["cov2d"], sc["color"][m], sc["alpha"][m], g["start"], g["end"], g["ids"],
                              cam.topleft, 1 / cam.fx, 1 / cam.fy, cam.h, cam.w)
    return torch.from_numpy(img)


def _cams(n):
    return [scenes.Camera(48, 32, fx=40.0, c2w=scenes.orbit(2.5, 10.0, 360.0 * i / n)) for i in range(n)]


def _worker(rank, world, port, n_cams, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = scenes.random_scene(200, seed=5, svec=0.08)
    cams = _cams(n_cams)
    out = D.render_batch_sharded(lambda c: _render(sc, c), cams)
    # the whole shard in one call (what BatchRenderer.render wants) must gather to the same batch
    out2 = D.render_cameras_sharded(lambda cs: torch.stack([_render(sc, c) for c in cs], 0), cams)
    assert torch.equal(out, out2)
    if rank == world - 1:
        q.put(out.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_cams", [(2, 4), (2, 5), (3, 4)])
def test_camera_sharding_matches_single_process(world, n_cams):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_cams, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sc = scenes.random_scene(200, seed=5, svec=0.08)
    want = np.stack([_render(sc, c).numpy() for c in _cams(n_cams)])
    assert got.shape == want.shape and np.array_equal(got, want)


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 64, 65):
        for w in (1, 2, 3, 8):
            spans = [D.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    grads = [torch.randn(50, 3, generator=g), torch.randn(50, generator=g), None, torch.randn(50, 3, 16, generator=g)]
    D.allreduce_gradients(grads)
    if rank == 0:
        q.put([x.numpy() if x is not None else None for x in grads])
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_gradient_allreduce():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = None
    for r in range(world):
        g = torch.Generator().manual_seed(100 + r)
        cur = [torch.randn(50, 3, generator=g), torch.randn(50, generator=g), None, torch.randn(50, 3, 16, generator=g)]
        want = cur if want is None else [a + b if a is not None else None for a, b in zip(want, cur)]
    for a, b in zip(got, want):
        if b is None:
            assert a is None
        else:
            assert np.allclose(a, (b / world).numpy(), atol=1e-6)


def _flat_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gsgen_amd.optim import FusedAdam
    fields = {"mean": torch.zeros(20, 3), "alpha": torch.zeros(20), "sh": torch.zeros(20, 3, 4)}
    fa = FusedAdam(fields, {k: 1e-3 for k in fields})
    # every rank back-propagates its own cameras' loss into the SAME flat gradient layout
    loss = sum(((rank + 1) * (i + 1)) * p.sum() for i, p in enumerate(fa.params.values()))
    loss.backward()
    fa.all_reduce_grad()
    if rank == 1:
        q.put(fa.grad.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def test_flat_parameter_buffer_one_allreduce():
    """optim.FusedAdam: field gradients are views of one flat buffer, reduced with ONE collective"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_flat_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.concatenate([np.full(60, 1.5 * 1), np.full(20, 1.5 * 2), np.full(240, 1.5 * 3)]).astype(np.float32)
    assert np.array_equal(got, want)


def _gather_grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_total = 5  # uneven shards at world 2: 3 + 2
    lo, hi = D.shard_bounds(n_total, rank, world)
    torch.manual_seed(0)
    full = torch.randn(n_total, 4, 6, 3)
    local = full[lo:hi].clone().requires_grad_(True)
    gathered = D.gather_images(local, n_total)
    assert torch.equal(gathered.detach(), full)
    w = torch.arange(gathered.numel(), dtype=torch.float32).reshape(gathered.shape)
    (gathered * w).sum().backward()  # every rank evaluates the loss on the whole batch
    q.put((rank, local.grad.numpy(), w[lo:hi].numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_gathered_batch_is_differentiable_wrt_the_local_shard():
    """a loss on the gathered batch must send gradients back to the rank's own images (as it does at world
    size 1), instead of silently dropping the local graph"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, grad, want in got:
        assert np.array_equal(grad, want), rank


def _gather_allreduce_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_total = 5
    lo, hi = D.shard_bounds(n_total, rank, world)
    torch.manual_seed(0)
    theta = torch.randn(7, requires_grad=True)          # the replicated parameters
    basis = torch.randn(n_total, 4, 6, 3, 7)            # "rendering": image_i = basis_i . theta
    target = torch.randn(n_total, 4, 6, 3)
    local = basis[lo:hi] @ theta
    gathered = D.gather_images(local, n_total)
    ((gathered - target) ** 2).mean().backward()        # the SAME loss on every rank, on the whole batch
    g_sum, g_avg = theta.grad.clone(), theta.grad.clone()
    D.allreduce_gradients([g_sum], average=False)
    D.allreduce_gradients([g_avg], average=True)
    q.put((rank, g_sum.numpy(), g_avg.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_gathered_loss_composes_with_a_summed_gradient_all_reduce():
    """a loss on the gathered batch followed by allreduce_gradients(average=False) is the world-size-1 gradient
    (every rank holds its cameras' share); the averaging form, meant for per-rank losses, would hand back 1 / world of it"""
    torch.manual_seed(0)
    theta = torch.randn(7, requires_grad=True)
    basis = torch.randn(5, 4, 6, 3, 7)
    target = torch.randn(5, 4, 6, 3)
    (((basis @ theta) - target) ** 2).mean().backward()
    want = theta.grad.numpy()
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_allreduce_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, g_sum, g_avg in got:
        assert np.allclose(g_sum, want, rtol=1e-5, atol=1e-6), rank
        assert np.allclose(g_avg * world, want, rtol=1e-5, atol=1e-6), rank


def test_bench_camera_shards_partition_the_64_poses():
    """bench.py --gpus N (BASELINE configs[3]): the ranks' camera lists are the contiguous shards of ONE seeded set of
    64 poses -- disjoint, complete, in order; and the self-launch command starts N ranks on 127.0.0.1"""
    import sys
    sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
    import bench
    allc = bench.random_pose_cameras(64, 0, 1, 512, 512)
    for world in (2, 4, 8, 3):
        parts = [bench.random_pose_cameras(64, r, world, 512, 512) for r in range(world)]
        assert sum(len(p) for p in parts) == 64
        flat = [c for p in parts for c in p]
        for a, b in zip(flat, allc):
            assert a.fx == b.fx and np.array_equal(a.c2w, b.c2w)
    # elevation stays within [-20, 90] degrees and the focal within [0.7, 1.35] x resolution (data/__init__.py:151-205)
    for c in allc:
        pos = c.c2w[:, 3]
        elev = np.rad2deg(np.arcsin(pos[2] / np.linalg.norm(pos)))
        assert -20.5 <= elev <= 90.0 and 0.7 * 512 <= c.fx <= 1.35 * 512 and 1.99 <= np.linalg.norm(pos) <= 2.51


def _densify_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import types
    g = torch.Generator().manual_seed(7 + rank)
    st = types.SimpleNamespace(max_radii2d=torch.rand(40, generator=g), grad_accum=torch.rand(40, generator=g),
                               cnt=torch.randint(0, 3, (40,), generator=g).float())
    D.allreduce_densify_stats(st)
    q.put((rank, st.max_radii2d.numpy(), st.grad_accum.numpy(), st.cnt.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_densify_statistics_are_identical_on_every_rank():
    """the densify / prune policy must decide identically on every rank -- the per-rank statistics are
    combined as one process rendering all cameras would have left them (max / sum / sum), bit-identical everywhere"""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_densify_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    per = []
    for r in range(world):
        g = torch.Generator().manual_seed(7 + r)
        per.append((torch.rand(40, generator=g), torch.rand(40, generator=g), torch.randint(0, 3, (40,), generator=g).float()))
    want_max = torch.stack([p_[0] for p_ in per]).max(0).values.numpy()
    want_cnt = torch.stack([p_[2] for p_ in per]).sum(0).numpy()
    for rank, mx, ga, cnt in got:
        assert np.array_equal(mx, want_max) and np.array_equal(cnt, want_cnt)
        assert np.array_equal(mx, got[0][1]) and np.array_equal(ga, got[0][2])  # the same bits on every rank
        assert np.allclose(ga, torch.stack([p_[1] for p_ in per]).sum(0).numpy(), rtol=1e-6)


def test_bench_accounting_and_launch_shape():
    """bench.py's host-side arithmetic (no GPU): SURVEY 8(d)'s algorithmic bytes, and the cameras-per-launch /
    launches-in-flight rule on the four BASELINE workloads' measured pair counts."""
    import bench
    total, parts = bench.b_alg_bytes(N=79_549, D=713_016, P=640_000, T=2_500, F=55)
    assert parts["composite_bwd"] == (4 + 4 * 55) * 713_016 + 28 * 640_000 + 4 * 55 * 713_016
    assert parts["composite_fwd"] == (4 + 4 * 55) * 713_016 + 16 * 640_000
    assert total == sum(parts.values()) and abs(total / 0.5457e9 - 1) < 0.01  # 0.546 GB per render at cfg2
    assert bench.choose_batch_and_slots(713_016) == (8, 3)      # cfg2: 5.7 M pairs per launch, three steps in flight
    assert bench.choose_batch_and_slots(2_470_000) == (4, 3)    # cfg3: four cameras per launch (9.9 M pairs), three in flight
    assert bench.choose_batch_and_slots(440_000) == (8, 3)      # cfg4: light launches
    assert bench.choose_batch_and_slots(2_700) == (8, 3)        # cfg1
    assert bench.choose_batch_and_slots(713_016, batch=4, slots=1) == (4, 1)


@pytest.mark.parametrize("world", [2, 3])
def test_bench_step_loop_dry_run_with_several_ranks(world):
    """nobody has run bench.py with more than one rank before the driver's 8-GPU run.  `--dry-run-lib` runs
    bench.py's OWN step loop -- slots, streams / events (stand-ins), the broadcast of the launch shape, the per-step
    all_gather of the rendered images behind each forward, the max-over-ranks timing, the reductions of the report -- with
    the host build of the kernels and gloo: a hang, a shape mismatch or a collective that not every rank reaches fails here.
    The ranks' cameras differ (camera sharding), the routed SH kernels take their polynomial form (narrow views)."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "-s", "emu"])
    emu = os.path.join(root, "oracle", "_build", "libgsgen_emu.so")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--dry-run-lib", emu, "--steps", "2",
                        "--warmup", "1"], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # ONE JSON line, from rank 0
    res = json.loads(lines[0])
    cfg = res["config"]
    assert res["n_gpus"] == world and cfg["rccl_world_size"] == world and res["steps"] == 2 and res["scaling"] == "weak"
    per = cfg["renders_per_s_per_rank"]
    assert len(per) == world and all(v > 0 for v in per)
    assert res["value"] <= sum(per) * 1.0001  # whole-job throughput over the SLOWEST rank's time
    assert f"{2 * world} of {2 * world} cameras" in cfg["sh_basis"] and "POLY6" in res["roofline"]["kernel"]
    assert "all_gather" in cfg["gather"]


def test_bench_step_loop_dry_run_eight_ranks_on_the_cfg4_camera_split():
    """the driver's scaling run starts EIGHT ranks; until now only 2 and 3 had ever run bench.py's step loop.  The
    same dry run (gloo + the host build of the kernels) with 8 ranks on BASELINE configs[3]'s camera partition: the 64 random
    poses split 8 per rank, one 8-camera step per rank and slot, the per-step all_gather of [8 ranks x 8 cameras] images, and the
    second timed region without the gather (`value_no_gather`: compute scaling and xGMI cost separate in the driver's record)."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "-s", "emu"])
    emu = os.path.join(root, "oracle", "_build", "libgsgen_emu.so")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--config", "cfg4", "--dry-run-lib", emu,
                        "--steps", "2", "--warmup", "1"], cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    cfg = res["config"]
    assert res["n_gpus"] == 8 and cfg["rccl_world_size"] == 8 and cfg["cameras_per_step"] == 8
    assert len(cfg["renders_per_s_per_rank"]) == 8 and all(v > 0 for v in cfg["renders_per_s_per_rank"])
    assert "64 of 64 cameras" in cfg["sh_basis"]  # every rank's 8 poses, counted over the job
    assert "all_gather" in cfg["gather"] and cfg["gather_bytes_per_step_per_rank"] == 8 * 24 * 40 * 3 * 4
    assert cfg["gather_ingest_bytes_per_step_per_rank"] == 7 * cfg["gather_bytes_per_step_per_rank"]
    assert res["value_no_gather"] > 0 and "gather_cost_fraction" in res["no_gather"]
    assert set(res["projected"]["by_n_gpus"]) == {"1", "2", "4", "8"} and "PROJECTED" in res["projected"]["label"]
