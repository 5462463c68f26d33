"""gsgen_amd.densify (second half): the reference's densify / prune bookkeeping on the replicated parameter
set.  Against the reference's OWN Python where /root/reference exists -- its GaussianSplattingRenderer.densify() (legacy
and "official") and .prune() run on the same parameters, Adam state and statistics, with the same random stream -- and
over gloo: ranks with different camera shards end up with bit-identical clouds."""
import os
import socket
import sys
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import refshim
from gsgen_amd import densify as DN
from gsgen_amd.optim import FusedAdam
from gsgen_amd.renderer import DensifyStats

FIELDS = DN.FIELDS
RAW = {"mean": "mean", "qvec": "qvec", "svec": "svec_before_activation", "color": "color_before_activation",
       "alpha": "alpha_before_activation"}


class _Cfg(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None


def _cloud(n, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    raw = {"mean": 0.5 * r(n, 3), "qvec": torch.nn.functional.normalize(r(n, 4), dim=-1),
           "svec": torch.log(0.02 * torch.exp(0.6 * r(n, 3))), "color": r(n, 3), "alpha": 1.5 * r(n)}
    mom = {k: (0.01 * r(*v.shape), 0.01 * r(*v.shape).abs()) for k, v in raw.items()}
    cnt = torch.randint(0, 4, (n,), generator=g).float()
    acc = torch.rand(n, generator=g) * 0.06 * cnt
    maxr = torch.rand(n, generator=g) * 2.0
    return raw, mom, acc, cnt, maxr


def _ours(raw, mom, acc, cnt, maxr, dcfg, pcfg, step, step_count=17):
    opt = FusedAdam({k: raw[k].clone() for k in FIELDS}, {k: 1e-3 for k in FIELDS}, eps=1e-15)
    opt.load_moments(mom, step_count)
    st = DensifyStats(raw["mean"].shape[0], torch.device("cpu"))
    st.grad_accum.copy_(acc); st.cnt.copy_(cnt); st.max_radii2d.copy_(maxr)
    ctl = DN.AdaptiveControl(dcfg, pcfg, use_global_rng=True)
    return ctl.step(step, opt, st) + (ctl,)


@pytest.fixture(scope="module")
def refmodule():
    if not refshim.available():
        pytest.skip("/root/reference is not present")
    refshim.install()
    dm = types.ModuleType("kornia.geometry.depth")
    dm.depth_to_3d = None
    sys.modules["kornia.geometry.depth"] = dm
    sys.modules["kornia"].__path__ = []
    sys.modules["kornia.geometry"].__path__ = []
    import gs.gaussian_splatting as M
    return M


class _TorchOnCpu:
    """the reference hard-codes device="cuda" in densify_by_split (gs/gaussian_splatting.py:557,604): same calls, no device"""

    def __getattr__(self, k):
        return getattr(torch, k)

    @staticmethod
    def zeros(*a, device=None, **k):
        return torch.zeros(*a, **k)


def _reference(M, raw, mom, acc, cnt, maxr, dens, prune, step):
    cfg = _Cfg(device="cpu", svec_act="exp", alpha_act="sigmoid", color_act="sigmoid", tile_size=16, frustum_culling_radius=6.0,
               tile_culling_type="aabb", tile_culling_thresh=0.01, tile_culling_radius=6.0, T_thresh=1e-4,
               skip_frustum_culling=False, normal_as_rgb=False, debug=False, depth_detach=True,
               background=_Cfg(type="fixed", device="cpu", color=[0.0, 0.0, 0.0], random_aug=False, random_aug_prob=0.0),
               densify=_Cfg(dens), prune=_Cfg(prune))
    model = M.GaussianSplattingRenderer(cfg, {**{k: raw[k].clone() for k in FIELDS}, "raw": True})
    for f in list(FIELDS) + ["bg"]:
        setattr(model, f"{f}_lr_scheduler", lambda step: 1e-3)
    model.set_optimizer(_Cfg(type="Adam", opt_args={"eps": 1e-15}))
    for k in FIELDS:  # give Adam the state a few steps of training leave behind
        p = getattr(model, RAW[k])
        p.grad = torch.zeros_like(p)
    model.optimizer.step()
    for grp in model.optimizer.param_groups:
        if grp["name"] in FIELDS:
            stt = model.optimizer.state[grp["params"][0]]
            stt["exp_avg"].copy_(mom[grp["name"]][0]); stt["exp_avg_sq"].copy_(mom[grp["name"]][1])
            grp["params"][0].data.copy_(raw[grp["name"]])  # (the zero-gradient step left the parameters alone; be sure)
    model.mean_2d_grad_accum = acc.clone(); model.cnt = cnt.clone(); model.max_radii2d = maxr.clone()
    old = M.torch
    M.torch = _TorchOnCpu()
    try:
        model.densify(step, verbose=False)
        model.prune(step, verbose=False)
    finally:
        M.torch = old
    return model


DENS = dict(enabled=True, type="official", warm_up=0, end=10 ** 6, period=100, mean2d_thresh=0.02, split_thresh=0.02, n_splits=2,
            split_shrink=0.8, use_legacy=False)
PRUNE = dict(enabled=True, warm_up=0, end=10 ** 6, period=100, radii2d_thresh=1.6, alpha_thresh=0.2, radii3d_thresh=0.0)


@pytest.mark.parametrize("mode", ["legacy", "official", "official+prune", "prune_only", "legacy+prune3d"])
def test_densify_and_prune_match_the_reference_python(refmodule, mode):
    raw, mom, acc, cnt, maxr = _cloud(700, seed=3)
    dens = dict(DENS, use_legacy=mode.startswith("legacy"), enabled=mode != "prune_only")
    prune = dict(PRUNE, enabled="prune" in mode, radii3d_thresh=0.012 if "3d" in mode else 0.0)
    step = 300
    torch.manual_seed(11)
    model = _reference(refmodule, raw, mom, acc, cnt, maxr, dens, prune, step)
    dcfg = DN.DensifyConfig(enabled=dens["enabled"], type="legacy" if dens["use_legacy"] else "official", warm_up=0, end=10 ** 6,
                            period=100, mean2d_thresh=0.02, split_thresh=0.02, n_splits=2, split_shrink=0.8)
    pcfg = DN.PruneConfig(enabled=prune["enabled"], warm_up=0, end=10 ** 6, period=100, radii2d_thresh=1.6, alpha_thresh=0.2,
                          radii3d_thresh=prune["radii3d_thresh"])
    torch.manual_seed(11)
    opt, st, changed, ctl = _ours(raw, mom, acc, cnt, maxr, dcfg, pcfg, step)
    assert changed and model.mean.shape[0] != 700  # the case does something
    assert opt.params["mean"].shape[0] == model.mean.shape[0] == model.N
    for k in FIELDS:
        want = getattr(model, RAW[k]).data
        assert torch.allclose(opt.params[k].detach(), want, rtol=0, atol=1e-6), (mode, k)
    # statistics: reset by densify (:817), carried through prune (:533-549)
    assert torch.equal(st.max_radii2d, model.max_radii2d) and torch.equal(st.cnt, model.cnt)
    assert torch.equal(st.grad_accum, model.mean_2d_grad_accum)
    # Adam state: kept for surviving rows / zero for new ones ("official", prune); a fresh optimiser after "legacy"
    for grp in model.optimizer.param_groups:
        k = grp["name"]
        if k not in FIELDS:
            continue
        stt = model.optimizer.state.get(grp["params"][0])
        ea, es = opt.moments(k)
        if stt is None or "exp_avg" not in stt:
            assert mode.startswith("legacy") and float(ea.abs().max()) == 0.0 and float(es.abs().max()) == 0.0 and opt.step_count == 0
        else:
            assert torch.equal(ea, stt["exp_avg"]) and torch.equal(es, stt["exp_avg_sq"]), (mode, k)
            assert opt.step_count == 17
    if "official" in mode:
        assert ctl.last_info["num_clone"] > 0 and ctl.last_info["num_split"] > 0


def test_schedule_and_noop():
    ctl = DN.AdaptiveControl(DN.DensifyConfig(warm_up=200, end=900, period=100), DN.PruneConfig(enabled=True, warm_up=0, end=1000, period=250))
    assert ctl.due(100) == (False, False) and ctl.due(200) == (True, False) and ctl.due(250) == (False, True)
    assert ctl.due(500) == (True, True) and ctl.due(1000) == (False, True) and ctl.due(0) == (False, False)
    raw, mom, acc, cnt, maxr = _cloud(50, seed=1)
    opt = FusedAdam({k: raw[k] for k in FIELDS}, {k: 1e-3 for k in FIELDS})
    st = DensifyStats(50, torch.device("cpu"))
    o2, s2, changed = ctl.step(101, opt, st)
    assert o2 is opt and s2 is st and not changed


def test_per_gaussian_clone_test_differs_from_the_reference_quirk():
    """densify_by_clone compares the norm of the WHOLE gradient vector with the threshold (gs/gaussian_splatting.py:616-618);
    clone_test="per_gaussian" is the evident intent.  Both are available; the default reproduces the reference."""
    raw, mom, acc, cnt, maxr = _cloud(400, seed=9)
    grads = torch.nan_to_num(acc / cnt, nan=0.0)
    svec = torch.exp(raw["svec"])
    a = DN.official_masks(grads, svec, DN.DensifyConfig(clone_test="reference"))
    b = DN.official_masks(grads, svec, DN.DensifyConfig(clone_test="per_gaussian"))
    assert int(a.sum()) > int(b.sum()) > 0 and bool((a | ~b).all())


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, dtype):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    raw, mom, acc, cnt, maxr = _cloud(600, seed=4)  # replicated parameters and Adam state
    g = torch.Generator().manual_seed(100 + rank)   # ... but every rank saw its own cameras
    vis = torch.rand(600, generator=g) < 0.6
    my_cnt = vis.float() * torch.randint(1, 3, (600,), generator=g).float()
    my_acc = my_cnt * torch.rand(600, generator=g) * 0.05
    my_maxr = vis.float() * torch.rand(600, generator=g) * 2.0
    opt = FusedAdam({k: raw[k] for k in FIELDS}, {k: 1e-3 for k in FIELDS})
    opt.load_moments(mom, 5)
    st = DensifyStats(600, torch.device("cpu"))
    st.grad_accum.copy_(my_acc); st.cnt.copy_(my_cnt); st.max_radii2d.copy_(my_maxr)
    torch.manual_seed(1000 + rank)  # the global RNG differs per rank: the split noise must not come from it
    ctl = DN.AdaptiveControl(DN.DensifyConfig(type=dtype, warm_up=0, end=10 ** 6, period=100),
                             DN.PruneConfig(enabled=True, warm_up=0, end=10 ** 6, period=100, radii2d_thresh=1000.0, alpha_thresh=0.1), seed=3)
    opt, st, changed = ctl.step(200, opt, st)
    flat = torch.cat([opt.flat, opt.exp_avg, opt.exp_avg_sq, st.cnt, st.grad_accum, st.max_radii2d])
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([flat.numel()]))
    assert len({int(s) for s in sizes}) == 1
    both = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    same = all(torch.equal(both[0], b) for b in both[1:])
    if rank == 0:
        q.put((same, int(opt.params["mean"].shape[0]), changed, dict(ctl.last_info)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dtype", ["legacy", "official"])
def test_ranks_with_different_camera_shards_end_up_with_identical_clouds(dtype):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, dtype)) for r in range(world)]
    for p in procs:
        p.start()
    same, n, changed, info = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert same and changed and n != 600 and info["num_split"] > 0 and info["pruned"]["alpha"] > 0


def _interval_stats(n, seed):
    g = torch.Generator().manual_seed(seed)
    vis = torch.rand(n, generator=g) < 0.6
    cnt = vis.float() * torch.randint(1, 3, (n,), generator=g).float()
    return cnt * torch.rand(n, generator=g) * 0.05, cnt, vis.float() * torch.rand(n, generator=g) * 2.0


def _prune_then_densify(world_ranks, group_rank=None):
    """Two intervals: statistics of interval 1, a PRUNE-ONLY step (prune period 50, densify period 100), statistics of
    interval 2 accumulated on top, then a densify step.  world_ranks: the ranks whose cameras this process renders (both
    for the single process, [rank] in the 2-rank job)."""
    n = 600
    raw, mom, acc, cnt, maxr = _cloud(n, seed=4)
    opt = FusedAdam({k: raw[k] for k in FIELDS}, {k: 1e-3 for k in FIELDS})
    opt.load_moments(mom, 5)
    st = DensifyStats(n, torch.device("cpu"))
    ctl = DN.AdaptiveControl(DN.DensifyConfig(type="official", warm_up=0, end=10 ** 6, period=100),
                             DN.PruneConfig(enabled=True, warm_up=0, end=10 ** 6, period=50, radii2d_thresh=1000.0, alpha_thresh=0.1), seed=3)
    assert ctl.due(150) == (False, True) and ctl.due(200) == (True, True)
    for r in world_ranks:  # interval 1
        a_, c_, m_ = _interval_stats(n, 100 + r)
        st.grad_accum += a_; st.cnt += c_; st.max_radii2d.copy_(torch.maximum(st.max_radii2d, m_))
    opt, st, changed = ctl.step(150, opt, st)  # prune only: the surviving rows keep their statistics
    n1 = opt.params["mean"].shape[0]
    assert changed and n1 < n
    for r in world_ranks:  # interval 2, on the pruned cloud
        a_, c_, m_ = _interval_stats(n1, 200 + r)
        st.grad_accum += a_; st.cnt += c_; st.max_radii2d.copy_(torch.maximum(st.max_radii2d, m_))
    opt, st, changed = ctl.step(200, opt, st)
    return torch.cat([opt.flat, opt.exp_avg, opt.exp_avg_sq]), dict(ctl.last_info), n1


def _worker_prune_only(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    flat, info, n1 = _prune_then_densify([rank])
    q.put((rank, flat, info, n1))
    dist.barrier()
    dist.destroy_process_group()


def test_prune_only_step_does_not_count_an_interval_twice_across_ranks():
    """the statistics used to be all-reduced IN PLACE at every due step; after a prune-only step the rows carried
    over held the cross-rank sum and were summed over the ranks again at the next densify step.  Two ranks with different
    camera shards must end exactly where ONE process that rendered both shards ends (sums of two fp32 numbers commute)."""
    want, want_info, want_n1 = _prune_then_densify([0, 1])
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_prune_only, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, flat, info, n1 in got:
        assert n1 == want_n1 and info == want_info, (rank, info, want_info)
        assert flat.shape == want.shape and torch.equal(flat, want), rank
    assert want_info["num_split"] + want_info["num_clone"] > 0


def test_sh_coefficients_and_other_per_gaussian_fields_ride_along():
    """the colour field may be SH coefficients [N,3,16] (and the optimiser may hold further per-Gaussian fields): they are
    payload -- copied for clones, repeated for split samples, pruned with their rows, Adam moments alike"""
    raw, mom, acc, cnt, maxr = _cloud(300, seed=8)
    g = torch.Generator().manual_seed(2)
    raw["sh"] = torch.randn(300, 3, 16, generator=g); del raw["color"]
    raw["extra"] = torch.randn(300, 2, generator=g)
    names = list(raw)
    opt = FusedAdam({k: raw[k].clone() for k in names}, {k: 1e-3 for k in names})
    for k in names:
        ea, es = opt.moments(k)
        ea.copy_(torch.randn(ea.shape, generator=g)); es.copy_(torch.rand(es.shape, generator=g))
    mom0 = {k: tuple(m.clone() for m in opt.moments(k)) for k in names}
    st = DensifyStats(300, torch.device("cpu"))
    st.grad_accum.copy_(acc); st.cnt.copy_(cnt); st.max_radii2d.copy_(maxr)
    ctl = DN.AdaptiveControl(DN.DensifyConfig(type="official", warm_up=0, end=10 ** 6, period=100, clone_test="per_gaussian"),
                             DN.PruneConfig(enabled=True, warm_up=0, end=10 ** 6, period=100, radii2d_thresh=1000.0, alpha_thresh=0.15))
    new, st2, changed = ctl.step(100, opt, st)
    assert changed and new.names == names
    n2 = new.params["mean"].shape[0]
    assert new.params["sh"].shape == (n2, 3, 16) and new.params["extra"].shape == (n2, 2) and st2.cnt.shape == (n2,)
    # every surviving row of the payload fields is a row of the old cloud, with the Adam moments of that row or zeros (new rows)
    old_rows = {tuple(r.tolist()) for r in raw["extra"]}
    assert all(tuple(r.tolist()) in old_rows for r in new.params["extra"].detach())
    key = {tuple(r.tolist()): i for i, r in enumerate(raw["extra"])}
    ea, _ = new.moments("sh")
    for j in range(0, n2, 7):
        i = key[tuple(new.params["extra"][j].tolist())]
        assert torch.equal(new.params["sh"][j].detach(), raw["sh"][i])
        assert torch.equal(ea[j], mom0["sh"][0][i]) or float(ea[j].abs().max()) == 0.0
