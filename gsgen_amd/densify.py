"""Adaptive density control on the replicated parameter set, deterministic across ranks.

The reference grows and shrinks its Gaussian set every `period` steps from three per-Gaussian statistics the render
loop maintains (gs/gaussian_splatting.py:464-479: mean_2d_grad_accum, cnt, max_radii2d -- here renderer.DensifyStats,
updated by the fused paths):

  densify, "legacy" (conf/base.yaml:163 `use_legacy: true`, gs/gaussian_splatting.py:820-948): Gaussians whose average
      screen-space gradient exceeds `mean2d_thresh` are SPLIT in two when any scale exceeds `split_thresh` (samples of
      the parent, scales divided by 2 * split_shrink), CLONED otherwise; the optimiser starts afresh.
  densify, "official" (:751-781 -> densify_by_clone :614-628, densify_by_split :551-612): clone, then split with the
      gradients zero-padded for the clones, Adam moments kept for surviving rows and zero for new ones (:481-522).
  prune (:1124-1177): by screen-space radius, by opacity, by 3-D scale; Adam moments follow (:421-449).

With camera sharding every rank renders its own cameras, so the statistics differ per rank while the parameters are
replicated.  `AdaptiveControl.step` first combines the statistics (dist.reduce_densify_stats: all-reduced into
temporaries, so that a prune-only step does not count an interval twice; every rank then holds the bits a single process would) and draws the split noise from a generator seeded by (seed, step): every rank takes
the same decisions and writes the same new parameters -- no parameter broadcast.  Pure torch (any device): this is the
caller-side bookkeeping around the HIP path, not a kernel.
"""
from dataclasses import dataclass

import torch

FIELDS = ("mean", "qvec", "svec", "color", "alpha")  # raw (pre-activation) fields, gs/gaussian_splatting.py:55-62
# mean / qvec / svec are what a split changes; every other per-Gaussian field of the optimiser (alpha, color -- or sh when
# the colours are SH coefficients, specular, normal, ...) is payload: copied for clones, repeated for split samples


def _payload(raw):
    return [k for k in raw if k not in ("mean", "qvec", "svec")]


def _rep(t, n):
    return t.repeat(n, *([1] * (t.dim() - 1)))


@dataclass
class DensifyConfig:  # conf/base.yaml:152-163
    enabled: bool = True
    type: str = "legacy"  # "legacy" | "official"
    warm_up: int = 2000
    end: int = 9999
    period: int = 1000
    mean2d_thresh: float = 0.02
    split_thresh: float = 0.02
    n_splits: int = 2
    split_shrink: float = 0.8
    clone_test: str = "reference"  # see official_masks


@dataclass
class PruneConfig:  # conf/base.yaml:164-171
    enabled: bool = False
    warm_up: int = 0
    end: int = 0
    period: int = 500
    radii2d_thresh: float = 1000.0
    alpha_thresh: float = 1000.0
    radii3d_thresh: float = 0.0


def step_check(step, step_size, run_at_zero=False):
    """gs/renderer.py:27-31"""
    return step_size != 0 and (run_at_zero or step != 0) and step % step_size == 0


def rotmat_of_qvec(q):
    """utils/transforms.py:34-38 -> kornia quaternion_to_rotation_matrix(q, WXYZ): normalise, then the rotation matrix"""
    q = torch.nn.functional.normalize(q, p=2.0, dim=-1, eps=1e-12)
    w, x, y, z = q.unbind(-1)
    tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    one = torch.ones_like(w)
    return torch.stack((one - (tyy + tzz), txy - twz, txz + twy, txy + twz, one - (txx + tzz), tyz - twx,
                        txz - twy, tyz + twx, one - (txx + tyy)), dim=-1).view(-1, 3, 3)


def _split_samples(raw, svec, mask, n_splits, generator):
    """n_splits samples of each selected Gaussian: mean + R(q)^T (N(0,1) * svec)  (:567-580 / :847-864)"""
    mean = raw["mean"][mask].repeat(n_splits, 1)
    qvec = raw["qvec"][mask].repeat(n_splits, 1)
    s = svec[mask].repeat(n_splits, 1)
    rot_t = rotmat_of_qvec(qvec).transpose(-1, -2)
    gn = torch.randn(mean.shape[0], 3, device=mean.device, generator=generator) * s
    return mean + torch.einsum("bij,bj->bi", rot_t, gn), qvec, s


def legacy_masks(stats_accum, stats_cnt, svec, cfg):
    """-> (split_mask, clone_mask)  (:827-836)"""
    mask = stats_accum / (stats_cnt + 1e-5) > cfg.mean2d_thresh
    split = torch.logical_and(mask, (svec > cfg.split_thresh).any(dim=-1))
    return split, torch.logical_and(mask, torch.logical_not(split))


def densify_legacy(raw, svec, svec_inv_act, stats_accum, stats_cnt, cfg, generator=None):
    """-> (new raw fields, info).  Row order of the reference: [not split | clones | 2 x split samples] (:871-912)."""
    split, clone = legacy_masks(stats_accum, stats_cnt, svec, cfg)
    smean, sqvec, ssvec = _split_samples(raw, svec, split, 2, generator)
    keep = torch.logical_not(split)
    new = {"mean": torch.cat([raw["mean"][keep], raw["mean"][clone], smean]),
           "qvec": torch.cat([raw["qvec"][keep], raw["qvec"][clone], sqvec]),
           "svec": torch.cat([raw["svec"][keep], raw["svec"][clone], svec_inv_act(ssvec / cfg.split_shrink / 2.0)])}
    for k in _payload(raw):
        new[k] = torch.cat([raw[k][keep], raw[k][clone], _rep(raw[k][split], 2)])
    return new, {"num_split": int(split.sum()), "num_clone": int(clone.sum())}


def official_masks(grads, svec, cfg):
    """clone / split selections of the "official" type.  grads = mean_2d_grad_accum / cnt with NaN -> 0 (:766-767).
    clone (:616-623): the reference writes `torch.norm(grads, dim=-1) >= thresh` on the 1-D `grads`, i.e. it compares the
    norm of the WHOLE gradient vector with the threshold: every small Gaussian is cloned once that norm passes it.
    cfg.clone_test = "reference" reproduces that; "per_gaussian" compares each Gaussian's own value (what split does)."""
    small = torch.max(svec, dim=1).values <= cfg.split_thresh
    if cfg.clone_test == "reference":
        hot = (torch.norm(grads, dim=-1) >= cfg.mean2d_thresh).expand_as(small)
    else:
        hot = grads >= cfg.mean2d_thresh
    return torch.logical_and(hot, small)


def densify_official(raw, moments, act, stats_accum, stats_cnt, cfg, generator=None):
    """-> (new raw, new moments, info).  moments: dict field -> (exp_avg, exp_avg_sq) or None.  act: dict of the svec
    activation pair {"svec": fn, "svec_inv": fn}."""
    grads = stats_accum / stats_cnt
    grads = torch.where(torch.isnan(grads), torch.zeros_like(grads), grads)
    # clone: append copies (:614-628), new Adam moments zero (:481-522)
    clone = official_masks(grads, act["svec"](raw["svec"]), cfg)
    names = list(raw)
    raw = {k: torch.cat([raw[k], raw[k][clone]]) for k in names}
    if moments is not None:
        moments = {k: tuple(torch.cat([m, torch.zeros_like(m[clone])]) for m in moments[k]) for k in names}
    num_clone = int(clone.sum())
    # split: gradients zero-padded behind the clones (:556-566)
    n = raw["mean"].shape[0]
    padded = torch.zeros(n, device=grads.device, dtype=grads.dtype)
    padded[:grads.shape[0]] = grads
    svec = act["svec"](raw["svec"])
    sel = torch.logical_and(padded >= cfg.mean2d_thresh, torch.max(svec, dim=1).values > cfg.split_thresh)
    smean, sqvec, ssvec = _split_samples(raw, svec, sel, cfg.n_splits, generator)
    add = {"mean": smean, "qvec": sqvec, "svec": act["svec_inv"](ssvec / (cfg.n_splits * cfg.split_shrink))}
    for k in _payload(raw):
        add[k] = _rep(raw[k][sel], cfg.n_splits)
    keep = torch.cat([torch.logical_not(sel), torch.ones(add["mean"].shape[0], dtype=torch.bool, device=sel.device)])
    raw = {k: torch.cat([raw[k], add[k]])[keep] for k in names}
    if moments is not None:
        moments = {k: tuple(torch.cat([m, torch.zeros_like(add[k])])[keep] for m in moments[k]) for k in names}
    return raw, moments, {"num_split": int(sel.sum()), "num_clone": num_clone}


def prune_masks(step, cfg, max_radii2d, alpha, svec):
    """the three tests of prune() in the reference's order (:1124-1150); each is applied to what the previous one left,
    so this returns ONE keep-mask over the current rows"""
    keep = torch.ones_like(alpha, dtype=torch.bool)
    counts = {}
    if cfg.radii2d_thresh > 0.0:
        m = max_radii2d > cfg.radii2d_thresh
        counts["scale"] = int(m.sum()); keep &= ~m
    if cfg.alpha_thresh > 0.0:
        m = (alpha < cfg.alpha_thresh) & keep
        counts["alpha"] = int(m.sum()); keep &= ~m
    if cfg.radii3d_thresh > 0.0:
        m = (svec > cfg.radii3d_thresh).all(dim=-1) & keep
        counts["svec"] = int(m.sum()); keep &= ~m
    return keep, counts


class AdaptiveControl:
    """densify() + prune() of the reference's renderer for a FusedAdam-held parameter set and a DensifyStats, with the
    cross-rank agreement described in the module docstring.

        ctl = AdaptiveControl(densify_cfg, prune_cfg, seed=0)
        ...every step, after backward and the optimiser step:
        opt, stats, changed = ctl.step(step, opt, stats)     # new objects when the Gaussian count changed

    activations: svec = exp, alpha = sigmoid (conf/base.yaml:141-143) unless given."""

    def __init__(self, densify=None, prune=None, seed=0, group=None, svec_act=torch.exp, svec_inv_act=torch.log,
                 alpha_act=torch.sigmoid, use_global_rng=False):
        self.densify_cfg, self.prune_cfg = densify or DensifyConfig(), prune or PruneConfig()
        self.seed, self.group = int(seed), group
        self.use_global_rng = bool(use_global_rng)  # torch's global generator, as the reference (single process only)
        self.svec_act, self.svec_inv_act, self.alpha_act = svec_act, svec_inv_act, alpha_act

    def _generator(self, step, device):
        g = torch.Generator(device=device)
        g.manual_seed(self.seed * 1_000_003 + int(step))  # the same stream on every rank
        return g

    def due(self, step):
        d, p = self.densify_cfg, self.prune_cfg
        dens = d.enabled and d.warm_up <= step <= d.end and step_check(step, d.period, True)
        prun = p.enabled and p.warm_up <= step <= p.end and step_check(step, p.period)
        return dens, prun

    def step(self, step, opt, stats, generator=None):
        from . import dist as gdist
        from .optim import FusedAdam
        from .renderer import DensifyStats
        dens, prun = self.due(step)  # trainer.py calls densify() then prune() every step
        if not (dens or prun):
            return opt, stats, False
        # the ranks' statistics combined into temporaries the decisions read; `stats` keeps this rank's own rows, which
        # is what a prune-only step carries over (summing in place would count this interval world_size times at the next
        # densify step)
        g_maxr, g_accum, g_cnt = gdist.reduce_densify_stats(stats, self.group, with_sums=dens)
        for k in ("mean", "qvec", "svec", "alpha"):
            if k not in opt.params:
                raise KeyError(f"AdaptiveControl needs the raw field {k!r} in the optimiser")
        raw = {k: opt.params[k].detach() for k in opt.names}
        moments = {k: opt.moments(k) for k in opt.names}
        dev = raw["mean"].device
        info = {}
        maxr = g_maxr
        if dens:
            gen = generator if generator is not None else (None if self.use_global_rng else self._generator(step, dev))
            d = self.densify_cfg
            if d.type == "legacy":
                raw, info = densify_legacy(raw, self.svec_act(raw["svec"]), self.svec_inv_act, g_accum, g_cnt, d, gen)
                moments = None  # set_optimizer(): the optimiser starts afresh (:935)
            elif d.type == "official":
                raw, moments, info = densify_official(raw, moments, {"svec": self.svec_act, "svec_inv": self.svec_inv_act},
                                                      g_accum, g_cnt, d, gen)
            else:
                raise NotImplementedError(f"densify type {d.type!r}")
            maxr = torch.zeros(raw["mean"].shape[0], device=dev)  # reset_densify_info (:476-479, :817)
        if prun:
            keep, counts = prune_masks(step, self.prune_cfg, maxr, self.alpha_act(raw["alpha"]), self.svec_act(raw["svec"]))
            info["pruned"] = counts
            raw = {k: v[keep] for k, v in raw.items()}
            if moments is not None:
                moments = {k: tuple(m[keep] for m in moments[k]) for k in moments}
            maxr = maxr[keep]
        n = raw["mean"].shape[0]
        new_opt = FusedAdam({k: raw[k] for k in opt.names}, opt.lrs, betas=opt.betas, eps=opt.eps)
        if moments is not None:
            new_opt.load_moments(moments, opt.step_count)
        new_stats = DensifyStats(n, dev)
        if prun and not dens:  # prune_by_mask keeps the surviving rows' statistics (:533-549): the job-wide maximum (a
            new_stats.max_radii2d.copy_(maxr)  # maximum may be taken twice) and THIS RANK'S gradient sums and counts
            new_stats.grad_accum.copy_(stats.grad_accum[keep]); new_stats.cnt.copy_(stats.cnt[keep])
        elif prun:
            new_stats.max_radii2d.copy_(maxr)
        self.last_info = info
        return new_opt, new_stats, True
