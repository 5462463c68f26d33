"""Camera sharding across the GPUs of one node (north_star: "partition across the
8 GPUs by sharding camera batches, no cross-GPU reduction, RCCL only to gather rendered images").

The reference renders the cameras of a batch one after another on one GPU
(gs/gaussian_splatting.py:1439-1462).  Here: one process per GPU (torchrun), the Gaussian
parameters replicated, the batch split contiguously, and ONE collective per batch -- an
all_gather of the rendered images over RCCL (`backend="nccl"` on ROCm) -- so every rank ends up
with the full [B, H, W, 3] batch exactly as `stack_dicts` would have produced it.  No gradient
collective is part of the rasterizer path (a data-parallel trainer all-reduces parameter grads
itself).

Pure torch.distributed: runs on gloo/CPU in the tests and on RCCL/xGMI on the GPU box.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_items, rank, world):
    """Contiguous [lo, hi) slice of `n_items` cameras owned by `rank`; the first n_items % world
    ranks take one extra camera (64 cameras on 8 GPUs -> 8 each, BASELINE configs[3])."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(n_items, world):
    return [shard_bounds(n_items, r, world)[1] - shard_bounds(n_items, r, world)[0] for r in range(world)]


class _AllGatherImages(torch.autograd.Function):
    """all_gather_into_tensor whose backward hands every rank the gradient of ITS OWN shard (each rank evaluates the
    loss on the gathered batch; the other shards' gradients belong to the ranks that rendered them).  Each rank's
    parameter gradient is then ITS cameras' share of d loss / d theta, and the shares SUM to the world-size-1 gradient:
    pair this with allreduce_gradients(..., average=False) / FusedAdam.all_reduce_grad(average=False).  (average=True is
    for the other data-parallel form -- every rank a loss on its own images only, the job's loss their mean; with a
    loss on the gathered batch it would hand back the gradient divided by the world size.  torch.distributed.nn.all_gather
    differs again: it reduce-scatters the gradients of every rank's copy of the loss.)"""

    @staticmethod
    def forward(ctx, send, group):
        world = dist.get_world_size(group)
        recv = torch.empty((world * send.shape[0],) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
        ctx.rank, ctx.n = dist.get_rank(group), send.shape[0]
        return recv

    @staticmethod
    def backward(ctx, grad):
        return grad[ctx.rank * ctx.n:(ctx.rank + 1) * ctx.n].contiguous(), None


def gather_images(local, n_total, group=None):
    """all_gather of per-rank image stacks.

    local: [b_local, H, W, C] tensor of this rank's rendered cameras (b_local from shard_bounds).
    Returns [n_total, H, W, C] on every rank, in camera order; gradients of a loss on the result flow back to this
    rank's own `local` (as they do at world size 1) -- SUM the parameter gradients over the ranks afterwards
    (allreduce_gradients(average=False)), see _AllGatherImages.  Uneven shards are padded to the
    largest shard so that a single all_gather_into_tensor moves everything (one collective per
    batch; on MI355X a direct all-gather uses all 7 xGMI links of a GPU at once)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = shard_sizes(n_total, world)
    bmax = max(sizes)
    if local.shape[0] != sizes[dist.get_rank(group)]:
        raise ValueError(f"rank {dist.get_rank(group)} holds {local.shape[0]} images, expected {sizes[dist.get_rank(group)]}")
    send = local
    if local.shape[0] < bmax:
        pad = torch.zeros((bmax - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send = torch.cat([local, pad], 0)
    recv = _AllGatherImages.apply(send, group)  # differentiable wrt this rank's own images
    if all(s == bmax for s in sizes):
        return recv
    parts = [recv[r * bmax:r * bmax + sizes[r]] for r in range(world)]
    return torch.cat(parts, 0)


def render_batch_sharded(render_one, cameras, group=None):
    """Renders this rank's shard of `cameras` with `render_one(camera) -> [H, W, C] tensor` and
    returns the gathered [len(cameras), H, W, C] batch on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(len(cameras), rank, world)
    imgs = [render_one(cameras[i]) for i in range(lo, hi)]
    if imgs:
        local = torch.stack(imgs, 0)
    else:  # more ranks than cameras: contribute an empty shard of the right shape
        probe = render_one(cameras[0])
        local = probe.new_zeros((0,) + tuple(probe.shape))
    return gather_images(local, len(cameras), group)


def render_cameras_sharded(render_many, cameras, group=None):
    """As render_batch_sharded, with the rank's whole shard handed to `render_many(list of cameras) ->
    [b, H, W, C]` in one call -- what BatchRenderer.render wants (one launch per stage for the shard),
    e.g. `lambda cams: br.render(mean, qvec, svec, alpha, sh, [c.info for c in cams], [c.c2w for c in cams], C=4)[0]`.
    A rank whose shard is empty renders camera 0 only to learn the image shape."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(len(cameras), rank, world)
    if hi > lo:
        local = render_many(list(cameras[lo:hi]))
        if local.shape[0] != hi - lo:
            raise ValueError(f"render_many returned {local.shape[0]} images for {hi - lo} cameras")
    else:
        probe = render_many([cameras[0]])
        local = probe.new_zeros((0,) + tuple(probe.shape[1:]))
    return gather_images(local, len(cameras), group)


def allreduce_gradients(tensors, group=None, average=True):
    """Data-parallel training on top of camera sharding: every rank has rendered
    and back-propagated its own cameras, the replicated parameters' gradients are summed (or
    averaged) with ONE bucketed all_reduce -- all gradient tensors are packed into a single flat
    buffer (24 MB for 100 k Gaussians at SH degree 3; on MI355X a reduce-scatter + all-gather over
    the 7 direct xGMI links moves that in tens of microseconds) and unpacked in place.
    average=True: every rank back-propagated a loss on ITS OWN images and the job's loss is their mean.
    average=False: every rank back-propagated the SAME loss on the gathered batch (gather_images): the ranks hold the
    shares of one gradient, which add up."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return tensors
    grads = [t for t in tensors if t is not None]
    if not grads:
        return tensors
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= dist.get_world_size(group)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n
    return tensors


def reduce_densify_stats(stats, group=None, with_sums=True):
    """-> (max_radii2d, grad_accum, cnt) combined over the ranks as TEMPORARIES; `stats` itself keeps this rank's own rows.
    This is what AdaptiveControl decides on.  Reducing in place would be wrong whenever the statistics survive the step:
    a prune-only step carries grad_accum / cnt over to the next interval, and rows that already hold the cross-rank sum
    would be summed over the ranks AGAIN at the next densify step (counted world_size times).  with_sums=False (a
    prune-only step never reads the gradient sums): only the MAX of max_radii2d travels."""
    maxr = stats.max_radii2d.clone()
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return maxr, stats.grad_accum, stats.cnt
    dist.all_reduce(maxr, op=dist.ReduceOp.MAX, group=group)
    if not with_sums:
        return maxr, stats.grad_accum, stats.cnt
    both = torch.stack([stats.grad_accum, stats.cnt], 0)
    dist.all_reduce(both, op=dist.ReduceOp.SUM, group=group)
    return maxr, both[0], both[1]


def allreduce_densify_stats(stats, group=None):
    """Camera sharding and the densify / prune policy (gs/gaussian_splatting.py:551-628, :1124-1132): every rank has
    accumulated the statistics of ITS cameras only (renderer.DensifyStats: max_radii2d, mean-2d-gradient sum, visit
    count).  The policy reads them and must take the same decisions on every rank -- the parameters are replicated
    -- so before it runs the three arrays are combined exactly as one process rendering all cameras would have left
    them: MAX for max_radii2d, SUM for the gradient sum and the count (fp32 sums in rank order: every rank receives
    the same bits, RCCL reduces deterministically for a fixed algorithm).  Two collectives on 3 N floats; a no-op
    without a process group.  With identical statistics, identical (seeded) RNG state and identical parameters the
    reference's densify_by_split / densify_by_clone / prune produce identical clouds on every rank.
    IN PLACE: only correct when the statistics are reset right afterwards (a densify step); AdaptiveControl uses
    reduce_densify_stats (temporaries) for that reason."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return stats
    dist.all_reduce(stats.max_radii2d, op=dist.ReduceOp.MAX, group=group)
    both = torch.stack([stats.grad_accum, stats.cnt], 0)
    dist.all_reduce(both, op=dist.ReduceOp.SUM, group=group)
    stats.grad_accum.copy_(both[0])
    stats.cnt.copy_(both[1])
    return stats
