"""Replicated optimiser step for camera-sharded data-parallel training.

The reference gives every parameter field (mean, qvec, svec, color, alpha) its own Adam param
group with a scheduled learning rate (gs/gaussian_splatting.py:398-419; torch.optim.Adam,
eps = 1e-15, conf/base.yaml:8-11).  Here all fields live back to back in ONE flat fp32 buffer:

  * the parameters handed to the renderer are views into it (so are their .grad's),
  * the data-parallel gradient reduction is ONE all_reduce of the flat gradient
    (RCCL over xGMI on MI355X: 24 MB for 100 k Gaussians at SH degree 3),
  * the Adam update is ONE pass over it (gsgen_adam_step: 28 B per parameter).

Every rank applies the same update to the same reduced gradient, so the replicas stay identical
without broadcasting parameters.
"""
import ctypes as C

import numpy as np
import torch

from . import _capi


class FusedAdam:
    def __init__(self, fields, lrs, betas=(0.9, 0.999), eps=1e-15, capturable=False):
        """fields: dict name -> initial tensor (same device, fp32); lrs: dict name -> float.
        capturable: the step's scalars (learning rate / (1 - beta1^t) per field, sqrt(1 - beta2^t)) are read from DEVICE memory
        (gsgen_adam_step_device_scalars) instead of riding in kernel arguments: a step() enqueued into a stream capture uploads
        nothing, and prepare_replay() -- which gsgen_amd.graph.CapturedStep calls before every replay -- counts the step, evaluates
        the scalars on the host and copies them there, so a replayed step is the step torch's Adam would take at that count."""
        self.names = list(fields)
        dev = next(iter(fields.values())).device
        sizes = [int(fields[k].numel()) for k in self.names]
        self.n = int(sum(sizes))
        self.flat = torch.empty(self.n, device=dev, dtype=torch.float32)
        self.grad = torch.zeros_like(self.flat)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.params, self._span = {}, {}
        ends, off = [], 0
        for k, sz in zip(self.names, sizes):
            self.flat[off:off + sz].copy_(fields[k].detach().reshape(-1).to(torch.float32))
            p = self.flat[off:off + sz].view(fields[k].shape).requires_grad_(True)
            p.grad = self.grad[off:off + sz].view(fields[k].shape)
            self.params[k] = p
            self._span[k] = (off, off + sz, tuple(fields[k].shape))
            off += sz
            ends.append(off)
        self._ends = np.array(ends, np.uint64)
        self.lrs = dict(lrs)
        self.betas, self.eps, self.step_count = betas, float(eps), 0
        self.capturable = bool(capturable)
        if self.capturable:
            self._scal = torch.zeros(9, device=dev, dtype=torch.float32)
            self._scal_host = np.zeros(9, np.float32)

    def zero_grad(self):
        self.grad.zero_()

    def moments(self, name):
        """-> (exp_avg, exp_avg_sq) of one field, shaped like the field (views into the flat buffers)"""
        a, b, shape = self._span[name]
        return self.exp_avg[a:b].view(shape), self.exp_avg_sq[a:b].view(shape)

    def load_moments(self, moments, step_count):
        """moments: dict name -> (exp_avg, exp_avg_sq) shaped like the fields (densify / prune carry the surviving rows'
        Adam state over, gs/gaussian_splatting.py:421-449, :481-522); step_count: the optimiser's step so far"""
        for k in self.names:
            ea, es = self.moments(k)
            ea.copy_(moments[k][0]); es.copy_(moments[k][1])
        self.step_count = int(step_count)

    def all_reduce_grad(self, group=None, average=True):
        """one collective for all fields (no-op without an initialised process group).  average=True for per-rank losses whose
        mean is the job's loss; average=False when every rank back-propagated one loss on the gathered batch
        (dist.gather_images: the ranks hold shares of ONE gradient, which add up -- see dist.allreduce_gradients)."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=group)
            if average:
                self.grad /= dist.get_world_size(group)

    def step(self, lrs=None):
        """lrs: this step's learning rates (the reference re-evaluates its schedulers every step)"""
        if self.capturable:
            if not torch.cuda.is_current_stream_capturing():
                self.prepare_replay(lrs)  # (eager: count the step and upload its scalars, then the launch below)
            with torch.cuda.device(self.flat.device):
                _capi.load().adam_step_device_scalars(self.n, self.flat.data_ptr(), self.grad.data_ptr(), self.exp_avg.data_ptr(),
                                                      self.exp_avg_sq.data_ptr(), len(self.names), self._ends.ctypes.data_as(C.c_void_p),
                                                      self.betas[0], self.betas[1], self.eps, self._scal.data_ptr(),
                                                      torch.cuda.current_stream(self.flat.device).cuda_stream)
            return
        if lrs is not None:
            self.lrs.update(lrs)
        self.step_count += 1
        lr = np.array([self.lrs[k] for k in self.names], np.float32)
        with torch.cuda.device(self.flat.device):
            _capi.load().adam_step(self.n, self.flat.data_ptr(), self.grad.data_ptr(), self.exp_avg.data_ptr(),
                                   self.exp_avg_sq.data_ptr(), len(self.names), self._ends.ctypes.data_as(C.c_void_p),
                                   lr.ctypes.data_as(C.c_void_p), self.betas[0], self.betas[1], self.eps,
                                   self.step_count, torch.cuda.current_stream(self.flat.device).cuda_stream)

    def prepare_replay(self, lrs=None):
        """capturable optimisers: count one step and put its scalars in place on the current stream (36 bytes through kernel arguments) -- before the replay of a captured step that contains step()"""
        if not self.capturable:
            raise RuntimeError("FusedAdam.prepare_replay: built without capturable=True")
        if lrs is not None:
            self.lrs.update(lrs)
        self.step_count += 1
        lr = np.array([self.lrs[n_] for n_ in self.names], np.float32)
        lib = _capi.load()
        lib.adam_step_scalars(len(self.names), lr.ctypes.data_as(C.c_void_p), self.betas[0], self.betas[1], self.step_count,
                              self._scal_host.ctypes.data_as(C.c_void_p))
        with torch.cuda.device(self.flat.device):
            lib.upload_small(self._scal.data_ptr(), self._scal_host.ctypes.data, 36, torch.cuda.current_stream(self.flat.device).cuda_stream)
