"""An optimisation step captured once as a hipGraph and replayed for other cameras (opt-in).

    br = BatchRenderer(N, W, H, dev, max_batch=B, device_cameras=True)

    def step(cam_infos, c2ws):          # everything the step enqueues: render, loss, backward, optimiser
        opt.zero_grad()
        rgb, depth, opacity, z_var, _ = br.render_heads(mean, qvec, raw_svec, raw_alpha, raw_color, cam_infos, c2ws, bg_rgb=bg, stats=stats,
                                                        z_var=True, activations=("exp", "sigmoid", "sigmoid"))
        loss = (rgb * image_gradient).sum() + ...
        loss.backward()
        opt.step()
        return loss

    cs = CapturedStep(br, step, cam_infos, c2ws)       # warmup + 1 = 3 REAL eager steps with these cameras (they size the pair lists), then the capture
    for it in range(steps):
        cam_infos, c2ws = sample_cameras()             # a pose AND a focal length per step, as the reference's loader draws them
        image_gradient.copy_(guidance(...))            # inputs of the step other than the cameras: update them IN PLACE
        loss = cs(cam_infos, c2ws)                     # one copy + one graph launch; `loss` is the captured tensor, refilled

What makes a replay render other cameras: a `device_cameras` renderer reads the camera blocks AND the pixel sizes from device memory
(gsgen_rgbd_view::pixel_size_dev) -- nothing that changes from step to step is a kernel argument -- and uploads nothing while a stream
is being captured; `CapturedStep.__call__` uploads the new cameras (one small enqueue outside the graph) and replays.  Every camera must have the renderer's (W, H), and the batch the captured size.

What a replay does NOT do is the host side of an eager call: the pair-list bookkeeping.  After every replay the renderer's report
words are read (no sync, as every eager render does): a camera that did not fit its list raises PairListOverflow exactly as in eager
mode (its image was NaN and it contributed no gradients: the captured step has been applied without it; the replay in flight at that
moment ran on the old lists too and is settled with it), and whenever the lists were
regrown -- quietly, with 25 % headroom left, or after an overflow -- the next call runs its step eagerly and records the graph again behind
it (the lists' addresses are baked into a graph): one step per call either way.  Densify / prune change N: build a new renderer and a new CapturedStep.

RGB + heads and RGB batches (render_heads; render with C = 0).  Reference: the loop this replaces is trainer.py:291-421 around
gs/gaussian_splatting.py:1423-1466."""
import torch

from .batch import BatchRenderer


class CapturedStep:
    def __init__(self, renderer, step, cam_infos, c2ws, warmup=2, frustum_radius=6.0, tile_radius=6.0, optimizers=()):
        """renderer: a BatchRenderer built with device_cameras=True (or a gsgen_amd.model.GaussianSplattingRenderer whose
        `device_cameras` attribute is True: its current BatchRenderer is used; a random background's colours are drawn on the host before
        every replay exactly as an eager forward draws them, random_aug backgrounds are not supported inside a capture).  step(cam_infos, c2ws): enqueues the whole step on the current stream and returns
        tensors (or None); it is called `warmup` + 1 times eagerly with the given cameras (real steps), then once more under capture
        (recorded, not executed).
        optimizers: the FusedAdam(capturable=True) objects whose step() the step contains -- before every replay their
        prepare_replay() counts the step and uploads its learning rates / bias corrections (an optimiser that bakes them into kernel
        arguments would repeat the captured step's).  A learning-rate schedule: set `opt.lrs[name]` before the call."""
        self._model = None
        if not isinstance(renderer, BatchRenderer):
            self._model, model = renderer, renderer
            if not getattr(model, "device_cameras", False):
                raise ValueError("CapturedStep: set model.device_cameras = True before the first forward")
            renderer = None
        elif not renderer.device_cameras:
            raise ValueError("CapturedStep: the BatchRenderer must be built with device_cameras=True")
        self._br = renderer
        self._step = step
        self._radii = (float(frustum_radius), float(tile_radius))
        self._warmup = max(1, int(warmup))
        self._B = len(cam_infos)
        self._optimizers = tuple(optimizers)
        for o in self._optimizers:
            if not getattr(o, "capturable", False):
                raise ValueError("CapturedStep: optimizers must be FusedAdam(capturable=True)")
        self.replays = self.captures = 0
        self._capture(list(cam_infos), c2ws, self._warmup + 1)

    # ------------------------------------------------------------------------------------------------------------------------
    def _renderer(self, cam_infos, c2ws):
        if self._model is not None:
            return self._model.batch_renderer({"camera_info": cam_infos, "c2w": c2ws})
        return self._br

    def _capture(self, cam_infos, c2ws, eager_steps):
        """`eager_steps` REAL steps with these cameras on the capture stream (they size the pair lists, build every table and buffer,
        fill the allocator's pools), then the capture, which records the step and does not run it -> the last eager step's outputs"""
        br = self._renderer(cam_infos, c2ws)
        dev = br.device
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, eager_steps)):
                out_e = self._step(cam_infos, c2ws)
            br = self._renderer(cam_infos, c2ws)
            side.synchronize()
            while not br.ensure_capacity(self._B):  # (one host sync; settles the report words: nothing is pending when the capture starts)
                out_e = self._step(cam_infos, c2ws)  # that step's images were NaN for a camera that did not fit: once more, as eager callers do
                side.synchronize()
            self._key = self._lists_key(br)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                out = self._step(cam_infos, c2ws)
        torch.cuda.current_stream(dev).wait_stream(side)
        self._graph, self.outputs, self._renderer_used = graph, out, br
        self.captures += 1
        return out_e

    @staticmethod
    def _lists_key(br):
        return (id(br), br.slots[0].D_cap, br.slots[0].ids.data_ptr())

    def __call__(self, cam_infos, c2ws):
        """upload the cameras, replay; -> the captured step's outputs (the same tensors every time, refilled by the replay)"""
        if len(cam_infos) != self._B:
            raise ValueError(f"CapturedStep: captured for {self._B} cameras, got {len(cam_infos)}")
        br = self._renderer(cam_infos, c2ws)
        if br is not self._renderer_used or self._lists_key(br) != self._key:
            # the lists moved (regrown) or the model rebuilt its renderer: THIS call's step runs eagerly -- one step per call, as ever --
            # and the graph is recorded again behind it (-> that eager step's outputs, valid until the next call like the captured ones)
            return self._capture(list(cam_infos), c2ws, 1)
        br.upload_cameras(cam_infos, c2ws, *self._radii)
        if self._model is not None and hasattr(self._model, "prepare_replay"):
            self._model.prepare_replay({"camera_info": cam_infos, "c2w": c2ws})  # (a random background's colours for this replay)
        for o in self._optimizers:
            o.prepare_replay()
        self._graph.replay()
        self.replays += 1
        try:
            br.check_overflow()  # (no sync; raises PairListOverflow for a camera of an earlier replay, regrows quietly near capacity)
        except Exception:
            # the lists have been regrown; the replay just enqueued still ran on the old ones and is lost with the reported one (its
            # report would otherwise raise once more, from the eager step of the next call): settle it here, once per episode
            torch.cuda.synchronize(br.device)
            br._report.clear()
            raise
        return self.outputs
