"""Saturated rate of the geometry stage alone (gsgen_frame_geometry: cull + project + bin + sort) on
cfg2: S streams each looping over the cameras; says how much GPU time per view the stage needs when
it cannot hide behind compositing."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench
from gsgen_amd import _capi, renderer as R
lib = _capi.load()
dev = torch.device("cuda:0")
sc, W, H = bench.make_workload("cfg2")
N = sc["mean"].shape[0]
cams = bench.camera_poses(8, 0, W, H)
cis = [R.CameraInfo(*c.intr) for c in cams]
t = {k: torch.tensor(sc[k], device=dev) for k in ("mean", "qvec", "svec")}
cam_dev = [torch.from_numpy(ci_.pack(c.c2w)).to(dev) for ci_, c in zip(cis, cams)]
p = lambda x: x.data_ptr()
out = {}
for S in (1, 2, 3, 4, 6):
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    bufs = [R.FrameBuffers(N, W, H, dev) for _ in range(S)]
    def go(i):
        b_, s = bufs[i % S], streams[i % S].cuda_stream
        k = i % len(cams)
        lib.frame_geometry(N, p(t["mean"]), p(t["qvec"]), p(t["svec"]), p(cam_dev[k]), W, H, b_.D_cap, p(b_.mean2d),
                           p(b_.cov2d), p(b_.depth), p(b_.mask), p(b_.ids), p(b_.start), p(b_.end), p(b_.total), p(b_.ws),
                           b_.ws.numel(), s)
    for i in range(48): go(i)
    torch.cuda.synchronize()
    for b_ in bufs: b_.ensure_capacity()
    for i in range(48): go(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 480
    for i in range(n): go(i)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    out[S] = round(el / n * 1e6, 1)
print(json.dumps({"geometry_us_per_view_by_streams": out}))
