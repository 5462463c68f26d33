"""The geometry chain of a camera batch ALONE (projection, count, scans, emit, sort: one gsgen_frame_geometry_batch_zero per step) on the
bench workload, for a kernel trace:  rocprofv3 --kernel-trace --stats ... -- python tools/prof_geometry_chain.py [steps]
Never composites: safe with experiment builds whose lists are not meant to be read (GSGEN_HIP_LIB)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
import bench
from gsgen_amd import renderer as R
from gsgen_amd.batch import BatchRenderer

K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
sc, W, H = bench.make_workload(cfg)
B = 8
cams = bench.camera_poses(B, 0, W, H)
dev = torch.device("cuda:0")
P = {k: torch.tensor(np.ascontiguousarray(sc[k]), device=dev) for k in ("mean", "qvec", "svec")}
cis, c2ws = [R.CameraInfo(*c.intr) for c in cams], np.stack([c.c2w for c in cams])
br = BatchRenderer(sc["mean"].shape[0], W, H, dev, max_batch=B)
gsh = torch.empty(br._Np, device=dev, dtype=torch.float32)


def step():
    br._upload(cis, c2ws, 6.0, 6.0)
    br._cis = cis
    br._begin_batch(B)
    with torch.cuda.device(dev):
        br._geometry("rgbd", B, br._fork(B), P["mean"], P["qvec"], P["svec"], gsh, None)


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    step()
torch.cuda.synchronize()
print(f"geometry chain: {(time.perf_counter() - t0) / K * 1e3:.4f} ms per {B} {cfg} views", file=sys.stderr)
