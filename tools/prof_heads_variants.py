"""BatchRenderer.render_heads + backward on the bench workload in one variant (argv: bg=0|1 zvar=0|1 stats=0|1), for a kernel trace per
variant:  rocprofv3 --kernel-trace --stats ... -- python tools/prof_heads_variants.py 1 1 1"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
import bench
from gsgen_amd import renderer as R
from gsgen_amd.batch import BatchRenderer

use_bg, zvar, use_stats = (int(a) for a in sys.argv[1:4])
K = 20
sc, W, H = bench.make_workload("cfg2")
cams = bench.camera_poses(8, 0, W, H)
dev = torch.device("cuda:0")
P = {k: torch.tensor(np.ascontiguousarray(sc[k]), device=dev, requires_grad=True) for k in ("mean", "qvec", "svec", "alpha", "color")}
cis, c2ws = [R.CameraInfo(*c.intr) for c in cams], np.stack([c.c2w for c in cams])
br = BatchRenderer(sc["mean"].shape[0], W, H, dev, max_batch=8)
bg = torch.tensor([0.1, 0.2, 0.3], device=dev, requires_grad=True) if use_bg else None
stats = R.DensifyStats(sc["mean"].shape[0], dev) if use_stats else None
go = [torch.randn(8, H, W, c, device=dev) for c in (3, 1, 1, 1)]


def step():
    outs = br.render_heads(P["mean"], P["qvec"], P["svec"], P["alpha"], P["color"], cis, c2ws, bg_rgb=bg, stats=stats, z_var=bool(zvar))[:4]
    torch.autograd.backward(outs, go)
    for q in P.values():
        q.grad = None


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    step()
torch.cuda.synchronize()
print(f"bg={use_bg} zvar={zvar} stats={use_stats}: {(time.perf_counter() - t0) / K * 1e3:.3f} ms per step", file=sys.stderr)
