"""Throughput of the public autograd path (gsgen_amd.batch.BatchRenderer): B cameras per step,
forward + backward of a summed loss.  Complements bench.py, which drives the C ABI directly."""
import argparse, json, sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import scenes
from gsgen_amd import renderer as R
from gsgen_amd.batch import BatchRenderer

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100_000)
ap.add_argument("--res", type=int, default=512)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--streams", type=int, default=3)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--no-stats", action="store_true", help="without the densify statistics")
ap.add_argument("--heads", action="store_true", help="RGB + depth + opacity + depth^2 (post-activation colours) instead of SH")
a = ap.parse_args()
dev = torch.device("cuda:0")
sc = scenes.pointe_scene(a.n, seed=0, C=4)
P = {k: torch.from_numpy(np.ascontiguousarray(sc[k])).to(dev).requires_grad_(True) for k in ("mean", "qvec", "svec", "alpha", "sh", "color")}
rng = np.random.default_rng(0)
cams = [scenes.Camera(a.res, a.res, fx=float(rng.uniform(0.7, 1.35) * a.res),
                      c2w=scenes.orbit(float(rng.uniform(2, 2.5)), float(rng.uniform(-20, 60)), float(rng.uniform(-180, 180))))
        for _ in range(a.batch)]
cis = [R.CameraInfo(*c.intr) for c in cams]
c2ws = [c.c2w for c in cams]
br = BatchRenderer(a.n, a.res, a.res, dev, max_batch=a.batch)
go = torch.randn(a.batch, a.res, a.res, 3, device=dev)
stats = None if a.no_stats else R.DensifyStats(a.n, dev)

def step():
    for p in P.values():
        p.grad = None
    if a.heads:
        rgb, dpt, opa, z2, _ = br.render_heads(P["mean"], P["qvec"], P["svec"], P["alpha"], P["color"], cis, c2ws, stats=stats)
        ((rgb * go).sum() + (dpt * go[..., :1]).sum() + (opa * go[..., 1:2]).sum() + (z2 * go[..., 2:]).sum()).backward()
    else:
        rgb, _ = br.render(P["mean"], P["qvec"], P["svec"], P["alpha"], P["sh"], cis, c2ws, C=4, stats=stats)
        (rgb * go).sum().backward()

step(); assert br.ensure_capacity(a.batch)
for _ in range(a.warmup):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps):
    step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize(); t1 = time.perf_counter() - t0
print(json.dumps({"path": "BatchRenderer autograd" + (" rgb+heads" if a.heads else " sh"), "n": a.n, "res": a.res, "batch": a.batch, 
                  "renders_per_s": a.batch * a.steps / t1, "ms_per_render": 1e3 * t1 / (a.batch * a.steps),
                  "host_enqueue_ms_per_render": 1e3 * t_host / (a.batch * a.steps)}))
