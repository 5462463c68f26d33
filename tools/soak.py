"""Soak test of the real usage pattern: the same batch of cameras rendered over and over through
BatchRenderer (forward, backward, densify statistics; cameras in flight on 3 streams, so forward,
matrix-core backward, sort and projection kernels of different cameras overlap), every iteration's
images and gradients compared with the first iteration's.  Prints the worst deviations."""
import argparse, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch
import scenes
from gsgen_amd import renderer as R
from gsgen_amd.batch import BatchRenderer

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=300)
ap.add_argument("--n", type=int, default=60_000)
ap.add_argument("--res", type=int, default=512)
ap.add_argument("--batch", type=int, default=6)
ap.add_argument("--C", type=int, default=4)
a = ap.parse_args()
dev = torch.device("cuda:0")
sc = scenes.pointe_scene(a.n, seed=1, C=a.C)
keys = ("mean", "qvec", "svec", "alpha", "sh")
P = {k: torch.from_numpy(np.ascontiguousarray(sc[k])).to(dev).requires_grad_(True) for k in keys}
rng = np.random.default_rng(3)
cams = [scenes.Camera(a.res, a.res, fx=float(rng.uniform(0.8, 1.3) * a.res),
                      c2w=scenes.orbit(float(rng.uniform(2.1, 2.6)), float(rng.uniform(-10, 50)), float(rng.uniform(0, 360))))
        for _ in range(a.batch)]
cis = [R.CameraInfo(*c.intr) for c in cams]; c2ws = [c.c2w for c in cams]
br = BatchRenderer(a.n, a.res, a.res, dev, max_batch=a.batch)
go = torch.randn(a.batch, a.res, a.res, 3, device=dev)
ref = None
worst = {k: 0.0 for k in keys}; worst["rgb"] = 0.0
for it in range(a.iters):
    for p in P.values():
        p.grad = None
    rgb, _ = br.render(P["mean"], P["qvec"], P["svec"], P["alpha"], P["sh"], cis, c2ws, C=a.C)
    (rgb * go).sum().backward()
    if it == 0:
        assert br.ensure_capacity(a.batch)
    cur = {"rgb": rgb.detach().clone(), **{k: P[k].grad.clone() for k in keys}}
    if ref is None or it == 1:
        ref = cur  # iteration 1 is the first one with settled buffers
        continue
    for k, v in cur.items():
        d = float((v - ref[k]).abs().max() / (ref[k].abs().max() + 1e-30))
        worst[k] = max(worst[k], d)
torch.cuda.synchronize()
print("iterations", a.iters, "worst relative deviation from the reference iteration:", {k: f"{v:.2e}" for k, v in worst.items()})
bad = {k: v for k, v in worst.items() if v > (0.0 if k == "rgb" else 2e-5) and k != "qvec"}
print("OK" if not bad else f"DEVIATIONS: {bad}")
sys.exit(1 if bad else 0)
