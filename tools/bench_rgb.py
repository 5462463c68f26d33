"""fused RGB path (C=0) fwd+bwd timing on cfg2-like scene, plus the 3 scalar heads via compat API."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, scenes
from gsgen_amd import renderer as R
dev = torch.device("cuda:0")
sc = scenes.pointe_scene(100_000, seed=0, C=1)
cam = scenes.Camera(800, 800, fx=800.0, c2w=scenes.orbit(2.5, 15, 30))
ci = R.CameraInfo(*cam.intr)
P = {k: torch.tensor(sc[k], device=dev, requires_grad=True) for k in ("mean", "qvec", "svec", "alpha", "color")}
buf = R.FrameBuffers(100_000, 800, 800, dev)
go = torch.randn(800, 800, 3, device=dev)
def step():
    rgb, T = R.render_frame(P["mean"], P["qvec"], P["svec"], P["alpha"], P["color"], ci, cam.c2w, buf, C=0)
    (rgb * go).sum().backward()
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize(); el = time.perf_counter() - t0
print("RGB fused fwd+bwd via autograd API: %.1f renders/s (%.3f ms)" % (50 / el, el / 50 * 1e3))
