"""Throughput and host cost of the per-camera autograd path (gsgen_amd.renderer.render_frame): one camera per call,
forward + backward, as the reference's render_one loop drives it (gs/gaussian_splatting.py:1423-1466)."""
import argparse, json, sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import scenes
from gsgen_amd import renderer as R

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100_000)
ap.add_argument("--res", type=int, default=800)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--warmup", type=int, default=20)
ap.add_argument("--segments", type=int, default=8)
a = ap.parse_args()
dev = torch.device("cuda:0")
sc = scenes.pointe_scene(a.n, seed=0, C=4)
P = {k: torch.from_numpy(np.ascontiguousarray(sc[k])).to(dev).requires_grad_(True) for k in ("mean", "qvec", "svec", "alpha", "sh")}
rng = np.random.default_rng(0)
cams = [scenes.Camera(a.res, a.res, fx=float(rng.uniform(0.7, 1.35) * a.res),
                      c2w=scenes.orbit(float(rng.uniform(2, 2.5)), float(rng.uniform(-20, 60)), float(rng.uniform(-180, 180))))
        for _ in range(8)]
cis = [R.CameraInfo(*c.intr) for c in cams]
buf = R.FrameBuffers(a.n, a.res, a.res, dev, segments=a.segments)
go = torch.randn(a.res, a.res, 3, device=dev)


def step(i):
    for p in P.values():
        p.grad = None
    c = cams[i % len(cams)]
    rgb, _ = R.render_frame(P["mean"], P["qvec"], P["svec"], P["alpha"], P["sh"], cis[i % len(cams)], c.c2w, buf, C=4)
    (rgb * go).sum().backward()


for i in range(len(cams)):
    step(i); buf.ensure_capacity()
for i in range(a.warmup):
    step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(a.steps):
    step(i)
t_host = time.perf_counter() - t0
torch.cuda.synchronize(); t1 = time.perf_counter() - t0
print(json.dumps({"path": "render_frame autograd sh", "n": a.n, "res": a.res, "segments": a.segments,
                  "renders_per_s": a.steps / t1, "ms_per_render": 1e3 * t1 / a.steps,
                  "host_enqueue_ms_per_render": 1e3 * t_host / a.steps}))
