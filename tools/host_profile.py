"""Where the host time of the autograd surface goes, measured as bench.py measures it (BatchRenderer.render + torch.autograd.grad,
8 cameras at 800x800, 100k Gaussians, three renderers on three streams, 20 steps between synchronisations so that the host never
waits for the device): perf_counter around the pieces, on whichever thread runs them (the autograd engine runs backward on its
own worker thread).  python tools/host_profile.py [heads]"""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import scenes
from bench import camera_poses
from gsgen_amd import renderer as R, batch as Bm, _capi
from gsgen_amd.batch import BatchRenderer

acc = {}
def timed(key, fn):
    def g(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); acc[key] = acc.get(key, 0.0) + time.perf_counter() - t; return r
    return g
def wrap(cls, name, key):
    raw = cls.__dict__[name]
    fn = raw.__func__ if isinstance(raw, staticmethod) else raw
    setattr(cls, name, staticmethod(timed(key, fn)) if isinstance(raw, staticmethod) else timed(key, fn))
heads = len(sys.argv) > 1 and sys.argv[1] == "heads"
F = Bm._render_batch_heads if heads else Bm._render_batch
wrap(F, "forward", "Function.forward")
wrap(F, "backward", "Function.backward (engine thread)")
wrap(BatchRenderer, "_upload", "  _upload")
wrap(BatchRenderer, "_begin_batch", "  _begin_batch")
R.sh_l1_bound_device = timed("  sh_l1_bound_device", R.sh_l1_bound_device)
lib = _capi.load()
for nm in ("frame_geometry_batch_zero", "vol_render_sh_batch_bounded", "vol_render_backward_sh_batch_bounded", "project_gaussians_backward_batch",
           "vol_render_rgbd_batch", "vol_render_rgbd_backward_batch", "project_gaussians_backward_batch_heads", "upload_small", "sh_l1_bound"):
    setattr(lib, nm, timed("    C ABI " + nm, getattr(lib, nm)))

N, W, H, B, K = 100_000, 800, 800, 8, 20
dev = torch.device("cuda:0")
sc = scenes.pointe_scene(N, seed=0, C=4)
names = ("mean", "qvec", "svec", "alpha", "color" if heads else "sh")
leaf = {k: torch.tensor(sc[k], device=dev).requires_grad_(True) for k in names}
cams = camera_poses(B, 0, W, H)
cis, c2ws = [R.CameraInfo(*c.intr) for c in cams], [c.c2w for c in cams]
bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
streams = [torch.cuda.Stream(dev) for _ in range(3)]
brs = []
for s in streams:
    with torch.cuda.stream(s):
        brs.append(BatchRenderer(N, W, H, dev, max_batch=B))
go = torch.randn(B, H, W, 3, device=dev)
go1 = torch.randn(B, H, W, 1, device=dev)
render = timed("render() / render_heads() call", lambda br, *a, **k: (br.render_heads if heads else br.render)(*a, **k))
grad = timed("torch.autograd.grad call", torch.autograd.grad)

def step(j):
    i = j % 3
    with torch.cuda.stream(streams[i]):
        if heads:
            rgb, d, o, z2, _ = render(brs[i], leaf["mean"], leaf["qvec"], leaf["svec"], leaf["alpha"], leaf["color"], cis, c2ws)
            return grad([rgb, d, o, z2], [leaf[n] for n in names], [go, go1, go1, go1])
        rgb, _ = render(brs[i], leaf["mean"], leaf["qvec"], leaf["svec"], leaf["alpha"], leaf["sh"], cis, c2ws, C=4, bg_rgb=bg)
        return grad([rgb], [leaf[n] for n in names], [go])

for j in range(12):
    step(j)
torch.cuda.synchronize()
assert all(b.ensure_capacity(B) for b in brs)
acc.clear()
host, wall, reps = 0.0, 0.0, 15
for r in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(K):
        step(r * K + j)
    host += time.perf_counter() - t0
    torch.cuda.synchronize()
    wall += time.perf_counter() - t0
n = reps * K
print(f"{'heads' if heads else 'sh'}: host enqueue {host / n * 1e6:.1f} us per step, wall {wall / n * 1e6:.1f} us per step ({B * n / wall:.0f} views/s)")
for k, v in acc.items():
    print(f"  {v / n * 1e6:8.1f} us per step  {k}")
