"""Where the host time of one BatchRenderer.render_heads step goes (perf_counter around the autograd Function's forward
and backward, the camera upload and the rest = autograd engine + the caller's loss)."""
import sys, os, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import scenes
from gsgen_amd import renderer as R, batch as Bm
from gsgen_amd.batch import BatchRenderer

acc = {}
def wrap(cls, name, key):
    f = getattr(cls, name)
    fn = f.__func__ if hasattr(f, "__func__") else f
    def g(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); acc[key] = acc.get(key, 0.0) + time.perf_counter() - t; return r
    setattr(cls, name, staticmethod(g) if isinstance(cls.__dict__[name], staticmethod) else g)
wrap(Bm._render_batch_heads, "forward", "heads.forward")
wrap(Bm._render_batch_heads, "backward", "heads.backward")
wrap(Bm._render_batch, "forward", "sh.forward")
wrap(Bm._render_batch, "backward", "sh.backward")
wrap(BatchRenderer, "_upload", "upload")
wrap(BatchRenderer, "_begin_batch", "begin_batch")
res, B = int(sys.argv[1]) if len(sys.argv) > 1 else 512, int(sys.argv[2]) if len(sys.argv) > 2 else 4
sh_mode = len(sys.argv) > 3 and sys.argv[3] == "sh"
dev = torch.device("cuda:0")
sc = scenes.pointe_scene(100_000, seed=0, C=4)
P = {k: torch.from_numpy(np.ascontiguousarray(sc[k])).to(dev).requires_grad_(True) for k in ("mean", "qvec", "svec", "alpha", "color", "sh")}
rng = np.random.default_rng(0)
cams = [scenes.Camera(res, res, fx=float(rng.uniform(0.7, 1.35) * res), c2w=scenes.orbit(float(rng.uniform(2, 2.5)), float(rng.uniform(-20, 60)), float(rng.uniform(-180, 180)))) for _ in range(B)]
cis = [R.CameraInfo(*c.intr) for c in cams]; c2ws = [c.c2w for c in cams]
br = BatchRenderer(100_000, res, res, dev, max_batch=B)
go = torch.randn(B, res, res, 3, device=dev)
bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
names = ("mean", "qvec", "svec", "alpha", "sh")
def step():
    if sh_mode:
        rgb, _ = br.render(P["mean"], P["qvec"], P["svec"], P["alpha"], P["sh"], cis, c2ws, C=4, bg_rgb=bg)
        return torch.autograd.grad([rgb], [P[n_] for n_ in names], [go])
    for p in P.values(): p.grad = None
    rgb, dpt, opa, z2, _ = br.render_heads(P["mean"], P["qvec"], P["svec"], P["alpha"], P["color"], cis, c2ws)
    ((rgb * go).sum() + (dpt * go[..., :1]).sum() + (opa * go[..., 1:2]).sum() + (z2 * go[..., 2:]).sum()).backward()
step(); br.ensure_capacity(B)
for _ in range(20): step()
torch.cuda.synchronize(); acc.clear(); n = 300; t0 = time.perf_counter()
for _ in range(n): step()
th = time.perf_counter() - t0; torch.cuda.synchronize(); tt = time.perf_counter() - t0
print(json.dumps({"res": res, "batch": B, "us_per_step_total": tt / n * 1e6, "us_host_loop": th / n * 1e6, **{k: v / n * 1e6 for k, v in acc.items()}}))
