"""A size far beyond BASELINE's configs through the public path (no oracle at this size: finite outputs, pair counts,
list growth, wall time): python tools/big_scene.py [N] [res] [B]"""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import scenes
from gsgen_amd import renderer as R
from gsgen_amd.batch import BatchRenderer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
res = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda:0")
sc = scenes.pointe_scene(N, seed=0, C=4)
P = {k: torch.from_numpy(np.ascontiguousarray(sc[k])).to(dev).requires_grad_(True) for k in ("mean", "qvec", "svec", "alpha", "sh")}
cams = [scenes.Camera(res, res, fx=float(res), c2w=scenes.orbit(2.5, 15.0 + 10 * i, 30.0 + 70 * i)) for i in range(B)]
cis, c2ws = [R.CameraInfo(*c.intr) for c in cams], [c.c2w for c in cams]
br = BatchRenderer(N, res, res, dev, max_batch=B)
out = {}
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rgb, _ = br.render(P["mean"], P["qvec"], P["svec"], P["alpha"], P["sh"], cis, c2ws, C=4)
    ok = br.ensure_capacity(B)
    if not ok:
        continue
    (rgb * rgb).sum().backward()
    torch.cuda.synchronize(); out["ms_fwd_bwd"] = 1e3 * (time.perf_counter() - t0)
    out["finite_image"] = bool(torch.isfinite(rgb).all())
    out["finite_grads"] = {k: bool(torch.isfinite(v.grad).all()) for k, v in P.items()}
    out["grad_norms"] = {k: float(v.grad.norm()) for k, v in P.items()}
    for v in P.values():
        v.grad = None
out["pairs_per_view"] = [int(br._report.last(i)) for i in range(B)]
out["max_mem_GB"] = torch.cuda.max_memory_allocated() / 2**30
out["N"], out["res"], out["B"] = N, res, B
print(json.dumps(out))
