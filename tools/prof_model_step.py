"""The model-level training call (bench.py's `model_surface`: forward(batch) -> dict, loss.backward(), post_backward()) for a kernel
trace:   rocprofv3 --kernel-trace --stats ... -- python tools/prof_model_step.py [steps]   -> per-kernel time of the step, torch's
own kernels (activations, z_var, the harness' loss) beside this library's."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
import bench
from gsgen_amd import renderer as R
from gsgen_amd.model import GaussianSplattingRenderer

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sc, W, H = bench.make_workload("cfg2")
cams = bench.camera_poses(8, 0, W, H)
dev = "cuda:0"
cfg = bench._Cfg(device=dev, svec_act="exp", alpha_act="sigmoid", color_act="sigmoid", tile_size=16, frustum_culling_radius=6.0,
                 tile_culling_type="aabb", tile_culling_thresh=0.01, tile_culling_radius=6.0, T_thresh=1e-4,
                 skip_frustum_culling=False, normal_as_rgb=False, debug=False, depth_detach=True,
                 background=bench._Cfg(type="fixed", device=dev, color=[0.1, 0.2, 0.3], random_aug=False, random_aug_prob=0.0),
                 densify=bench._Cfg(enabled=True), prune=bench._Cfg(enabled=False))
init = {k: torch.tensor(np.ascontiguousarray(sc[k])) for k in ("mean", "qvec", "svec", "color", "alpha")}
init["alpha"] = init["alpha"].clamp(1e-4, 1 - 1e-4)
model = GaussianSplattingRenderer(cfg, init); model.train()
batch = {"c2w": torch.tensor(np.stack([c.c2w for c in cams])), "camera_info": [R.CameraInfo(*c.intr) for c in cams]}
go = {k: torch.randn(8, H, W, c, device=dev) for k, c in (("rgb", 3), ("depth", 1), ("opacity", 1), ("z_var", 1))}


def step():
    out = model(batch)
    sum((out[k] * go[k]).sum() for k in out).backward()
    model.post_backward()
    for q in model.parameters():
        q.grad = None


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    step()
host = time.perf_counter() - t0
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"model step: {el / K * 1e3:.3f} ms ({8 * K / el:.0f} views/s), host {host / K * 1e3:.3f} ms", file=sys.stderr)
