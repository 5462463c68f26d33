import sys, os, cProfile, pstats, io, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ["GSGEN_TEST_REFPY"] = "staged"
import numpy as np, torch, types
import scenes, refshim, gsgen_amd
import bench
backend = gsgen_amd.compiled_gs()
refshim.install(); sys.modules["_gs"] = backend
import refpy_cases as RC
import gs.renderer as GR
GR._backend = backend
stub = types.SimpleNamespace(cudaProfilerStart=lambda: 0, cudaProfilerStop=lambda: 0)
torch.cuda.profiler.cudart = lambda: stub
M = RC.import_reference_model(backend)
from utils.camera import CameraInfo
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
model = M.GaussianSplattingRenderer(cfg, init); model.train()
batch = {"c2w": torch.tensor(np.stack([c.c2w for c in cams])), "camera_info": [CameraInfo(*c.intr) for c in cams]}
go = {k: torch.randn(8, H, W, c, device=dev) for k, c in (("rgb", 3), ("depth", 1), ("opacity", 1), ("z_var", 1))}
def step():
    out = model(batch)
    sum((out[k] * go[k]).sum() for k in out).backward()
    model.post_backward()
    for q in model.parameters(): q.grad = None
step(); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35); print(s.getvalue()[:6000])
