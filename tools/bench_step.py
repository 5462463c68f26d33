"""The renderer's share of one GSGEN optimisation step (BASELINE configs[4], guidance stubbed): what
trainer.py:291-421 does around the diffusion model, on this repo's public autograd path.

    raw parameters --activations--> BatchRenderer.render_heads (4 views at 512x512: rgb + depth + opacity + depth^2,
    one enqueue per stage) --> loss = <rgb, g_sds> + sparsity + z_var terms --> backward --> densify statistics
    --> FusedAdam step on the five raw fields (one flat buffer, one kernel)

The StableDiffusion guidance (diffusers + weights) is not available offline; its output as far as the renderer is
concerned is a gradient on the rendered rgb batch (SDS: grad = w(t) (eps_hat - eps), back through the VAE encoder), so
the stub is a fixed random image gradient of that shape.  Everything else is the real step.  Prints one JSON line:
iterations per second and the host time per iteration.  A tool next to bench.py (which measures BASELINE's headline
metric); not part of the driver's contract."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import scenes  # noqa: E402
from gsgen_amd import renderer as R  # noqa: E402
from gsgen_amd.batch import BatchRenderer  # noqa: E402
from gsgen_amd.optim import FusedAdam  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100_000)
ap.add_argument("--res", type=int, default=512)
ap.add_argument("--batch", type=int, default=4, help="views per step (conf/base.yaml:2)")
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--warmup", type=int, default=20)
ap.add_argument("--graph", action="store_true", help="replay the whole step from one hipGraph")
ap.add_argument("--profile", action="store_true", help="torch.profiler CPU table of 20 eager steps on stderr (where the host time goes)")
ap.add_argument("--cprofile", action="store_true", help="cProfile of 200 eager steps on stderr (the Python side of the host time)")
ap.add_argument("--pipeline", choices=["auto", "on", "off"], default="off", help="BatchRenderer(pipeline=...): two half-batches on two streams")
ap.add_argument("--torch-ops", action="store_true", help="activations / z_var as torch operations around render_heads (rounds 1-5) instead of inside its node")
a = ap.parse_args()
dev = torch.device("cuda:0")
sc = scenes.pointe_scene(a.n, seed=0, C=1)
logit = lambda x: np.log(x / (1 - x))  # noqa: E731
raw0 = {"mean": sc["mean"], "qvec": sc["qvec"], "svec": np.log(sc["svec"]), "color": logit(np.clip(sc["color"], 1e-3, 1 - 1e-3)),
        "alpha": logit(np.clip(sc["alpha"], 1e-3, 1 - 1e-3))}
opt = FusedAdam({k: torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(dev) for k, v in raw0.items()},
                {"mean": 5e-3, "qvec": 3e-3, "svec": 3e-3, "color": 1e-2, "alpha": 3e-3}, eps=1e-15,  # conf/base.yaml:8-30
                capturable=a.graph)
P = opt.params
rng = np.random.default_rng(0)
# a pool of camera batches, a fresh one per step: pose AND focal length drawn per step as the reference's loader draws them
# (data/__init__.py:187-200; conf/base.yaml:71-85)
pool = []
for _ in range(64):
    cams = [scenes.Camera(a.res, a.res, fx=float(rng.uniform(0.75, 1.35) * a.res),
                          c2w=scenes.orbit(2.5, float(rng.uniform(-20, 90)), float(rng.uniform(-180, 180)))) for _ in range(a.batch)]
    pool.append(([R.CameraInfo(*c.intr) for c in cams], np.stack([c.c2w for c in cams])))
tick = [0]
br = BatchRenderer(a.n, a.res, a.res, dev, max_batch=a.batch, pipeline={'auto': 'auto', 'on': True, 'off': False}[a.pipeline],
                   device_cameras=a.graph)
stats = R.DensifyStats(a.n, dev)
g_sds = torch.randn(a.batch, a.res, a.res, 3, device=dev) * 1e-4  # the guidance's gradient on the rendered batch
bg = torch.tensor([0.5, 0.5, 0.5], device=dev)


def step(cis=None, c2ws=None):
    if cis is None:
        cis, c2ws = pool[tick[0] % len(pool)]
        tick[0] += 1
    opt.zero_grad()
    if a.torch_ops:  # rounds 1-5: the activations, the background and z_var as torch kernels / autograd nodes around the launches
        rgb, dpt, opa, z2, _ = br.render_heads(P["mean"], P["qvec"], torch.exp(P["svec"]), torch.sigmoid(P["alpha"]),
                                               torch.sigmoid(P["color"]), cis, c2ws, bg_rgb=bg, stats=stats)
        z_var = z2 - dpt * dpt
    else:  # round 6 (what gsgen_amd.model.GaussianSplattingRenderer.forward does): raw parameters in, z_var out
        rgb, dpt, opa, z_var, _ = br.render_heads(P["mean"], P["qvec"], P["svec"], P["alpha"], P["color"], cis, c2ws, bg_rgb=bg, stats=stats,
                                                  z_var=True, activations=("exp", "sigmoid", "sigmoid"))
    loss = (rgb * g_sds).sum() + 1e-3 * (opa * opa + 0.01).sqrt().mean() + 1e-3 * z_var.mean()  # trainer.py:305-388
    loss.backward()
    opt.step()


step()
torch.cuda.synchronize()
assert br.ensure_capacity(a.batch)
for _ in range(a.warmup):
    step()
run, graph_error = step, None
if a.graph:
    # gsgen_amd.graph.CapturedStep: the whole step as one hipGraph, replayed for a FRESH camera batch every time (camera blocks and pixel
    # sizes from device memory, the optimiser's per-step scalars likewise: round 6)
    try:
        from gsgen_amd.graph import CapturedStep
        cs = CapturedStep(br, step, *pool[0], optimizers=[opt])

        def run():
            cis, c2ws = pool[tick[0] % len(pool)]
            tick[0] += 1
            cs(cis, c2ws)
        for _ in range(5):
            run()
    except Exception as e:  # report, and time the eager step instead
        graph_error = f"{type(e).__name__}: {str(e)[:300]}"
if a.profile:
    from torch.profiler import profile, ProfilerActivity
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        for _ in range(20):
            step()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=40, max_name_column_width=60), file=sys.stderr)
if a.cprofile:
    import cProfile
    import pstats
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr, stream=sys.stderr)
    st.sort_stats("tottime").print_stats(45)
    st.sort_stats("cumulative").print_stats(45)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    run()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t1 = time.perf_counter() - t0
print(json.dumps({"metric": "optimisation-step iters/sec, renderer + optimiser share (guidance stubbed)", "value": a.steps / t1,
                  "unit": "iters/s", "ms_per_iter": 1e3 * t1 / a.steps, "host_ms_per_iter": 1e3 * t_host / a.steps,
                  "views_per_s": a.batch * a.steps / t1, "hipgraph": bool(a.graph) and graph_error is None,
                  "cameras": "a fresh batch per step (pose and focal length)", "graph_captures": (cs.captures if a.graph and graph_error is None else None), "pipeline": a.pipeline, "hipgraph_error": graph_error,
                  "config": {"workload": "BASELINE configs[4] without the diffusion model: 100k Gaussians, "
                                         f"{a.batch} views at {a.res}x{a.res}, rgb + depth + opacity + z_var, "
                                         "densify statistics, Adam on the five raw fields", "gaussians": a.n}}))
