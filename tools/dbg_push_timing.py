"""Experiment (with the instrumented build gsgen_amd/lib_alt/push_dbg.so through GSGEN_HIP_LIB): per-workgroup start / end wall clock
of the push binning's two launches on the bench workload, read from the (otherwise unused) per-wavefront counter block of each view's
workspace.  Prints the spread: launch window, per-workgroup durations, the slowest chunks."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
import bench
from gsgen_amd import renderer as R
from gsgen_amd.batch import BatchRenderer

sc, W, H = bench.make_workload("cfg2")
cams = bench.camera_poses(8, 0, W, H)
dev = torch.device("cuda:0")
P = {k: torch.tensor(np.ascontiguousarray(sc[k]), device=dev, requires_grad=True) for k in ("mean", "qvec", "svec", "alpha", "color")}
cis, c2ws = [R.CameraInfo(*c.intr) for c in cams], np.stack([c.c2w for c in cams])
N = sc["mean"].shape[0]
br = BatchRenderer(N, W, H, dev, max_batch=8)
for _ in range(4):
    outs = br.render_heads(P["mean"], P["qvec"], P["svec"], P["alpha"], P["color"], cis, c2ws)[:4]
    torch.autograd.backward(outs, [torch.ones_like(o) for o in outs])
torch.cuda.synchronize()
T = br.slots[0].nth * br.slots[0].ntw
nchunks = (N + 2047) // 2048
al = lambda b: (b + 255) // 256 * 256
off = al(4 * (T + 4)) + al(4 * (T + 1)) + al(4 * T) + al(4 * nchunks * T)
allv = []
for i in range(8):
    raw = br.slots[i].ws[off: off + nchunks * 256].cpu().numpy().view(np.uint64).reshape(nchunks, 2, 16).astype(np.int64)
    allv.append(raw)
a = np.stack(allv)  # [view, chunk, (emit, count), 16]: t0, after the counters' setup, after each of wave 0's 8 slices, ..., [11] = end
for k, name in enumerate(("emit", "count")):
    t = a[:, :, k, :12]
    base = t[..., 0].min()
    d = (t[..., 11] - t[..., 0]) * 0.01
    print(f"{name}: launch window {(t[..., 11].max() - base) * 0.01:.1f} us; workgroup duration mean {d.mean():.1f} max {d.max():.1f} min {d.min():.1f} us")
    ph = np.diff(t[..., :11], axis=-1) * 0.01  # setup, slice 1..8 (thread 0's wavefront), then [10]->[11] = waiting for the other wavefronts
    print("   mean us per phase: setup", round(float(ph[..., 0].mean()), 2), "| slices", np.round(ph[..., 1:9].mean((0, 1)), 2).tolist(),
          "| tail (barrier)", round(float(((t[..., 11] - t[..., 9]) * 0.01).mean()), 2))
