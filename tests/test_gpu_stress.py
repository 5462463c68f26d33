"""Run-to-run stability of the SH backward under load (-m gpu).

A matrix-core form of the backward (removed in round 3) once produced timing-dependent garbage when
several of its wavefronts shared a matrix core (profiles/r01_notes.md, "MFMA chain hazard"): a
single launch over a few thousand tiles, or a few launches in flight on different streams, was
enough to see per-Gaussian gradients move by percents between identical runs.  The guard stays for
the vector kernels that ship, for every SH degree: identical launches must agree to
atomics-reordering noise, alone and with other cameras' launches in flight."""
import numpy as np
import pytest
import torch

import scenes
from gsgen_amd import _capi, renderer as R
from gsgen_amd.renderer import _p

pytestmark = pytest.mark.gpu


def _setup(C, N, W, H, n_cam):
    dev = torch.device("cuda:0")
    sc = scenes.pointe_scene(N, seed=3, C=C)
    P = {k: torch.from_numpy(np.ascontiguousarray(sc[k])).to(dev) for k in ("mean", "qvec", "svec", "alpha", "sh")}
    lib = _capi.load()
    cams = []
    for i in range(n_cam):
        cam = scenes.Camera(W, H, fx=float(W) * (0.9 + 0.1 * i), c2w=scenes.orbit(2.4 + 0.05 * i, 10 + 5 * i, 40.0 * i))
        ci = R.CameraInfo(*cam.intr)
        buf = R.FrameBuffers(N, W, H, dev)
        cam_dev = torch.from_numpy(ci.pack(cam.c2w)).to(dev)
        R.frame_geometry(P["mean"], P["qvec"], P["svec"], cam_dev, buf)
        topleft = torch.tensor(cam.topleft, device=dev)
        rot = torch.from_numpy(np.ascontiguousarray(cam.c2w[:3, :3].reshape(-1))).to(dev)
        out = torch.zeros(H, W, 3, device=dev)
        T = torch.ones(H, W, 1, device=dev)
        lib.vol_render_sh_ordered(N, buf.D_cap, _p(buf.mean2d), _p(buf.cov2d), _p(P["sh"]), _p(P["alpha"]), _p(buf.start),
                                  _p(buf.end), _p(buf.ids), _p(out), _p(topleft), _p(rot), 16, buf.nth, buf.ntw, 1 / ci.fx,
                                  1 / ci.fy, H, W, C, 1e-4, None, _p(T), buf.tile_order(), None)
        go = torch.randn(H, W, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(i))
        cams.append((buf, out, topleft, rot, ci, go))
    torch.cuda.synchronize()
    return dev, lib, P, cams


def _backward_all(dev, lib, P, cams, C, N, W, H, streams):
    gsh = torch.zeros(N, 3 * C * C, device=dev)
    ga = torch.zeros(N, device=dev)
    per_cam = [(torch.zeros(N, 2, device=dev), torch.zeros(N, 4, device=dev)) for _ in cams]
    torch.cuda.synchronize()
    for i, (buf, out, topleft, rot, ci, go) in enumerate(cams):
        s = streams[i % len(streams)].cuda_stream if streams else None
        gm, gc = per_cam[i]
        lib.vol_render_backward_sh_ordered(N, buf.D_cap, _p(buf.mean2d), _p(buf.cov2d), _p(P["sh"]), _p(P["alpha"]),
                                           _p(buf.start), _p(buf.end), _p(buf.ids), _p(out), _p(gm), _p(gc), _p(gsh), _p(ga),
                                           _p(go), _p(topleft), _p(rot), 16, buf.nth, buf.ntw, 1 / ci.fx, 1 / ci.fy, H, W, C,
                                           1e-4, None, buf.tile_order(), s)
    torch.cuda.synchronize()
    return [x.cpu().numpy() for pair in per_cam for x in pair] + [gsh.cpu().numpy(), ga.cpu().numpy()]


def _worst(a, b):
    return max(float(np.abs(x - y).max() / (np.abs(x).max() + 1e-30)) for x, y in zip(a, b))


@pytest.mark.parametrize("C", [1, 2, 3, 4])
def test_sh_backward_identical_launches_agree(C):
    """one big launch (2500 tiles: every CU holds several wavefronts of the kernel), 25 times"""
    N, W, H = 60_000, 800, 800
    dev, lib, P, cams = _setup(C, N, W, H, 1)
    ref = _backward_all(dev, lib, P, cams, C, N, W, H, None)
    for _ in range(24):  # the hazard this guards against showed up about once in 15 launches
        again = _backward_all(dev, lib, P, cams, C, N, W, H, None)
        assert _worst(ref, again) < 5e-6


@pytest.mark.parametrize("C", [1, 2, 3, 4])
def test_sh_backward_unchanged_by_launches_in_flight(C):
    """five cameras' launches on three streams == the same launches one after another"""
    N, W, H = 5000, 176, 128
    dev, lib, P, cams = _setup(C, N, W, H, 5)
    seq = _backward_all(dev, lib, P, cams, C, N, W, H, None)
    streams = [torch.cuda.Stream() for _ in range(3)]
    for _ in range(16):
        con = _backward_all(dev, lib, P, cams, C, N, W, H, streams)
        assert _worst(seq, con) < 5e-6


def test_batched_geometry_chain_identical_with_other_batches_in_flight():
    """The binning count pass finishes its own job: the last workgroup of every tile group scans the group's counts, the last of
    those the tile offsets -- data crossing workgroups (and XCDs, each with its own L2) INSIDE one launch.  Three camera
    batches of the bench workload on three streams, many times: the forward has no atomics and the per-tile order is unique,
    so every image must equal the batch rendered alone, bit for bit (a stale count read by a scanning workgroup would move
    list entries between tiles)."""
    from gsgen_amd.batch import BatchRenderer
    dev = torch.device("cuda:0")
    N, W, H, B = 100_000, 800, 800, 8
    sc = scenes.pointe_scene(N, seed=5, C=4)
    P = {k: torch.from_numpy(np.ascontiguousarray(sc[k])).to(dev) for k in ("mean", "qvec", "svec", "alpha", "sh")}
    cams = [scenes.Camera(W, H, fx=float(W), c2w=scenes.orbit(2.4, 12.0 + 9 * i, 45.0 * i)) for i in range(B)]
    cis = [R.CameraInfo(*c.intr) for c in cams]
    c2ws = [c.c2w for c in cams]
    brs = [BatchRenderer(N, W, H, dev, max_batch=B) for _ in range(3)]
    streams = [torch.cuda.Stream() for _ in brs]

    def render(br):
        with torch.no_grad():
            return br.render(P["mean"], P["qvec"], P["svec"], P["alpha"], P["sh"], cis, c2ws, C=4)[0]
    for br in brs:  # size the pair buffers
        for _ in range(3):
            ref = render(br)
            if br.ensure_capacity(B):
                break
    torch.cuda.synchronize()
    ref = render(brs[0]).clone()
    torch.cuda.synchronize()
    assert float(ref.abs().max()) > 0.1
    for it in range(12):
        outs = []
        for br, st in zip(brs, streams):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                outs.append(render(br))
        torch.cuda.synchronize()
        for k, o in enumerate(outs):
            assert torch.equal(o, ref), (it, k, float((o - ref).abs().max()))
