"""Pair-list overflow on the MI355X: a camera that needs more (tile, Gaussian) pairs than its
list holds is NEVER rendered as a finite blank image.  Through render_frame, BatchRenderer.render, BatchRenderer.render_heads and
a replayed hipGraph it either reproduces the oracle's image (default sizing and strict=True: the count is read back, the list
regrown, the frame binned again) or comes out as NaN and the next render / check_overflow() raises PairListOverflow with the
lists regrown -- every batch, every view (nothing is sampled), no host sync on the reporting path."""
import numpy as np
import pytest
import torch

import scenes
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def T_(a):
    return torch.tensor(np.ascontiguousarray(a), device=dev())


@pytest.fixture(scope="module")
def setup():
    from gsgen_amd import renderer as R
    sc = scenes.random_scene(3000, seed=4, svec=0.03, C=2)
    W, H = 96, 64
    # camera 2 has the whole cloud in view (3 808 pairs), the others are zoomed in on its middle (about 2 170 each)
    cams = [scenes.Camera(W, H, fx=90.0 if i == 2 else 260.0, c2w=scenes.orbit(2.4, 10 + 5 * i, 60.0 * i)) for i in range(4)]
    cis = [R.CameraInfo(*c.intr) for c in cams]
    refs, refs_rgb, Ds = [], [], []
    for cam in cams:
        g = scenes.oracle_geometry(sc, cam)
        m = g["mask"]
        rot = cam.c2w[:3, :3].reshape(-1)
        refs.append(O.render_sh_fwd(g["mean2d"], g["cov2d"], sc["sh"][m], sc["alpha"][m], g["start"], g["end"], g["ids"],
                                    cam.topleft, rot, 2, 1 / cam.fx, 1 / cam.fy, H, W))
        refs_rgb.append(O.render_rgb_fwd(g["mean2d"], g["cov2d"], sc["color"][m], sc["alpha"][m], g["start"], g["end"], g["ids"],
                                         cam.topleft, 1 / cam.fx, 1 / cam.fy, H, W)[0])
        Ds.append(g["D"])
    P = {k: T_(sc[k]) for k in ("mean", "qvec", "svec", "alpha", "sh", "color")}
    return dict(sc=sc, W=W, H=H, cams=cams, cis=cis, refs=refs, refs_rgb=refs_rgb, Ds=Ds, P=P, N=sc["mean"].shape[0])


def test_host_visible_report_words_are_written_by_the_geometry_launch(setup):
    """the mechanism itself: pinned, device-mapped words receive every frame's count and keep the largest overflow"""
    from gsgen_amd import renderer as R
    s = setup
    P, cam, ci = s["P"], s["cams"][0], s["cis"][0]
    buf = R.FrameBuffers(s["N"], s["W"], s["H"], dev(), D_cap=64)
    assert buf.report_ptr() is not None, "pinned host memory is not mapped into the device's address space"
    R.frame_geometry(P["mean"], P["qvec"], P["svec"], T_(ci.pack(cam.c2w)), buf)
    torch.cuda.synchronize()
    assert buf._report.last(0) == s["Ds"][0] and buf._report.overflow(0) == s["Ds"][0]
    assert int((buf.start == -2).sum().item()) == buf.start.numel()  # GSGEN_LIST_OVERFLOW, not "empty"
    big = R.FrameBuffers(s["N"], s["W"], s["H"], dev(), D_cap=s["Ds"][0] * 4)
    R.frame_geometry(P["mean"], P["qvec"], P["svec"], T_(ci.pack(cam.c2w)), big)
    torch.cuda.synchronize()
    assert big._report.last(0) == s["Ds"][0] and big._report.overflow(0) == 0


@pytest.mark.parametrize("mode", ["default", "strict"])
def test_render_frame_is_lossless_by_default_and_under_strict(setup, mode):
    from gsgen_amd import renderer as R
    s = setup
    P = s["P"]
    if mode == "default":  # nobody has sized the list: the first frame is rendered synchronously and sizes it
        buf = R.FrameBuffers(s["N"], s["W"], s["H"], dev())
        buf._alloc_pairs(64)  # (a default far too small for this scene)
        buf.sized = False
    else:
        buf = R.FrameBuffers(s["N"], s["W"], s["H"], dev(), D_cap=64, strict=True)
    for k in range(3):
        img, T = R.render_frame(P["mean"], P["qvec"], P["svec"], P["alpha"], P["sh"], s["cis"][k], s["cams"][k].c2w, buf, C=2)
        assert np.abs(img.cpu().numpy() - s["refs"][k]).max() <= 1e-4, (mode, k)
    assert buf.D_cap >= max(s["Ds"][:3])


def test_render_frame_overflow_is_nan_and_raises(setup):
    from gsgen_amd import renderer as R
    from gsgen_amd import PairListOverflow
    s = setup
    P = {k: v.clone().requires_grad_(True) for k, v in s["P"].items()}
    small = R.FrameBuffers(s["N"], s["W"], s["H"], dev(), D_cap=64)  # the caller's own capacity: it answers for it
    img, T = R.render_frame(P["mean"], P["qvec"], P["svec"], P["alpha"], P["sh"], s["cis"][0], s["cams"][0].c2w, small, C=2)
    assert bool(torch.isnan(img).all()) and bool(torch.isnan(T).all())
    img.nan_to_num().sum().backward()  # the backward of such a frame runs and contributes nothing
    assert float(P["sh"].grad.abs().max()) == 0.0 and float(P["mean"].grad.abs().max()) == 0.0
    torch.cuda.synchronize()
    with pytest.raises(PairListOverflow, match="NaN"):
        R.render_frame(P["mean"], P["qvec"], P["svec"], P["alpha"], P["sh"], s["cis"][0], s["cams"][0].c2w, small, C=2)
    assert small.D_cap >= s["Ds"][0]  # regrown before the exception left: repeating the step succeeds
    img2, _ = R.render_frame(P["mean"], P["qvec"], P["svec"], P["alpha"], P["sh"], s["cis"][0], s["cams"][0].c2w, small, C=2)
    assert np.abs(img2.detach().cpu().numpy() - s["refs"][0]).max() <= 1e-4


@pytest.mark.parametrize("pipeline", [False, True])
@pytest.mark.parametrize("heads", [False, True])
def test_batch_renderer_overflow_every_batch_every_view(setup, heads, pipeline):
    """an explicit capacity that fits three of the four cameras' neighbours but not camera `bad`: that view alone is NaN, in
    EVERY batch (nothing is sampled), the next call raises, the lists are regrown, the repeated step is the oracle's"""
    from gsgen_amd.batch import BatchRenderer
    from gsgen_amd import PairListOverflow
    s = setup
    P, cis, c2ws = s["P"], s["cis"], [c.c2w for c in s["cams"]]
    bad = int(np.argmax(s["Ds"]))
    cap = int(1.3 * sorted(s["Ds"])[-2])  # (beyond the 25 % margin inside which the lists are regrown BEFORE they overflow)
    assert bad == 2 and s["Ds"][bad] > 1.2 * cap

    ok = (bad + 1) % 4  # a batch in which camera `ok` stands in for camera `bad` fits

    def render(br, with_bad=True):
        idx = list(range(4)) if with_bad else [i if i != bad else ok for i in range(4)]
        ci_, cw_ = [cis[i] for i in idx], [c2ws[i] for i in idx]
        if heads:
            return br.render_heads(P["mean"], P["qvec"], P["svec"], P["alpha"], P["color"], ci_, cw_)[0], idx
        return br.render(P["mean"], P["qvec"], P["svec"], P["alpha"], P["sh"], ci_, cw_, C=2)[0], idx

    refs = s["refs_rgb"] if heads else s["refs"]
    for trial in range(4):  # the overflowing batch is the 1st, 2nd, 3rd, 4th of a fresh renderer: each is caught
        br = BatchRenderer(s["N"], s["W"], s["H"], dev(), max_batch=4, D_cap=cap, pipeline=pipeline)
        with torch.no_grad():
            for _ in range(trial):
                img, idx = render(br, with_bad=False)
                for i in range(4):
                    assert np.abs(img[i].cpu().numpy() - refs[idx[i]]).max() <= 1e-4
            img, _ = render(br)
            for i in range(4):
                if i == bad:
                    assert bool(torch.isnan(img[i]).all()), (trial, i)
                else:
                    assert np.abs(img[i].cpu().numpy() - refs[i]).max() <= 1e-4, (trial, i)
            torch.cuda.synchronize()
            with pytest.raises(PairListOverflow, match=rf"camera\(s\) \[{bad}\]"):
                render(br)
            img, _ = render(br)
            for i in range(4):
                assert np.abs(img[i].cpu().numpy() - refs[i]).max() <= 1e-4, (trial, i)
            torch.cuda.synchronize()
            assert br.check_overflow()


@pytest.mark.parametrize("mode", ["default", "strict"])
def test_batch_renderer_lossless_modes(setup, mode):
    from gsgen_amd.batch import BatchRenderer
    s = setup
    P, cis, c2ws = s["P"], s["cis"], [c.c2w for c in s["cams"]]
    br = BatchRenderer(s["N"], s["W"], s["H"], dev(), max_batch=4, D_cap=(None if mode == "default" else 64), strict=(mode == "strict"))
    if mode == "default":
        for sl in br.slots:
            sl._alloc_pairs(64)
            sl.sized = False
    with torch.no_grad():
        img = br.render(P["mean"], P["qvec"], P["svec"], P["alpha"], P["sh"], cis, c2ws, C=2)[0]
        rgb = br.render_heads(P["mean"], P["qvec"], P["svec"], P["alpha"], P["color"], cis, c2ws)[0]
    for i in range(4):
        assert np.abs(img[i].cpu().numpy() - s["refs"][i]).max() <= 1e-4
        assert np.abs(rgb[i].cpu().numpy() - s["refs_rgb"][i]).max() <= 1e-4
    assert br.slots[0].D_cap >= max(s["Ds"])


def test_lists_are_regrown_before_a_growing_scene_overflows(setup):
    """proactive regrowth from the counts every batch reports: the scene's scales grow 4 % a step -- no frame is ever NaN,
    nothing raises, no host sync after the first batch"""
    from gsgen_amd.batch import BatchRenderer
    s = setup
    P, cis, c2ws = s["P"], s["cis"], [c.c2w for c in s["cams"]]
    br = BatchRenderer(s["N"], s["W"], s["H"], dev(), max_batch=4, D_cap=int(1.3 * max(s["Ds"])))
    caps = []
    with torch.no_grad():
        sv = P["svec"].clone()
        for step in range(40):
            img = br.render(P["mean"], P["qvec"], sv, P["alpha"], P["sh"], cis, c2ws, C=2)[0]
            assert bool(torch.isfinite(img).all()), step
            torch.cuda.synchronize()  # (a training step's worth of time passes: the report has landed)
            sv = sv * 1.04
            caps.append(br.slots[0].D_cap)
    assert caps[-1] > 2 * caps[0]


@pytest.mark.parametrize("heads", [False, True])
def test_a_quiet_regrow_protects_the_step_that_triggered_it(setup, heads):
    """on the C++ fast path the Plan used to be fetched BEFORE the overflow check, so the batch whose check regrew the
    lists still ran on the old Plan (old buffers, old capacity) while slots[i].ids / tile_order() already named the new ones.  Now
    the check comes first: the step that triggers the regrow bins into the NEW lists -- they hold the oracle's lists afterwards,
    ensure_capacity() speaks about the frame that was actually rendered, and the backward runs."""
    from gsgen_amd.batch import BatchRenderer
    s = setup
    P, cis, c2ws = s["P"], s["cis"], [c.c2w for c in s["cams"]]
    br = BatchRenderer(s["N"], s["W"], s["H"], dev(), max_batch=4, D_cap=int(1.3 * max(s["Ds"])))
    assert br.use_ext
    sv = P["svec"].clone()
    regrown = 0
    for step in range(24):
        cap0 = br.slots[0].D_cap
        mean = P["mean"].clone().requires_grad_(True)
        if heads:
            img = br.render_heads(mean, P["qvec"], sv, P["alpha"], P["color"], cis, c2ws)[0]
        else:
            img = br.render(mean, P["qvec"], sv, P["alpha"], P["sh"], cis, c2ws, C=2)[0]
        img.sum().backward()
        torch.cuda.synchronize()
        assert bool(torch.isfinite(img).all()) and bool(torch.isfinite(mean.grad).all()), step
        if br.slots[0].D_cap != cap0:  # this step's own check regrew the lists
            regrown += 1
            sc2 = dict(s["sc"]); sc2["svec"] = sv.cpu().numpy()
            for i in (0, 2):
                g = scenes.oracle_geometry(sc2, s["cams"][i])
                nz = np.nonzero(g["mask"])[0]
                sl = br.slots[i]
                assert int(sl.total.item()) == g["D"] <= sl.D_cap
                assert np.array_equal(sl.start.cpu().numpy(), g["start"]) and np.array_equal(sl.end.cpu().numpy(), g["end"])
                assert np.array_equal(sl.ids[:g["D"]].cpu().numpy(), nz[g["ids"]])
            assert br.ensure_capacity(4)
        sv = sv * 1.05
    assert regrown >= 2


def test_capture_requires_sized_lists_and_a_replay_reports_overflow(setup):
    from gsgen_amd.batch import BatchRenderer
    from gsgen_amd import PairListOverflow
    s = setup
    P, cis, c2ws = s["P"], s["cis"], [c.c2w for c in s["cams"]]
    sv = P["svec"].clone()

    def step(br):
        return br.render(P["mean"], P["qvec"], sv, P["alpha"], P["sh"], cis, c2ws, C=2)[0]

    side = torch.cuda.Stream()
    # (1) nobody has sized the lists: capturing is an error, not a silent risk
    br0 = BatchRenderer(s["N"], s["W"], s["H"], dev(), max_batch=4)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        g0 = torch.cuda.CUDAGraph()
        with pytest.raises(RuntimeError, match="nobody has sized"):
            with torch.cuda.graph(g0, stream=side):
                step(br0)
    torch.cuda.synchronize()
    # (2) sized eagerly, captured, replayed on a scene that has outgrown the lists meanwhile: NaN + a report
    br = BatchRenderer(s["N"], s["W"], s["H"], dev(), max_batch=4, D_cap=int(1.5 * max(s["Ds"])))
    with torch.no_grad():
        step(br)
        torch.cuda.synchronize()
        assert br.ensure_capacity(4)
        cap0 = br.slots[0].D_cap
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step(br)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                img = step(br)
            graph.replay()
            torch.cuda.synchronize()
            for i in range(4):
                assert np.abs(img[i].cpu().numpy() - s["refs"][i]).max() <= 1e-4
            assert br.check_overflow()
            sv.mul_(3.0)  # the captured step reads `sv` in place
            graph.replay()
            torch.cuda.synchronize()
            assert bool(torch.isnan(img).any())
            with pytest.raises(PairListOverflow):
                br.check_overflow()
            assert br.slots[0].D_cap > cap0
