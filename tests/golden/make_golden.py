"""Generates tests/golden/*.npz from the REFERENCE ITSELF (run in the authoring container only).

Two sources, both the reference's own code executed here, nothing restated:
  * Python half: gs/renderer.py project_gaussians, gs/culling.py tile_culling_aabb_count,
    utils/camera.py CameraInfo.get_frustum -- imported from /root/reference through
    tests/refshim.py (torch CPU), forward and autograd backward.
  * CUDA half: gs/src/include/*.h compiled for the CPU by oracle/ref_build.py
    (oracle/_ref/libgs_ref.so): cull mask, per-tile sorted lists, RGB / scalar / SH images,
    transmittance and all gradients.
Scenes: the reference's MockRenderer pair of Gaussians (gs/debug.py:52-68, camera :380-398,
downsampled so the fixture stays small) and seeded random clouds.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import refshim  # noqa: E402
import scenes  # noqa: E402
from oracle import ref as Rf, ref_build  # noqa: E402


def mock_scene():
    """gs/debug.py:52-68 (default layout), parameters after the reference's activations
    (svec = exp, alpha = sigmoid: conf/base.yaml:141-143)."""
    s = {}
    s["mean"] = np.array([[0.0, 0.0, 0.0], [0.1, 0.07, 0.0]], np.float32)
    s["qvec"] = np.array([[1.0, 0, 0, 0], [1.0, 0, 0, 0]], np.float32)
    log_svec = np.array([[1.0, 1.0, 0.5], [1.1, 0.5, 1.1]], np.float32) * np.float32(np.log(20e-3))
    s["svec"] = np.exp(log_svec).astype(np.float32)
    s["color"] = np.array([[0.01, 0.01, 0.99], [0.01, 0.99, 0.01]], np.float32)
    s["alpha"] = (1.0 / (1.0 + np.exp(-np.array([10000.0, 10000.0])))).astype(np.float32)
    sh = np.zeros((2, 3, 4), np.float32)
    sh[:, :, 0] = np.log(s["color"] / (1 - s["color"])) / 0.28209479177387814
    sh[:, :, 1:] = np.random.default_rng(0).normal(0, 0.3, (2, 3, 3))
    s["sh"], s["C"] = sh, 2
    # CameraInfo(961.22, 963.09, 648.38, 420.12, 1297, 840, 0, 1000) looking from (1,0,0) at the origin, /8
    cam = scenes.Camera(162, 105, fx=961.22 / 8, fy=963.09 / 8, cx=648.38 / 8, cy=420.12 / 8, near=0.0, far=1000.0,
                        c2w=scenes.look_at((1.0, 0.0, 0.0)))
    return s, cam


def long_list_scene():
    """tile lists that cross every LDS batch boundary of the reference's kernels -- 1 200 entries in the RGB
    forward, 472 in its backward (vol_render.h:501, :442), 218 / 109 in the SH degree-3 forward / backward
    (vol_render_sh.h:171-248, :353-455) -- on a 32 x 32 image (4 tiles) that nearly every splat of a dense cloud touches.  The
    splats on the left of the view are almost transparent (pixels there walk their whole list: deep walks), those on the right
    are ordinary (pixels saturate and stop early: early termination), so both exits of the entry loop are taken."""
    sc = scenes.random_scene(3200, seed=21, svec=0.09, spread=0.5, C=4)
    left = sc["mean"][:, 1] < 0.0
    sc["alpha"] = np.where(left, sc["alpha"] * 0.02, sc["alpha"]).astype(np.float32)
    return sc, scenes.Camera(32, 32, fx=26.0, c2w=scenes.look_at((2.6, 0.0, 0.3)))


CASES = {
    "long_lists": long_list_scene,
    "mock2": mock_scene,
    "rand_c1": lambda: (scenes.random_scene(400, seed=11, svec=0.05, C=1), scenes.Camera(96, 64, fx=90.0)),
    "rand_c3": lambda: (scenes.random_scene(500, seed=12, svec=0.06, C=3),
                        scenes.Camera(80, 72, fx=70.0, fy=75.0, cx=41.3, cy=35.2, c2w=scenes.orbit(2.3, 25, 130))),
    "rand_c4": lambda: (scenes.random_scene(600, seed=13, svec=0.05, C=4),
                        scenes.Camera(70, 50, fx=64.0, c2w=scenes.orbit(2.6, -10, 300))),
}


def generate(name):
    project_gaussians, tile_count, CameraInfo = refshim.reference_api()
    sc, cam = CASES[name]()
    out = {"cam_intr": np.array(cam.intr, np.float64), "c2w": cam.c2w}
    for k in ("mean", "qvec", "svec", "color", "alpha", "sh"):
        out["in_" + k] = sc[k]
    C = sc["C"]
    ci = CameraInfo(*cam.intr)
    c2w_t = torch.tensor(cam.c2w)
    fn, fp = ci.get_frustum(c2w_t)
    out["frustum_normals"], out["frustum_pts"] = fn.numpy(), fp.numpy()
    mask = Rf.cull_bsphere(sc["mean"], sc["qvec"], sc["svec"], fn.numpy(), fp.numpy(), 6.0)
    out["mask"] = mask
    m = mask
    tm, tq, ts = (torch.tensor(sc[k][m], requires_grad=True) for k in ("mean", "qvec", "svec"))
    mean2d, cov2d, JW, depth = project_gaussians(tm, tq, ts, c2w_t, True)
    out.update(mean2d=mean2d.detach().numpy(), cov2d=cov2d.detach().numpy(), JW=JW.detach().numpy(),
               depth=depth.detach().numpy())
    D, tl, br = tile_count(mean2d.detach(), cov2d.detach(), 16, ci, 6.0)
    out.update(D=np.int64(D), tl=tl.numpy(), br=br.numpy())
    nth, ntw = cam.tiles
    ids, start, end = Rf.bin_sort(tl.numpy(), br.numpy(), out["depth"], nth, ntw, int(D))
    out.update(ids=ids, start=start, end=end)
    H, W = cam.h, cam.w
    a = (out["mean2d"], out["cov2d"])
    col, al = sc["color"][m], sc["alpha"][m]
    geo = (start, end, ids, cam.topleft, 1 / cam.fx, 1 / cam.fy, H, W)
    rng = np.random.default_rng(7)
    go = rng.normal(size=(H, W, 3)).astype(np.float32)
    bgimg = rng.uniform(size=(H, W, 3)).astype(np.float32)
    out.update(grad_out=go, bg_img=bgimg)
    rgb, T = Rf.render_rgb_fwd(*a, col, al, *geo)
    final = (rgb + T * bgimg).astype(np.float32)
    g = Rf.render_rgb_bwd(*a, col, al, start, end, ids, final, go, cam.topleft, 1 / cam.fx, 1 / cam.fy, H, W)
    out.update(rgb=rgb, T=T, rgb_gmean=g[0], rgb_gcov=g[1], rgb_gcol=g[2], rgb_galpha=g[3])
    # projection backward (torch autograd through the reference's python) fed with the rgb 2-D grads
    (mean2d * torch.tensor(g[0])).sum().add((cov2d * torch.tensor(g[1])).sum()).backward()
    out.update(proj_gmean=tm.grad.numpy(), proj_gqvec=tq.grad.numpy(), proj_gsvec=ts.grad.numpy())
    sv = out["depth"].ravel()
    s_img, sT = Rf.render_scalar_fwd(*a, sv, al, *geo)
    g = Rf.render_scalar_bwd(*a, sv, al, start, end, ids, s_img, go[..., 0].copy(), cam.topleft, 1 / cam.fx,
                             1 / cam.fy, H, W)
    out.update(depth_img=s_img, depth_T=sT, sc_gmean=g[0], sc_gcov=g[1], sc_gscalar=g[2], sc_galpha=g[3])
    rot = cam.c2w[:3, :3].reshape(-1).copy()
    bg = np.array([0.2, 0.5, 0.7], np.float32)
    sh = np.ascontiguousarray(sc["sh"][m])
    for tag, b in (("sh", None), ("shbg", bg)):
        img = Rf.render_sh_fwd(*a, sh, al, start, end, ids, cam.topleft, rot, C, 1 / cam.fx, 1 / cam.fy, H, W, bg=b)
        g = Rf.render_sh_bwd(*a, sh, al, start, end, ids, img, go, cam.topleft, rot, C, 1 / cam.fx, 1 / cam.fy, H, W, bg=b)
        out.update({tag + "_img": img, tag + "_gmean": g[0], tag + "_gcov": g[1], tag + "_gsh": g[2],
                    tag + "_galpha": g[3]})
    out["bg_rgb"] = bg
    out["C"] = np.int64(C)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    return out


if __name__ == "__main__":
    if not refshim.available():
        raise SystemExit("needs /root/reference")
    ref_build.build()
    for n in (sys.argv[1:] or CASES):
        o = generate(n)
        print(n, "N_visible", int(o["mask"].sum()), "D", int(o["D"]), "longest list", int((o["end"] - o["start"]).max()),
              "T range", float(o["T"].min()), float(o["T"].max()), "rgb mean", float(o["rgb"].mean()),
              os.path.getsize(os.path.join(HERE, n + ".npz")) // 1024, "KiB")
