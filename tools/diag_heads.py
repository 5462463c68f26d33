import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import scenes
from oracle import oracle as O
from gsgen_amd import renderer as R
from gsgen_amd.batch import BatchRenderer
dev = torch.device("cuda:0")
T_ = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
sc = scenes.pointe_scene(100_000, seed=0, C=1)
rng = np.random.default_rng(3)
sc["svec"] = (sc["svec"] * np.exp(rng.normal(0, 0.3, sc["svec"].shape))).astype(np.float32)
N = sc["mean"].shape[0]; W = H = 800
cams = [scenes.Camera(W, H, fx=800.0, c2w=scenes.orbit(2.5, 15, 30)), scenes.Camera(W, H, fx=640.0, c2w=scenes.orbit(2.2, 40, -75))]
cis = [R.CameraInfo(*c.intr) for c in cams]
keys = ("mean", "qvec", "svec", "alpha", "color")
for packed in (1, 0):
    from gsgen_amd import _capi
    _capi.load().set_variant("chan_packed", packed)
    P = {k: T_(sc[k]).requires_grad_(True) for k in keys}
    br = BatchRenderer(N, W, H, dev, max_batch=2)
    for _ in range(2):
        rgb, dimg, opac, z2, T = br.render_heads(P["mean"], P["qvec"], P["svec"], P["alpha"], P["color"], cis, [c.c2w for c in cams], detach_depth=False)
        if br.ensure_capacity(2): break
    for i, cam in enumerate(cams):
        g = scenes.oracle_geometry(sc, cam); m = g["mask"]
        a = (g["mean2d"], g["cov2d"]); al, col, dv = sc["alpha"][m], sc["color"][m], g["depth"].ravel()
        geo = (g["start"], g["end"], g["ids"], cam.topleft, 1 / cam.fx, 1 / cam.fy, H, W)
        o_rgb, o_T = O.render_rgb_fwd(*a, col, al, *geo)
        e = np.abs(rgb[i].detach().cpu().numpy() - o_rgb).max(-1)
        print("packed", packed, "cam", i, "rgb max err", e.max(), "n>1e-4", int((e > 1e-4).sum()), "T err", np.abs(T[i, ..., 0].cpu().numpy() - o_T.reshape(H, W)).max())
        for name, val, img in (("depth", dv, dimg), ("opac", np.ones_like(dv), opac), ("z2", dv * dv, z2)):
            o_s, _ = O.render_scalar_fwd(*a, val, al, *geo)
            e = np.abs(img[i, ..., 0].detach().cpu().numpy() - o_s)
            tol = 1e-4 * max(1.0, np.abs(o_s).max())
            bad = np.argwhere(e > tol)
            print("   ", name, "max", np.abs(o_s).max(), "max err", e.max(), "n>tol", len(bad), "tol", tol, bad[:3].tolist(), [float(o_T.reshape(H, W)[y, x]) for y, x in bad[:3]])
