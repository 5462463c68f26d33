"""GPU: the HIP path (C ABI) against the golden vectors produced by the reference itself
(tests/golden/*.npz, see make_golden.py).  /root/reference is not needed at run time."""
import glob
import os

import numpy as np
import pytest
import torch

import scenes

pytestmark = pytest.mark.gpu
GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def dev():
    return torch.device("cuda:0")


_KEEP = []


def T_(a):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    _KEEP.append(t)
    return t


def p(t):
    return t.data_ptr() if t is not None else None


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(q)[:-4] for q in GOLD])
def test_hip_path_reproduces_reference_golden(path):
    from gsgen_amd import _capi
    L = _capi.load()
    s = torch.cuda.current_stream().cuda_stream
    g = np.load(path)
    fx, fy, cx, cy, w, h, near, far = g["cam_intr"]
    w, h = int(w), int(h)
    m = g["mask"]
    mean, q, sv = T_(g["in_mean"]), T_(g["in_qvec"]), T_(g["in_svec"])
    mask = torch.zeros(mean.shape[0], dtype=torch.bool, device=dev())
    L.culling_gaussian_bsphere(mean.shape[0], p(mean), p(q), p(sv), p(T_(g["frustum_normals"])), p(T_(g["frustum_pts"])),
                               p(mask), 6.0, s)
    assert np.array_equal(mask.cpu().numpy(), m)
    mean, q, sv = T_(g["in_mean"][m]), T_(g["in_qvec"][m]), T_(g["in_svec"][m])
    N = mean.shape[0]
    m2 = torch.empty(N, 2, device=dev()); c2 = torch.empty(N, 2, 2, device=dev())
    JW = torch.empty(N, 3, 3, device=dev()); dep = torch.empty(N, 1, device=dev())
    c2w = T_(g["c2w"])
    L.project_gaussians(N, p(mean), p(q), p(sv), p(c2w), p(m2), p(c2), p(JW), p(dep), s)
    for a, k in ((m2, "mean2d"), (c2, "cov2d"), (dep, "depth"), (JW, "JW")):
        # torch's BLAS sums the 3-term dot products in its own order: a few ulp, not bit-exact
        assert np.abs(a.cpu().numpy() - g[k]).max() <= 4e-6 * max(1e-3, np.abs(g[k]).max()), k
    # downstream stages are fed the reference's own projection so that they can be exact
    m2, c2, dep = T_(g["mean2d"]), T_(g["cov2d"]), T_(g["depth"])
    tl = torch.empty(N, 2, dtype=torch.int32, device=dev()); br = torch.empty_like(tl)
    tot = torch.zeros(1, dtype=torch.int32, device=dev())
    L.tile_culling_aabb_count(N, p(m2), p(c2), 16, fx, fy, cx, cy, w, h, 6.0, p(tl), p(br), p(tot), s)
    D = int(tot.item())
    assert D == int(g["D"]) and np.array_equal(tl.cpu().numpy(), g["tl"]) and np.array_equal(br.cpu().numpy(), g["br"])
    nth, ntw = (h + 15) // 16, (w + 15) // 16
    ids = torch.zeros(D, dtype=torch.int32, device=dev())
    st = -torch.ones(nth * ntw, dtype=torch.int32, device=dev()); en = -torch.ones_like(st)
    nb = L.tile_culling_workspace_bytes(N, D, nth * ntw)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev())
    L.tile_culling_aabb_start_end(N, D, nth, ntw, p(tl), p(br), p(dep), p(ids), p(st), p(en), p(ws), nb, s)
    assert np.array_equal(ids.cpu().numpy(), g["ids"])
    assert np.array_equal(st.cpu().numpy(), g["start"]) and np.array_equal(en.cpu().numpy(), g["end"])
    topleft = T_(np.array([-cx / fx, -cy / fy], np.float32))
    col, al = T_(g["in_color"][m]), T_(g["in_alpha"][m])
    out = torch.zeros(h, w, 3, device=dev()); T = torch.ones(h, w, 1, device=dev())
    L.vol_render_start_end_with_T(N, D, p(m2), p(c2), p(col), p(al), p(st), p(en), p(ids), p(out), p(topleft), 16, nth,
                                  ntw, 1 / fx, 1 / fy, h, w, 1e-4, p(T), s)
    assert np.abs(out.cpu().numpy() - g["rgb"]).max() <= 1e-4
    assert np.abs(T.cpu().numpy() - g["T"]).max() <= 1e-4
    final = T_((g["rgb"] + g["T"] * g["bg_img"]).astype(np.float32))
    go = T_(g["grad_out"])
    gm = torch.zeros(N, 2, device=dev()); gc = torch.zeros(N, 2, 2, device=dev())
    gcol = torch.zeros(N, 3, device=dev()); ga = torch.zeros(N, device=dev())
    L.vol_render_backward_start_end(N, D, p(m2), p(c2), p(col), p(al), p(st), p(en), p(ids), p(final), p(gm), p(gc),
                                    p(gcol), p(ga), p(go), p(topleft), 16, nth, ntw, 1 / fx, 1 / fy, h, w, 1e-4, s)
    for a, k in ((gm, "rgb_gmean"), (gc, "rgb_gcov"), (gcol, "rgb_gcol"), (ga, "rgb_galpha")):
        assert rel(a.cpu().numpy(), g[k]) < 1e-3, k
    g3m, g3q, g3s = torch.empty_like(mean), torch.empty_like(q), torch.empty_like(sv)
    L.project_gaussians_backward(N, p(mean), p(q), p(sv), p(c2w), 1, p(T_(g["rgb_gmean"])), p(T_(g["rgb_gcov"])), None,
                                 p(g3m), p(g3q), p(g3s), s)
    for a, k in ((g3m, "proj_gmean"), (g3q, "proj_gqvec"), (g3s, "proj_gsvec")):
        assert rel(a.cpu().numpy(), g[k]) < 1e-3, k
    C = int(g["C"])
    rot = T_(np.ascontiguousarray(g["c2w"][:3, :3]).reshape(-1).copy())
    sh = T_(g["in_sh"][m])
    for tag, bg in (("sh", None), ("shbg", T_(g["bg_rgb"]))):
        img = torch.zeros(h, w, 3, device=dev())
        L.vol_render_sh(N, D, p(m2), p(c2), p(sh), p(al), p(st), p(en), p(ids), p(img), p(topleft), p(rot), 16, nth, ntw,
                        1 / fx, 1 / fy, h, w, C, 1e-4, p(bg), None, s)
        # every pixel within 1e-4 of the reference's own output (north_star); scenes.assert_sh_image_parity names the
        # one admissible kind of exception (a decision within a few ulps of its threshold in the reference itself)
        scenes.assert_sh_image_parity(img.cpu().numpy(), g[tag + "_img"], g["mean2d"], g["cov2d"], g["in_alpha"][m],
                                      g["start"], g["end"], g["ids"], np.array([-cx / fx, -cy / fy], np.float32),
                                      1 / fx, 1 / fy, what=tag)
        gm = torch.zeros(N, 2, device=dev()); gc = torch.zeros(N, 2, 2, device=dev())
        gsh = torch.zeros(N, 3, C * C, device=dev()); ga = torch.zeros(N, device=dev())
        L.vol_render_backward_sh(N, D, p(m2), p(c2), p(sh), p(al), p(st), p(en), p(ids), p(T_(g[tag + "_img"])), p(gm),
                                 p(gc), p(gsh), p(ga), p(go), p(topleft), p(rot), 16, nth, ntw, 1 / fx, 1 / fy, h, w, C,
                                 1e-4, p(bg), s)
        tol = 1e-3
        for a, k in ((gm, "_gmean"), (gc, "_gcov"), (gsh, "_gsh"), (ga, "_galpha")):
            assert rel(a.cpu().numpy(), g[tag + k]) < tol, tag + k
