"""numpy front-end of the CPU oracle (oracle/gs_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under gsgen_amd/ may import this module.

Each function mirrors one stage of the reference hot path; the
reference file:line each follows is cited in gs_oracle.c next to the C function.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libgs_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "gs_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.gso_aabb_count.restype = C.c_longlong
        _lib.gso_bin_sort.restype = C.c_int
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def camera_consts(fx, fy, cx, cy, w, h, near, far):
    """yfov/aspect/half sides exactly as utils/camera.py:225-226,265-267 (python doubles)."""
    yfov = 2 * np.arctan(h / (2 * fy))
    aspect = w / h
    half_v = far * np.tan(yfov * 0.5)
    half_h = half_v * aspect
    return float(half_v), float(half_h)


def frustum(c2w, fx, fy, cx, cy, w, h, near, far):
    c2w = _f(c2w)
    hv, hh = camera_consts(fx, fy, cx, cy, w, h, near, far)
    normals = np.zeros((6, 3), np.float32)
    pts = np.zeros((6, 3), np.float32)
    lib().gso_frustum(_p(c2w), C.c_float(near), C.c_float(far), C.c_float(hv), C.c_float(hh),
                      _p(normals), _p(pts))
    return normals, pts


def cull_bsphere(mean, svec, normals, pts, thresh):
    mean, svec, normals, pts = _f(mean), _f(svec), _f(normals), _f(pts)
    mask = np.zeros(mean.shape[0], np.uint8)
    lib().gso_cull_bsphere(mean.shape[0], _p(mean), _p(svec), _p(normals), _p(pts),
                           C.c_float(thresh), _p(mask))
    return mask.astype(bool)


def project(mean, qvec, svec, c2w, detach_depth=True):
    mean, qvec, svec, c2w = _f(mean), _f(qvec), _f(svec), _f(c2w)
    N = mean.shape[0]
    mean2d = np.zeros((N, 2), np.float32)
    cov2d = np.zeros((N, 2, 2), np.float32)
    JW = np.zeros((N, 3, 3), np.float32)
    depth = np.zeros((N, 1), np.float32)
    lib().gso_project(N, _p(mean), _p(qvec), _p(svec), _p(c2w), int(detach_depth), _p(mean2d),
                      _p(cov2d), _p(JW), _p(depth))
    return mean2d, cov2d, JW, depth


def project_bwd(mean, qvec, svec, c2w, g_mean2d, g_cov2d, g_depth=None, detach_depth=True):
    mean, qvec, svec, c2w = _f(mean), _f(qvec), _f(svec), _f(c2w)
    g_mean2d, g_cov2d = _f(g_mean2d), _f(g_cov2d)
    g_depth = _f(g_depth) if g_depth is not None else None
    N = mean.shape[0]
    gm, gq, gs = np.zeros((N, 3), np.float32), np.zeros((N, 4), np.float32), np.zeros((N, 3), np.float32)
    lib().gso_project_bwd(N, _p(mean), _p(qvec), _p(svec), _p(c2w), int(detach_depth), _p(g_mean2d),
                          _p(g_cov2d), _p(g_depth), _p(gm), _p(gq), _p(gs))
    return gm, gq, gs


def aabb_count(mean2d, cov2d, tile_size, fx, fy, cx, cy, w, h, D=6.0):
    mean2d, cov2d = _f(mean2d), _f(cov2d)
    N = mean2d.shape[0]
    tl, br = np.zeros((N, 2), np.int32), np.zeros((N, 2), np.int32)
    n = lib().gso_aabb_count(N, _p(mean2d), _p(cov2d), int(tile_size), C.c_float(fx), C.c_float(fy),
                             C.c_float(cx), C.c_float(cy), int(w), int(h), C.c_float(D), _p(tl), _p(br))
    return int(n), tl, br


def densify_update(cov2d, grad_mean2d, mask, max_radii2d, grad_accum, cnt):
    """In place on the three float32 [N] statistics arrays (any pair may be None)."""
    N = (cov2d if cov2d is not None else grad_mean2d).shape[0]
    keep = [None if a is None else _f(a) for a in (cov2d, grad_mean2d)]
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    for a in (max_radii2d, grad_accum, cnt):
        assert a is None or (a.dtype == np.float32 and a.flags.c_contiguous)
    lib().gso_densify_update(N, _p(keep[0]), _p(keep[1]), _p(m), _p(max_radii2d), _p(grad_accum), _p(cnt))


def bin_sort(tl, br, depth, n_tiles_h, n_tiles_w, D):
    tl, br, depth = _i(tl), _i(br), _f(depth).reshape(-1)
    N = tl.shape[0]
    T = n_tiles_h * n_tiles_w
    ids = np.zeros(max(D, 1), np.int32)
    start, end = np.zeros(T, np.int32), np.zeros(T, np.int32)
    rc = lib().gso_bin_sort(N, C.c_longlong(D), n_tiles_h, n_tiles_w, _p(tl), _p(br), _p(depth),
                            _p(ids), _p(start), _p(end))
    if rc != 0:
        raise RuntimeError("pair count mismatch (aabb_culling.h:228 assert)")
    return ids[:D], start, end


def _tiles(H, W, ts):
    return (H + ts - 1) // ts, (W + ts - 1) // ts


def render_rgb_fwd(mean2d, cov2d, color, alpha, start, end, ids, topleft, psx, psy, H, W,
                   thresh=1e-4, tile_size=16, with_T=True):
    nth, ntw = _tiles(H, W, tile_size)
    out = np.zeros((H, W, 3), np.float32)
    T = np.ones((H, W, 1), np.float32)
    a = [_f(mean2d), _f(cov2d), _f(color), _f(alpha), _i(start), _i(end), _i(ids), _f(topleft)]
    lib().gso_render_rgb_fwd(*[_p(x) for x in a], tile_size, nth, ntw, C.c_float(psx), C.c_float(psy),
                             H, W, C.c_float(thresh), _p(out), _p(T) if with_T else None)
    return out, T


def render_rgb_bwd(mean2d, cov2d, color, alpha, start, end, ids, final, grad_out, topleft, psx, psy,
                   H, W, thresh=1e-4, tile_size=16):
    nth, ntw = _tiles(H, W, tile_size)
    N = np.asarray(mean2d).shape[0]
    gm, gc = np.zeros((N, 2), np.float32), np.zeros((N, 2, 2), np.float32)
    gcol, ga = np.zeros((N, 3), np.float32), np.zeros((N,), np.float32)
    a = [_f(mean2d), _f(cov2d), _f(color), _f(alpha), _i(start), _i(end), _i(ids), _f(final),
         _f(grad_out), _f(topleft)]
    lib().gso_render_rgb_bwd(N, *[_p(x) for x in a], tile_size, nth, ntw, C.c_float(psx),
                             C.c_float(psy), H, W, C.c_float(thresh), _p(gm), _p(gc), _p(gcol), _p(ga))
    return gm, gc, gcol, ga


def render_scalar_fwd(mean2d, cov2d, scalar, alpha, start, end, ids, topleft, psx, psy, H, W,
                      thresh=1e-4, tile_size=16):
    nth, ntw = _tiles(H, W, tile_size)
    out = np.zeros((H, W), np.float32)
    T = np.ones((H, W, 1), np.float32)
    a = [_f(mean2d), _f(cov2d), _f(scalar), _f(alpha), _i(start), _i(end), _i(ids), _f(topleft)]
    lib().gso_render_scalar_fwd(*[_p(x) for x in a], tile_size, nth, ntw, C.c_float(psx),
                                C.c_float(psy), H, W, C.c_float(thresh), _p(out), _p(T))
    return out, T


def render_scalar_bwd(mean2d, cov2d, scalar, alpha, start, end, ids, final, grad_out, topleft, psx,
                      psy, H, W, thresh=1e-4, tile_size=16):
    nth, ntw = _tiles(H, W, tile_size)
    N = np.asarray(mean2d).shape[0]
    gm, gc = np.zeros((N, 2), np.float32), np.zeros((N, 2, 2), np.float32)
    gs, ga = np.zeros((N,), np.float32), np.zeros((N,), np.float32)
    a = [_f(mean2d), _f(cov2d), _f(scalar), _f(alpha), _i(start), _i(end), _i(ids), _f(final),
         _f(grad_out), _f(topleft)]
    lib().gso_render_scalar_bwd(N, *[_p(x) for x in a], tile_size, nth, ntw, C.c_float(psx),
                                C.c_float(psy), H, W, C.c_float(thresh), _p(gm), _p(gc), _p(gs), _p(ga))
    return gm, gc, gs, ga


def sh_basis(dirs, Cb):
    dirs = _f(dirs).reshape(-1, 3)
    Y = np.zeros((dirs.shape[0], Cb * Cb), np.float32)
    for k in range(dirs.shape[0]):
        lib().gso_sh_basis(_p(dirs[k]), int(Cb), _p(Y[k]))
    return Y


def render_sh_fwd(mean2d, cov2d, sh, alpha, start, end, ids, topleft, rot9, Cb, psx, psy, H, W,
                  thresh=1e-4, bg=None, tile_size=16, want_T=False):
    nth, ntw = _tiles(H, W, tile_size)
    out = np.zeros((H, W, 3), np.float32)
    T = np.ones((H, W, 1), np.float32) if want_T else None
    rot9 = _f(rot9).reshape(-1)[:9].copy()
    bg_ = _f(bg) if bg is not None else None
    a = [_f(mean2d), _f(cov2d), _f(sh), _f(alpha), _i(start), _i(end), _i(ids), _f(topleft), rot9]
    lib().gso_render_sh_fwd(*[_p(x) for x in a], int(Cb), _p(bg_), tile_size, nth, ntw,
                            C.c_float(psx), C.c_float(psy), H, W, C.c_float(thresh), _p(out), _p(T))
    return (out, T) if want_T else out


def sh_decision_margin(mean2d, cov2d, alpha, start, end, ids, topleft, psx, psy, H, W, thresh=1e-4, tile_size=16):
    """[H,W,2]: per pixel, how close the SH forward came to flipping a skip test (|a G - 1/255| relative) and a stop
    test (|T - thresh| relative).  Test aid: a pixel off by more than the tolerance must have a margin of a few ulps."""
    nth, ntw = _tiles(H, W, tile_size)
    margin = np.zeros((H, W, 2), np.float32)
    a = [_f(mean2d), _f(cov2d), _f(alpha), _i(start), _i(end), _i(ids), _f(topleft)]
    lib().gso_sh_decision_margin(*[_p(x) for x in a], tile_size, nth, ntw, C.c_float(psx), C.c_float(psy), H, W,
                                 C.c_float(thresh), _p(margin))
    return margin


def render_sh_bwd(mean2d, cov2d, sh, alpha, start, end, ids, final, grad_out, topleft, rot9, Cb, psx,
                  psy, H, W, thresh=1e-4, tile_size=16):
    nth, ntw = _tiles(H, W, tile_size)
    N = np.asarray(mean2d).shape[0]
    gm, gc = np.zeros((N, 2), np.float32), np.zeros((N, 2, 2), np.float32)
    gsh, ga = np.zeros((N, 3, Cb * Cb), np.float32), np.zeros((N,), np.float32)
    rot9 = _f(rot9).reshape(-1)[:9].copy()
    a = [_f(mean2d), _f(cov2d), _f(sh), _f(alpha), _i(start), _i(end), _i(ids), _f(final),
         _f(grad_out), _f(topleft), rot9]
    lib().gso_render_sh_bwd(N, *[_p(x) for x in a], int(Cb), tile_size, nth, ntw, C.c_float(psx),
                            C.c_float(psy), H, W, C.c_float(thresh), _p(gm), _p(gc), _p(gsh), _p(ga))
    return gm, gc, gsh, ga


# ---- legacy binning (tile_ops.h; used by the reference's older GaussianRenderer / debug / benchmarks) -----
def legacy_count(mode, mean2d, shape, topleft, tile_size, nth, ntw, psx, psy, thresh=0.0, num=None):
    mean2d, shape, topleft = _f(mean2d), _f(shape), _f(topleft)
    num = np.zeros(nth * ntw, np.int32) if num is None else num
    lib().gso_legacy_count(int(mode), mean2d.shape[0], _p(mean2d), _p(shape), _p(topleft), C.c_uint(tile_size), nth, ntw,
                           C.c_float(psx), C.c_float(psy), C.c_float(thresh), _p(num))
    return num


def legacy_image_sort(mode, depth, tile_n, mean2d, shape, topleft, tile_size, nth, ntw, psx, psy, thresh=0.0):
    mean2d, shape, topleft, depth = _f(mean2d), _f(shape), _f(topleft), _f(depth).reshape(-1)
    tile_n = _i(tile_n).copy()
    D = int(tile_n.sum())
    ids = np.zeros(max(D, 1), np.int32)
    td = np.zeros(max(D, 1), np.uint64)
    offset = np.zeros(nth * ntw, np.int32)
    lib().gso_legacy_image_sort.restype = C.c_longlong
    tot = lib().gso_legacy_image_sort(int(mode), mean2d.shape[0], C.c_longlong(D), _p(ids), _p(td), _p(depth), _p(tile_n),
                                      _p(offset), _p(mean2d), _p(shape), _p(topleft), C.c_uint(tile_size), nth, ntw,
                                      C.c_float(psx), C.c_float(psy), C.c_float(thresh))
    assert tot == D, (tot, D)
    return ids[:D], td[:D], tile_n, offset


def part_workstats(mean2d, cov2d, alpha, start, end, ids, topleft, psx, psy, H, W, rows_of_part=None, thresh=1e-4):
    """Work statistics of the compositing walk (bench.py's cpu_baseline leg, tools/workstats.py): -> (walked, contributing,
    pixel_pairs) -- (wavefront, list entry) pairs some pixel of the wavefront's part is still alive for, those of them in
    which some pixel's a*G reaches 1/255, and the contributing (pixel, entry) pairs.  rows_of_part[16]: tile row -> wavefront
    (default: one wavefront per tile, the backward's shape)."""
    import ctypes as C
    rows = np.zeros(16, np.int32) if rows_of_part is None else np.ascontiguousarray(rows_of_part, np.int32)
    nth, ntw = _tiles(H, W, 16)
    w, c, pp = C.c_longlong(), C.c_longlong(), C.c_longlong()
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    m2, c2, al, tl = _f(mean2d), _f(cov2d), _f(alpha), _f(topleft)
    st, en, idv = _i(start), _i(end), _i(ids)
    lib().gso_part_workstats(vp(m2), vp(c2), vp(al), vp(st), vp(en), vp(idv), vp(tl), nth, ntw, C.c_float(psx), C.c_float(psy),
                             int(H), int(W), C.c_float(thresh), vp(rows), int(rows.max()) + 1, C.byref(w), C.byref(c), C.byref(pp))
    return int(w.value), int(c.value), int(pp.value)
