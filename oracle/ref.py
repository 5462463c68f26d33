"""numpy front-end of oracle/_ref/libgs_ref.so: the REFERENCE's own CUDA kernels
(/root/reference/gs/src/include/*.h) compiled for the CPU by oracle/ref_build.py and executed on
the SIMT emulator.  Calling conventions are those of the reference's pybind wrappers
(gs/src/render.cu): caller-allocated, pre-initialised outputs, gradients accumulated.

TEST INFRASTRUCTURE ONLY (the checker the C oracle is pinned against).  The reference uses fp32
atomics; on the emulator threads run in a fixed order, so its results are deterministic here.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "_ref", "libgs_ref.so")
_lib = None


def available():
    return os.path.exists(LIB)


def lib():
    global _lib
    if _lib is None:
        if not available():
            from . import ref_build
            ref_build.build_if_possible()
        _lib = C.CDLL(LIB)
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


u32, f32 = C.c_uint32, C.c_float


def cull_bsphere(mean, qvec, svec, normals, pts, thresh):
    mean, qvec, svec, normals, pts = _f(mean), _f(qvec), _f(svec), _f(normals), _f(pts)
    mask = np.zeros(mean.shape[0], np.bool_)
    lib().ref_culling_gaussian_bsphere(u32(mean.shape[0]), _p(mean), _p(qvec), _p(svec), _p(normals), _p(pts),
                                       _p(mask), f32(thresh))
    return mask


def bin_sort(tl, br, depth, nth, ntw, D):
    tl, br, depth = _i(tl), _i(br), _f(depth).reshape(-1)
    ids = np.zeros(max(D, 1), np.int32)
    start = -np.ones(nth * ntw, np.int32)
    end = -np.ones(nth * ntw, np.int32)
    lib().ref_tile_culling_aabb_start_end(u32(tl.shape[0]), u32(D), u32(nth), u32(ntw), _p(ids), _p(start), _p(end),
                                          _p(tl), _p(br), _p(depth))
    return ids[:D], start, end


def _geo(H, W, ts=16):
    return (H + ts - 1) // ts, (W + ts - 1) // ts


def render_rgb_fwd(mean2d, cov2d, color, alpha, start, end, ids, topleft, psx, psy, H, W, thresh=1e-4):
    nth, ntw = _geo(H, W)
    out = np.zeros((H, W, 3), np.float32)
    T = np.ones((H, W, 1), np.float32)
    a = [_f(mean2d), _f(cov2d), _f(color), _f(alpha), _i(start), _i(end), _i(ids)]
    tl = _f(topleft)
    lib().ref_vol_render_start_end_with_T(u32(a[0].shape[0]), u32(len(a[6])), *[_p(x) for x in a], _p(out), _p(tl),
                                          u32(16), u32(nth), u32(ntw), f32(psx), f32(psy), u32(H), u32(W),
                                          f32(thresh), _p(T))
    return out, T


def render_rgb_bwd(mean2d, cov2d, color, alpha, start, end, ids, final, grad_out, topleft, psx, psy, H, W,
                   thresh=1e-4):
    nth, ntw = _geo(H, W)
    N = np.asarray(mean2d).shape[0]
    gm, gc = np.zeros((N, 2), np.float32), np.zeros((N, 2, 2), np.float32)
    gcol, ga = np.zeros((N, 3), np.float32), np.zeros(N, np.float32)
    a = [_f(mean2d), _f(cov2d), _f(color), _f(alpha), _i(start), _i(end), _i(ids), _f(final)]
    go, tl = _f(grad_out), _f(topleft)
    lib().ref_vol_render_backward_start_end(u32(N), u32(len(a[6])), *[_p(x) for x in a], _p(gm), _p(gc), _p(gcol),
                                            _p(ga), _p(go), _p(tl), u32(16), u32(nth), u32(ntw), f32(psx), f32(psy),
                                            u32(H), u32(W), f32(thresh))
    return gm, gc, gcol, ga


def render_scalar_fwd(mean2d, cov2d, scalar, alpha, start, end, ids, topleft, psx, psy, H, W, thresh=1e-4):
    nth, ntw = _geo(H, W)
    out = np.zeros((H, W), np.float32)
    T = np.ones((H, W, 1), np.float32)
    a = [_f(mean2d), _f(cov2d), _f(scalar), _f(alpha), _i(start), _i(end), _i(ids)]
    tl = _f(topleft)
    lib().ref_vol_render_scalar(u32(a[0].shape[0]), u32(len(a[6])), *[_p(x) for x in a], _p(out), _p(tl), u32(16),
                                u32(nth), u32(ntw), f32(psx), f32(psy), u32(H), u32(W), f32(thresh), _p(T))
    return out, T


def render_scalar_bwd(mean2d, cov2d, scalar, alpha, start, end, ids, final, grad_out, topleft, psx, psy, H, W,
                      thresh=1e-4):
    nth, ntw = _geo(H, W)
    N = np.asarray(mean2d).shape[0]
    gm, gc = np.zeros((N, 2), np.float32), np.zeros((N, 2, 2), np.float32)
    gs, ga = np.zeros(N, np.float32), np.zeros(N, np.float32)
    a = [_f(mean2d), _f(cov2d), _f(scalar), _f(alpha), _i(start), _i(end), _i(ids), _f(final)]
    go, tl = _f(grad_out), _f(topleft)
    lib().ref_vol_render_scalar_backward(u32(N), u32(len(a[6])), *[_p(x) for x in a], _p(gm), _p(gc), _p(gs), _p(ga),
                                         _p(go), _p(tl), u32(16), u32(nth), u32(ntw), f32(psx), f32(psy), u32(H),
                                         u32(W), f32(thresh))
    return gm, gc, gs, ga


def render_sh_fwd(mean2d, cov2d, sh, alpha, start, end, ids, topleft, rot9, Cb, psx, psy, H, W, thresh=1e-4, bg=None):
    nth, ntw = _geo(H, W)
    out = np.zeros((H, W, 3), np.float32)
    a = [_f(mean2d), _f(cov2d), _f(sh), _f(alpha), _i(start), _i(end), _i(ids)]
    tl, rot = _f(topleft), _f(rot9).reshape(-1)[:9].copy()
    bg_ = _f(bg) if bg is not None else None
    lib().ref_vol_render_sh(u32(a[0].shape[0]), u32(len(a[6])), *[_p(x) for x in a], _p(out), _p(tl), _p(rot), u32(16),
                            u32(nth), u32(ntw), f32(psx), f32(psy), u32(H), u32(W), u32(Cb), f32(thresh), _p(bg_))
    return out


def render_sh_bwd(mean2d, cov2d, sh, alpha, start, end, ids, final, grad_out, topleft, rot9, Cb, psx, psy, H, W,
                  thresh=1e-4, bg=None):
    nth, ntw = _geo(H, W)
    N = np.asarray(mean2d).shape[0]
    gm, gc = np.zeros((N, 2), np.float32), np.zeros((N, 2, 2), np.float32)
    gsh, ga = np.zeros((N, 3, Cb * Cb), np.float32), np.zeros(N, np.float32)
    a = [_f(mean2d), _f(cov2d), _f(sh), _f(alpha), _i(start), _i(end), _i(ids), _f(final)]
    go, tl, rot = _f(grad_out), _f(topleft), _f(rot9).reshape(-1)[:9].copy()
    bg_ = _f(bg) if bg is not None else None
    lib().ref_vol_render_backward_sh(u32(N), u32(len(a[6])), *[_p(x) for x in a], _p(gm), _p(gc), _p(gsh), _p(ga),
                                     _p(go), _p(tl), _p(rot), u32(16), u32(nth), u32(ntw), f32(psx), f32(psy), u32(H),
                                     u32(W), u32(Cb), f32(thresh), _p(bg_))
    return gm, gc, gsh, ga


# ---- legacy binning (render.cu:46-176) ---------------------------------------------------------
def legacy_count(mode, mean2d, shape, topleft, tile_size, nth, ntw, psx, psy, thresh=0.0, num=None):
    """mode 0: count_num_gaussians_each_tile (shape = cov2d, thresh); 1: ..._bcircle (shape = radius)"""
    mean2d, shape, topleft = _f(mean2d), _f(shape), _f(topleft)
    num = np.zeros(nth * ntw, np.int32) if num is None else num
    N = mean2d.shape[0]
    if mode == 0:
        lib().ref_count_num_gaussians_each_tile(u32(N), _p(mean2d), _p(shape), _p(topleft), u32(tile_size), u32(nth),
                                                u32(ntw), f32(psx), f32(psy), _p(num), f32(thresh))
    else:
        lib().ref_count_num_gaussians_each_tile_bcircle(u32(N), _p(mean2d), _p(shape), _p(topleft), u32(tile_size),
                                                        u32(nth), u32(ntw), f32(psx), f32(psy), _p(num))
    return num


def legacy_image_sort(mode, depth, tile_n, mean2d, shape, topleft, tile_size, nth, ntw, psx, psy, thresh=0.0):
    """-> gaussian_ids, tiledepth (u64 keys, unsorted), tile_n_gaussians, offset  (image_sort /
    prepare_image_sort; the caller's zero-initialised outputs are allocated here)"""
    mean2d, shape, topleft, depth = _f(mean2d), _f(shape), _f(topleft), _f(depth).reshape(-1)
    tile_n = _i(tile_n).copy()
    D = int(tile_n.sum())
    ids = np.zeros(max(D, 1), np.int32)
    td = np.zeros(max(D, 1), np.float64)
    offset = np.zeros(nth * ntw, np.int32)
    N = mean2d.shape[0]
    if mode == 0:
        lib().ref_image_sort(u32(N), u32(D), _p(ids), _p(td), _p(depth), _p(tile_n), _p(offset), _p(mean2d), _p(shape),
                             _p(topleft), u32(tile_size), u32(nth), u32(ntw), f32(psx), f32(psy), f32(thresh))
    else:
        lib().ref_prepare_image_sort(u32(N), u32(D), _p(ids), _p(td), _p(depth), _p(tile_n), _p(offset), _p(mean2d),
                                     _p(shape), _p(topleft), u32(tile_size), u32(nth), u32(ntw), f32(psx), f32(psy))
    return ids[:D], td[:D].view(np.uint64), tile_n, offset
