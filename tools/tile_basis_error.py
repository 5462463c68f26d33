"""How well does a tile-local polynomial basis reproduce the per-pixel SH basis?  (A round-3 candidate: the view direction
dir(pixel) = normalize(R (qx, qy, 1)) varies by ~1e-2 rad across a 16x16 tile, so Y[pixel, 0..15] is, within a tile, very
nearly a low-degree polynomial in the tile-local pixel offsets (u, v).  With Y ~ U(u, v) V the two SH contractions of the
compositing kernels shrink from 16 to 6 (degree 2) or 10 (degree 3) terms per pixel and channel, and the per-pixel basis
leaves the registers.)  This script measures the interpolation error in fp64 for the bench cameras: exact real SH basis
(degree 3, the torch-ngp table the reference uses) against the degree-d bivariate polynomial through fixed nodes of the
tile.  CPU only, numpy only."""
import sys
import numpy as np


def sh16(d):
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
    return np.stack([
        0.28209479177387814 + 0 * x, -0.48860251190291987 * y, 0.48860251190291987 * z, -0.48860251190291987 * x,
        1.0925484305920792 * xy, -1.0925484305920792 * yz, 0.94617469575755997 * z2 - 0.31539156525251999,
        -1.0925484305920792 * xz, 0.54627421529603959 * x2 - 0.54627421529603959 * y2,
        0.59004358992664352 * y * (-3.0 * x2 + y2), 2.8906114426405538 * xy * z, 0.45704579946446572 * y * (1.0 - 5.0 * z2),
        0.3731763325901154 * z * (5.0 * z2 - 3.0), 0.45704579946446572 * x * (1.0 - 5.0 * z2),
        1.4453057213202769 * z * (x2 - y2), 0.59004358992664352 * x * (-x2 + 3.0 * y2)], -1)


def monomials(u, v, deg):
    return np.stack([u ** i * v ** (j - i) for j in range(deg + 1) for i in range(j + 1)], -1)


def worst(W, H, f, deg, rot=np.eye(3)):
    worst_err = 0.0
    t = (np.arange(16) - 7.5) / 7.5
    uu, vv = np.meshgrid(t, t, indexing="xy")
    U = monomials(uu.ravel(), vv.ravel(), deg)                       # [256, n_terms]
    n = U.shape[1]
    # fixed interpolation nodes: a triangular Chebyshev-like subset of the 16x16 grid
    idx = np.linspace(0, 15, deg + 1).round().astype(int)
    nodes = [(idx[i], idx[j]) for j in range(deg + 1) for i in range(deg + 1 - j)]
    sel = [b * 16 + a for a, b in nodes]
    Minv = np.linalg.inv(U[sel])                                     # a compile-time constant of the kernel
    for ty in range(0, (H + 15) // 16, max(1, H // 16 // 12)):
        for tx in range(0, (W + 15) // 16, max(1, W // 16 // 12)):
            gx, gy = tx * 16 + np.arange(16), ty * 16 + np.arange(16)
            qx, qy = np.meshgrid((gx - W / 2) / f, (gy - H / 2) / f, indexing="xy")
            d = np.stack([qx, qy, np.ones_like(qx)], -1).reshape(-1, 3) @ rot.T
            d /= np.linalg.norm(d, axis=-1, keepdims=True)
            Y = sh16(d)                                              # [256, 16] exact
            V = Minv @ Y[sel]                                        # [n, 16]
            worst_err = max(worst_err, float(np.abs(U @ V - Y).max()))
    return worst_err, n


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    for name, W, H, f in (("cfg2 800x800 f=800", 800, 800, 800.0), ("cfg3 1024^2 f=1024", 1024, 1024, 1024.0),
                          ("cfg4 512^2 f=0.7x512", 512, 512, 0.7 * 512), ("cfg4 512^2 f=1.35x512", 512, 512, 1.35 * 512),
                          ("wide 256^2 f=128", 256, 256, 128.0)):
        print(name, " | ".join(f"degree {dg}: {worst(W, H, f, dg, q)[1]} terms, max |Y - UV| = {worst(W, H, f, dg, q)[0]:.2e}" for dg in (2, 3, 4)))
