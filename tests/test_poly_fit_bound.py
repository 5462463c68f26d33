"""The routing constant of the tile-local polynomial SH basis against the fit that actually ships.

composite_common.hpp routes a splat to the polynomial form where  0.25 * S * kPolyFitErr * delta^3 <= 1e-5 - 8.7e-7  (poly_ok /
poly_row_ok; the 8.7e-7 are the Taylor tier's share of the budget, kTaylorErr): 0.25 = the sigmoid's largest slope, S = the largest sum_{k>=1} |sh| of one (splat, channel) row, delta = a tile's
half diagonal in camera space, and "kPolyFitErr delta^3" stands for the largest error of one basis function under the degree-2
fit.  Rounds 2-4 shipped 0.7 there, calibrated with a different interpolation and a single rotation (tools/tile_basis_error.py),
which made the real guarantee 1.4e-5; since round 5 the constant is the measured one, 1.0.  This test evaluates the SHIPPED fit
-- the 6 x 9 least-squares matrix kPolyFit, parsed from the header, through the exact basis at the kernels' nine nodes -- in
fp64 over random rotations, tile positions and focal lengths, and pins what the library may promise:

    max_k |Y_k - Y^_k|  <=  1.0 * delta^3      (measured: ~0.94 delta^3, worst at the image centre)

so a routed splat's colours are within  1e-5  of the exact kernels' -- the figure DESIGN.md and include/gsgen_hip.h state -- a
tenth of the 1e-4 image contract, which the full-size GPU tests hold on every pixel."""
import os
import re

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def shipped_fit():
    src = open(os.path.join(ROOT, "gsgen_amd", "csrc", "composite_common.hpp")).read()
    body = src[src.index("constexpr float kPolyFit[kPolyNB][kPolyNodes] = {"):]
    body = body[body.index("{", body.index("=")) + 1:body.index("};")]
    rows = re.findall(r"\{([^{}]*)\}", body)
    M = np.array([[float(x.strip().rstrip("f")) for x in r.split(",")] for r in rows])
    assert M.shape == (6, 9)
    return M


def sh_basis_deg3(d):
    """shencoder.h:13-62 (bands 1..4), fp64; d: [..., 3] unit vectors -> [..., 16]"""
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
    Y = [0.28209479177387814 + 0 * x, -0.48860251190291987 * y, 0.48860251190291987 * z, -0.48860251190291987 * x,
         1.0925484305920792 * xy, -1.0925484305920792 * yz, 0.94617469575755997 * z2 - 0.31539156525251999,
         -1.0925484305920792 * xz, 0.54627421529603959 * x2 - 0.54627421529603959 * y2,
         0.59004358992664352 * y * (-3.0 * x2 + y2), 2.8906114426405538 * xy * z, 0.45704579946446572 * y * (1.0 - 5.0 * z2),
         0.3731763325901154 * z * (5.0 * z2 - 3.0), 0.45704579946446572 * x * (1.0 - 5.0 * z2), 1.4453057213202769 * z * (x2 - y2),
         0.59004358992664352 * x * (-x2 + 3.0 * y2)]
    return np.stack(Y, -1)


def basis_of(R, qx, qy):
    d = np.stack([R[0] * qx + R[1] * qy + R[2], R[3] * qx + R[4] * qy + R[5], R[6] * qx + R[7] * qy + R[8]], -1)
    return sh_basis_deg3(d / np.linalg.norm(d, axis=-1, keepdims=True))


def worst_constant(n_rot, rng, focal_over_size):
    """max over rotations, tiles of an 800-pixel-wide image at the given focal / size, pixels and basis functions of
    |Y - Y^| / delta^3"""
    M = shipped_fit()
    size, ps = 800.0, 1.0 / (focal_over_size * 800.0)
    delta = 7.5 * np.sqrt(2.0) * ps
    l = (np.arange(16) - 7.5) / 7.5                      # poly_offset
    u, v = np.meshgrid(l, l)                             # u: column offset, v: row offset
    U = np.stack([np.ones_like(u), v, u, v * v, u * v, u * u], -1)  # monomials (1, v, u, v^2, uv, u^2): composite_common.hpp
    nodes = [(float(t % 3 - 1), float(t // 3 - 1)) for t in range(9)]  # poly_tile_setup: u = t % 3 - 1, v = t / 3 - 1
    worst = 0.0
    for _ in range(n_rot):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        w, x, y, z = q
        R = np.array([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z),
                      2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)])
        # tile centres: the image centre, an edge, a corner and random tiles (the error is largest where |q| is smallest)
        centres = [(0.0, 0.0), (0.0, (size / 2 - 8) * ps), ((size / 2 - 8) * ps, (size / 2 - 8) * ps)]
        centres += [tuple(rng.uniform(-size / 2 + 8, size / 2 - 8, 2) * ps) for _ in range(5)]
        for cx, cy in centres:
            Yn = np.stack([basis_of(R, cx + 7.5 * nu * ps, cy + 7.5 * nv * ps) for nu, nv in nodes])  # [9, 16]
            V = M @ Yn                                                                                # [6, 16]
            exact = basis_of(R, cx + 7.5 * u * ps, cy + 7.5 * v * ps)                                 # [16, 16, 16]
            worst = max(worst, float(np.abs(exact - U @ V)[..., 1:].max()) / delta ** 3)
    return worst


def test_shipped_fit_error_constant():
    rng = np.random.default_rng(0)
    c_wide = worst_constant(400, rng, 0.7)    # the widest camera of BASELINE configs[3]
    c_norm = worst_constant(400, rng, 1.0)    # f = image size (configs[1], [2])
    print(f"largest basis error of the shipped fit: {c_wide:.3f} delta^3 at f = 0.7 x size, {c_norm:.3f} delta^3 at f = size")
    assert max(c_wide, c_norm) <= 1.0, (c_wide, c_norm)   # => kPolyFitErr = 1.0 bounds it: routed colours within 1e-5 of the exact kernels'
    assert max(c_wide, c_norm) >= 0.5                      # (the sweep does reach the regime the constant describes)


def test_documented_guarantee_matches_the_routing_rule():
    """poly_ok's constants as shipped, and the guarantee the documents state for them"""
    src = open(os.path.join(ROOT, "gsgen_amd", "csrc", "composite_common.hpp")).read()
    assert src.count("return 0.25f * S * kPolyFitErr * delta * delta * delta <= kPolyFitTol;") == 2 and "constexpr float kPolyFitErr = 1.0f;" in src
    # the fit's share and the Taylor tier's share add up to the promise
    assert "constexpr float kPolyFitTol = 1e-5f - kTaylorErr;" in src and "constexpr float kTaylorErr = 8.7e-7f;" in src
    assert "0.00694f * d * d * d + 0.02312f * q * (2.0f * l + q) <= kTaylorErr" in src
    for doc in ("DESIGN.md", os.path.join("include", "gsgen_hip.h")):
        text = open(os.path.join(ROOT, doc)).read()
        assert "within 1e-5 of the exact" in text, doc
