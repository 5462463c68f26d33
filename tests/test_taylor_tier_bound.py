"""The Taylor tier's remainder bound against the arithmetic that ships (composite_common.hpp: taylor_row_ok, taylor_convert).

A staged row (w0 .. w5) of a (splat, channel) is the tile-local polynomial of its logit,
    s(u, v) = w0 + w1 v + w2 u + w3 v^2 + w4 u v + w5 u^2,   |u|, |v| <= 1,
and the colour is f(s) = 1 / (1 + 2^s).  Where  0.00694 (l + q)^3 + 0.02312 q (2 l + q) <= kTaylorErr  (l = |w1| + |w2|,
q = |w3| + |w4| + |w5|) the kernels replace the six coefficients by those of the colour's own quadratic
    c(u, v) = f0 + f1 (L + Q) + f2 L^2
and evaluate that instead of the sigmoid.  This test parses the constants from the header, draws rows ON the boundary of the
rule (the worst the rule admits), forms c exactly as taylor_convert does, and compares it with the sigmoid in fp64 over the
tile: the documented 8.7e-7 -- the Taylor tier's share of the 1e-5 colour promise -- must hold."""
import os
import re

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
SRC = open(os.path.join(ROOT, "gsgen_amd", "csrc", "composite_common.hpp")).read()


def constants():
    m = re.search(r"return ([0-9.e-]+)f \* d \* d \* d \+ ([0-9.e-]+)f \* q \* \(2\.0f \* l \+ q\) <= kTaylorErr;", SRC)
    err = re.search(r"constexpr float kTaylorErr = ([0-9.e-]+)f;", SRC)
    assert m and err
    return float(m.group(1)), float(m.group(2)), float(err.group(1))


def colour_rows(w):
    """taylor_convert, in fp64: logit coefficients (w0, w1 v, w2 u, w3 v^2, w4 uv, w5 u^2) -> colour coefficients"""
    w0, w1, w2, w3, w4, w5 = w.T
    f0 = 1.0 / (1.0 + np.exp2(w0))
    ff = f0 * (1.0 - f0)
    f1 = -np.log(2.0) * ff
    f2 = 0.5 * np.log(2.0) ** 2 * ff * (1.0 - 2.0 * f0)
    return np.stack([f0, f1 * w1, f1 * w2, f1 * w3 + f2 * w1 * w1, f1 * w4 + 2.0 * f2 * w1 * w2, f1 * w5 + f2 * w2 * w2], 1)


def poly(c, u, v):
    return (c[:, 0, None] + c[:, 1, None] * v + c[:, 2, None] * u + c[:, 3, None] * v * v + c[:, 4, None] * u * v
            + c[:, 5, None] * u * u)


def test_header_constants_are_the_derived_ones():
    a3, a2, err = constants()
    ln2 = np.log(2.0)
    f = np.linspace(0.0, 1.0, 200001)
    g = f * (1 - f)
    # |f'''| / 6 and |f''| / 2 of f(z) = 1 / (1 + 2^z), maximised over z (as functions of f)
    third = ln2 ** 3 / 6.0 * np.abs(g * (1 - 6 * g)).max()
    second = ln2 ** 2 / 2.0 * np.abs(g * (1 - 2 * f)).max()
    assert third <= a3 <= 1.01 * third, (third, a3)
    assert second <= a2 <= 1.01 * second, (second, a2)
    assert err == 8.7e-7 and "constexpr float kPolyFitTol = 1e-5f - kTaylorErr;" in SRC
    # the rows the kernels convert and the rows this test converts are the same arithmetic
    for line in ("row[1] = f1 * w1;", "row[2] = f1 * w2;", "row[3] = fmaf(f1, w4, 2.0f * f2 * w1 * w2);",
                 "row[4] = fmaf(f1, w5, f2 * w2 * w2);", "row[6] = fmaf(f1, w3, f2 * w1 * w1);"):
        assert line in SRC, line


def test_colour_quadratic_stays_within_the_documented_remainder_on_the_rule_s_boundary():
    a3, a2, err = constants()
    rng = np.random.default_rng(5)
    n = 4000
    w = rng.normal(size=(n, 6))
    w[:, 0] = rng.uniform(-6.0, 6.0, n)                      # the tile-centre logit: anywhere
    w[:, 3:] *= 10.0 ** rng.uniform(-3.0, 0.0, (n, 1))      # quadratic part: from negligible to as large as the linear one
    w[: n // 8, 3:] = 0.0
    # signs that make the bound tight: everything pulling the same way at a tile corner
    w[n // 2:, 1:] = np.abs(w[n // 2:, 1:]) * rng.choice([-1.0, 1.0], (n - n // 2, 1))

    def bound(scale):
        l_ = scale * (np.abs(w[:, 1]) + np.abs(w[:, 2]))
        q_ = scale * np.abs(w[:, 3:]).sum(1)
        return a3 * (l_ + q_) ** 3 + a2 * q_ * (2 * l_ + q_)
    lo, hi = np.zeros(n), np.ones(n)
    for _ in range(60):                                       # scale each row onto bound == kTaylorErr
        mid = 0.5 * (lo + hi)
        ok = bound(mid) <= err
        lo, hi = np.where(ok, mid, lo), np.where(ok, hi, mid)
    w[:, 1:] *= lo[:, None]
    assert np.all(bound(np.ones(n)) <= err * (1 + 1e-9)) and np.all(bound(np.ones(n)) >= 0.999 * err)
    dmax = np.abs(w[:, 1:]).sum(1)
    assert 0.0499 < dmax.max() <= 0.0501                     # (the cubic term alone reaches the bound at l + q = 0.05)
    g = np.linspace(-1.0, 1.0, 33)
    u, v = [a.ravel()[None, :] for a in np.meshgrid(g, g)]
    exact = 1.0 / (1.0 + np.exp2(poly(w, u, v)))
    approx = poly(colour_rows(w), u, v)
    worst = np.abs(exact - approx).max(1)
    assert worst.max() <= err, worst.max()
    # ... and the bound is not vacuous: the worst rows use a good part of it
    assert worst.max() >= 0.3 * err, worst.max()
