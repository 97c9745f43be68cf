#!/usr/bin/env python3
"""Derive the polynomial coefficients of the deterministic f32 elementary functions.

The arithmetic contract of this repo ("Arithmetic contract") is that the
oracle (oracle/oracle_math.h) and the HIP kernels (digital_earth_amd/csrc/de_math.h)
evaluate exp/log/sin/cos/atan2/asin with the SAME sequence of IEEE-754 binary32
operations (+, -, *, fma, sqrt, /), so that CPU and GPU agree bit for bit.  The
coefficients are not copied from any library: this script fits them (Lawson-weighted
least squares in f64 on Chebyshev nodes -> near-minimax), rounds them to binary32 and
prints them as C hex-float literals together with the measured max relative error.

Run:  python tools/fit_f32_polys.py        (prints a C snippet; numbers are pasted into
the two math headers by hand and pinned by tests/test_oracle_math.py)
"""
import numpy as np


def cheb_nodes(a, b, n):
    k = np.arange(n)
    x = np.cos(np.pi * (k + 0.5) / n)
    return 0.5 * (a + b) + 0.5 * (b - a) * x


def lawson_fit(basis, target, weight, iters=60):
    """minimise max |weight * (basis @ c - target)| approximately."""
    w = np.ones_like(target)
    c = None
    for _ in range(iters):
        A = basis * (w * weight)[:, None]
        y = target * (w * weight)
        c, *_ = np.linalg.lstsq(A, y, rcond=None)
        err = np.abs(weight * (basis @ c - target))
        w = w * (err / err.max() + 1e-3)
        w /= w.max()
    return c


def f32(x):
    return np.asarray(x, dtype=np.float32)


def hexf(v):
    return float(np.float32(v)).hex()


def report(name, coefs, relerr):
    print(f"/* {name}: max rel err of the f64 polynomial with f32-rounded coefs = {relerr:.3e} */")
    for i, c in enumerate(coefs):
        print(f"#define {name}_C{i} {hexf(c)}f /* {float(np.float32(c)):.9e} */")
    print()


def fit_exp():
    # e^r = 1 + r + r^2 * q(r),  r in [-ln2/2, ln2/2]
    h = np.log(2.0) / 2 * 1.0001
    x = cheb_nodes(-h, h, 4001)
    tgt = (np.exp(x) - 1 - x) / (x * x)
    deg = 4
    B = np.vander(x, deg + 1, increasing=True)
    # relative error of the final e^r: weight = r^2 / e^r
    c = lawson_fit(B, tgt, x * x / np.exp(x))
    c32 = f32(c).astype(np.float64)
    approx = 1 + x + x * x * (B @ c32)
    rel = np.max(np.abs(approx / np.exp(x) - 1))
    report("DE_EXP", c32, rel)


def fit_log():
    # log(1+f) = f - f^2/2 + f^3 * P(f),  f in [sqrt(.5)-1, sqrt(2)-1]
    a, b = np.sqrt(0.5) - 1, np.sqrt(2.0) - 1
    x = cheb_nodes(a * 1.0001, b * 1.0001, 6001)
    x = x[np.abs(x) > 1e-6]
    tgt = (np.log1p(x) - x + 0.5 * x * x) / x ** 3
    deg = 8
    B = np.vander(x, deg + 1, increasing=True)
    c = lawson_fit(B, tgt, np.abs(x ** 3 / np.log1p(x)))
    c32 = f32(c).astype(np.float64)
    approx = x - 0.5 * x * x + x ** 3 * (B @ c32)
    rel = np.max(np.abs(approx / np.log1p(x) - 1))
    report("DE_LOG", c32, rel)


def fit_sin():
    # sin(r) = r + r^3 * S(r^2),  |r| <= pi/4
    h = np.pi / 4 * 1.0001
    x = cheb_nodes(1e-4, h, 4001)
    s = x * x
    tgt = (np.sin(x) - x) / x ** 3
    deg = 3
    B = np.vander(s, deg + 1, increasing=True)
    c = lawson_fit(B, tgt, x ** 3 / np.sin(x))
    c32 = f32(c).astype(np.float64)
    approx = x + x ** 3 * (B @ c32)
    rel = np.max(np.abs(approx / np.sin(x) - 1))
    report("DE_SIN", c32, rel)


def fit_cos():
    # cos(r) = 1 - r^2/2 + r^4 * C(r^2),  |r| <= pi/4
    h = np.pi / 4 * 1.0001
    x = cheb_nodes(1e-3, h, 4001)
    s = x * x
    tgt = (np.cos(x) - 1 + 0.5 * s) / (s * s)
    deg = 3
    B = np.vander(s, deg + 1, increasing=True)
    c = lawson_fit(B, tgt, s * s / np.cos(x))
    c32 = f32(c).astype(np.float64)
    approx = 1 - 0.5 * s + s * s * (B @ c32)
    rel = np.max(np.abs(approx / np.cos(x) - 1))
    report("DE_COS", c32, rel)


def fit_atan():
    # atan(a) = a + a^3 * A(a^2),  a in [0, 1]
    x = cheb_nodes(1e-4, 1.0, 8001)
    s = x * x
    tgt = (np.arctan(x) - x) / x ** 3
    deg = 8
    B = np.vander(s, deg + 1, increasing=True)
    c = lawson_fit(B, tgt, x ** 3 / np.arctan(x), iters=120)
    c32 = f32(c).astype(np.float64)
    approx = x + x ** 3 * (B @ c32)
    rel = np.max(np.abs(approx / np.arctan(x) - 1))
    report("DE_ATAN", c32, rel)


def fit_asin():
    # asin(x) = x + x^3 * R(x^2),  |x| <= 0.5
    x = cheb_nodes(1e-4, 0.5, 4001)
    s = x * x
    tgt = (np.arcsin(x) - x) / x ** 3
    deg = 5
    B = np.vander(s, deg + 1, increasing=True)
    c = lawson_fit(B, tgt, x ** 3 / np.arcsin(x))
    c32 = f32(c).astype(np.float64)
    approx = x + x ** 3 * (B @ c32)
    rel = np.max(np.abs(approx / np.arcsin(x) - 1))
    report("DE_ASIN", c32, rel)


def constants():
    ln2 = np.log(2.0)
    hi = np.float32(0.693359375)  # 12 significant bits: n*hi exact for |n| < 2^11
    lo = np.float32(ln2 - float(hi))
    print(f"#define DE_LN2_HI {hexf(hi)}f /* {float(hi):.9e} */")
    print(f"#define DE_LN2_LO {hexf(lo)}f /* {float(lo):.9e} */")
    print(f"#define DE_LOG2E {hexf(1 / ln2)}f")
    p = np.pi / 2
    p_hi = np.float32(1.5703125)  # 8 significant bits -> k*hi exact for |k| < 2^16
    p_mid = np.float32(p - float(p_hi))
    # make mid short too
    p_mid = np.float32(float.fromhex(float(p_mid).hex()))
    mid_bits = np.frombuffer(np.float32(p_mid).tobytes(), dtype=np.uint32)[0] & np.uint32(0xFFFFF000)
    p_mid = np.frombuffer(np.uint32(mid_bits).tobytes(), dtype=np.float32)[0]
    p_lo = np.float32(p - float(p_hi) - float(p_mid))
    print(f"#define DE_PIO2_HI {hexf(p_hi)}f /* {float(p_hi):.9e} */")
    print(f"#define DE_PIO2_MID {hexf(p_mid)}f /* {float(p_mid):.9e} */")
    print(f"#define DE_PIO2_LO {hexf(p_lo)}f /* {float(p_lo):.9e} */")
    print(f"#define DE_2OPI {hexf(2 / np.pi)}f")
    print(f"#define DE_PI {hexf(np.pi)}f")
    print(f"#define DE_PIO2 {hexf(np.pi / 2)}f")
    print()


if __name__ == "__main__":
    np.set_printoptions(precision=17)
    constants()
    fit_exp()
    fit_log()
    fit_sin()
    fit_cos()
    fit_atan()
    fit_asin()


# ---------------------------------------------------------------------------------------------------------------
# Contract 2 (round 2): the hot-loop functions, fitted to the accuracy the path needs instead of to 2 ulp.
#   atan2 / asin feed the sphere UV map: the map coordinate is an f32 in [0, 1] (ulp 6e-8 = 1.3e-3 texel at 21600), so an
#   angle error of a few 1e-7 rad is at the level of the coordinate's own rounding;
#   log feeds free-flight distances -log(xi)/majorant: 1e-6 relative is far below the Monte-Carlo noise of any frame.
def fit_c2(deg_atan=5, deg_asin=4, deg_log=5):
    x = cheb_nodes(1e-4, 1.0, 8001)
    s = x * x
    tgt = (np.arctan(x) - x) / x ** 3
    B = np.vander(s, deg_atan + 1, increasing=True)
    c = lawson_fit(B, tgt, x ** 3, iters=200)          # absolute error of the angle
    c32 = f32(c).astype(np.float64)
    approx = x + x ** 3 * (B @ c32)
    print(f"/* C2 atan, {deg_atan + 1} coefficients: max abs err {np.max(np.abs(approx - np.arctan(x))):.3e} rad */")
    for i, v in enumerate(c32): print(f"#define DE_ATAN2_K{i} {hexf(v)}f")
    x = cheb_nodes(1e-4, 0.5, 4001)
    s = x * x
    tgt = (np.arcsin(x) - x) / x ** 3
    B = np.vander(s, deg_asin + 1, increasing=True)
    c = lawson_fit(B, tgt, x ** 3, iters=200)
    c32 = f32(c).astype(np.float64)
    approx = x + x ** 3 * (B @ c32)
    print(f"/* C2 asin, {deg_asin + 1} coefficients: max abs err {np.max(np.abs(approx - np.arcsin(x))):.3e} rad (x2 on the |x| > 0.5 branch) */")
    for i, v in enumerate(c32): print(f"#define DE_ASIN2_K{i} {hexf(v)}f")
    a, b = np.sqrt(0.5) - 1, np.sqrt(2.0) - 1
    x = cheb_nodes(a * 1.0001, b * 1.0001, 6001)
    x = x[np.abs(x) > 1e-6]
    tgt = (np.log1p(x) - x + 0.5 * x * x) / x ** 3
    B = np.vander(x, deg_log + 1, increasing=True)
    c = lawson_fit(B, tgt, np.abs(x ** 3 / np.log1p(x)), iters=200)
    c32 = f32(c).astype(np.float64)
    approx = x - 0.5 * x * x + x ** 3 * (B @ c32)
    print(f"/* C2 log(1+f), {deg_log + 1} coefficients: max rel err {np.max(np.abs(approx / np.log1p(x) - 1)):.3e} */")
    for i, v in enumerate(c32): print(f"#define DE_LOG2_K{i} {hexf(v)}f")
