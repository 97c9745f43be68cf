// de_math.h — deterministic f32 elementary functions for the gfx950 kernels.
//
// Arithmetic contract: every function is a fixed sequence of correctly rounded binary32 operations
// (+, -, *, /, sqrt, fma) plus floor and bit manipulation, so results do not depend on the device's
// transcendental units (v_exp_f32 / v_log_f32 / v_sin_f32 are ~1 ulp and unspecified) and are reproducible
// bit for bit on any IEEE-754 machine.  tests/test_gpu_parity.py (test_math_bit_exact, test_log_of_a_random_draw) checks
// these against the CPU oracle's independent statement of the same sequences.  Coefficients: tools/fit_f32_polys.py.
//
// The translation unit must be compiled with -ffp-contract=off (the only fused operations are the explicit
// de_fma calls) and without fast-math; hipcc's default correctly rounded f32 divide/sqrt is relied upon.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DE_DEV __device__ __forceinline__

DE_DEV float de_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
DE_DEV uint32_t de_f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
DE_DEV float de_u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
DE_DEV float de_abs(float x) { return __builtin_fabsf(x); }
DE_DEV float de_floor(float x) { return __builtin_floorf(x); }
DE_DEV float de_sqrt(float x) { return __builtin_sqrtf(x); }  // correctly rounded; NaN for x < 0
// comparison-defined min/max: a NaN in the second operand is dropped, a NaN in the first is returned
DE_DEV float de_max(float a, float b) { return (b > a) ? b : a; }
DE_DEV float de_min(float a, float b) { return (b < a) ? b : a; }
DE_DEV float de_clamp(float x, float lo, float hi) { return de_min(de_max(x, lo), hi); }
DE_DEV float de_saturate(float x) { return de_clamp(x, 0.0f, 1.0f); }

#ifdef DE_FAST_MATH
// ---------------------------------------------------------------------------------------------------- DE_FLAG_FAST_MATH (round 5, opt-in)
// The same functions on the hardware's transcendental units — v_exp_f32 / v_log_f32 / v_sin_f32 / v_cos_f32 / v_sqrt_f32 / v_rcp_f32 / v_rsq_f32
// (about 1 ulp each, unspecified roundings) — and shorter polynomials for atan2 / asin: what the arithmetic contract costs, priced
// (profiles/r5_fast_math.md).  NOT bit-reproducible against the CPU oracle: a sample whose `rand < p` test lands on the other side follows a
// different path from there on.  Compiled into a second instantiation of render_kernel_v6 only (de_fast.hip); the default kernels never see this.
DE_DEV float de_div_rc_ng(float x, float c, float rc) { (void)c; return x * rc; }
#define DE_DIVC_NG(x, c) ((x) * (1.0f / (c)))
DE_DEV float de_sqrt_nr(float x) { return __builtin_amdgcn_sqrtf(x); }
DE_DEV float de_rcp_nr(float b) { return __builtin_amdgcn_rcpf(b); }
#define DE_PI 0x1.921fb6p+1f
#define DE_PIO2 0x1.921fb6p+0f
#define DE_LOG2E 0x1.715476p+0f
#define DE_LN2 0x1.62e430p-1f
#define DE_INF __builtin_inff()
#define DE_INV_PI ((float)(1.0 / 3.14159265358979323846))
#define DE_INV_2PI ((float)(0.5 / 3.14159265358979323846))
#define DE_INV_255 ((float)(1.0 / 255.0))
DE_DEV float de_exp(float x) { return __builtin_amdgcn_exp2f(x * DE_LOG2E); }
DE_DEV float de_exp_nonpos(float x) { return __builtin_amdgcn_exp2f(x * DE_LOG2E); }
DE_DEV float de_log(float x) { return __builtin_amdgcn_logf(x) * DE_LN2; }
DE_DEV float de_log_unit(float x) { return __builtin_amdgcn_logf(x) * DE_LN2; }      // log2(0) = -inf: the free flight of a zero draw stays +inf
DE_DEV void de_sincos(float x, float* s_out, float* c_out) {
    const float r = x * DE_INV_2PI;                       // v_sin_f32 / v_cos_f32 take revolutions
    *s_out = __builtin_amdgcn_sinf(r); *c_out = __builtin_amdgcn_cosf(r);
}
DE_DEV float de_sin(float x) { return __builtin_amdgcn_sinf(x * DE_INV_2PI); }
DE_DEV float de_cos(float x) { return __builtin_amdgcn_cosf(x * DE_INV_2PI); }
// atan2: four coefficients (1.7e-5 rad: 0.06 texel of a 21 600-wide map); asin: the contract's four with the hardware sqrt
DE_DEV float de_atan2(float y, float x) {
    float ax = de_abs(x), ay = de_abs(y);
    bool swap = ay > ax;
    float mx = swap ? ay : ax, mn = swap ? ax : ay;
    float a = mn * __builtin_amdgcn_rcpf(mx);
    float s = a * a;
    float p = 0x1.79d11ep-6f;                             // four coefficients fitted on [0, 1] (odd polynomial a + a s p(s)): 1.7e-5 rad
    p = de_fma(p, s, -0x1.727c04p-4f);
    p = de_fma(p, s, 0x1.79d652p-3f);
    p = de_fma(p, s, -0x1.53a54cp-2f);
    float r = de_fma(a * s, p, a);
    if (swap) r = DE_PIO2 - r;
    if (x < 0.0f) r = DE_PI - r;
    if (y < 0.0f) r = -r;
    return (mx == 0.0f) ? 0.0f : r;
}
DE_DEV float de_asin(float x) {
    float ax = de_abs(x);
    if (ax > 1.0f) ax = 1.0f;
    bool big = ax > 0.5f;
    float s = big ? (1.0f - ax) * 0.5f : ax * ax;
    float t = big ? __builtin_amdgcn_sqrtf(s) : ax;
    float p = 0x1.a7813ap-5f;
    p = de_fma(p, s, 0x1.409a72p-5f);
    p = de_fma(p, s, 0x1.35737ap-4f);
    p = de_fma(p, s, 0x1.554bf8p-3f);
    float r = de_fma(t * s, p, t);
    if (big) r = de_fma(-2.0f, r, DE_PIO2);
    return (x < 0.0f) ? -r : r;
}
DE_DEV float de_pow(float a, float b) {
    if (b == 0.0f) return 1.0f;
    if (b == 1.0f) return a;
    return __builtin_amdgcn_exp2f(b * __builtin_amdgcn_logf(a));
}
#else
// x / c in 3 operations instead of the ~12 of an IEEE divide, for a LITERAL divisor c with rc = RN(1/c) (the density
// profiles: 49, 532307548.4168, 1.225, 8136.646).  That q below IS the correctly rounded x / c is a property of the divisor:
// tools/verify_exact_div.c checks it exhaustively (all 2^23 significands) for every literal used, and
// tests/test_gpu_parity.py::test_three_operation_division_by_literals on the device.  The numerator must be 0 or in
// 2^-100 <= |x| <= 2^100 (stated at each use).
DE_DEV float de_div_rc_ng(float x, float c, float rc) {
    float q0 = x * rc;
    float r = de_fma(-q0, c, x);
    return de_fma(r, rc, q0);
}
#define DE_DIVC_NG(x, c) de_div_rc_ng((x), (c), 1.0f / (c))

// sqrt for x = 0 or a normal x in [2^-63, 2^64]: v_rsq_f32 (<= 1 ulp), s = x * y, and one coupled Newton step whose residual
// x - s*s is exact in an fma.  On gfx950 that IS the correctly rounded root for every significand at every exponent of
// that range (tools/exhaustive_math.py sqrt_rsq: 128 exponents x 2^23 significands, 0 mismatches; also
// tests/test_gpu_parity.py::test_fast_sqrt_is_correctly_rounded) — 7 operations instead of the 21 of the compiler's
// IEEE expansion.  NaN and negative arguments give NaN; zero is selected explicitly (0 * rsq(0) would be NaN).
DE_DEV float de_sqrt_nr(float x) {
    float y = __builtin_amdgcn_rsqf(x);
    float s = x * y;
    float h = 0.5f * y;
    float e = de_fma(-s, s, x);
    float r = de_fma(e, h, s);
    return (x == 0.0f) ? 0.0f : r;
}

// 1 / b for a normal b in [2^-60, 2^60]: v_rcp_f32 (<= 1 ulp) and one Newton step with the residual exact in an fma.
// That yields RN(1/b) for every significand on gfx950 — checked exhaustively by tests/test_gpu_parity.py::
// test_fast_reciprocal_is_correctly_rounded — in 3 operations instead of the 10 of the compiler's IEEE divide.
DE_DEV float de_rcp_nr(float b) {
    float y = __builtin_amdgcn_rcpf(b);
    float e = de_fma(-b, y, 1.0f);
    return de_fma(e, y, y);
}

#define DE_PI 0x1.921fb6p+1f
#define DE_PIO2 0x1.921fb6p+0f
#define DE_LN2_HI 0x1.63p-1f
#define DE_LN2_LO -0x1.bd0106p-13f
#define DE_LOG2E 0x1.715476p+0f
#define DE_PIO2_HI 0x1.92p+0f
#define DE_PIO2_MID 0x1.fb4p-12f
#define DE_PIO2_LO 0x1.4442d2p-24f
#define DE_2OPI 0x1.45f306p-1f
#define DE_INF __builtin_inff()

// e^x: x > 88 -> +inf, x < -87 -> 0, NaN -> NaN
DE_DEV float de_exp(float x) {
    float n = de_floor(de_fma(x, DE_LOG2E, 0.5f));
    float r = de_fma(-n, DE_LN2_HI, x);
    r = de_fma(-n, DE_LN2_LO, r);
    float q = 0x1.6b9e2p-10f;
    q = de_fma(q, r, 0x1.1239bep-7f);
    q = de_fma(q, r, 0x1.5556d6p-5f);
    q = de_fma(q, r, 0x1.555492p-3f);
    q = de_fma(q, r, 0x1.fffffep-2f);
    float p = de_fma(r * r, q, r);
    p = p + 1.0f;
    int ni = (int)n;
    float res = p * de_u2f((uint32_t)(ni + 127) << 23);
    res = (x > 88.0f) ? DE_INF : res;
    res = (x < -87.0f) ? 0.0f : res;
    return (x != x) ? x : res;
}

// de_exp for an argument that is <= 0 or NaN (the negated squares of the density profiles): the overflow select of
// de_exp cannot fire and is left out; every other operation is de_exp's.
DE_DEV float de_exp_nonpos(float x) {
    float n = de_floor(de_fma(x, DE_LOG2E, 0.5f));
    float r = de_fma(-n, DE_LN2_HI, x);
    r = de_fma(-n, DE_LN2_LO, r);
    float q = 0x1.6b9e2p-10f;
    q = de_fma(q, r, 0x1.1239bep-7f);
    q = de_fma(q, r, 0x1.5556d6p-5f);
    q = de_fma(q, r, 0x1.555492p-3f);
    q = de_fma(q, r, 0x1.fffffep-2f);
    float p = de_fma(r * r, q, r);
    p = p + 1.0f;
    int ni = (int)n;
    float res = p * de_u2f((uint32_t)(ni + 127) << 23);
    res = (x < -87.0f) ? 0.0f : res;
    return (x != x) ? x : res;
}

// natural log: 0 -> -inf, x < 0 -> NaN, +inf -> +inf; subnormals scaled first
DE_DEV float de_log(float x) {
    float xs = x;
    int e = 0;
    uint32_t ix = de_f2u(xs);
    if (ix < 0x00800000u) { xs = xs * 8388608.0f; ix = de_f2u(xs); e = -23; }
    e += (int)(ix >> 23) - 127;
    float f = de_u2f((ix & 0x007fffffu) | 0x3f800000u);
    if (f > 0x1.6a09e6p+0f) { f = f * 0.5f; e += 1; }
    f = f - 1.0f;
    float p = 0x1.16d79ep-4f;
    p = de_fma(p, f, -0x1.dfb3a4p-4f);
    p = de_fma(p, f, 0x1.e69804p-4f);
    p = de_fma(p, f, -0x1.fbdbap-4f);
    p = de_fma(p, f, 0x1.2338e2p-3f);
    p = de_fma(p, f, -0x1.555e68p-3f);
    p = de_fma(p, f, 0x1.99a4b4p-3f);
    p = de_fma(p, f, -0x1.000002p-2f);
    p = de_fma(p, f, 0x1.555548p-2f);
    float f2 = f * f;
    float fe = (float)e;
    float y = (f * f2) * p;
    y = de_fma(fe, DE_LN2_LO, y);
    y = de_fma(-0.5f, f2, y);
    float z = f + y;
    float res = de_fma(fe, DE_LN2_HI, z);
    res = (x == DE_INF) ? x : res;
    res = (x == 0.0f) ? -DE_INF : res;
    return (x != x || x < 0.0f) ? __builtin_nanf("") : res;
}

DE_DEV void de_sincos(float x, float* s_out, float* c_out) {
    float k = de_floor(de_fma(x, DE_2OPI, 0.5f));
    float r = de_fma(-k, DE_PIO2_HI, x);
    r = de_fma(-k, DE_PIO2_MID, r);
    r = de_fma(-k, DE_PIO2_LO, r);
    int q = ((int)k) & 3;
    float r2 = r * r;
    float ps = 0x1.6cca94p-19f;
    ps = de_fma(ps, r2, -0x1.a00f5ap-13f);
    ps = de_fma(ps, r2, 0x1.111108p-7f);
    ps = de_fma(ps, r2, -0x1.555556p-3f);
    float sn = de_fma(r * r2, ps, r);
    float pc = -0x1.241daap-22f;
    pc = de_fma(pc, r2, 0x1.a010dap-16f);
    pc = de_fma(pc, r2, -0x1.6c16b8p-10f);
    pc = de_fma(pc, r2, 0x1.555556p-5f);
    float cs = de_fma(r2 * r2, pc, de_fma(-0.5f, r2, 1.0f));
    float s = (q & 1) ? cs : sn;
    float c = (q & 1) ? sn : cs;
    if (q & 2) s = -s;
    if ((q + 1) & 2) c = -c;
    *s_out = s;
    *c_out = c;
}
DE_DEV float de_sin(float x) { float s, c; de_sincos(x, &s, &c); return s; }
DE_DEV float de_cos(float x) { float s, c; de_sincos(x, &s, &c); return c; }

// ------------------------------------------------------------------------------------------------ contract 2 (round 2)
// The functions of the hot loops, re-stated for the accuracy the path needs (tools/fit_f32_polys.py fit_c2)
// instead of 2 ulp, and with every quotient by a shared or constant divisor written as a product with its correctly
// rounded reciprocal.  Same rule as before: a fixed sequence of correctly rounded binary32 operations, stated
// independently in oracle/oracle_math.h and compared bit for bit.
#define DE_INV_PI ((float)(1.0 / 3.14159265358979323846))
#define DE_INV_2PI ((float)(0.5 / 3.14159265358979323846))
#define DE_INV_255 ((float)(1.0 / 255.0))
#define DE_LN2 0x1.62e430p-1f

// atan2(y, x) for |x|, |y| <= 2^60: q = min * RN(1 / max), 6-coefficient odd polynomial (3.5e-7 rad).  Both arguments
// below 2^-60 (or zero) give 0.
DE_DEV float de_atan2(float y, float x) {
    float ax = de_abs(x), ay = de_abs(y);
    bool swap = ay > ax;
    float mx = swap ? ay : ax;
    float mn = swap ? ax : ay;
    float a = mn * de_rcp_nr(mx);
    float s = a * a;
    float p = 0x1.e50bcap-8f;
    p = de_fma(p, s, -0x1.23a7dep-5f);
    p = de_fma(p, s, 0x1.50f1cep-4f);
    p = de_fma(p, s, -0x1.12821cp-3f);
    p = de_fma(p, s, 0x1.96cb56p-3f);
    p = de_fma(p, s, -0x1.5540c2p-2f);
    float r = de_fma(a * s, p, a);
    if (swap) r = DE_PIO2 - r;
    if (x < 0.0f) r = DE_PI - r;
    if (y < 0.0f) r = -r;
    return (mx < 0x1p-60f) ? 0.0f : r;
}

// asin, argument clamped to [-1, 1]: 4-coefficient polynomial (6e-8 rad)
DE_DEV float de_asin(float x) {
    float ax = de_abs(x);
    if (ax > 1.0f) ax = 1.0f;
    bool big = ax > 0.5f;
    float s = big ? (1.0f - ax) * 0.5f : ax * ax;
    float t = big ? de_sqrt_nr(s) : ax;           // s is 0 or in [2^-25, 1/4]
    float p = 0x1.a7813ap-5f;
    p = de_fma(p, s, 0x1.409a72p-5f);
    p = de_fma(p, s, 0x1.35737ap-4f);
    p = de_fma(p, s, 0x1.554bf8p-3f);
    float r = de_fma(t * s, p, t);
    if (big) r = de_fma(-2.0f, r, DE_PIO2);
    r = (x < 0.0f) ? -r : r;
    return (x != x) ? x : r;
}

// log of a random draw x = k * 2^-24, k in [0, 2^24): mantissa reduced to [sqrt(1/2), sqrt(2)) by integer arithmetic on
// the bits, 6-coefficient polynomial (2.8e-7 relative), one ln 2 constant.  log(0) = -inf.
DE_DEV float de_log_unit(float x) {
    const uint32_t t = de_f2u(x) + 0x004afb0du;                    // + (bits(1) - bits(sqrt(1/2)))
    const int e = (int)(t >> 23) - 127;
    const float f = de_u2f((t & 0x007fffffu) + 0x3f3504f3u) - 1.0f;
    float p = -0x1.a34a8ap-4f;
    p = de_fma(p, f, 0x1.4875bep-3f);
    p = de_fma(p, f, -0x1.5ed6f8p-3f);
    p = de_fma(p, f, 0x1.980734p-3f);
    p = de_fma(p, f, -0x1.ffa98cp-3f);
    p = de_fma(p, f, 0x1.5557bep-2f);
    const float f2 = f * f;
    float y = (f * f2) * p;
    y = de_fma(-0.5f, f2, y);
    const float z = f + y;
    const float res = de_fma((float)e, DE_LN2, z);
    return (x == 0.0f) ? -DE_INF : res;
}

DE_DEV float de_pow(float a, float b) {
    if (b == 0.0f) return 1.0f;
    if (b == 1.0f) return a;
    return de_exp(b * de_log(a));
}

#endif  // DE_FAST_MATH
