// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path:
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
//
// oracle_math.h — the arithmetic contract of the CPU restatement.
//
// The reference (Taichi on Vulkan) gets exp/log/sin/cos/atan2/asin/pow from the GPU driver;
// their rounding is not specified anywhere in /root/reference ("parity
// unpinned").  The oracle therefore DEFINES them: every function below is a fixed sequence
// of IEEE-754 binary32 operations (+, -, *, /, sqrt, fma — all correctly rounded — plus
// floor and bit manipulation), so any conforming implementation reproduces it bit for bit.
// The HIP kernels carry their own implementation of the same sequences
// (digital_earth_amd/csrc/de_math.h); tests/test_gpu_parity.py::test_math_bit_exact compares the two bit-exactly.
//
// Polynomial coefficients were derived by tools/fit_f32_polys.py (not taken from any libm).
// Must be compiled with -ffp-contract=off -fno-fast-math -mfma (see oracle/Makefile).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

namespace deo {

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// comparison-defined min/max (NaN in the second operand is dropped, like maxnum for (x, NaN))
static inline float max_(float a, float b) { return (b > a) ? b : a; }
static inline float min_(float a, float b) { return (b < a) ? b : a; }
static inline float clamp_(float x, float lo, float hi) { return min_(max_(x, lo), hi); }
static inline float saturate_(float x) { return clamp_(x, 0.0f, 1.0f); }
static inline float sqrt_(float x) { return __builtin_sqrtf(x); }  // IEEE: NaN for x < 0

#define OM_PI 0x1.921fb6p+1f
#define OM_PIO2 0x1.921fb6p+0f
#define OM_LN2_HI 0x1.63p-1f
#define OM_LN2_LO -0x1.bd0106p-13f
#define OM_LOG2E 0x1.715476p+0f
#define OM_PIO2_HI 0x1.92p+0f
#define OM_PIO2_MID 0x1.fb4p-12f
#define OM_PIO2_LO 0x1.4442d2p-24f
#define OM_2OPI 0x1.45f306p-1f

// e^x.  x > 88 -> +inf, x < -87 -> 0 (no subnormal tail), NaN -> NaN.
static inline float exp_(float x) {
    if (x != x) return x;
    if (x > 88.0f) return __builtin_inff();
    if (x < -87.0f) return 0.0f;
    float n = floorf(fma_(x, OM_LOG2E, 0.5f));
    float r = fma_(-n, OM_LN2_HI, x);
    r = fma_(-n, OM_LN2_LO, r);
    float q = 0x1.6b9e2p-10f;
    q = fma_(q, r, 0x1.1239bep-7f);
    q = fma_(q, r, 0x1.5556d6p-5f);
    q = fma_(q, r, 0x1.555492p-3f);
    q = fma_(q, r, 0x1.fffffep-2f);
    float p = fma_(r * r, q, r);
    p = p + 1.0f;
    int ni = (int)n;
    return p * u2f((uint32_t)(ni + 127) << 23);
}

// natural log.  0 -> -inf, x < 0 -> NaN, +inf -> +inf.  Subnormal inputs are scaled first.
static inline float log_(float x) {
    if (x != x || x < 0.0f) return __builtin_nanf("");
    if (x == 0.0f) return -__builtin_inff();
    if (x == __builtin_inff()) return x;
    int e = 0;
    uint32_t ix = f2u(x);
    if (ix < 0x00800000u) { x = x * 8388608.0f; ix = f2u(x); e = -23; }
    e += (int)(ix >> 23) - 127;
    float f = u2f((ix & 0x007fffffu) | 0x3f800000u);
    if (f > 0x1.6a09e6p+0f) { f = f * 0.5f; e += 1; }
    f = f - 1.0f;
    float p = 0x1.16d79ep-4f;
    p = fma_(p, f, -0x1.dfb3a4p-4f);
    p = fma_(p, f, 0x1.e69804p-4f);
    p = fma_(p, f, -0x1.fbdbap-4f);
    p = fma_(p, f, 0x1.2338e2p-3f);
    p = fma_(p, f, -0x1.555e68p-3f);
    p = fma_(p, f, 0x1.99a4b4p-3f);
    p = fma_(p, f, -0x1.000002p-2f);
    p = fma_(p, f, 0x1.555548p-2f);
    float f2 = f * f;
    float fe = (float)e;
    float y = (f * f2) * p;
    y = fma_(fe, OM_LN2_LO, y);
    y = fma_(-0.5f, f2, y);
    float z = f + y;
    return fma_(fe, OM_LN2_HI, z);
}

// sin and cos together (|x| up to a few thousand keeps full accuracy).
static inline void sincos_(float x, float* s_out, float* c_out) {
    float k = floorf(fma_(x, OM_2OPI, 0.5f));
    float r = fma_(-k, OM_PIO2_HI, x);
    r = fma_(-k, OM_PIO2_MID, r);
    r = fma_(-k, OM_PIO2_LO, r);
    int q = ((int)k) & 3;
    float r2 = r * r;
    float ps = 0x1.6cca94p-19f;
    ps = fma_(ps, r2, -0x1.a00f5ap-13f);
    ps = fma_(ps, r2, 0x1.111108p-7f);
    ps = fma_(ps, r2, -0x1.555556p-3f);
    float sn = fma_(r * r2, ps, r);
    float pc = -0x1.241daap-22f;
    pc = fma_(pc, r2, 0x1.a010dap-16f);
    pc = fma_(pc, r2, -0x1.6c16b8p-10f);
    pc = fma_(pc, r2, 0x1.555556p-5f);
    float cs = fma_(r2 * r2, pc, fma_(-0.5f, r2, 1.0f));
    float s = (q & 1) ? cs : sn;
    float c = (q & 1) ? sn : cs;
    if (q & 2) s = -s;
    if ((q + 1) & 2) c = -c;
    *s_out = s;
    *c_out = c;
}
static inline float sin_(float x) { float s, c; sincos_(x, &s, &c); return s; }
static inline float cos_(float x) { float s, c; sincos_(x, &s, &c); return c; }

// ---- contract 2 (round 2): atan2, asin and the log of a random draw are fitted to what the path needs
// (tools/fit_f32_polys.py fit_c2: 3.5e-7 rad, 6e-8 rad, 2.8e-7 relative) and every quotient by a shared or constant
// divisor is the product with its correctly rounded reciprocal (1.0f / d is IEEE here).
static const float OM_INV_PI = (float)(1.0 / M_PI);
static const float OM_INV_2PI = (float)(0.5 / M_PI);
static const float OM_INV_255 = (float)(1.0 / 255.0);
static const float OM_LN2 = 0x1.62e430p-1f;

// atan2(y, x) in (-pi, pi]: ratio = smaller magnitude times the reciprocal of the larger; 0 when both are below 2^-60.
static inline float atan2_(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const bool swap = ay > ax;
    const float mx = swap ? ay : ax;
    const float mn = swap ? ax : ay;
    const float inv = 1.0f / mx;
    const float a = mn * inv;
    const float s = a * a;
    static const float K[6] = {-0x1.5540c2p-2f, 0x1.96cb56p-3f, -0x1.12821cp-3f, 0x1.50f1cep-4f, -0x1.23a7dep-5f, 0x1.e50bcap-8f};
    float p = K[5];
    for (int i = 4; i >= 0; --i) p = fma_(p, s, K[i]);
    float r = fma_(a * s, p, a);
    if (swap) r = OM_PIO2 - r;
    if (x < 0.0f) r = OM_PI - r;
    if (y < 0.0f) r = -r;
    if (mx < 0x1p-60f) return 0.0f;
    return r;
}

// asin with the argument clamped to [-1, 1] (a normalised f32 vector component can exceed 1 by an ulp).
static inline float asin_(float x) {
    if (x != x) return x;
    float ax = fabsf(x);
    if (ax > 1.0f) ax = 1.0f;
    const bool big = ax > 0.5f;
    const float s = big ? (1.0f - ax) * 0.5f : ax * ax;
    const float t = big ? sqrt_(s) : ax;
    static const float K[4] = {0x1.554bf8p-3f, 0x1.35737ap-4f, 0x1.409a72p-5f, 0x1.a7813ap-5f};
    float p = K[3];
    for (int i = 2; i >= 0; --i) p = fma_(p, s, K[i]);
    float r = fma_(t * s, p, t);
    if (big) r = fma_(-2.0f, r, OM_PIO2);
    return (x < 0.0f) ? -r : r;
}

// log of a uniform draw x = k * 2^-24 (0 <= k < 2^24).  The significand is brought into [sqrt(1/2), sqrt(2)) by adding
// bits(1) - bits(sqrt(1/2)) to the representation: the carry into the exponent field does the "halve and increment".
static inline float log_unit_(float x) {
    if (x == 0.0f) return -__builtin_inff();
    const uint32_t shifted = f2u(x) + (0x3f800000u - 0x3f3504f3u);
    const int e = (int)(shifted >> 23) - 127;
    const float m = u2f((shifted & 0x007fffffu) + 0x3f3504f3u);
    const float f = m - 1.0f;
    static const float K[6] = {0x1.5557bep-2f, -0x1.ffa98cp-3f, 0x1.980734p-3f, -0x1.5ed6f8p-3f, 0x1.4875bep-3f, -0x1.a34a8ap-4f};
    float p = K[5];
    for (int i = 4; i >= 0; --i) p = fma_(p, f, K[i]);
    const float f2 = f * f;
    float y = (f * f2) * p;
    y = fma_(-0.5f, f2, y);
    return fma_((float)e, OM_LN2, f + y);
}

// a^b for a >= 0 via exp(b*log(a)); exact identities for b == 0 and b == 1.  a < 0 -> NaN.
static inline float pow_(float a, float b) {
    if (b == 0.0f) return 1.0f;
    if (b == 1.0f) return a;
    return exp_(b * log_(a));
}

}  // namespace deo
