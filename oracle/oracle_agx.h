// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// oracle_agx.h — restatement of lib/AgX.py:1-159 (Troy Sobotka's AgX as a shader, the alternative display transform
// the reference keeps behind a comment at renderer.py:356: `output = srgb_transfer(agx.display_transform(linear))`).
// lib/AgX.py:130-159 `display_transform` returns clamp(workingColor, 0, 1) (its last line, :159 — rounds 1 and 2 had missed it
// and returned the un-clamped saturation mix; found in round 3 by executing the reference's text, tests/golden/ref_leaves.npz).  Matrix set-up (PrimariesToMatrix, ComputeCompressionMatrix, InverseMat, :22-85) and the two
// AgXScale constants (:99-101) depend on module constants only: evaluated in double, rounded to f32 once.
#pragma once
#include "oracle_lib.h"

namespace deo {
namespace agx {

static const double MIDDLE_GREY = 0.18, SLOPE = 2.3, TOE_POWER = 1.9, SHOULDER_POWER = 3.1, COMPRESSION = 0.15;   // :11-15
static const double MIN_EV = -10.0, MAX_EV = 6.5, SATURATION = 1.4;                                               // :18-20

struct M3 { double m[9]; };
static inline M3 inverse(const M3& a) {                                                                           // :22-41
    const double* m = a.m;
    double d = m[0] * (m[4] * m[8] - m[7] * m[5]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    double id = 1.0 / d;
    M3 c;
    c.m[0] = id * (m[4] * m[8] - m[7] * m[5]); c.m[1] = id * (m[2] * m[7] - m[1] * m[8]); c.m[2] = id * (m[1] * m[5] - m[2] * m[4]);
    c.m[3] = id * (m[5] * m[6] - m[3] * m[8]); c.m[4] = id * (m[0] * m[8] - m[2] * m[6]); c.m[5] = id * (m[3] * m[2] - m[0] * m[5]);
    c.m[6] = id * (m[3] * m[7] - m[6] * m[4]); c.m[7] = id * (m[6] * m[1] - m[0] * m[7]); c.m[8] = id * (m[0] * m[4] - m[3] * m[1]);
    return c;
}
static inline void unproject(double x, double y, double* X, double* Y, double* Z) {                               // :43-57 with Y = 1
    *X = 0; *Y = 0; *Z = 0;
    if (y != 0.0) { *Y = 1.0; *X = x / y; *Z = (1.0 - x - y) / y; }
}
static inline M3 primaries_to_matrix(const double r[2], const double g[2], const double b[2], const double w[2]) {   // :59-74
    double Xr, Yr, Zr, Xg, Yg, Zg, Xb, Yb, Zb, Xw, Yw, Zw;
    unproject(r[0], r[1], &Xr, &Yr, &Zr); unproject(g[0], g[1], &Xg, &Yg, &Zg);
    unproject(b[0], b[1], &Xb, &Yb, &Zb); unproject(w[0], w[1], &Xw, &Yw, &Zw);
    M3 t = {{Xr, Xg, Xb, 1.0, 1.0, 1.0, Zr, Zg, Zb}};
    M3 inv = inverse(t);
    double sx = inv.m[0] * Xw + inv.m[1] * Yw + inv.m[2] * Zw;
    double sy = inv.m[3] * Xw + inv.m[4] * Yw + inv.m[5] * Zw;
    double sz = inv.m[6] * Xw + inv.m[7] * Yw + inv.m[8] * Zw;
    M3 o = {{sx * Xr, sy * Xg, sz * Xb, sx * Yr, sy * Yg, sz * Yb, sx * Zr, sy * Zg, sz * Zb}};
    return o;
}
static inline double agx_scale(double x_pivot, double y_pivot, double slope_pivot, double power) {                // :99-101
    return pow(pow(slope_pivot * x_pivot, -power) * (pow(slope_pivot * (x_pivot / y_pivot), power) - 1.0), -1.0 / power);
}
struct Consts { float srgb_to_xyz[9], xyz_to_adjusted[9]; float x_pivot, y_pivot, toe_scale, shoulder_scale; };
static inline Consts consts() {
    const double R[2] = {0.64, 0.33}, G[2] = {0.3, 0.6}, B[2] = {0.15, 0.06}, W[2] = {0.3127, 0.3290};
    M3 s2x = primaries_to_matrix(R, G, B, W);                                                                     // :133-136
    const double sf = 1.0 / (1.0 - COMPRESSION);                                                                  // :76-85
    double r2[2] = {(R[0] - W[0]) * sf + W[0], (R[1] - W[1]) * sf + W[1]}, g2[2] = {(G[0] - W[0]) * sf + W[0], (G[1] - W[1]) * sf + W[1]};
    double b2[2] = {(B[0] - W[0]) * sf + W[0], (B[1] - W[1]) * sf + W[1]};
    M3 x2a = inverse(primaries_to_matrix(r2, g2, b2, W));                                                         // :138-144
    Consts k;
    for (int i = 0; i < 9; ++i) { k.srgb_to_xyz[i] = (float)s2x.m[i]; k.xyz_to_adjusted[i] = (float)x2a.m[i]; }
    const double xp = fabs(MIN_EV) / (MAX_EV - MIN_EV), yp = 0.5;                                                 // :149-150
    k.x_pivot = (float)xp; k.y_pivot = (float)yp;
    k.toe_scale = (float)agx_scale(xp, yp, SLOPE, TOE_POWER);                         // x <  x_pivot: scale_x_pivot = x_pivot        :119-123
    k.shoulder_scale = (float)agx_scale(1.0 - xp, 1.0 - yp, SLOPE, SHOULDER_POWER);   // x >= x_pivot: scale_x_pivot = 1 - x_pivot
    return k;
}
static inline float log2_(float x) { return log_(x) / (float)0.6931471805599453; }
static inline float hyperbolic(float x, float power) { return x / pow_(1.0f + pow_(x, power), 1.0f / power); }   // :103-105
// :111-127 for one channel
static inline float full_curve(const Consts& k, float x) {
    const bool upper = x >= k.x_pivot;
    const float scale = upper ? k.shoulder_scale : -k.toe_scale;
    const float term = ((float)SLOPE * (x - k.x_pivot)) / scale;                                                  // :107-109
    const float power = (scale < 0.0f) ? (float)TOE_POWER : (float)SHOULDER_POWER;
    return scale * hyperbolic(term, power) + k.y_pivot;
}
// :130-159
static inline vec3 display_transform(vec3 c) {
    const Consts k = consts();
    vec3 xyz = mat3_mul(k.srgb_to_xyz, c);
    vec3 adj = mat3_mul(k.xyz_to_adjusted, xyz);
    const float lo = (float)MIN_EV, hi = (float)MAX_EV, total = (float)(MAX_EV - MIN_EV);
    float v[3] = {adj.x, adj.y, adj.z}, o[3];
    for (int i = 0; i < 3; ++i) {
        // OpenDomainToNormalizedLog2 (:88-95); a non-positive channel has no logarithm: NaN / -inf clamp to MIN_EV
        float l = min_(max_(lo, log2_(v[i] / (float)MIDDLE_GREY)), hi);
        float x = (l - lo) / total;
        o[i] = clamp_(full_curve(k, x), 0.0f, 1.0f);
    }
    vec3 w(o[0], o[1], o[2]);
    w = mix(lum3(w), w, (float)SATURATION);
    return vec3(clamp_(w.x, 0.0f, 1.0f), clamp_(w.y, 0.0f, 1.0f), clamp_(w.z, 0.0f, 1.0f));   // :159 `return clamp(workingColor, 0., 1.)`
}

}  // namespace agx
}  // namespace deo
