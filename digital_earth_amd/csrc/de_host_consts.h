// de_host_consts.h — host-side constant expressions of the reference, evaluated once in double and rounded to f32: binary16 round trips of the
// LUT uploads, default parameters, OpenDRT tonescale constants, AgX matrices and curve scales.
#pragma once
#include "de_context.h"

namespace {

uint32_t f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// tex.store into an rgba16f texture (renderer.py:213-216): binary32 -> binary16 -> binary32, RNE
float quantize_f16(float f) {
    uint32_t x = f2bits(f);
    uint32_t sign = x & 0x80000000u, ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) return f;
    if (ax >= 0x477ff000u) return bits2f(sign | 0x7f800000u);
    if (ax < 0x33000001u) return bits2f(sign);
    int e = (int)(ax >> 23) - 127;
    uint32_t m = (ax & 0x007fffffu) | 0x00800000u;
    int shift = (e >= -14) ? 13 : (13 + (-14 - e));
    uint32_t kept = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (kept & 1u))) kept += 1;
    float v = ldexpf((float)kept, e - 23 + shift);
    return bits2f(f2bits(v) | sign);
}
float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 1023u;
    if (e == 31u) return bits2f(sign | 0x7f800000u | (m << 13));
    if (e == 0u) return bits2f(f2bits((float)m * 0x1p-24f) | sign);
    return bits2f(sign | ((e + 112u) << 23) | (m << 13));
}

void default_params(de_params* p) {
    memset(p, 0, sizeof(*p));
    p->camera_pos[0] = -15000000.0f; p->camera_pos[2] = 15000000.0f;   // earth_viewer.py:27
    p->up[1] = 1.0f;
    p->fov = (float)(27.0 * M_PI / 180.0 * 0.5);                        // renderer.py:50
    p->aspect_scale = 1.0f; p->exposure = 2.5f; p->gamma = 1.0f;       // :51-53
    p->sun_angle = (float)(60.0 * M_PI / 180.0);                       // :55
    p->sun_path_rot = (float)(-45.0 * M_PI / 180.0);                   // :56
    p->land_height_scale = 7800.0f;                                    // :58
    p->vignette_strength = 0.9f; p->vignette_radius = 0.0f;            // :20-21
    p->vignette_center[0] = 0.5f; p->vignette_center[1] = 0.5f;        // :22
}

// lib/OpenDRT.py:257-319,404 for the live module constants (Lp=100, gb=.12, c=1, fl=.005, dch=.35, EOTF=lin):
// constant expressions, evaluated once in double and rounded to f32.
void opendrt_consts(float* m, float* s, float* fl, float* ds, float* clamp_max, float* dch_s) {
    const double Lp = 100.0, gb = 0.12, c = 1.0, flare = 0.005, dch = 0.35;
    double ds_d = 100.0 / Lp, cm = ds_d * Lp / 100.0;
    double px = 128.0 * log10(Lp) / log10(100.0) - 64.0, py = Lp / 100.0, gx = 0.18;
    double gy = 11.696 / 100.0 * (1.0 + gb * log10(py) / log10(2.0));
    double s0 = (gy + sqrt(gy * (4.0 * flare + gy))) / 2.0, m0 = (py + sqrt(py * (4.0 * flare + py))) / 2.0;
    double ip = 1.0 / c;
    double s_d = (px * gx * (pow(m0, ip) - pow(s0, ip))) / (px * pow(s0, ip) - gx * pow(m0, ip));
    double m_d = pow(m0, ip) * (s_d + px) / px;
    *m = (float)m_d; *s = (float)s_d; *fl = (float)flare; *ds = (float)ds_d; *clamp_max = (float)cm; *dch_s = (float)(dch / s_d);
}

// lib/AgX.py:22-85, 99-101, 133-150: matrices and curve scales of the AgX display transform — constant expressions of
// module constants (primaries, COMPRESSION = 0.15, SLOPE = 2.3, TOE_POWER = 1.9, SHOULDER_POWER = 3.1, EV range
// [-10, 6.5]), evaluated in double and rounded to f32.
struct M3d { double m[9]; };
M3d m3_inverse(const M3d& a) {
    const double* m = a.m;
    double d = m[0] * (m[4] * m[8] - m[7] * m[5]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    double id = 1.0 / d;
    M3d c;
    c.m[0] = id * (m[4] * m[8] - m[7] * m[5]); c.m[1] = id * (m[2] * m[7] - m[1] * m[8]); c.m[2] = id * (m[1] * m[5] - m[2] * m[4]);
    c.m[3] = id * (m[5] * m[6] - m[3] * m[8]); c.m[4] = id * (m[0] * m[8] - m[2] * m[6]); c.m[5] = id * (m[3] * m[2] - m[0] * m[5]);
    c.m[6] = id * (m[3] * m[7] - m[6] * m[4]); c.m[7] = id * (m[6] * m[1] - m[0] * m[7]); c.m[8] = id * (m[0] * m[4] - m[3] * m[1]);
    return c;
}
void xy_unproject(double x, double y, double* X, double* Y, double* Z) {
    *X = 0; *Y = 0; *Z = 0;
    if (y != 0.0) { *Y = 1.0; *X = x / y; *Z = (1.0 - x - y) / y; }
}
M3d primaries_to_matrix(const double r[2], const double g[2], const double b[2], const double w[2]) {
    double Xr, Yr, Zr, Xg, Yg, Zg, Xb, Yb, Zb, Xw, Yw, Zw;
    xy_unproject(r[0], r[1], &Xr, &Yr, &Zr); xy_unproject(g[0], g[1], &Xg, &Yg, &Zg);
    xy_unproject(b[0], b[1], &Xb, &Yb, &Zb); xy_unproject(w[0], w[1], &Xw, &Yw, &Zw);
    M3d t = {{Xr, Xg, Xb, 1.0, 1.0, 1.0, Zr, Zg, Zb}};
    M3d inv = m3_inverse(t);
    double sx = inv.m[0] * Xw + inv.m[1] * Yw + inv.m[2] * Zw;
    double sy = inv.m[3] * Xw + inv.m[4] * Yw + inv.m[5] * Zw;
    double sz = inv.m[6] * Xw + inv.m[7] * Yw + inv.m[8] * Zw;
    M3d o = {{sx * Xr, sy * Xg, sz * Xb, sx * Yr, sy * Yg, sz * Yb, sx * Zr, sy * Zg, sz * Zb}};
    return o;
}
double agx_scale(double x_pivot, double y_pivot, double slope_pivot, double power) {
    return pow(pow(slope_pivot * x_pivot, -power) * (pow(slope_pivot * (x_pivot / y_pivot), power) - 1.0), -1.0 / power);
}
void agx_consts(float* out22) {
    const double R[2] = {0.64, 0.33}, G[2] = {0.3, 0.6}, B[2] = {0.15, 0.06}, W[2] = {0.3127, 0.3290};
    M3d s2x = primaries_to_matrix(R, G, B, W);
    const double sf = 1.0 / (1.0 - 0.15);
    double r2[2] = {(R[0] - W[0]) * sf + W[0], (R[1] - W[1]) * sf + W[1]}, g2[2] = {(G[0] - W[0]) * sf + W[0], (G[1] - W[1]) * sf + W[1]};
    double b2[2] = {(B[0] - W[0]) * sf + W[0], (B[1] - W[1]) * sf + W[1]};
    M3d x2a = m3_inverse(primaries_to_matrix(r2, g2, b2, W));
    for (int i = 0; i < 9; ++i) { out22[i] = (float)s2x.m[i]; out22[9 + i] = (float)x2a.m[i]; }
    const double xp = 10.0 / (6.5 + 10.0), yp = 0.5;
    out22[18] = (float)xp; out22[19] = (float)yp;
    out22[20] = (float)agx_scale(xp, yp, 2.3, 1.9);
    out22[21] = (float)agx_scale(1.0 - xp, 1.0 - yp, 2.3, 3.1);
}

}  // namespace

