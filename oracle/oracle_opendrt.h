// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// oracle_opendrt.h — restatement of lib/OpenDRT.py:77-131,191-218,221-485 (OpenDRT v0.2.2, GPLv3 upstream)
// for the configuration the reference actually runs (lib/OpenDRT.py:39-55): in_gamut = Rec709,
// display_gamut = Rec709, EOTF = lin, Lp = 100.  Branches that are dead for those module constants
// (other gamuts, PQ/HLG encodes, the inverse tonescale) are not restated.
#pragma once
#include "oracle_lib.h"

namespace deo {
namespace opendrt {

// lib/OpenDRT.py:60,72 — mat3(vec3, vec3, vec3) = rows
static const float rec709_to_xyz[9] = {0.412390917540f, 0.357584357262f, 0.180480793118f,
                                       0.212639078498f, 0.715168714523f, 0.072192311287f,
                                       0.019330825657f, 0.119194783270f, 0.950532138348f};
static const float xyz_to_rec709[9] = {3.2409699419f,   -1.53738317757f, -0.498610760293f,
                                       -0.969243636281f, 1.87596750151f,  0.041555057407f,
                                       0.055630079697f, -0.203976958889f, 1.05697151424f};
// :86-88  vdot(m, v) = v @ m  (row vector times matrix)
static inline vec3 vdot(const float* m, vec3 v) {
    return vec3((v.x * m[0] + v.y * m[3]) + v.z * m[6], (v.x * m[1] + v.y * m[4]) + v.z * m[7],
                (v.x * m[2] + v.y * m[5]) + v.z * m[8]);
}
// :92-97  (the `result == 0.0` typo leaves result at 0)
static inline float sdivf(float a, float b) { return (fabsf(b) < 1e-4f) ? 0.0f : a / b; }
static inline vec3 sdivf3f(vec3 a, float b) { return vec3(sdivf(a.x, b), sdivf(a.y, b), sdivf(a.z, b)); }   // :100-102
// :111-116  spowf for the three exponents that are live: 1.0 (contrast), 2.0 (flare), 0.5 (v_p)
static inline float spowf_1(float a) { return a; }
static inline float spowf_2(float a) { return (a <= 0.0f) ? a : a * a; }
static inline float spowf_half(float a) { return (a <= 0.0f) ? a : sqrt_(a); }
static inline vec3 maxf3(float mn, vec3 a) { return vec3(max_(a.x, mn), max_(a.y, mn), max_(a.z, mn)); }   // :124-126
static inline vec3 minf3(float mx, vec3 a) { return vec3(min_(a.x, mx), min_(a.y, mx), min_(a.z, mx)); }   // :129-131
static inline vec3 narrow_hue_angles(vec3 v) {                                                              // :191-197
    return vec3(min_(2.0f, max_(0.0f, v.x - (v.y + v.z))), min_(2.0f, max_(0.0f, v.y - (v.x + v.z))),
                min_(2.0f, max_(0.0f, v.z - (v.x + v.y))));
}

// Parameter setup, :257-319.  These are constant expressions of module constants; they are evaluated in
// double here and rounded to f32 once (the reference evaluates them in f32 per pixel; unpinned either way).
struct Consts { float ds, clamp_max, m, s, fl, dch_s; float px, gy, s0, m0; };
static inline Consts consts() {
    const double Lp = 100.0, gb = 0.12, c = 1.0, fl = 0.005, dch = 0.35;
    double ds = 100.0 / Lp;                                    // eotf == lin                  :270
    double clamp_max = ds * Lp / 100.0;                        //                              :271
    double px = 128.0 * log10(Lp) / log10(100.0) - 64.0;       //                              :306
    double py = Lp / 100.0;                                    //                              :308
    double gx = 0.18;                                          //                              :310
    double gy = 11.696 / 100.0 * (1.0 + gb * log10(py) / log10(2.0));                       // :312
    double s0 = (gy + sqrt(gy * (4.0 * fl + gy))) / 2.0;       // flare(gy, fl, 1)             :315,217
    double m0 = (py + sqrt(py * (4.0 * fl + py))) / 2.0;       // flare(py, fl, 1)             :316
    double ip = 1.0 / c;
    double s = (px * gx * (pow(m0, ip) - pow(s0, ip))) / (px * pow(s0, ip) - gx * pow(m0, ip));  // :318
    double m = pow(m0, ip) * (s + px) / px;                    //                              :319
    Consts k;
    k.ds = (float)ds; k.clamp_max = (float)clamp_max; k.m = (float)m; k.s = (float)s; k.fl = (float)fl;
    k.dch_s = (float)(dch / s);                                //                              :404
    k.px = (float)px; k.gy = (float)gy; k.s0 = (float)s0; k.m0 = (float)m0;
    return k;
}

// :221-485, rendering code :325-473
static inline vec3 openDR_transform(float p_R, float p_G, float p_B) {
    const Consts k = consts();
    const float rw = 0.25f, bw = 0.35f, dch_toe = 0.0f, hs_r = 0.3f, hs_g = -0.1f, hs_b = -0.2f;
    vec3 rgb(p_R, p_G, p_B);
    rgb = vdot(rec709_to_xyz, rgb);                                                          // :328
    rgb = vdot(xyz_to_rec709, rgb);                                                          // :329
    float mx = max_(rgb.x, max_(rgb.y, rgb.z));                                              // :333
    float mn = min_(rgb.x, min_(rgb.y, rgb.z));                                              // :334
    vec3 h_rgb = narrow_hue_angles(sdivf3f(rgb - vec3(mn, mn, mn), mx));                     // :354-355
    vec3 w(rw, 1.0f, bw);                                                                    // :369
    w = w / length(w);                                                                       // :370
    w = w * maxf3(1e-5f, rgb);                                                               // :371
    float lum = length(w);                                                                   // :372
    vec3 rats = sdivf3f(rgb, lum);                                                           // :376
    float ts = spowf_1(k.m * lum / (lum + k.s));                                             // :379,204
    ts = spowf_2(ts) / (ts + k.fl);                                                          // :380,215
    ts *= k.ds;                                                                              // :387
    float ccf = sdivf(1.0f, lum * k.dch_s + 1.0f);                                           // :407
    float toe_ccf = (dch_toe + 1.0f) * sdivf(lum, lum + dch_toe) * ccf;                      // :417
    vec3 hs_w = (1.0f - ccf) * h_rgb;                                                        // :432
    rats = vec3(rats.x + hs_w.z * hs_b - hs_w.y * hs_g, rats.y + hs_w.x * hs_r - hs_w.z * hs_b,
                rats.z + hs_w.y * hs_g - hs_w.x * hs_r);                                     // :435
    rats = vec3(1.0f - toe_ccf + rats.x * toe_ccf, 1.0f - toe_ccf + rats.y * toe_ccf,
                1.0f - toe_ccf + rats.z * toe_ccf);                                          // :438
    rats = maxf3(0.0f, rats);                                                                // :441
    float rats_mx = max_(rats.x, max_(rats.y, rats.z));                                      // :454
    float rats_mn = min_(rats.x, min_(rats.y, rats.z));                                      // :455
    float rats_ch = sdivf(rats_mx - rats_mn, rats_mx);                                       // :458
    float chf = spowf_half(rats_ch * ts);                                                    // :461
    vec3 rats_n = sdivf3f(rats, rats_mx);                                                    // :464
    rats = rats_n * chf + rats * (1.0f - chf);                                               // :467
    rgb = rats * ts;                                                                         // :470
    rgb = minf3(k.clamp_max, rgb);                                                           // :473
    return rgb;                                                                              // eotf == 0: no encode :477-483
}

}  // namespace opendrt
}  // namespace deo
