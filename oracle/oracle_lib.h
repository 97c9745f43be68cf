// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md). the reference has no
// tests or golden vectors for this path and cannot be executed here (needs Taichi + Vulkan).
//
// oracle_lib.h — CPU restatement of the reference's leaf libraries, one function per reference
// function, in the reference's expression order, in f32:
//   lib/math_utils.py, lib/sampling.py, lib/volume_rendering_models.py,
//   lib/surface_rendering_models.py, lib/colour.py   (paths relative to /root/reference)
// Conventions: Python-float constant sub-expressions fold in double and are then rounded to f32 (Taichi
// evaluates them in Python before they meet an f32 Expr); pow() with an integer-valued literal exponent is a
// multiply chain; NaN-as-miss is kept literally — C comparisons with NaN are false, as in Taichi.
#pragma once
#include "oracle_math.h"
#include <vector>

namespace deo {

// ------------------------------------------------------------------ vectors
struct vec2 { float x, y; };
struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
};
struct vec4 { float x, y, z, w; };
static inline vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 operator-(vec3 a) { return vec3(-a.x, -a.y, -a.z); }
static inline vec3 operator*(vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
static inline vec3 operator*(float s, vec3 a) { return vec3(s * a.x, s * a.y, s * a.z); }
static inline vec3 operator*(vec3 a, vec3 b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline vec3 operator/(vec3 a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
// Taichi Vector.dot = sum of products, left to right
static inline float dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float length(vec3 a) { return sqrt_(dot(a, a)); }
// v.normalized(): no epsilon (Appendix B).  Contract 2: the three quotients by |v| are products with RN(1 / |v|).
static inline vec3 normalized(vec3 a) { const float inv = 1.0f / length(a); return vec3(a.x * inv, a.y * inv, a.z * inv); }
static inline vec3 cross(vec3 a, vec3 b) {
    return vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// taichi.math.mix: a*(1-t) + b*t, t not clamped (Appendix B)
static inline float mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }
static inline vec3 mix(vec3 a, vec3 b, float t) { return vec3(mix(a.x, b.x, t), mix(a.y, b.y, t), mix(a.z, b.z, t)); }
static inline float smoothstep(float e0, float e1, float x) {
    float t = clamp_((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
static inline float sqr(float x) { return x * x; }                                   // lib/math_utils.py:9-11
static inline float pow5(float x) { float x2 = x * x; float x4 = x2 * x2; return x4 * x; }  // pow(x, 5.0)

// ------------------------------------------------------------------ RNG
// Stands in for ti.random() (Taichi's stream is runtime-internal and not reproducible).
// One xoroshiro64* stream per (seed, absolute pixel index, sample index); the n-th call of next_f32() on
// that stream is "draw# n" of SURVEY Appendix A.  24-bit uniform in [0,1), like ti.random(f32).
struct Rng {
    uint32_t s0, s1;
    uint32_t draws;
    static uint32_t fmix32(uint32_t h) {
        h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
        return h;
    }
    static uint32_t rotl(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
    void seed(uint64_t seed, uint32_t pixel, uint32_t sample) {
        uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
        uint32_t a = fmix32((pixel * 0x9E3779B1u) ^ lo);
        uint32_t b = fmix32((sample * 0x85EBCA77u) ^ hi ^ a);
        s0 = fmix32(a ^ rotl(b, 16) ^ 0x68E31DA4u);
        s1 = fmix32(b + 0xB5297A4Du);
        if ((s0 | s1) == 0u) s0 = 1u;
        draws = 0;
    }
    uint32_t next_u32() {
        uint32_t r = s0 * 0x9E3779BBu;
        uint32_t t = s1 ^ s0;
        s0 = rotl(s0, 26) ^ t ^ (t << 9);
        s1 = rotl(t, 13);
        ++draws;
        return r;
    }
    float next_f32() { return (float)(next_u32() >> 8) * 0x1p-24f; }
};

// ------------------------------------------------------------------ textures
// Texel (i, j): i along u (width), j along v (height); j = 0 is v = 0 = south pole
// (renderer.py:171-210 store (i, j) from imread's [x][y] with y up; lib/math_utils.py:25-28).
// sample_lod(uv, 0): normalised coordinates, texel centres at (i+0.5)/N, bilinear with f32 weights.
// Address mode is Taichi-internal: REPEAT by default, CLAMP selectable.
struct Texture {
    int w = 0, h = 0, c = 0;       // c = channels stored (1, 3 or 4)
    bool is_float = false;
    bool clamp = false;
    std::vector<uint8_t> u8;       // unorm8: value = u8 / 255 (renderer.py:173-210)
    std::vector<float> f32;        // LUT textures (CIE rgba16f values already quantised, CRF rgba32f)
    int wrap(int i, int n) const {
        if (clamp) return i < 0 ? 0 : (i >= n ? n - 1 : i);
        i %= n;
        return i < 0 ? i + n : i;
    }
    // unorm8 maps: the stored byte as a float (0..255); sample_lod filters the bytes and scales the result by RN(1/255)
    // (contract 2 — the sampler's own arithmetic is Taichi-internal either way)
    float texel(int i, int j, int ch) const {
        size_t idx = ((size_t)j * w + i) * c + ch;
        return is_float ? f32[idx] : (float)u8[idx];
    }
};
struct TapCounters { uint64_t taps_r8 = 0, taps_rgba8 = 0, taps_lut = 0; };
static thread_local TapCounters g_taps;

static inline void sample_lod(const Texture& t, float u, float v, float* out, int nch) {
    float x = u * (float)t.w - 0.5f;
    float y = v * (float)t.h - 0.5f;
    float x0f = floorf(x), y0f = floorf(y);
    float fx = x - x0f, fy = y - y0f;
    int i0 = t.wrap((int)x0f, t.w), i1 = t.wrap((int)x0f + 1, t.w);
    int j0 = t.wrap((int)y0f, t.h), j1 = t.wrap((int)y0f + 1, t.h);
    for (int ch = 0; ch < nch; ++ch) {
        float t00 = t.texel(i0, j0, ch), t10 = t.texel(i1, j0, ch);
        float t01 = t.texel(i0, j1, ch), t11 = t.texel(i1, j1, ch);
        float a = fma_(t10 - t00, fx, t00);
        float b = fma_(t11 - t01, fx, t01);
        float filtered = fma_(b - a, fy, a);
        out[ch] = t.is_float ? filtered : filtered * OM_INV_255;
    }
    if (t.is_float) g_taps.taps_lut++;
    else if (t.c == 1) g_taps.taps_r8++;
    else g_taps.taps_rgba8++;
}

// ------------------------------------------------------------------ lib/math_utils.py
static inline float cone_angle_to_solid_angle(float x) {                              // :13-15
    return (float)(M_PI * 2) * (1.0f - cos_(x));
}
// :17-23  — sqrt of a negative discriminant gives NaN; `discr < 0` is then false, so a miss returns (NaN, NaN)
static inline vec2 rsi(vec3 pos, vec3 dir, float r) {
    float b = dot(pos, dir);
    float discr = b * b - dot(pos, pos) + r * r;
    discr = sqrt_(discr);
    if (discr < 0.0f) return vec2{-1.0f, -1.0f};
    return vec2{-b + -discr, -b + discr};
}
static inline vec2 sphere_UV_map(vec3 n) {                                            // :25-28
    // (atan2 / pi + 1) / 2  and  asin / pi + 0.5, each as ONE fused multiply-add with RN(1/2pi) / RN(1/pi) (contract 2)
    return vec2{fma_(atan2_(n.z, -n.x), OM_INV_2PI, 0.5f), fma_(asin_(n.y), OM_INV_PI, 0.5f)};
}
static inline float fract(float x) { return x - floorf(x); }
static inline void sample_sphere_texture(const Texture& t, vec3 pos, float* out, int nch) {  // :38-44, scale = 1
    vec2 uv = sphere_UV_map(normalized(pos));
    sample_lod(t, fract(uv.x * 1.0f), fract(uv.y * 1.0f), out, nch);
}
static inline float sample_sphere_texture_r(const Texture& t, vec3 pos) {
    float o; sample_sphere_texture(t, pos, &o, 1); return o;
}
static inline void make_orthonormal_basis(vec3 n, vec3* x, vec3* y) {                 // :55-60
    vec3 h = (fabsf(n.y) > 0.9f) ? vec3(1.0f, 0.0f, 0.0f) : vec3(0.0f, 1.0f, 0.0f);
    *y = normalized(cross(n, h));
    *x = cross(n, *y);
}
// :62-65 make_tangent_space(n) @ v  ==  columns (x, y, n): row i = (x_i*v0 + y_i*v1) + n_i*v2
static inline vec3 tangent_space_apply(vec3 n, vec3 v) {
    vec3 x, y; make_orthonormal_basis(n, &x, &y);
    return vec3((x.x * v.x + y.x * v.y) + n.x * v.z, (x.y * v.x + y.y * v.y) + n.y * v.z,
                (x.z * v.x + y.z * v.y) + n.z * v.z);
}
static inline vec3 spherical_direction(float sin_theta, float cos_theta, float phi, vec3 x, vec3 y, vec3 z) {  // :67-69
    float s, c; sincos_(phi, &s, &c);
    return (sin_theta * c) * x + (sin_theta * s) * y + cos_theta * z;
}

// ------------------------------------------------------------------ lib/sampling.py
static inline vec3 sample_cone(Rng& rng, float cos_theta_max) {                        // :13-23
    float u0 = rng.next_f32();
    float u1 = rng.next_f32();
    float cos_theta = (1.0f - u0) + u0 * cos_theta_max;
    float sin_theta = sqrt_(1.0f - cos_theta * cos_theta);
    float phi = (float)(2.0 * M_PI) * u1;
    float s, c; sincos_(phi, &s, &c);
    return vec3(sin_theta * c, sin_theta * s, cos_theta);
}
static inline vec3 sample_cone_oriented(Rng& rng, float cos_theta_max, vec3 n) {       // :25-28
    return tangent_space_apply(n, sample_cone(rng, cos_theta_max));
}
static inline vec3 sample_hemisphere_cosine_weighted(Rng& rng, vec3 n) {               // :30-39
    float u0 = rng.next_f32();
    float u1 = rng.next_f32();
    float a = 1.0f - 2.0f * u0;
    float b = sqrt_(1.0f - a * a);
    a *= (float)(1.0 - 1e-5);
    b *= (float)(1.0 - 1e-5);
    float phi = (float)(2.0 * M_PI) * u1;
    float s, c; sincos_(phi, &s, &c);
    return normalized(vec3(n.x + b * c, n.y + b * s, n.z + a));
}
static inline vec3 sample_sphere(float r0, float r1) {                                 // :41-45
    float rx = r0 * (float)(M_PI * 2.0);
    float ry = r1 * 2.0f - 1.0f;
    float s, c; sincos_(rx, &s, &c);
    float k = sqrt_(1.0f - ry * ry);
    return normalized(vec3(s * k, c * k, ry));
}

// ------------------------------------------------------------------ lib/volume_rendering_models.py
namespace volume {
static const double air_num_density_d = 2.5035422e25;                                  // :9
static const double ozone_num_density_d = air_num_density_d * 0.012588 * 8e-6;          // :10-11
static const float ozone_peak_height = 25000.0f;                                       // :13
static const float mie_asymmetry = 3000.0f;                                            // :18
static const double turbidity_d = 1.06;                                                // :19
enum { RAYLEIGH_ID = 0, MIE_ID = 1, OZONE_ID = 2, CLOUD_ID = 3, ISOTROPIC_CLOUD_ID = 4 };  // :21-25
static const float rayleigh_albedo = 1.0f, aerosol_albedo = 0.95f, ozone_albedo = 0.0f, cloud_albedo = 0.99f;  // :27-30
static const float planet_r = 6371e3f;                                                 // :34
static const float atmos_upper_limit = (float)(6371e3 + 110e3);                        // :35-36
static const float clouds_extinct = 0.1f;                                              // :39
static const float clouds_density = 0.029f;                                            // :40
static const float clouds_thickness = 6000.0f;                                         // :42
static const float clouds_lower_limit = (float)(6371e3 + 4000.0);                      // :43
static const float clouds_upper_limit = (float)(6371e3 + 4000.0 + 6000.0);             // :44

static inline float air(float wavelength) {                                            // :48-51
    float rcp = 1.0f / (wavelength * wavelength);
    return (float)(1.0 + 8.06051e-5) + 2.480990e-2f / (132.274f - rcp) + 1.74557e-4f / (39.32957f - rcp);
}
static inline float rayleigh_phase(float c) {                                          // :61-63
    return (float)(3.0 / (16.0 * M_PI)) * (1.0f + c * c);
}
static inline float hg_phase(float c, float g) {                                       // :73-76, pow(x,1.5) = x*sqrt(x)
    float d = 1.0f + g * g - 2.0f * g * c;
    return (1.0f - g * g) / ((float)(4.0 * M_PI) * (d * sqrt_(d)));
}
static inline vec3 sample_hg_phase(Rng& rng, vec3 view, float g) {                     // :78-85
    float sqr_term = (1.0f - g * g) / (1.0f - g + 2.0f * g * rng.next_f32());
    float cos_theta = (1.0f + g * g - sqr_term * sqr_term) / (2.0f * g);
    float sin_theta = sqrt_(max_(0.0f, 1.0f - cos_theta * cos_theta));
    float phi = (float)(2.0 * M_PI) * rng.next_f32();
    vec3 tang, bitang; make_orthonormal_basis(view, &tang, &bitang);
    return spherical_direction(sin_theta, cos_theta, phi, tang, bitang, view);
}
static inline float klein_nishina_phase(float c, float e) {                            // :87-89
    return e / ((float)(2.0 * M_PI) * (e * (1.0f - c) + 1.0f) * log_(2.0f * e + 1.0f));
}
static inline vec3 sample_klein_nishina_phase(Rng& rng, vec3 view, float e) {          // :91-98
    float cos_theta = (-pow_(2.0f * e + 1.0f, 1.0f - rng.next_f32()) + e + 1.0f) / e;
    float sin_theta = sqrt_(max_(0.0f, 1.0f - cos_theta * cos_theta));
    float phi = (float)(2.0 * M_PI) * rng.next_f32();
    vec3 tang, bitang; make_orthonormal_basis(view, &tang, &bitang);
    return spherical_direction(sin_theta, cos_theta, phi, tang, bitang, view);
}
static inline float mie_phase(float c) { return klein_nishina_phase(c, mie_asymmetry); }           // :65-67
static inline vec3 sample_mie_phase(Rng& rng, vec3 view) { return sample_klein_nishina_phase(rng, view, mie_asymmetry); }  // :69-71

// The Draine phase function and its analytic CDF inversion below restate formulas that the reference carries under this
// notice (lib/volume_rendering_models.py:98-120):
//   SPDX-FileCopyrightText: Copyright (c) <2023> NVIDIA CORPORATION & AFFILIATES. All rights reserved.
//   SPDX-License-Identifier: MIT
//   Permission is hereby granted, free of charge, to any person obtaining a copy of this software and associated
//   documentation files (the "Software"), to deal in the Software without restriction, including without limitation the
//   rights to use, copy, modify, merge, publish, distribute, sublicense, and/or sell copies of the Software, and to permit
//   persons to whom the Software is furnished to do so, subject to the following conditions: The above copyright notice and
//   this permission notice shall be included in all copies or substantial portions of the Software.
//   THE SOFTWARE IS PROVIDED "AS IS", WITHOUT WARRANTY OF ANY KIND, EXPRESS OR IMPLIED, INCLUDING BUT NOT LIMITED TO THE
//   WARRANTIES OF MERCHANTABILITY, FITNESS FOR A PARTICULAR PURPOSE AND NONINFRINGEMENT. IN NO EVENT SHALL THE AUTHORS OR
//   COPYRIGHT HOLDERS BE LIABLE FOR ANY CLAIM, DAMAGES OR OTHER LIABILITY, WHETHER IN AN ACTION OF CONTRACT, TORT OR
//   OTHERWISE, ARISING FROM, OUT OF OR IN CONNECTION WITH THE SOFTWARE OR THE USE OR OTHER DEALINGS IN THE SOFTWARE.
// :121-123 (Draine phase; the reference's copy carries NVIDIA's MIT notice at :101-120 — formula restated)
static inline float draine_phase(float c, float g, float a) {
    float d = 1.0f + g * g - 2.0f * g * c;
    return ((1.0f - g * g) * (1.0f + a * c * c)) /
           (4.0f * (1.0f + (a * (1.0f + 2.0f * g * g)) / 3.0f) * (float)M_PI * (d * sqrt_(d)));
}
// :125-152  closed-form inverse-CDF sampling of the Draine phase function, expression order kept
static inline vec3 sample_draine(Rng& rng, vec3 view, float g, float a) {
    float xi = rng.next_f32();
    float g2 = g * g, g3 = g * g2, g4 = g2 * g2, g6 = g2 * g4;
    float pgp1_2 = (1.0f + g2) * (1.0f + g2);
    float T1a = -a + a * g4;
    float T1a3 = T1a * T1a * T1a;
    float T2 = -1296.0f * (-1.0f + g2) * (a - a * g2) * (T1a) * (4.0f * g2 + a * pgp1_2);
    float T3 = 3.0f * g2 * (1.0f + g * (-1.0f + 2.0f * xi)) + a * (2.0f + g2 + g3 * (1.0f + 2.0f * g2) * (-1.0f + 2.0f * xi));
    float T4a = 432.0f * T1a3 + T2 + 432.0f * (a - a * g2) * T3 * T3;
    float T4b = -144.0f * a * g2 + 288.0f * a * g4 - 144.0f * a * g6;
    float T4b3 = T4b * T4b * T4b;
    float T4 = T4a + sqrt_(-4.0f * T4b3 + T4a * T4a);
    float T4p3 = pow_(T4, (float)(1.0 / 3.0));
    const float cbrt2 = (float)1.2599210498948732;  // pow(2, 1.0/3.0) folded in Python
    float T6 = (2.0f * T1a + (48.0f * cbrt2 * (-(a * g2) + 2.0f * a * g4 - a * g6)) / T4p3 + T4p3 / (3.0f * cbrt2)) / (a - a * g2);
    float T5 = 6.0f * (1.0f + g2) + T6;
    float inner = -0.5f * sqrt_(T5) + sqrt_(6.0f * (1.0f + g2) - (8.0f * T3) / (a * (-1.0f + g2) * sqrt_(T5)) - T6) / 2.0f;
    float cos_theta = (1.0f + g2 - inner * inner) / (2.0f * g);
    float sin_theta = sqrt_(max_(0.0f, 1.0f - cos_theta * cos_theta));
    float phi = (float)(2.0 * M_PI) * rng.next_f32();
    vec3 tang, bitang; make_orthonormal_basis(view, &tang, &bitang);
    return spherical_direction(sin_theta, cos_theta, phi, tang, bitang, view);
}
// :156-183  droplet size d = 8: the four constants are f32 exp() of Python-folded arguments
struct CloudPhaseConsts { float g_hg, g_draine, alpha_draine, w_draine; };
static inline CloudPhaseConsts cloud_consts() {
    const double d = 8.0;
    CloudPhaseConsts k;
    k.g_hg = exp_((float)(-0.0990567 / (d - 1.67154)));
    k.g_draine = exp_((float)(-2.20679 / (d + 3.91029) - 0.428934));
    k.alpha_draine = exp_((float)(3.62489 - 8.29288 / (d + 5.52825)));
    k.w_draine = exp_((float)(-0.599085 / (d - 0.641583) - 0.665888));
    return k;
}
static inline float cloud_phase(float c, bool reduce_peak) {                            // :156-164
    CloudPhaseConsts k = cloud_consts();
    float g_hg = reduce_peak ? 0.91f : k.g_hg;
    return mix(hg_phase(c, g_hg), draine_phase(c, k.g_draine, k.alpha_draine), k.w_draine);
}
static inline vec3 sample_cloud_phase(Rng& rng, vec3 view, bool reduce_peak) {          // :166-183
    CloudPhaseConsts k = cloud_consts();
    float g_hg = reduce_peak ? 0.91f : k.g_hg;
    if (rng.next_f32() < k.w_draine) return sample_draine(rng, view, k.g_draine, k.alpha_draine);
    return sample_hg_phase(rng, view, g_hg);
}

static inline float spectra_extinction_mie(float wavelength) {                         // :194-200, junge-2 = 2
    const double c = (0.6544 * turbidity_d - 0.6510) * 4e-18;
    float K = (0.773335f - 0.00386891f * wavelength) / (1.0f - 0.00546759f * wavelength);
    float x = (float)(2.0 * M_PI) / (wavelength * 1e-9f);
    return (float)(0.434 * c * M_PI) * (x * x) * K;   // 0.434*c*np.pi folds in Python
}
static inline float spectra_extinction_rayleigh(float wavelength) {                    // :203-217
    float wl_nm = wavelength * 1e-9f;
    float w2 = wavelength * wavelength;
    float F_N2 = 1.034f + 3.17e-4f * (1.0f / w2);
    float F_O2 = 1.096f + 1.385e-3f * (1.0f / w2) + 1.448e-4f * (1.0f / (w2 * w2));
    float king = (78.084f * F_N2 + 20.946f * F_O2 + 0.934f + (float)(0.0421 * 1.15)) / (float)(78.084 + 20.946 + 0.934 + 0.0421);
    float n = sqr(air(wavelength * 1e-3f)) - 1.0f;
    const float pi3 = (float)(M_PI * M_PI * M_PI);
    float wl2 = wl_nm * wl_nm;
    return ((8.0f * pi3 * (n * n)) / ((float)(3.0 * air_num_density_d) * (wl2 * wl2))) * king;
}
static inline float spectra_extinction_ozone(float wavelength, const float* o3_lut) {  // :219-224
    float ext = 0.0f;
    if (wavelength >= 390.0f && wavelength < 831.0f)
        ext = (float)(0.0001 * ozone_num_density_d) * o3_lut[(int)(wavelength - 390.0f)];
    return ext;
}
static inline float get_ozone_density(float h) {                                       // :229-247
    float h_km = h * 0.001f;
    float rel = h_km - (float)(25000.0 * 0.001);
    rel = rel * rel;
    float d = (float)(1.0 - 0.375) * exp_(-rel / 49.0f);
    d += 0.375f * exp_(-rel / 256.0f);
    float t = h_km - 15.0f;
    d += max_(0.0f, -0.000015f * (t * t * t));
    return d;
}
static inline float get_rayl_density(float h) {                                        // :249-253
    float t = h + 24239.99f;
    return 3.68082f * exp_(-(t * t) / 532307548.4168f) / 1.225f;
}
static inline float get_mie_density(float h) {                                         // :255-268
    float dens;
    if (h > 11500.0f)      { float t = h - 11500.0f; dens = 0.0918f * exp_(-1.0e-6f * (t * t)); }
    else if (h > 2400.0f)  { float t = h + 2500.0f;  dens = 0.3000f * exp_(-2.5e-9f * (t * t)) - 0.092f; }
    else if (h > 1300.0f)  { float t = h - 1300.0f;  dens = 0.6500f * exp_(-5.0e-6f * (t * t)) + 0.18899f; }
    else                   { dens = 1.0f - h / 8136.646f; }
    return dens * (float)turbidity_d;
}
static inline vec3 get_density(float h) {                                              // :271-274
    h = max_(h, 0.0f);
    return vec3(get_rayl_density(h), get_mie_density(h), get_ozone_density(h));
}
static inline float get_elevation(vec3 p) {                                            // :276-278
    return sqrt_(p.x * p.x + p.y * p.y + p.z * p.z) - planet_r;
}
}  // namespace volume

// ------------------------------------------------------------------ lib/surface_rendering_models.py
namespace surface {
static const float DIFFUSE_FACTOR = 0.28f, SPECULAR_FACTOR = 0.5f;                     // :6-7
static inline float disney_diffuse(float roughness, float ndl, float ndv, float ldh) { // :40-52
    float R_R = 2.0f * roughness * sqr(ldh);
    float F_L = pow5(1.0f - ndl);
    float F_V = pow5(1.0f - ndv);
    const float f_lambert = (float)(1.0 / M_PI);
    float f_retro = f_lambert * R_R * (F_L + F_V + F_L * F_V * (R_R - 1.0f));
    return f_lambert * (1.0f - 0.5f * F_L) * (1.0f - 0.5f * F_V) + f_retro;
}
static inline float fresnel_dielectric(float vdh, float F_0) {                          // :110-122
    F_0 = sqrt_(F_0);
    F_0 = (1.0f + F_0) / (1.0f - F_0);
    float sin_i = sqrt_(saturate_(1.0f - sqr(vdh)));
    float sin_t = sin_i / max_(F_0, 1e-8f);
    float cos_t = sqrt_(1.0f - sqr(sin_t));
    float R_s = sqr((vdh - (F_0 * cos_t)) / max_(vdh + (F_0 * cos_t), 1e-8f));
    float R_p = sqr((cos_t - (F_0 * vdh)) / max_(cos_t + (F_0 * vdh), 1e-8f));
    return saturate_((R_s + R_p) * 0.5f);
}
static inline float GGX_D(float ndh, float alpha2) {                                    // :82-85
    float den = (alpha2 - 1.0f) * ndh * ndh + 1.0f;
    return alpha2 / ((float)M_PI * den * den);
}
static inline float lambda_smith(float ndx, float alpha2) {                             // :88-91
    float n2 = ndx * ndx;
    return (-1.0f + sqrt_(alpha2 * (1.0f - n2) / n2 + 1.0f)) * 0.5f;
}
static inline float G2_smith(float ndl, float ndv, float alpha2) {                      // :100-104
    float lv = lambda_smith(ndv, alpha2);
    float ll = lambda_smith(ndl, alpha2);
    return 1.0f / (1.0f + lv + ll);
}
static inline float GGX_smith_specular(float roughness, float F_0, float ndl, float ndv, float ldh, float ndh) {  // :70-80
    float alpha2 = roughness * roughness;
    float D = GGX_D(ndh, alpha2);
    float G = G2_smith(ndl, ndv, alpha2);
    float F = fresnel_dielectric(ldh, F_0);
    return D * G * F / max_(4.0f * ndl * ndv, 1e-5f);
}
static inline float beckmann_isotropic_ndf(float ndh, float alpha) {                    // :146-152
    float c2 = ndh * ndh;
    float a2 = alpha * alpha;
    float exponent = (1.0f - c2) / (a2 * c2);
    float denom = (float)M_PI * a2 * c2 * c2;
    return exp_(-exponent) / max_(denom, 1e-5f);
}
static inline float G2_VCavity(float ndl, float ndv, float ndh, float vdh) {            // :169-171
    return min_(1.0f, min_(2.0f * ndv * ndh / vdh, 2.0f * ndl * ndh / vdh));
}
static inline float beckmann_specular(float roughness, float F_0, float ndl, float ndv, float ldh, float ndh) {  // :54-68
    float alpha = roughness;
    alpha *= alpha * 2.0f;
    float D = beckmann_isotropic_ndf(ndh, alpha);
    float V = G2_VCavity(ndl, ndv, ndh, ldh);
    float F = fresnel_dielectric(ldh, F_0);
    return D * V * F;
}
// :9-38  returns brdf, writes n_dot_l
static inline float earth_brdf(float albedo, float oceanness, float bathymetry, vec3 v, vec3 n, vec3 l, float* n_dot_l_out) {
    vec3 h = normalized(v + l);
    float ndl = saturate_(dot(n, l));
    float ndv = saturate_(dot(n, v));
    float ldh = saturate_(dot(l, h));
    float ndh = saturate_(dot(n, h));
    const float land_roughness = 0.73f;
    float ocean_roughness = mix((float)(0.23 + 0.02), (float)(0.23 - 0.04), smoothstep(0.3f, 0.7f, bathymetry));
    const float land_F_0 = 0.04f, ocean_F_0 = 0.02f;
    float diffuse = disney_diffuse(land_roughness, ndl, ndv, ldh);
    float land_spec = GGX_smith_specular(land_roughness, land_F_0, ndl, ndv, ldh, ndh);
    float ocean_ggx = GGX_smith_specular(ocean_roughness, ocean_F_0, ndl, ndv, ldh, ndh);
    float ocean_beck = 0.65f * beckmann_specular(ocean_roughness, ocean_F_0, ndl, ndv, ldh, ndh);
    float ocean_spec = mix(ocean_beck, ocean_ggx, clamp_(smoothstep(0.2f, 0.95f, ndv), 0.05f, 0.94f));
    float blender = smoothstep(0.6f, 1.0f, oceanness);
    float brdf = albedo * diffuse * DIFFUSE_FACTOR + mix(land_spec, ocean_spec, blender) * SPECULAR_FACTOR;
    *n_dot_l_out = ndl;
    return brdf;
}
}  // namespace surface

// ------------------------------------------------------------------ lib/colour.py
static const float xyzToRGBMatrix_D65[9] = {                                            // :6-10
    3.2409699419f, -1.5373831776f, -0.4986107603f,
    -0.9692436363f, 1.8759675015f, 0.0415550574f,
    0.0556300797f, -0.2039769589f, 1.0569715142f};
static inline vec3 mat3_mul(const float* m, vec3 v) {  // M @ v, rows left to right
    return vec3((m[0] * v.x + m[1] * v.y) + m[2] * v.z, (m[3] * v.x + m[4] * v.y) + m[5] * v.z,
                (m[6] * v.x + m[7] * v.y) + m[8] * v.z);
}
struct SpectrumSample { float wavelength; vec3 response; float rcp_pdf; };
// :13-48.  cie: 441x2 rgba16f texture (values quantised to f16 at upload).  8 bisection steps (Q3).
// fixed_wavelength > 0 (cfg1): no draw; mid = (lambda-390)/441.
static inline SpectrumSample spectrum_sample(Rng& rng, const Texture& cie, float fixed_wavelength) {
    float mid;
    if (fixed_wavelength > 0.0f) {
        mid = (fixed_wavelength - 390.0f) / 441.0f;
    } else {
        float sample = rng.next_f32();
        float lo = 0.0f, hi = 1.0f;
        mid = (lo + hi) / 2.0f;
        for (int it = 0; it < 8; ++it) {                                  // range(0, log2(441)) -> 8
            float c3[3]; sample_lod(cie, mid, 0.25f, c3, 3);
            const float third = (float)(1.0 / 3.0);
            float val = saturate_((third * c3[0] + third * c3[1]) + third * c3[2]);
            if (val < sample) lo = mid;
            else if (val > sample) hi = mid;
            else break;
            mid = (lo + hi) / 2.0f;
        }
    }
    SpectrumSample s;
    s.wavelength = (fixed_wavelength > 0.0f) ? fixed_wavelength : 390.0f + 441.0f * mid;
    float r3[3]; sample_lod(cie, mid, 0.75f, r3, 3);
    float m3[3]; sample_lod(cie, 1.0f, 0.25f, m3, 3);
    s.response = vec3(r3[0], r3[1], r3[2]);
    float pdf = dot(s.response, vec3(m3[0], m3[1], m3[2]));
    s.rcp_pdf = 0.0f;
    if (pdf > 1e-3f && !(isinf(pdf) || isnan(pdf))) s.rcp_pdf = 1.0f / pdf;
    return s;
}
static inline float plancks(float temperature, float wavelength) {                      // :51-60
    const float h = 6.62607015e-16f, c = 2.9e17f, k = 1.38e-5f;
    float w2 = wavelength * wavelength;
    float p1 = 2.0f * h * (c * c) / (w2 * w2 * wavelength);
    float p2 = exp_((float)(6.62607015e-16 * 2.9e17) / (wavelength * k * temperature)) - 1.0f;   // h*c folds in Python
    return p1 / p2;
}
// :62-71  lut: 300 x 3 values (f16 quantised).  f = w - (lambda-400) <= 0 (Q4, kept).
static inline float srgb_to_spectrum(const float* lut, vec3 rgb, float wavelength) {
    int w = (int)(wavelength - 400.0f);
    float f = (float)w - (wavelength - 400.0f);
    float power = 0.0f;
    if (w > 0 && w < 299) {
        vec3 a(lut[w * 3], lut[w * 3 + 1], lut[w * 3 + 2]);
        vec3 b(lut[w * 3 + 3], lut[w * 3 + 4], lut[w * 3 + 5]);
        power = dot(rgb, mix(a, b, f));
    }
    return power;
}
static inline float srgb_transfer1(float linear) {                                      // :74-79
    float lo = linear * 12.92f;
    float hi = (pow_(fabsf(linear), (float)(1.0 / 2.4)) * 1.055f) - 0.055f;
    float t = (0.0031308f < linear) ? 0.0f : 1.0f;   // step(edge=linear, x=0.0031308)
    return mix(hi, lo, t);
}
static inline float lum(vec3 x) { return dot(x, vec3(0.2126729f, 0.7151522f, 0.0721750f)); }   // :88-90
static inline vec3 lum3(vec3 x) { float y = lum(x); return vec3(y, y, y); }                    // :92-95

}  // namespace deo
