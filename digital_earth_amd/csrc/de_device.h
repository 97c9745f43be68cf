// de_device.h — device-side building blocks of the Digital Earth hot path for gfx950:
// vectors, the per-path RNG, the footprint-packed texture fetch, and the leaf physics of
//   lib/math_utils.py, lib/sampling.py, lib/volume_rendering_models.py, lib/surface_rendering_models.py,
//   lib/colour.py, lib/OpenDRT.py            (paths relative to the reference repository)
// written for the arithmetic contract in de_math.h.  Expression ORDER is part of the contract: it decides the
// f32 roundings, and with them which side of a `rand < p` test a path falls on.
#pragma once
#include "de_math.h"

// ------------------------------------------------------------------ vectors
struct float3_ { float x, y, z; };
typedef float3_ vec3;
DE_DEV vec3 v3(float x, float y, float z) { vec3 r; r.x = x; r.y = y; r.z = z; return r; }
DE_DEV vec3 operator+(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
DE_DEV vec3 operator-(vec3 a, vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
DE_DEV vec3 operator-(vec3 a) { return v3(-a.x, -a.y, -a.z); }
DE_DEV vec3 operator*(vec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
DE_DEV vec3 operator*(float s, vec3 a) { return v3(s * a.x, s * a.y, s * a.z); }
DE_DEV vec3 operator*(vec3 a, vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
DE_DEV vec3 operator/(vec3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
DE_DEV float dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
DE_DEV float length(vec3 a) { return de_sqrt(dot(a, a)); }
// |a| for positions (|a|^2 ~ 4e13) and other vectors known to be far from the under/overflow ranges
DE_DEV float length_nr(vec3 a) { return de_sqrt_nr(dot(a, a)); }
// Vector.normalized() under contract 2: v * RN(1 / |v|) — one reciprocal shared by the three components
DE_DEV vec3 normalized_ieee(vec3 a) { return a * (1.0f / length(a)); }    // any argument (set-up code: camera basis)
// Every normalisation inside the integrators has |a| = 0 or |a| in [1e-6, 1e9] (directions, positions, SDF differences):
// the short correctly rounded sqrt and reciprocal apply.  |a| = 0 gives NaN components on both sides (0 * inf there,
// 0 * NaN here); the reference divides 0 by 0.
DE_DEV vec3 normalized(vec3 a) { return a * de_rcp_nr(length_nr(a)); }
DE_DEV vec3 cross(vec3 a, vec3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
DE_DEV float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
DE_DEV vec3 mix3(vec3 a, vec3 b, float t) { return v3(mixf(a.x, b.x, t), mixf(a.y, b.y, t), mixf(a.z, b.z, t)); }
DE_DEV float smoothstep_(float e0, float e1, float x) {
    float t = de_clamp((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
DE_DEV float sqr(float x) { return x * x; }
DE_DEV float pow5(float x) { float x2 = x * x; float x4 = x2 * x2; return x4 * x; }

// ------------------------------------------------------------------ RNG: xoroshiro64* stream per (seed, pixel, sample)
struct Rng {
    uint32_t s0, s1;
    uint32_t draws;
};
DE_DEV uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
DE_DEV uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
DE_DEV void rng_seed(Rng& r, uint32_t seed_lo, uint32_t seed_hi, uint32_t pixel, uint32_t sample) {
    uint32_t a = fmix32((pixel * 0x9E3779B1u) ^ seed_lo);
    uint32_t b = fmix32((sample * 0x85EBCA77u) ^ seed_hi ^ a);
    r.s0 = fmix32(a ^ rotl32(b, 16) ^ 0x68E31DA4u);
    r.s1 = fmix32(b + 0xB5297A4Du);
    if ((r.s0 | r.s1) == 0u) r.s0 = 1u;
    r.draws = 0;
}
DE_DEV float rng_next(Rng& r) {
    uint32_t v = r.s0 * 0x9E3779BBu;
    uint32_t t = r.s1 ^ r.s0;
    r.s0 = rotl32(r.s0, 26) ^ t ^ (t << 9);
    r.s1 = rotl32(t, 13);
    r.draws++;
    return (float)(v >> 8) * 0x1p-24f;
}

// ------------------------------------------------------------------ textures
// Single-channel maps are stored "footprint packed": entry (i, j) is one dword holding the four texels of the
// bilinear footprint whose lower-left texel is (i, j) — t(i,j) | t(i+1,j) << 8 | t(i,j+1) << 16 | t(i+1,j+1) << 24 —
// with the address mode (REPEAT / CLAMP) already applied to i+1 and j+1.  One bilinear tap = ONE global_load_dword
// and one cache line.  Entries are tiled 8 x 4 (one 128-byte line per tile) so that neighbouring taps share lines
// in both directions.  4x the bytes of an r8 map (933 MB for 21600 x 10800): HBM capacity is spent to buy
// instruction count and line locality.
struct TexR8 {
    const uint32_t* data;
    int w, h, tiles_x;
    float wf, hf;
};
// colour maps (albedo, stars: one tap per surface hit / per miss) stay linear, one rgbx dword per texel
struct TexRGB {
    const uint32_t* data;
    int w, h;
    float wf, hf;
};
// float LUT textures: CIE (441 x 2 x 3, f16-quantised) and CRF (1024 x n x 3)
struct TexF3 {
    const float* data;
    int w, h;
};


struct TapCount { uint32_t r8, rgb; };

// A dword of a map by index: the pointer is cast to the GLOBAL address space, so that the load is a global_load_dword with the
// (wave-uniform) base in scalar registers and the 32-bit byte offset in one vector register — a generic pointer makes it a flat_load with
// a 64-bit vector address: two more VGPRs live through every tracking loop (at 80 VGPRs that was the difference between the cloud map's
// base staying in registers and being reloaded from scratch on every tap: round 5) and a 64-bit vector add per tap.  Same dword either way.
DE_DEV uint32_t map_dword(const uint32_t* base, uint32_t idx) {
    typedef const __attribute__((address_space(1))) uint32_t* de_gptr;
    typedef const __attribute__((address_space(1))) char* de_gbytes;
    const uint32_t byte_off = idx << 2;      // 32-bit on purpose (every map is below 4 GB: de_api.hip refuses larger ones): the offset fits the instruction's one-VGPR form
    return *(de_gptr)((de_gbytes)base + byte_off);
}

// A tap in two halves, so that the caller can put independent work between the load and its first use.
template <bool CLAMP>
DE_DEV uint32_t tap_r8_issue(const TexR8& t, float u, float v, float* fx_out, float* fy_out) {
    float x = u * t.wf - 0.5f;
    float y = v * t.hf - 0.5f;
    float x0f = de_floor(x), y0f = de_floor(y);
    float fx = x - x0f, fy = y - y0f;
    int i0 = (int)x0f, j0 = (int)y0f;
    if (CLAMP) {
        if (i0 < 0) { i0 = 0; fx = 0.0f; }
        if (j0 < 0) { j0 = 0; fy = 0.0f; }
    } else {
        if (i0 < 0) i0 = t.w - 1;
        if (j0 < 0) j0 = t.h - 1;
    }
    uint32_t idx = ((uint32_t)(j0 >> 2) * (uint32_t)t.tiles_x + (uint32_t)(i0 >> 3)) * 32u + (uint32_t)(((j0 & 3) << 3) | (i0 & 7));
    *fx_out = fx; *fy_out = fy;
    return map_dword(t.data, idx);
}
// The same tap for a caller that samples the map at points a fraction of a texel apart (tracking through the cloud shell: a free
// flight of ~345 m against texels of 1.85 km): the footprint dword of the previous tap is kept, and a tap whose footprint is the
// same entry reuses it — same dword, same result, no load.
template <bool CLAMP>
DE_DEV uint32_t tap_r8_issue_cached(const TexR8& t, float u, float v, float* fx_out, float* fy_out, uint32_t& last_idx, uint32_t& last_q) {
    float x = u * t.wf - 0.5f;
    float y = v * t.hf - 0.5f;
    float x0f = de_floor(x), y0f = de_floor(y);
    float fx = x - x0f, fy = y - y0f;
    int i0 = (int)x0f, j0 = (int)y0f;
    if (CLAMP) {
        if (i0 < 0) { i0 = 0; fx = 0.0f; }
        if (j0 < 0) { j0 = 0; fy = 0.0f; }
    } else {
        if (i0 < 0) i0 = t.w - 1;
        if (j0 < 0) j0 = t.h - 1;
    }
    const uint32_t idx = ((uint32_t)(j0 >> 2) * (uint32_t)t.tiles_x + (uint32_t)(i0 >> 3)) * 32u + (uint32_t)(((j0 & 3) << 3) | (i0 & 7));
    *fx_out = fx; *fy_out = fy;
    if (idx != last_idx) { last_q = map_dword(t.data, idx); last_idx = idx; }
    return last_q;
}
DE_DEV float tap_r8_finish(uint32_t q, float fx, float fy) {
    // contract 2: the bilinear filter runs on the byte values (exact small integers), the result is scaled by RN(1/255)
    float t00 = (float)(q & 255u), t10 = (float)((q >> 8) & 255u), t01 = (float)((q >> 16) & 255u), t11 = (float)(q >> 24);
    float a = de_fma(t10 - t00, fx, t00);
    float b = de_fma(t11 - t01, fx, t01);
    return de_fma(b - a, fy, a) * DE_INV_255;
}
template <bool CLAMP>
DE_DEV float tap_r8(const TexR8& t, float u, float v) {
    float fx, fy;
    uint32_t q = tap_r8_issue<CLAMP>(t, u, v, &fx, &fy);
    return tap_r8_finish(q, fx, fy);
}
DE_DEV int wrap_idx(int i, int n, bool clamp) {
    if (clamp) return i < 0 ? 0 : (i >= n ? n - 1 : i);
    i %= n;
    return i < 0 ? i + n : i;
}
template <bool CLAMP>
DE_DEV vec3 tap_rgb(const TexRGB& t, float u, float v) {
    float x = u * t.wf - 0.5f;
    float y = v * t.hf - 0.5f;
    float x0f = de_floor(x), y0f = de_floor(y);
    float fx = x - x0f, fy = y - y0f;
    int i0 = wrap_idx((int)x0f, t.w, CLAMP), i1 = wrap_idx((int)x0f + 1, t.w, CLAMP);
    int j0 = wrap_idx((int)y0f, t.h, CLAMP), j1 = wrap_idx((int)y0f + 1, t.h, CLAMP);
    uint32_t q00 = t.data[(size_t)j0 * t.w + i0], q10 = t.data[(size_t)j0 * t.w + i1];
    uint32_t q01 = t.data[(size_t)j1 * t.w + i0], q11 = t.data[(size_t)j1 * t.w + i1];
    float o[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        int sh = ch * 8;
        float t00 = (float)((q00 >> sh) & 255u), t10 = (float)((q10 >> sh) & 255u);
        float t01 = (float)((q01 >> sh) & 255u), t11 = (float)((q11 >> sh) & 255u);
        float a = de_fma(t10 - t00, fx, t00);
        float b = de_fma(t11 - t01, fx, t01);
        o[ch] = de_fma(b - a, fy, a) * DE_INV_255;
    }
    return v3(o[0], o[1], o[2]);
}
DE_DEV vec3 tap_f3(const TexF3& t, float u, float v, bool clamp) {
    float x = u * (float)t.w - 0.5f;
    float y = v * (float)t.h - 0.5f;
    float x0f = de_floor(x), y0f = de_floor(y);
    float fx = x - x0f, fy = y - y0f;
    int i0 = wrap_idx((int)x0f, t.w, clamp), i1 = wrap_idx((int)x0f + 1, t.w, clamp);
    int j0 = wrap_idx((int)y0f, t.h, clamp), j1 = wrap_idx((int)y0f + 1, t.h, clamp);
    float o[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float t00 = t.data[((size_t)j0 * t.w + i0) * 3 + ch], t10 = t.data[((size_t)j0 * t.w + i1) * 3 + ch];
        float t01 = t.data[((size_t)j1 * t.w + i0) * 3 + ch], t11 = t.data[((size_t)j1 * t.w + i1) * 3 + ch];
        float a = de_fma(t10 - t00, fx, t00);
        float b = de_fma(t11 - t01, fx, t01);
        o[ch] = de_fma(b - a, fy, a);
    }
    return v3(o[0], o[1], o[2]);
}

// ------------------------------------------------------------------ lib/math_utils.py
DE_DEV float cone_angle_to_solid_angle(float x) { return (float)(M_PI * 2) * (1.0f - de_cos(x)); }   // :13-15
struct vec2_ { float x, y; };
// :17-23 — a miss yields (NaN, NaN): sqrt of a negative discriminant; every caller relies on NaN comparisons being false
DE_DEV vec2_ rsi(vec3 pos, vec3 dir, float r) {
    float b = dot(pos, dir);
    float discr = b * b - dot(pos, pos) + r * r;
    // r*r >= 4e13 has a spacing of 2^22: the sum is 0 or at least 2^21 in magnitude (and below 2^64 for |pos| < 4e9 m), inside
    // the range where the short sqrt is the correctly rounded one; a negative discriminant gives NaN either way
    discr = de_sqrt_nr(discr);
    vec2_ o;
    o.x = -b + -discr;
    o.y = -b + discr;
    if (discr < 0.0f) { o.x = -1.0f; o.y = -1.0f; }
    return o;
}
DE_DEV vec2_ sphere_UV_map(vec3 n) {                                                                   // :25-28
    vec2_ o;
    // contract 2: (atan2 / pi + 1) / 2 and asin / pi + 0.5 as one fused multiply-add each with RN(1/2pi), RN(1/pi)
    o.x = de_fma(de_atan2(n.z, -n.x), DE_INV_2PI, 0.5f);
    o.y = de_fma(de_asin(n.y), DE_INV_PI, 0.5f);
    return o;
}
DE_DEV float fract_(float x) { return x - de_floor(x); }
// sample_sphere_texture (:38-44) for a direction that is ALREADY the normalised position
template <bool CLAMP>
DE_DEV float sphere_tap_r8(const TexR8& t, vec3 n) {
    vec2_ uv = sphere_UV_map(n);
    return tap_r8<CLAMP>(t, fract_(uv.x * 1.0f), fract_(uv.y * 1.0f));
}
template <bool CLAMP>
DE_DEV vec3 sphere_tap_rgb(const TexRGB& t, vec3 n) {
    vec2_ uv = sphere_UV_map(n);
    return tap_rgb<CLAMP>(t, fract_(uv.x * 1.0f), fract_(uv.y * 1.0f));
}
DE_DEV void make_orthonormal_basis(vec3 n, vec3* x, vec3* y) {                                         // :55-60
    vec3 h = (de_abs(n.y) > 0.9f) ? v3(1.0f, 0.0f, 0.0f) : v3(0.0f, 1.0f, 0.0f);
    *y = normalized(cross(n, h));
    *x = cross(n, *y);
}
DE_DEV vec3 tangent_space_apply(vec3 n, vec3 v) {                                                      // :62-65
    vec3 x, y;
    make_orthonormal_basis(n, &x, &y);
    return v3((x.x * v.x + y.x * v.y) + n.x * v.z, (x.y * v.x + y.y * v.y) + n.y * v.z, (x.z * v.x + y.z * v.y) + n.z * v.z);
}
DE_DEV vec3 spherical_direction(float sin_theta, float cos_theta, float phi, vec3 x, vec3 y, vec3 z) { // :67-69
    float s, c;
    de_sincos(phi, &s, &c);
    return (sin_theta * c) * x + (sin_theta * s) * y + cos_theta * z;
}

// ------------------------------------------------------------------ lib/sampling.py
DE_DEV vec3 sample_cone(Rng& rng, float cos_theta_max) {                                               // :13-23
    float u0 = rng_next(rng);
    float u1 = rng_next(rng);
    float cos_theta = (1.0f - u0) + u0 * cos_theta_max;
    float sin_theta = de_sqrt_nr(1.0f - cos_theta * cos_theta);      // 0 or in [2^-25, 1]
    float phi = (float)(2.0 * M_PI) * u1;
    float s, c;
    de_sincos(phi, &s, &c);
    return v3(sin_theta * c, sin_theta * s, cos_theta);
}
DE_DEV vec3 sample_hemisphere_cosine_weighted(Rng& rng, vec3 n) {                                      // :30-39
    float u0 = rng_next(rng);
    float u1 = rng_next(rng);
    float a = 1.0f - 2.0f * u0;
    float b = de_sqrt(1.0f - a * a);
    a *= (float)(1.0 - 1e-5);
    b *= (float)(1.0 - 1e-5);
    float phi = (float)(2.0 * M_PI) * u1;
    float s, c;
    de_sincos(phi, &s, &c);
    return normalized(v3(n.x + b * c, n.y + b * s, n.z + a));
}
DE_DEV vec3 sample_sphere(float r0, float r1) {                                                        // :41-45
    float rx = r0 * (float)(M_PI * 2.0);
    float ry = r1 * 2.0f - 1.0f;
    float s, c;
    de_sincos(rx, &s, &c);
    float k = de_sqrt(1.0f - ry * ry);
    return normalized(v3(s * k, c * k, ry));
}

// ------------------------------------------------------------------ lib/volume_rendering_models.py
#define DE_PLANET_R 6371e3f
#define DE_ATMOS_UPPER ((float)(6371e3 + 110e3))
#define DE_CLOUDS_LOWER ((float)(6371e3 + 4000.0))
#define DE_CLOUDS_UPPER ((float)(6371e3 + 4000.0 + 6000.0))
#define DE_CLOUDS_THICKNESS 6000.0f
#define DE_CLOUDS_EXTINCT 0.1f
#define DE_CLOUDS_DENSITY 0.029f
#define DE_MIE_ASYMMETRY 3000.0f
#define DE_OZONE_PEAK_HEIGHT 25000.0f
enum { RAYLEIGH_ID = 0, MIE_ID = 1, OZONE_ID = 2, CLOUD_ID = 3, ISOTROPIC_CLOUD_ID = 4 };

DE_DEV float air_index(float wavelength) {                                                             // :48-51
    float rcp = 1.0f / (wavelength * wavelength);
    return (float)(1.0 + 8.06051e-5) + 2.480990e-2f / (132.274f - rcp) + 1.74557e-4f / (39.32957f - rcp);
}
DE_DEV float rayleigh_phase(float c) { return (float)(3.0 / (16.0 * M_PI)) * (1.0f + c * c); }         // :61-63
DE_DEV float hg_phase(float c, float g) {                                                              // :73-76
    float d = 1.0f + g * g - 2.0f * g * c;
    return (1.0f - g * g) / ((float)(4.0 * M_PI) * (d * de_sqrt(d)));
}
DE_DEV vec3 sample_hg_phase(Rng& rng, vec3 view, float g) {                                            // :78-85
    float sqr_term = (1.0f - g * g) / (1.0f - g + 2.0f * g * rng_next(rng));
    float cos_theta = (1.0f + g * g - sqr_term * sqr_term) / (2.0f * g);
    float sin_theta = de_sqrt(de_max(0.0f, 1.0f - cos_theta * cos_theta));
    float phi = (float)(2.0 * M_PI) * rng_next(rng);
    vec3 tang, bitang;
    make_orthonormal_basis(view, &tang, &bitang);
    return spherical_direction(sin_theta, cos_theta, phi, tang, bitang, view);
}
DE_DEV float klein_nishina_phase(float c, float e, float log_2e1) {                                    // :87-89
    return e / ((float)(2.0 * M_PI) * (e * (1.0f - c) + 1.0f) * log_2e1);
}
DE_DEV vec3 sample_klein_nishina_phase(Rng& rng, vec3 view, float e) {                                 // :91-98
    float cos_theta = (-de_pow(2.0f * e + 1.0f, 1.0f - rng_next(rng)) + e + 1.0f) / e;
    float sin_theta = de_sqrt(de_max(0.0f, 1.0f - cos_theta * cos_theta));
    float phi = (float)(2.0 * M_PI) * rng_next(rng);
    vec3 tang, bitang;
    make_orthonormal_basis(view, &tang, &bitang);
    return spherical_direction(sin_theta, cos_theta, phi, tang, bitang, view);
}
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
DE_DEV float draine_phase(float c, float g, float a) {                                                 // :121-123
    float d = 1.0f + g * g - 2.0f * g * c;
    return ((1.0f - g * g) * (1.0f + a * c * c)) / (4.0f * (1.0f + (a * (1.0f + 2.0f * g * g)) / 3.0f) * (float)M_PI * (d * de_sqrt(d)));
}
DE_DEV vec3 sample_draine(Rng& rng, vec3 view, float g, float a) {                                     // :125-152
    float xi = rng_next(rng);
    float g2 = g * g, g3 = g * g2, g4 = g2 * g2, g6 = g2 * g4;
    float pgp1_2 = (1.0f + g2) * (1.0f + g2);
    float T1a = -a + a * g4;
    float T1a3 = T1a * T1a * T1a;
    float T2 = -1296.0f * (-1.0f + g2) * (a - a * g2) * (T1a) * (4.0f * g2 + a * pgp1_2);
    float T3 = 3.0f * g2 * (1.0f + g * (-1.0f + 2.0f * xi)) + a * (2.0f + g2 + g3 * (1.0f + 2.0f * g2) * (-1.0f + 2.0f * xi));
    float T4a = 432.0f * T1a3 + T2 + 432.0f * (a - a * g2) * T3 * T3;
    float T4b = -144.0f * a * g2 + 288.0f * a * g4 - 144.0f * a * g6;
    float T4b3 = T4b * T4b * T4b;
    float T4 = T4a + de_sqrt(-4.0f * T4b3 + T4a * T4a);
    float T4p3 = de_pow(T4, (float)(1.0 / 3.0));
    const float cbrt2 = (float)1.2599210498948732;
    float T6 = (2.0f * T1a + (48.0f * cbrt2 * (-(a * g2) + 2.0f * a * g4 - a * g6)) / T4p3 + T4p3 / (3.0f * cbrt2)) / (a - a * g2);
    float T5 = 6.0f * (1.0f + g2) + T6;
    float inner = -0.5f * de_sqrt(T5) + de_sqrt(6.0f * (1.0f + g2) - (8.0f * T3) / (a * (-1.0f + g2) * de_sqrt(T5)) - T6) / 2.0f;
    float cos_theta = (1.0f + g2 - inner * inner) / (2.0f * g);
    float sin_theta = de_sqrt(de_max(0.0f, 1.0f - cos_theta * cos_theta));
    float phi = (float)(2.0 * M_PI) * rng_next(rng);
    vec3 tang, bitang;
    make_orthonormal_basis(view, &tang, &bitang);
    return spherical_direction(sin_theta, cos_theta, phi, tang, bitang, view);
}
struct CloudPhaseConsts { float g_hg, g_draine, alpha_draine, w_draine; };
DE_DEV CloudPhaseConsts cloud_consts() {                                                               // :156-162
    const double d = 8.0;
    CloudPhaseConsts k;
    k.g_hg = de_exp((float)(-0.0990567 / (d - 1.67154)));
    k.g_draine = de_exp((float)(-2.20679 / (d + 3.91029) - 0.428934));
    k.alpha_draine = de_exp((float)(3.62489 - 8.29288 / (d + 5.52825)));
    k.w_draine = de_exp((float)(-0.599085 / (d - 0.641583) - 0.665888));
    return k;
}
DE_DEV float cloud_phase(const CloudPhaseConsts& k, float c, bool reduce_peak) {                       // :156-164
    float g_hg = reduce_peak ? 0.91f : k.g_hg;
    return mixf(hg_phase(c, g_hg), draine_phase(c, k.g_draine, k.alpha_draine), k.w_draine);
}
DE_DEV vec3 sample_cloud_phase(const CloudPhaseConsts& k, Rng& rng, vec3 view, bool reduce_peak) {      // :166-183
    float g_hg = reduce_peak ? 0.91f : k.g_hg;
    if (rng_next(rng) < k.w_draine) return sample_draine(rng, view, k.g_draine, k.alpha_draine);
    return sample_hg_phase(rng, view, g_hg);
}
DE_DEV float spectra_extinction_mie(float wavelength) {                                                // :194-200
    const double c = (0.6544 * 1.06 - 0.6510) * 4e-18;
    float K = (0.773335f - 0.00386891f * wavelength) / (1.0f - 0.00546759f * wavelength);
    float x = (float)(2.0 * M_PI) / (wavelength * 1e-9f);
    return (float)(0.434 * c * M_PI) * (x * x) * K;
}
DE_DEV float spectra_extinction_rayleigh(float wavelength) {                                           // :203-217
    float wl_nm = wavelength * 1e-9f;
    float w2 = wavelength * wavelength;
    float F_N2 = 1.034f + 3.17e-4f * (1.0f / w2);
    float F_O2 = 1.096f + 1.385e-3f * (1.0f / w2) + 1.448e-4f * (1.0f / (w2 * w2));
    float king = (78.084f * F_N2 + 20.946f * F_O2 + 0.934f + (float)(0.0421 * 1.15)) / (float)(78.084 + 20.946 + 0.934 + 0.0421);
    float n = sqr(air_index(wavelength * 1e-3f)) - 1.0f;
    const float pi3 = (float)(M_PI * M_PI * M_PI);
    float wl2 = wl_nm * wl_nm;
    return ((8.0f * pi3 * (n * n)) / ((float)(3.0 * 2.5035422e25) * (wl2 * wl2))) * king;
}
DE_DEV float spectra_extinction_ozone(float wavelength, const float* o3_lut) {                         // :219-224
    float ext = 0.0f;
    if (wavelength >= 390.0f && wavelength < 831.0f)
        ext = (float)(0.0001 * (2.5035422e25 * 0.012588 * 8e-6)) * o3_lut[(int)(wavelength - 390.0f)];
    return ext;
}
// Densities are evaluated for 0 <= h <= ~120 km (inside the atmosphere segment), h a multiple of 0.5 m (|pos| ~ 6.4e6):
// every constant division below has a numerator that is 0 or far above 2^-100, so the unguarded 3-operation form applies.
DE_DEV float get_ozone_density(float h) {                                                              // :229-247
    float h_km = h * 0.001f;
    float rel = h_km - (float)(25000.0 * 0.001);
    rel = rel * rel;
    float d = (float)(1.0 - 0.375) * de_exp_nonpos(DE_DIVC_NG(-rel, 49.0f));
    d += 0.375f * de_exp_nonpos(-rel / 256.0f);
    float t = h_km - 15.0f;
    d += de_max(0.0f, -0.000015f * (t * t * t));
    return d;
}
DE_DEV float get_rayl_density(float h) {                                                               // :249-253
    float t = h + 24239.99f;
    return DE_DIVC_NG(3.68082f * de_exp_nonpos(DE_DIVC_NG(-(t * t), 532307548.4168f)), 1.225f);
}
// :255-268.  The reference's four-way branch, with the three exponential branches folded into ONE exp call on
// per-lane selected (offset, scale, coefficient, addend): a wave whose lanes sit in different altitude bands would
// otherwise execute all three.  Same operations per lane: 0.0918*e + 0 == 0.0918*e, x - 0.092 == x + (-0.092).
DE_DEV float get_mie_density(float h) {
    const bool bA = h > 11500.0f, bB = h > 2400.0f, bC = h > 1300.0f;
    const float off = bA ? -11500.0f : (bB ? 2500.0f : -1300.0f);
    const float k = bA ? -1.0e-6f : (bB ? -2.5e-9f : -5.0e-6f);
    const float coef = bA ? 0.0918f : (bB ? 0.3000f : 0.6500f);
    const float add = bA ? 0.0f : (bB ? -0.092f : 0.18899f);
    float t = h + off;
    float e = de_exp_nonpos(k * (t * t));
    float dens = coef * e + add;
    float lin = 1.0f - DE_DIVC_NG(h, 8136.646f);
    dens = bC ? dens : lin;
    return dens * (float)1.06;
}
DE_DEV vec3 get_density(float h) {                                                                     // :271-274
    h = de_max(h, 0.0f);
    return v3(get_rayl_density(h), get_mie_density(h), get_ozone_density(h));
}

// One read of the altitude table of get_density (de_kernels.h: DE_DENS_TABLE_N entries of DE_DENS_STRIDE floats at h = i / 2 m).
// DE_DENS_NT: with a non-temporal hint, so that the 3 MB table does not compete with the map taps for L2 lines (A/B experiment).
DE_DEV vec3 dens_table_read(const float* table, uint32_t i) {
#if DE_DENS_STRIDE == 4
    typedef float de_f4v __attribute__((ext_vector_type(4)));
    const de_f4v* e = reinterpret_cast<const de_f4v*>(table) + i;
#ifdef DE_DENS_NT
    const de_f4v q = __builtin_nontemporal_load(e);
#else
    const de_f4v q = *e;
#endif
    return v3(q.x, q.y, q.z);
#else
    const float* dt = table + (size_t)i * 3u;
#ifdef DE_DENS_NT
    return v3(__builtin_nontemporal_load(dt), __builtin_nontemporal_load(dt + 1), __builtin_nontemporal_load(dt + 2));
#else
    return v3(dt[0], dt[1], dt[2]);
#endif
#endif
}

// ------------------------------------------------------------------ lib/surface_rendering_models.py
DE_DEV float disney_diffuse(float roughness, float ndl, float ndv, float ldh) {                        // :40-52
    float R_R = 2.0f * roughness * sqr(ldh);
    float F_L = pow5(1.0f - ndl);
    float F_V = pow5(1.0f - ndv);
    const float f_lambert = (float)(1.0 / M_PI);
    float f_retro = f_lambert * R_R * (F_L + F_V + F_L * F_V * (R_R - 1.0f));
    return f_lambert * (1.0f - 0.5f * F_L) * (1.0f - 0.5f * F_V) + f_retro;
}
DE_DEV float fresnel_dielectric(float vdh, float F_0) {                                                // :110-122
    F_0 = de_sqrt(F_0);
    F_0 = (1.0f + F_0) / (1.0f - F_0);
    float sin_i = de_sqrt(de_saturate(1.0f - sqr(vdh)));
    float sin_t = sin_i / de_max(F_0, 1e-8f);
    float cos_t = de_sqrt(1.0f - sqr(sin_t));
    float R_s = sqr((vdh - (F_0 * cos_t)) / de_max(vdh + (F_0 * cos_t), 1e-8f));
    float R_p = sqr((cos_t - (F_0 * vdh)) / de_max(cos_t + (F_0 * vdh), 1e-8f));
    return de_saturate((R_s + R_p) * 0.5f);
}
DE_DEV float GGX_D(float ndh, float alpha2) {                                                          // :82-85
    float den = (alpha2 - 1.0f) * ndh * ndh + 1.0f;
    return alpha2 / ((float)M_PI * den * den);
}
DE_DEV float lambda_smith(float ndx, float alpha2) {                                                   // :88-91
    float n2 = ndx * ndx;
    return (-1.0f + de_sqrt(alpha2 * (1.0f - n2) / n2 + 1.0f)) * 0.5f;
}
DE_DEV float G2_smith(float ndl, float ndv, float alpha2) {                                            // :100-104
    float lv = lambda_smith(ndv, alpha2);
    float ll = lambda_smith(ndl, alpha2);
    return 1.0f / (1.0f + lv + ll);
}
DE_DEV float GGX_smith_specular(float roughness, float F_0, float ndl, float ndv, float ldh, float ndh) {  // :70-80
    float alpha2 = roughness * roughness;
    float D = GGX_D(ndh, alpha2);
    float G = G2_smith(ndl, ndv, alpha2);
    float F = fresnel_dielectric(ldh, F_0);
    return D * G * F / de_max(4.0f * ndl * ndv, 1e-5f);
}
DE_DEV float beckmann_isotropic_ndf(float ndh, float alpha) {                                          // :146-152
    float c2 = ndh * ndh;
    float a2 = alpha * alpha;
    float exponent = (1.0f - c2) / (a2 * c2);
    float denom = (float)M_PI * a2 * c2 * c2;
    return de_exp(-exponent) / de_max(denom, 1e-5f);
}
DE_DEV float G2_VCavity(float ndl, float ndv, float ndh, float vdh) {                                  // :169-171
    return de_min(1.0f, de_min(2.0f * ndv * ndh / vdh, 2.0f * ndl * ndh / vdh));
}
DE_DEV float beckmann_specular(float roughness, float F_0, float ndl, float ndv, float ldh, float ndh) {   // :54-68
    float alpha = roughness;
    alpha *= alpha * 2.0f;
    float D = beckmann_isotropic_ndf(ndh, alpha);
    float V = G2_VCavity(ndl, ndv, ndh, ldh);
    float F = fresnel_dielectric(ldh, F_0);
    return D * V * F;
}
DE_DEV float earth_brdf(float albedo, float oceanness, float bathymetry, vec3 v, vec3 n, vec3 l, float* n_dot_l_out) {  // :9-38
    vec3 h = normalized(v + l);
    float ndl = de_saturate(dot(n, l));
    float ndv = de_saturate(dot(n, v));
    float ldh = de_saturate(dot(l, h));
    float ndh = de_saturate(dot(n, h));
    const float land_roughness = 0.73f;
    float ocean_roughness = mixf((float)(0.23 + 0.02), (float)(0.23 - 0.04), smoothstep_(0.3f, 0.7f, bathymetry));
    const float land_F_0 = 0.04f, ocean_F_0 = 0.02f;
    float diffuse = disney_diffuse(land_roughness, ndl, ndv, ldh);
    float land_spec = GGX_smith_specular(land_roughness, land_F_0, ndl, ndv, ldh, ndh);
    float ocean_ggx = GGX_smith_specular(ocean_roughness, ocean_F_0, ndl, ndv, ldh, ndh);
    float ocean_beck = 0.65f * beckmann_specular(ocean_roughness, ocean_F_0, ndl, ndv, ldh, ndh);
    float ocean_spec = mixf(ocean_beck, ocean_ggx, de_clamp(smoothstep_(0.2f, 0.95f, ndv), 0.05f, 0.94f));
    float blender = smoothstep_(0.6f, 1.0f, oceanness);
    float brdf = albedo * diffuse * 0.28f + mixf(land_spec, ocean_spec, blender) * 0.5f;
    *n_dot_l_out = ndl;
    return brdf;
}

// ------------------------------------------------------------------ lib/colour.py
DE_DEV vec3 xyz_to_rgb_d65(vec3 v) {                                                                   // :6-10
    return v3((3.2409699419f * v.x + -1.5373831776f * v.y) + -0.4986107603f * v.z,
              (-0.9692436363f * v.x + 1.8759675015f * v.y) + 0.0415550574f * v.z,
              (0.0556300797f * v.x + -0.2039769589f * v.y) + 1.0569715142f * v.z);
}
DE_DEV float plancks(float temperature, float wavelength) {                                            // :51-60
    const float h = 6.62607015e-16f, c = 2.9e17f, k = 1.38e-5f;
    float w2 = wavelength * wavelength;
    float p1 = 2.0f * h * (c * c) / (w2 * w2 * wavelength);
    float p2 = de_exp((float)(6.62607015e-16 * 2.9e17) / (wavelength * k * temperature)) - 1.0f;
    return p1 / p2;
}
DE_DEV float srgb_transfer1(float linear) {                                                            // :74-79
    float lo = linear * 12.92f;
    float hi = (de_pow(de_abs(linear), (float)(1.0 / 2.4)) * 1.055f) - 0.055f;
    float t = (0.0031308f < linear) ? 0.0f : 1.0f;
    return mixf(hi, lo, t);
}
DE_DEV float lum(vec3 x) { return dot(x, v3(0.2126729f, 0.7151522f, 0.0721750f)); }                   // :88-90
DE_DEV vec3 lum3(vec3 x) { float y = lum(x); return v3(y, y, y); }                                     // :92-95
