// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// oracle_synth.h — CPU statement of the synthetic stand-in Earth maps.  The NASA maps the reference samples
// (lib/textures.py:10-27) are not shipped with it (textures/ holds only a placeholder), so the bench
// and the tests use procedural maps of the reference's resolutions.  The product generates them on the GPU
// (digital_earth_amd/csrc/synth_textures.hip); this file is the independent CPU statement of the same
// specification, used by tests to check the GPU generator byte for byte.
//
// Specification (all f32, same primitive set as oracle_math.h):
//   texel (i, j) of a W x H map -> u = (i+.5)/W, v = (j+.5)/H -> unit direction n with
//   sphere_UV_map(n) == (u, v)  (lib/math_utils.py:25-28):  phi = pi*(2u-1), lat = pi*(v-.5),
//   n = (-cos(lat)cos(phi), sin(lat), cos(lat)sin(phi)).
//   vnoise(p, seed): trilinear value noise on the integer lattice with smoothstep weights, lattice values
//   (fmix32(ix*0x8DA6B343 ^ iy*0xD8163841 ^ iz*0xCB1AB31F ^ seed) >> 8) * 2^-24.
//   fbm(p, seed, oct) = sum_o 0.5^o * (vnoise(p * 2^o + 17.0*o, seed + o) - 0.5)
//   elevation  e = fbm(1.7 n, seed, 8) - 0.04          (about 65-70 % below 0 = ocean)
//   topography = min(1, 2.5 e)^2 for e > 0 else 0
//   ocean mask = clamp(0.5 - 40 e, 0, 1);  bathymetry = clamp(-3 e, 0, 1)
//   clouds     = clamp(gain * (0.5 + fbm(3 n, seed ^ 0xC10D, 6)) - 0.3 * gain - bias, 0, 1)^2, gain 1.6 (2.2 cloud-heavy)
//   albedo     = ocean / vegetation / desert / ice ramp by |lat|, e and a moisture fbm
//   emissive   = 1 % of land texels lit (per-texel hash), stars = 0.02 % of texels lit
//   byte = (int)(value * 255 + 0.5)
#pragma once
#include "oracle_math.h"

namespace deo {
namespace synth {

static inline uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
static inline float lattice(int ix, int iy, int iz, uint32_t seed) {
    uint32_t h = fmix32(((uint32_t)ix * 0x8DA6B343u) ^ ((uint32_t)iy * 0xD8163841u) ^ ((uint32_t)iz * 0xCB1AB31Fu) ^ seed);
    return (float)(h >> 8) * 0x1p-24f;
}
static inline float lerp_(float a, float b, float t) { return fma_(b - a, t, a); }
static inline float vnoise(float px, float py, float pz, uint32_t seed) {
    float fx0 = floorf(px), fy0 = floorf(py), fz0 = floorf(pz);
    int ix = (int)fx0, iy = (int)fy0, iz = (int)fz0;
    float fx = px - fx0, fy = py - fy0, fz = pz - fz0;
    float wx = fx * fx * (3.0f - 2.0f * fx), wy = fy * fy * (3.0f - 2.0f * fy), wz = fz * fz * (3.0f - 2.0f * fz);
    float c000 = lattice(ix, iy, iz, seed), c100 = lattice(ix + 1, iy, iz, seed);
    float c010 = lattice(ix, iy + 1, iz, seed), c110 = lattice(ix + 1, iy + 1, iz, seed);
    float c001 = lattice(ix, iy, iz + 1, seed), c101 = lattice(ix + 1, iy, iz + 1, seed);
    float c011 = lattice(ix, iy + 1, iz + 1, seed), c111 = lattice(ix + 1, iy + 1, iz + 1, seed);
    float x00 = lerp_(c000, c100, wx), x10 = lerp_(c010, c110, wx);
    float x01 = lerp_(c001, c101, wx), x11 = lerp_(c011, c111, wx);
    float y0 = lerp_(x00, x10, wy), y1 = lerp_(x01, x11, wy);
    return lerp_(y0, y1, wz);
}
static inline float fbm(float x, float y, float z, float base, uint32_t seed, int octaves) {
    float sum = 0.0f, amp = 1.0f, freq = base;
    for (int o = 0; o < octaves; ++o) {
        float off = 17.0f * (float)o;
        sum = fma_(amp, vnoise(fma_(x, freq, off), fma_(y, freq, off), fma_(z, freq, off), seed + (uint32_t)o) - 0.5f, sum);
        amp *= 0.5f;
        freq *= 2.0f;
    }
    return sum;
}
static inline uint8_t to_byte(float v) {
    v = clamp_(v, 0.0f, 1.0f);
    return (uint8_t)(int)fma_(v, 255.0f, 0.5f);
}
static inline void direction(int i, int j, int W, int H, float* nx, float* ny, float* nz, float* lat_out) {
    float u = ((float)i + 0.5f) / (float)W;
    float v = ((float)j + 0.5f) / (float)H;
    float phi = OM_PI * fma_(2.0f, u, -1.0f);
    float lat = OM_PI * (v - 0.5f);
    float sp, cp, sl, cl;
    sincos_(phi, &sp, &cp);
    sincos_(lat, &sl, &cl);
    *nx = -(cl * cp); *ny = sl; *nz = cl * sp; *lat_out = lat;
}
static inline float elevation(float nx, float ny, float nz, uint32_t seed) {
    return fbm(nx, ny, nz, 1.7f, seed, 8) - 0.04f;
}
// slot numbering = include/digital_earth.h DE_TEX_*; out receives `channels` bytes
static inline void texel(int slot, int i, int j, int W, int H, uint32_t seed, int variant, uint8_t* out) {
    float nx, ny, nz, lat;
    direction(i, j, W, H, &nx, &ny, &nz, &lat);
    uint32_t th = fmix32(((uint32_t)i * 0x9E3779B1u) ^ fmix32((uint32_t)j + 0x7F4A7C15u) ^ (seed * 0x85EBCA77u) ^ (uint32_t)slot);
    float trand = (float)(th >> 8) * 0x1p-24f;
    if (slot == 6) {  // stars
        uint8_t b = 0;
        if (trand < 0.0002f) b = (uint8_t)(128u + ((th >> 1) & 127u));
        out[0] = b; out[1] = b; out[2] = (uint8_t)(b ? (b - (th & 31u)) : 0);
        return;
    }
    if (slot == 3) {  // clouds
        float gain = (variant == 1) ? 2.2f : 1.6f;
        float bias = (variant == 1) ? -0.25f : 0.0f;
        float c = 0.5f + fbm(nx, ny, nz, 3.0f, seed ^ 0xC10Du, 6);
        float t = clamp_(fma_(gain, c, -0.3f * gain - 0.3f) - bias, 0.0f, 1.0f);
        out[0] = to_byte(t * t);
        return;
    }
    float e = elevation(nx, ny, nz, seed);
    float topo = 0.0f;
    if (e > 0.0f) { float t = min_(1.0f, 2.5f * e); topo = t * t; }
    if (slot == 1) { out[0] = to_byte(topo); return; }
    if (slot == 2) { out[0] = to_byte(clamp_(fma_(-40.0f, e, 0.5f), 0.0f, 1.0f)); return; }
    if (slot == 4) { out[0] = to_byte(clamp_(-3.0f * e, 0.0f, 1.0f)); return; }
    float alat = fabsf(lat) * (2.0f / OM_PI);  // 0 equator .. 1 pole
    float ice = clamp_((alat + 0.35f * topo - 0.80f) * 12.0f, 0.0f, 1.0f);
    if (slot == 5) {  // emissive: sparse city lights on ice-free land
        float lit = (e > 0.002f && ice < 0.5f && trand < 0.01f) ? fma_(0.75f, (float)((th >> 3) & 255u) * (1.0f / 255.0f), 0.25f) : 0.0f;
        out[0] = to_byte(lit);
        return;
    }
    // slot 0: albedo
    float moist = 0.5f + fbm(nx, ny, nz, 2.3f, seed ^ 0xA1BEu, 4);
    float dry = clamp_((0.55f - moist) * 4.0f + (0.35f - fabsf(alat - 0.28f)) * 2.0f, 0.0f, 1.0f);
    float r = lerp_(40.0f, 170.0f, dry), g = lerp_(70.0f, 140.0f, dry), b = lerp_(30.0f, 100.0f, dry);
    float rocky = clamp_(topo * 2.0f, 0.0f, 1.0f);
    r = lerp_(r, 120.0f, rocky * 0.5f); g = lerp_(g, 110.0f, rocky * 0.5f); b = lerp_(b, 100.0f, rocky * 0.5f);
    float sea = clamp_(fma_(-40.0f, e, 0.5f), 0.0f, 1.0f);
    float depth = clamp_(-3.0f * e, 0.0f, 1.0f);
    float sr = lerp_(30.0f, 10.0f, depth), sg = lerp_(70.0f, 25.0f, depth), sb = lerp_(90.0f, 60.0f, depth);
    r = lerp_(r, sr, sea); g = lerp_(g, sg, sea); b = lerp_(b, sb, sea);
    r = lerp_(r, 240.0f, ice); g = lerp_(g, 240.0f, ice); b = lerp_(b, 245.0f, ice);
    out[0] = to_byte(r * (1.0f / 255.0f)); out[1] = to_byte(g * (1.0f / 255.0f)); out[2] = to_byte(b * (1.0f / 255.0f));
}

}  // namespace synth
}  // namespace deo
