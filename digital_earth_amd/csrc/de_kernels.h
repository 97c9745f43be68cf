// de_kernels.h — device structures shared by the kernels and the host API (de_api.hip).
#pragma once
#include "de_device.h"
#include "../../include/digital_earth_debug.h"      // includes digital_earth.h: the library defines every entry point of both headers

// Per-frame constants: everything the reference recomputes per pixel (or per launch) from scalar state that is
// uniform over the image.  Filled ON THE DEVICE by setup_kernel with the contract arithmetic, so the values are the
// ones a per-pixel evaluation would produce (renderer.py:270-302, pathtracer.py:332-337, lib/OpenDRT.py:257-319).
struct FrameConsts {
    vec3 cam_pos, d, du, dv;                 // get_cast_dir basis            renderer.py:272,276-277
    float fov, aspect_ratio, aspect_scale;   //                               renderer.py:271-275
    vec3 light_dir;                          //                               renderer.py:301-302
    vec3 sun_x, sun_y;                       // make_orthonormal_basis(light_dir), lib/math_utils.py:55-60 (same for every light sample)
    float sun_cos_angle, sun_angular_radius; //                               renderer.py:297-300
    float land_height_scale;
    float normal_eps;                        // pi * planet_r / TOPO_W        pathtracer.py:20
    float offset_scale;                      // 1 + 1e-4*scale/12000          pathtracer.py:421
    float kn_log;                            // log(2*3000 + 1)               lib/volume_rendering_models.py:89
    float max_dens_r, max_dens_m, max_dens_o;//                               pathtracer.py:336
    CloudPhaseConsts cloud;
    // display
    float exposure_scale;                    // 2^exposure                    renderer.py:355
    float gamma;
    float crf_slice_v;                       //                               renderer.py:338
    float vig_strength, vig_radius, vig_cx, vig_cy;
    float drt_m, drt_s, drt_fl, drt_ds, drt_clamp_max, drt_dch_s;
    vec3 drt_w;                              // normalised (rw, 1, bw)        lib/OpenDRT.py:369-370
    // AgX alternative display transform (lib/AgX.py; DE_FLAG_AGX): matrices and curve scales, host double -> f32
    float agx_srgb_to_xyz[9], agx_xyz_to_adjusted[9];
    float agx_x_pivot, agx_y_pivot, agx_toe_scale, agx_shoulder_scale;
    int use_agx;
};

// One node of spectrum_sample's bisection tree (lib/colour.py:24-39): 8 steps reach one of 256 leaves; an exact
// tie stops early at an inner node.  Heap numbering: root 1, children 2n / 2n+1; index 0 = the fixed-wavelength
// entry of BASELINE cfg1.  Everything path_tracer derives from the wavelength alone (pathtracer.py:332-343) is here.
struct LambdaNode {
    float wavelength;
    float rx, ry, rz;        // CMF response                     lib/colour.py:40
    float rcp_pdf;           //                                  lib/colour.py:42-46
    float ext_r, ext_m, ext_o;
    float sun_power, night_power, sun_irradiance;
    float max_ext_rmo;       //                                  pathtracer.py:355
    float c0, c1, c2;        // srgb_to_spectrum coefficients mix(lut[w], lut[w+1], f)   lib/colour.py:64-70
    float s2s_valid;         // 1 if 0 < w < 299 else 0
    float inv_max_ext_rmo;   // RN(1 / max_ext_rmo): tracking divides by the majorant as a product with this (contract 2)
    float pad0, pad1, pad2;
};
#define DE_N_NODES 512
// Altitude table of get_density: entry i = (Rayleigh, Mie, ozone) density at h = i / 2 metres, for every f32-representable
// altitude of a point with |pos| in [2^22, 2^23) below 131 072 m (3 MB).
#ifndef DE_DENS_TABLE_N
#define DE_DENS_TABLE_N 262144
#endif
// floats per entry: 3 (12 bytes: 9 % of the reads straddle two 128-byte lines) or 4 (16 bytes, aligned) — A/B in profiles/experiments
#ifndef DE_DENS_STRIDE
#define DE_DENS_STRIDE 3
#endif

struct RenderArgs {
    const FrameConsts* fc;
    const LambdaNode* nodes;
    const float* node_val;        // [256] CDF value tested at inner node n (lib/colour.py:28)
    const float* dens_table;      // [DE_DENS_TABLE_N][3]
    TexR8 topo, ocean, clouds, bathy, emissive;
    TexRGB albedo, stars;
    float* hdr;                   // [H][W][3]
    int W, H;
    const uint32_t* tiles;        // owned 8x8 tiles (tile id = ty * tiles_x + tx)
    int n_tiles, tiles_x;
    int tiles_identity;           // tiles[k] == k (tile_world == 1)
    uint32_t seed_lo, seed_hi;
    int spp_begin, spp_count;     // this launch renders sample indices spp_begin + k * spp_stride, k = 0 .. spp_count - 1
    int spp_stride;               // 1, or the world size of a sample partition (de_set_sample_partition: renderer.py:371-380 dealt round robin)
    int fixed_wavelength;
    unsigned long long* counters; // [8] when counting
    float* debug_out;             // [H][W][4] when tracing single samples
    uint32_t* work_counter;       // v2: device-wide work-item counter (zeroed before each launch)
    unsigned long long spp_magic; // v2: floor(2^40 / spp_count) + 1
    uint2* contrib;               // v2: per-sample (radiance bits, wavelength node), index = work item
    int pend_thr;                 // v2: lanes that must wait before the transition pass runs
    int heavy_thr;                // v2: stashed paths of one kind that trigger a batch (env DE_V2_A)
    int chunk;                    // v2: work items per counter fetch (env DE_V2_CHUNK)
    int gas_thr;                  // v2: lanes waiting in the GAS class that make its body run (env DE_V2_G)
    int tune_b;                   // v2: below this many iterating lanes everything that waits is processed (env DE_V2_B)
    // v2 as the FINISHER of the wavefront pipeline (render_kernel_v3.hip): work items are not fresh samples but the path rays the
    // pipeline queued for its next vertex round — 64-byte records (resume_rec, *resume_count of them, holes included) plus the
    // paths' cold records (throughput, radiance so far).  nullptr = start from (pixel, sample) as usual.
    const uint4* resume_rec;
    const uint32_t* resume_count;
    const float* resume_cold;     // 16 floats per path: thr, Ls, ...
    uint32_t resume_capacity;
};

struct SetupArgs {
    de_params p;
    int W, H;
    int topo_w;
    int n_crf;
    TexF3 cie;
    const float* srgb2spec;   // 300 x 3
    const float* o3;          // 441
    float drt[6];             // OpenDRT m, s, fl, ds, clamp_max, dch_s (host, double -> f32)
    float agx[22];            // AgX: srgb_to_xyz[9], xyz_to_adjusted[9], x_pivot, y_pivot, toe_scale, shoulder_scale
    FrameConsts* fc;
    LambdaNode* nodes;
    float* node_val;
    int do_nodes;             // 0: frame constants only (camera / sun / display change)
};

struct DisplayArgs {
    const FrameConsts* fc;
    const float* hdr;   // [H][W][3]
    float* image;       // (W, H, 3)
    TexF3 crf;
    int W, H, samples, clamp;
};
