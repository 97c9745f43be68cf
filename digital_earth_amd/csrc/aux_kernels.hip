// aux_kernels.hip — everything around the hot kernel:
//   dens_table_kernel   altitude table of the gas density profiles
//   setup_kernel        per-frame constants + the 512-node wavelength table (lib/colour.py:13-48, pathtracer.py:332-343)
//   display_kernel      Renderer._render_to_image + camera_response + openDR_transform (renderer.py:333-365,
//                       lib/OpenDRT.py:325-473) + srgb_transfer (lib/colour.py:74-79)
//   pack / unpack       linear u8 maps <-> footprint-packed tiles (the upload kernels of renderer.py:171-210)
//   synth_kernel        procedural stand-in Earth maps (the reference ships none)
//   math_kernel         de_debug_math
#include "de_kernels.h"

// ------------------------------------------------------------------------------------------------ setup
__global__ void setup_kernel(SetupArgs s) {
    const int n = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const bool clamp = (s.p.flags & DE_FLAG_CLAMP_SAMPLER) != 0;
    if (n == 0) {
        FrameConsts fc;
        const de_params& p = s.p;
        fc.cam_pos = v3(p.camera_pos[0], p.camera_pos[1], p.camera_pos[2]);
        vec3 look_at = v3(p.look_at[0], p.look_at[1], p.look_at[2]);
        vec3 up = v3(p.up[0], p.up[1], p.up[2]);
        fc.d = normalized_ieee(look_at - fc.cam_pos);                       // renderer.py:272
        fc.du = normalized_ieee(cross(fc.d, up));                           // :276
        fc.dv = normalized_ieee(cross(fc.du, fc.d));                        // :277
        fc.fov = p.fov;
        fc.aspect_ratio = (float)((double)s.W / (double)s.H);          // renderer.py:19
        fc.aspect_scale = p.aspect_scale;
        fc.sun_angular_radius = (float)(6.95e8 / 1.4959e11);           // renderer.py:297-299
        fc.sun_cos_angle = de_cos(fc.sun_angular_radius);              // :300
        float s_rot, c_rot, s_ang, c_ang;
        de_sincos(p.sun_path_rot, &s_rot, &c_rot);
        de_sincos(p.sun_angle, &s_ang, &c_ang);
        fc.light_dir = v3(-s_ang, c_ang * -s_rot, c_ang * c_rot);      // :301-302
        make_orthonormal_basis(fc.light_dir, &fc.sun_x, &fc.sun_y);
        fc.land_height_scale = p.land_height_scale;
        int topo_res = p.topo_res_override > 0 ? p.topo_res_override : s.topo_w;
        fc.normal_eps = (float)(M_PI * 6371e3 / (double)topo_res);     // pathtracer.py:20
        fc.offset_scale = 1.0f + 0.0001f * p.land_height_scale / 12000.0f;   // pathtracer.py:421
        fc.kn_log = de_log(2.0f * DE_MIE_ASYMMETRY + 1.0f);
        vec3 d0 = get_density(0.0f);
        fc.max_dens_r = d0.x; fc.max_dens_m = d0.y; fc.max_dens_o = get_ozone_density(DE_OZONE_PEAK_HEIGHT);   // pathtracer.py:336
        fc.cloud = cloud_consts();
        fc.exposure_scale = de_pow(2.0f, p.exposure);                  // renderer.py:355
        fc.gamma = p.gamma;
        fc.crf_slice_v = ((float)p.selected_crf + 0.5f) / (float)s.n_crf;   // renderer.py:338
        fc.vig_strength = p.vignette_strength; fc.vig_radius = p.vignette_radius;
        fc.vig_cx = p.vignette_center[0]; fc.vig_cy = p.vignette_center[1];
        // OpenDRT parameter setup (lib/OpenDRT.py:257-319, 404): constant expressions of module constants, evaluated in
        // double on the host — see de_api.hip:opendrt_consts
        fc.drt_m = s.drt[0]; fc.drt_s = s.drt[1]; fc.drt_fl = s.drt[2];
        fc.drt_ds = s.drt[3]; fc.drt_clamp_max = s.drt[4]; fc.drt_dch_s = s.drt[5];
        for (int i = 0; i < 9; ++i) { fc.agx_srgb_to_xyz[i] = s.agx[i]; fc.agx_xyz_to_adjusted[i] = s.agx[9 + i]; }
        fc.agx_x_pivot = s.agx[18]; fc.agx_y_pivot = s.agx[19]; fc.agx_toe_scale = s.agx[20]; fc.agx_shoulder_scale = s.agx[21];
        fc.use_agx = (p.flags & DE_FLAG_AGX) ? 1 : 0;
        vec3 w = v3(0.25f, 1.0f, 0.35f);                               // lib/OpenDRT.py:369
        fc.drt_w = w / length(w);                                      // :370
        *s.fc = fc;
    }
    if (n >= DE_N_NODES || !s.do_nodes) return;
    // node n of the bisection tree: depth d = floor(log2 n), mid = (2*(n - 2^d) + 1) / 2^(d+1) — the same dyadic
    // value the reference reaches by mid = (lo + hi)/2 (exact in f32)
    float mid;
    float wavelength;
    if (n == 0) {
        mid = (s.p.fixed_wavelength - 390.0f) / 441.0f;
        wavelength = s.p.fixed_wavelength;
    } else {
        int d = 31 - __builtin_clz((unsigned)n);
        mid = (float)(2 * (n - (1 << d)) + 1) / (float)(1 << (d + 1));
        wavelength = 390.0f + 441.0f * mid;                            // lib/colour.py:39
    }
    if (n >= 1 && n < 256) {
        vec3 c = tap_f3(s.cie, mid, 0.25f, clamp);                     // lib/colour.py:28
        const float third = (float)(1.0 / 3.0);
        s.node_val[n] = de_saturate((third * c.x + third * c.y) + third * c.z);
    }
    if (n == 0) s.node_val[0] = 0.0f;
    LambdaNode L;
    L.wavelength = wavelength;
    vec3 response = tap_f3(s.cie, mid, 0.75f, clamp);                  // :40
    vec3 cmf_max = tap_f3(s.cie, 1.0f, 0.25f, clamp);                  // :41
    float pdf = dot(response, cmf_max);                                // :42
    float rcp_pdf = 0.0f;
    if (pdf > 1e-3f && !(__builtin_isinf(pdf) || pdf != pdf)) rcp_pdf = 1.0f / pdf;   // :44-46
    L.rx = response.x; L.ry = response.y; L.rz = response.z; L.rcp_pdf = rcp_pdf;
    L.ext_r = spectra_extinction_rayleigh(wavelength);                 // pathtracer.py:340-342
    L.ext_m = spectra_extinction_mie(wavelength);
    L.ext_o = spectra_extinction_ozone(wavelength, s.o3);
    L.sun_power = plancks(5778.0f, wavelength);                        // :332
    L.night_power = plancks(2700.0f, wavelength) * 0.0001f;            // :333
    L.sun_irradiance = L.sun_power * cone_angle_to_solid_angle((float)(6.95e8 / 1.4959e11));   // :334
    vec3 d0 = get_density(0.0f);
    float mo = get_ozone_density(DE_OZONE_PEAK_HEIGHT);
    L.max_ext_rmo = (L.ext_r * d0.x + L.ext_m * d0.y) + L.ext_o * mo;  // :355
    int w = (int)(wavelength - 400.0f);                                // lib/colour.py:64
    float f = (float)w - (wavelength - 400.0f);                        // :65
    L.c0 = 0.0f; L.c1 = 0.0f; L.c2 = 0.0f; L.s2s_valid = 0.0f;
    if (w > 0 && w < 299) {                                            // :67
        const float* l0 = s.srgb2spec + w * 3;
        L.c0 = mixf(l0[0], l0[3], f); L.c1 = mixf(l0[1], l0[4], f); L.c2 = mixf(l0[2], l0[5], f);
        L.s2s_valid = 1.0f;
    }
    L.inv_max_ext_rmo = 1.0f / L.max_ext_rmo;
    L.pad0 = 0.0f; L.pad1 = 0.0f; L.pad2 = 0.0f;
    s.nodes[n] = L;
}

// get_density for every altitude a tracking step can see (DE_DENS_TABLE_N): the table IS the function, not an approximation.
__global__ void __launch_bounds__(256) dens_table_kernel(float* table) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= DE_DENS_TABLE_N) return;
    const vec3 d = get_density((float)i * 0.5f);
    table[i * DE_DENS_STRIDE] = d.x; table[i * DE_DENS_STRIDE + 1u] = d.y; table[i * DE_DENS_STRIDE + 2u] = d.z;
}

// ------------------------------------------------------------------------------------------------ display
namespace {
// The functions from here to openDR_transform restate the algorithm of OpenDRT v0.2.2 ("Open Display Transform", written by Jed
// Smith, https://github.com/jedypod/open-display-transform), which the reference carries as a Taichi port in lib/OpenDRT.py under
// the notice "License: GPL v3" (lib/OpenDRT.py:5-10).  Only the configuration the reference runs is restated (in_gamut = Rec709,
// display_gamut = Rec709, EOTF = lin, Lp = 100; lib/OpenDRT.py:39-55); the same notice is on the oracle's restatement
// (oracle/oracle_opendrt.h).
DE_DEV float sdivf(float a, float b) { return (de_abs(b) < 1e-4f) ? 0.0f : a / b; }                    // lib/OpenDRT.py:92-97
DE_DEV vec3 sdivf3f(vec3 a, float b) { return v3(sdivf(a.x, b), sdivf(a.y, b), sdivf(a.z, b)); }
DE_DEV vec3 vdot_rows(const float* m, vec3 v) {                                                        // :86-88, v @ m
    return v3((v.x * m[0] + v.y * m[3]) + v.z * m[6], (v.x * m[1] + v.y * m[4]) + v.z * m[7], (v.x * m[2] + v.y * m[5]) + v.z * m[8]);
}
DE_DEV vec3 narrow_hue_angles(vec3 v) {                                                                // :191-197
    return v3(de_min(2.0f, de_max(0.0f, v.x - (v.y + v.z))), de_min(2.0f, de_max(0.0f, v.y - (v.x + v.z))),
              de_min(2.0f, de_max(0.0f, v.z - (v.x + v.y))));
}
DE_DEV vec3 openDR_transform(const FrameConsts& k, float p_R, float p_G, float p_B) {                  // :221-485 (live config)
    const float rec709_to_xyz[9] = {0.412390917540f, 0.357584357262f, 0.180480793118f, 0.212639078498f, 0.715168714523f,
                                    0.072192311287f, 0.019330825657f, 0.119194783270f, 0.950532138348f};
    const float xyz_to_rec709[9] = {3.2409699419f, -1.53738317757f, -0.498610760293f, -0.969243636281f, 1.87596750151f,
                                    0.041555057407f, 0.055630079697f, -0.203976958889f, 1.05697151424f};
    const float dch_toe = 0.0f, hs_r = 0.3f, hs_g = -0.1f, hs_b = -0.2f;
    vec3 rgb = v3(p_R, p_G, p_B);
    rgb = vdot_rows(rec709_to_xyz, rgb);
    rgb = vdot_rows(xyz_to_rec709, rgb);
    float mx = de_max(rgb.x, de_max(rgb.y, rgb.z));
    float mn = de_min(rgb.x, de_min(rgb.y, rgb.z));
    vec3 h_rgb = narrow_hue_angles(sdivf3f(rgb - v3(mn, mn, mn), mx));
    vec3 w = k.drt_w * v3(de_max(rgb.x, 1e-5f), de_max(rgb.y, 1e-5f), de_max(rgb.z, 1e-5f));
    float lum_ = length(w);
    vec3 rats = sdivf3f(rgb, lum_);
    float ts = k.drt_m * lum_ / (lum_ + k.drt_s);                         // spowf(., c = 1)
    ts = ((ts <= 0.0f) ? ts : ts * ts) / (ts + k.drt_fl);                 // flare: spowf(x, 2)/(x + fl)
    ts *= k.drt_ds;
    float ccf = sdivf(1.0f, lum_ * k.drt_dch_s + 1.0f);
    float toe_ccf = (dch_toe + 1.0f) * sdivf(lum_, lum_ + dch_toe) * ccf;
    vec3 hs_w = (1.0f - ccf) * h_rgb;
    rats = v3(rats.x + hs_w.z * hs_b - hs_w.y * hs_g, rats.y + hs_w.x * hs_r - hs_w.z * hs_b, rats.z + hs_w.y * hs_g - hs_w.x * hs_r);
    rats = v3(1.0f - toe_ccf + rats.x * toe_ccf, 1.0f - toe_ccf + rats.y * toe_ccf, 1.0f - toe_ccf + rats.z * toe_ccf);
    rats = v3(de_max(rats.x, 0.0f), de_max(rats.y, 0.0f), de_max(rats.z, 0.0f));
    float rats_mx = de_max(rats.x, de_max(rats.y, rats.z));
    float rats_mn = de_min(rats.x, de_min(rats.y, rats.z));
    float rats_ch = sdivf(rats_mx - rats_mn, rats_mx);
    float chf_in = rats_ch * ts;
    float chf = (chf_in <= 0.0f) ? chf_in : de_sqrt(chf_in);             // spowf(., v_p = 0.5)
    vec3 rats_n = sdivf3f(rats, rats_mx);
    rats = rats_n * chf + rats * (1.0f - chf);
    rgb = rats * ts;
    return v3(de_min(rgb.x, k.drt_clamp_max), de_min(rgb.y, k.drt_clamp_max), de_min(rgb.z, k.drt_clamp_max));
}
// lib/AgX.py:87-159 (the alternative display transform kept behind a comment at renderer.py:356); display_transform ends
// with `return clamp(workingColor, 0., 1.)` (:159).
DE_DEV float agx_hyperbolic(float x, float power) { return x / de_pow(1.0f + de_pow(x, power), 1.0f / power); }   // :103-105
DE_DEV float agx_full_curve(const FrameConsts& k, float x) {                                                       // :111-127
    const bool upper = x >= k.agx_x_pivot;
    const float scale = upper ? k.agx_shoulder_scale : -k.agx_toe_scale;
    const float term = ((float)2.3 * (x - k.agx_x_pivot)) / scale;                                                  // :107-109
    const float power = (scale < 0.0f) ? (float)1.9 : (float)3.1;
    return scale * agx_hyperbolic(term, power) + k.agx_y_pivot;
}
DE_DEV vec3 agx_display_transform(const FrameConsts& k, vec3 c) {                                                  // :130-159
    const float* m1 = k.agx_srgb_to_xyz;
    const float* m2 = k.agx_xyz_to_adjusted;
    vec3 xyz = v3((m1[0] * c.x + m1[1] * c.y) + m1[2] * c.z, (m1[3] * c.x + m1[4] * c.y) + m1[5] * c.z, (m1[6] * c.x + m1[7] * c.y) + m1[8] * c.z);
    vec3 adj = v3((m2[0] * xyz.x + m2[1] * xyz.y) + m2[2] * xyz.z, (m2[3] * xyz.x + m2[4] * xyz.y) + m2[5] * xyz.z,
                  (m2[6] * xyz.x + m2[7] * xyz.y) + m2[8] * xyz.z);
    const float lo = -10.0f, hi = 6.5f, total = (float)(6.5 - -10.0);
    float v[3] = {adj.x, adj.y, adj.z}, o[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float l = de_min(de_max(lo, de_log(v[i] / (float)0.18) / (float)0.6931471805599453), hi);   // :88-95; log of a non-positive channel -> MIN_EV
        float x = (l - lo) / total;
        o[i] = de_clamp(agx_full_curve(k, x), 0.0f, 1.0f);
    }
    vec3 w = v3(o[0], o[1], o[2]);
    w = mix3(lum3(w), w, (float)1.4);
    return v3(de_clamp(w.x, 0.0f, 1.0f), de_clamp(w.y, 0.0f, 1.0f), de_clamp(w.z, 0.0f, 1.0f));   // :159 `return clamp(workingColor, 0., 1.)`
}
}  // namespace

DE_DEV void display_pixel(const DisplayArgs& a, int i, int j, float* o) {
    const int idx = j * a.W + i;
    const FrameConsts& k = *a.fc;
    float u = 1.0f * (float)i / (float)a.W;                            // renderer.py:349-350
    float v = 1.0f * (float)j / (float)a.H;
    float du = u - k.vig_cx, dv = v - k.vig_cy;
    float darken = 1.0f - k.vig_strength * de_max(de_sqrt(du * du + dv * dv) - k.vig_radius, 0.0f);   // :352-354
    const float* px = a.hdr + (size_t)idx * 3;
    float samples = (float)a.samples;
    vec3 linear = v3(px[0] / samples, px[1] / samples, px[2] / samples) * darken * k.exposure_scale;   // :355
    if (k.use_agx) {                                                   // :356: srgb_transfer(agx.display_transform(linear))
        vec3 ax = agx_display_transform(k, linear);
        o[0] = srgb_transfer1(ax.x); o[1] = srgb_transfer1(ax.y); o[2] = srgb_transfer1(ax.z);
        return;
    }
    vec3 t = openDR_transform(k, linear.x, linear.y, linear.z);        // :357
    // camera_response, renderer.py:333-344
    t = v3(de_clamp(t.x, 0.0f, 1.0f), de_clamp(t.y, 0.0f, 1.0f), de_clamp(t.z, 0.0f, 1.0f));
    const float u_offset = (float)(0.5 / 1024.0), u_hi = (float)(1.0 - 0.5 / 1024.0);
    float red = tap_f3(a.crf, de_min(t.x + u_offset, u_hi), k.crf_slice_v, a.clamp != 0).x;
    float green = tap_f3(a.crf, de_min(t.y + u_offset, u_hi), k.crf_slice_v, a.clamp != 0).y;
    float blue = tap_f3(a.crf, de_min(t.z + u_offset, u_hi), k.crf_slice_v, a.clamp != 0).z;
    vec3 cam = v3(de_clamp(red, 0.0f, 1.0f), de_clamp(green, 0.0f, 1.0f), de_clamp(blue, 0.0f, 1.0f));
    vec3 g = v3(de_pow(cam.x, k.gamma), de_pow(cam.y, k.gamma), de_pow(cam.z, k.gamma));   // :360
    o[0] = srgb_transfer1(g.x); o[1] = srgb_transfer1(g.y); o[2] = srgb_transfer1(g.z);    // :362-365
}

// One 256-thread block = one 32 x 32 pixel tile.  The accumulation buffer is [H][W][3] (pixels of a row contiguous), the
// image (W, H, 3) like the reference field (pixels of a COLUMN contiguous): each thread transforms 4 pixels read along rows
// and stages them in LDS; the tile is then written out along columns, so both sides move whole lines.
__global__ void __launch_bounds__(256) display_kernel(DisplayArgs a) {
    __shared__ float tile[32][32 * 3 + 1];
    const int tx = (int)threadIdx.x & 31, ty = (int)threadIdx.x >> 5;      // 32 x 8 threads
    const int i0 = (int)blockIdx.x * 32, j0 = (int)blockIdx.y * 32;
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + tx, j = j0 + ty + 8 * r;
        if (i < a.W && j < a.H) {
            float o[3];
            display_pixel(a, i, j, o);
            tile[ty + 8 * r][tx * 3 + 0] = o[0]; tile[ty + 8 * r][tx * 3 + 1] = o[1]; tile[ty + 8 * r][tx * 3 + 2] = o[2];
        }
    }
    __syncthreads();
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + ty + 8 * r, j = j0 + tx;                        // consecutive threads: consecutive j of one column i
        if (i < a.W && j < a.H) {
            float* o = a.image + ((size_t)i * a.H + j) * 3;
            o[0] = tile[tx][(ty + 8 * r) * 3 + 0]; o[1] = tile[tx][(ty + 8 * r) * 3 + 1]; o[2] = tile[tx][(ty + 8 * r) * 3 + 2];
        }
    }
}

// Second half of render_kernel_v2: add the samples of every owned pixel to the HDR buffer IN SAMPLE ORDER
// (renderer.py:329-330: color_buffer[u, v] += xyzToRGB @ (sample * response * rcp_pdf), one launch per sample).
// One thread per pixel of an owned tile; item index of (tile k, sample s, lane-pixel sub) = (k*spp + s)*64 + sub.
__global__ void __launch_bounds__(256) accumulate_kernel(RenderArgs a) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t k = gid >> 6;
    if (k >= (uint32_t)a.n_tiles) return;
    const int sub = (int)(gid & 63u);
    const uint32_t tile = a.tiles[k];
    const int u = (int)(tile % (uint32_t)a.tiles_x) * 8 + (sub & 7);
    const int v = (int)(tile / (uint32_t)a.tiles_x) * 8 + (sub >> 3);
    float* px = a.hdr + ((size_t)v * a.W + u) * 3;
    float acc_r = px[0], acc_g = px[1], acc_b = px[2];
    const uint2* c = a.contrib + (size_t)k * 64u * (uint32_t)a.spp_count + sub;
    for (int s = 0; s < a.spp_count; ++s) {
        const uint2 q = c[(size_t)s * 64u];
        const float sample = __builtin_bit_cast(float, q.x);
        const LambdaNode& L = a.nodes[q.y];
        vec3 xyz = (sample * v3(L.rx, L.ry, L.rz)) * L.rcp_pdf;
        vec3 rgb = xyz_to_rgb_d65(xyz);
        acc_r += rgb.x; acc_g += rgb.y; acc_b += rgb.z;
    }
    px[0] = acc_r; px[1] = acc_g; px[2] = acc_b;
}

// Root of a SAMPLE partition's collective (de_reduce_ordered): the ranks' partial sums added in rank order — ((p0 + p1) + p2) + ... — whatever
// the interconnect did: parts[r] is the gathered buffer of rank r (the root's own accumulation buffer for r = root).  Streaming, float4.
struct OrderedSumArgs { const float* part[16]; float* out; int n_parts; size_t n4, n; };
__global__ void __launch_bounds__(256) ordered_sum_kernel(OrderedSumArgs a) {
    typedef float de_f4 __attribute__((ext_vector_type(4)));
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n4; i += stride) {
        de_f4 acc = reinterpret_cast<const de_f4*>(a.part[0])[i];
        for (int r = 1; r < a.n_parts; ++r) acc += reinterpret_cast<const de_f4*>(a.part[r])[i];
        reinterpret_cast<de_f4*>(a.out)[i] = acc;
    }
    for (size_t i = a.n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
        float acc = a.part[0][i];
        for (int r = 1; r < a.n_parts; ++r) acc += a.part[r][i];
        a.out[i] = acc;
    }
}

// (W, H, 3) view of the [H][W][3] accumulation buffer: 32 x 32 tiles through LDS, contiguous on both sides
__global__ void __launch_bounds__(256) hdr_transpose_kernel(const float* hdr, float* out, int W, int H) {
    __shared__ float tile[32][32 * 3 + 1];
    const int tx = (int)threadIdx.x & 31, ty = (int)threadIdx.x >> 5;
    const int i0 = (int)blockIdx.x * 32, j0 = (int)blockIdx.y * 32;
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + tx, j = j0 + ty + 8 * r;
        if (i < W && j < H)
            for (int c = 0; c < 3; ++c) tile[ty + 8 * r][tx * 3 + c] = hdr[((size_t)j * W + i) * 3 + c];
    }
    __syncthreads();
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + ty + 8 * r, j = j0 + tx;
        if (i < W && j < H)
            for (int c = 0; c < 3; ++c) out[((size_t)i * H + j) * 3 + c] = tile[tx][(ty + 8 * r) * 3 + c];
    }
}

// ------------------------------------------------------------------------------------------------ texture packing
__global__ void pack_r8_kernel(const uint8_t* src, uint32_t* dst, int w, int h, int tiles_x, int clamp) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int j = (int)blockIdx.y;
    if (i >= w || j >= h) return;
    int i1 = i + 1, j1 = j + 1;
    if (i1 >= w) i1 = clamp ? w - 1 : 0;
    if (j1 >= h) j1 = clamp ? h - 1 : 0;
    uint32_t q = (uint32_t)src[(size_t)j * w + i] | ((uint32_t)src[(size_t)j * w + i1] << 8) |
                 ((uint32_t)src[(size_t)j1 * w + i] << 16) | ((uint32_t)src[(size_t)j1 * w + i1] << 24);
    size_t idx = ((size_t)(j >> 2) * tiles_x + (size_t)(i >> 3)) * 32u + (size_t)(((j & 3) << 3) | (i & 7));
    dst[idx] = q;
}
__global__ void pack_rgb_kernel(const uint8_t* src, uint32_t* dst, size_t n) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    dst[k] = (uint32_t)src[k * 3] | ((uint32_t)src[k * 3 + 1] << 8) | ((uint32_t)src[k * 3 + 2] << 16);
}

// ------------------------------------------------------------------------------------------------ synthetic maps
namespace {
DE_DEV float lattice(int ix, int iy, int iz, uint32_t seed) {
    uint32_t h = fmix32(((uint32_t)ix * 0x8DA6B343u) ^ ((uint32_t)iy * 0xD8163841u) ^ ((uint32_t)iz * 0xCB1AB31Fu) ^ seed);
    return (float)(h >> 8) * 0x1p-24f;
}
DE_DEV float lerp_(float a, float b, float t) { return de_fma(b - a, t, a); }
DE_DEV float vnoise(float px, float py, float pz, uint32_t seed) {
    float fx0 = de_floor(px), fy0 = de_floor(py), fz0 = de_floor(pz);
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
DE_DEV float fbm(float x, float y, float z, float base, uint32_t seed, int octaves) {
    float sum = 0.0f, amp = 1.0f, freq = base;
    for (int o = 0; o < octaves; ++o) {
        float off = 17.0f * (float)o;
        sum = de_fma(amp, vnoise(de_fma(x, freq, off), de_fma(y, freq, off), de_fma(z, freq, off), seed + (uint32_t)o) - 0.5f, sum);
        amp *= 0.5f;
        freq *= 2.0f;
    }
    return sum;
}
DE_DEV uint8_t to_byte(float v) {
    v = de_clamp(v, 0.0f, 1.0f);
    return (uint8_t)(int)de_fma(v, 255.0f, 0.5f);
}
}  // namespace

// One thread per texel; writes the LINEAR u8 map (row j = 0 is the south edge), `ch` bytes per texel.
__global__ void __launch_bounds__(256) synth_kernel(uint8_t* dst, int slot, int W, int H, uint32_t seed, int variant) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int j = (int)blockIdx.y;
    if (i >= W || j >= H) return;
    const int ch = (slot == DE_TEX_ALBEDO || slot == DE_TEX_STARS) ? 3 : 1;
    uint8_t* out = dst + ((size_t)j * W + i) * ch;
    float u = ((float)i + 0.5f) / (float)W;
    float v = ((float)j + 0.5f) / (float)H;
    float phi = DE_PI * de_fma(2.0f, u, -1.0f);
    float lat = DE_PI * (v - 0.5f);
    float sp, cp, sl, cl;
    de_sincos(phi, &sp, &cp);
    de_sincos(lat, &sl, &cl);
    float nx = -(cl * cp), ny = sl, nz = cl * sp;
    uint32_t th = fmix32(((uint32_t)i * 0x9E3779B1u) ^ fmix32((uint32_t)j + 0x7F4A7C15u) ^ (seed * 0x85EBCA77u) ^ (uint32_t)slot);
    float trand = (float)(th >> 8) * 0x1p-24f;
    if (slot == DE_TEX_STARS) {
        uint8_t b = 0;
        if (trand < 0.0002f) b = (uint8_t)(128u + ((th >> 1) & 127u));
        out[0] = b; out[1] = b; out[2] = (uint8_t)(b ? (b - (th & 31u)) : 0);
        return;
    }
    if (slot == DE_TEX_CLOUDS) {
        float gain = (variant == 1) ? 2.2f : 1.6f;
        float bias = (variant == 1) ? -0.25f : 0.0f;
        float c = 0.5f + fbm(nx, ny, nz, 3.0f, seed ^ 0xC10Du, 6);
        float t = de_clamp(de_fma(gain, c, -0.3f * gain - 0.3f) - bias, 0.0f, 1.0f);
        out[0] = to_byte(t * t);
        return;
    }
    float e = fbm(nx, ny, nz, 1.7f, seed, 8) - 0.04f;
    float topo = 0.0f;
    if (e > 0.0f) { float t = de_min(1.0f, 2.5f * e); topo = t * t; }
    if (slot == DE_TEX_TOPOGRAPHY) { out[0] = to_byte(topo); return; }
    if (slot == DE_TEX_OCEAN) { out[0] = to_byte(de_clamp(de_fma(-40.0f, e, 0.5f), 0.0f, 1.0f)); return; }
    if (slot == DE_TEX_BATHYMETRY) { out[0] = to_byte(de_clamp(-3.0f * e, 0.0f, 1.0f)); return; }
    float alat = de_abs(lat) * (2.0f / DE_PI);
    float ice = de_clamp((alat + 0.35f * topo - 0.80f) * 12.0f, 0.0f, 1.0f);
    if (slot == DE_TEX_EMISSIVE) {
        float lit = (e > 0.002f && ice < 0.5f && trand < 0.01f) ? de_fma(0.75f, (float)((th >> 3) & 255u) * (1.0f / 255.0f), 0.25f) : 0.0f;
        out[0] = to_byte(lit);
        return;
    }
    float moist = 0.5f + fbm(nx, ny, nz, 2.3f, seed ^ 0xA1BEu, 4);
    float dry = de_clamp((0.55f - moist) * 4.0f + (0.35f - de_abs(alat - 0.28f)) * 2.0f, 0.0f, 1.0f);
    float r = lerp_(40.0f, 170.0f, dry), g = lerp_(70.0f, 140.0f, dry), b = lerp_(30.0f, 100.0f, dry);
    float rocky = de_clamp(topo * 2.0f, 0.0f, 1.0f);
    r = lerp_(r, 120.0f, rocky * 0.5f); g = lerp_(g, 110.0f, rocky * 0.5f); b = lerp_(b, 100.0f, rocky * 0.5f);
    float sea = de_clamp(de_fma(-40.0f, e, 0.5f), 0.0f, 1.0f);
    float depth = de_clamp(-3.0f * e, 0.0f, 1.0f);
    float sr = lerp_(30.0f, 10.0f, depth), sg = lerp_(70.0f, 25.0f, depth), sb = lerp_(90.0f, 60.0f, depth);
    r = lerp_(r, sr, sea); g = lerp_(g, sg, sea); b = lerp_(b, sb, sea);
    r = lerp_(r, 240.0f, ice); g = lerp_(g, 240.0f, ice); b = lerp_(b, 245.0f, ice);
    out[0] = to_byte(r * (1.0f / 255.0f)); out[1] = to_byte(g * (1.0f / 255.0f)); out[2] = to_byte(b * (1.0f / 255.0f));
}

// ------------------------------------------------------------------------------------------------ debug math
__global__ void math_kernel(int fn, const float* a, const float* b, float* out, size_t n) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    float x = a[k], y = b ? b[k] : 0.0f, r;
    switch (fn) {
        case 0: r = de_exp(x); break;
        case 1: r = de_log(x); break;
        case 2: r = de_sin(x); break;
        case 3: r = de_cos(x); break;
        case 4: r = de_atan2(x, y); break;
        case 5: r = de_asin(x); break;
        case 6: r = de_pow(x, y); break;
        case 7: r = x / y; break;
        case 8: r = de_sqrt(x); break;
        case 9: r = de_sqrt_nr(x); break;
        case 10: r = de_div_rc_ng(x, y, 1.0f / y); break;
        case 11: r = x * de_rcp_nr(y); break;                  // a quotient under contract 2: product with the correctly rounded reciprocal
        case 12: r = de_log_unit(x); break;
        case 13: r = de_rcp_nr(x); break;
        case 14: r = de_exp_nonpos(x); break;
        default: r = 0.0f;
    }
    out[k] = r;
}
