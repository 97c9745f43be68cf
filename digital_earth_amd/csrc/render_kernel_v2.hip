// render_kernel_v2.hip — path_tracer (pathtracer.py:316-469) as a WAVEFRONT-LEVEL STATE MACHINE.
//
// Why: in the straightforward kernel (render_kernel.hip) every lane walks the reference's nested loops
// (250-step sphere trace, two delta-tracking loops, two ratio-tracking loops per light sample, 25 vertices) and the
// 64 lanes of a wave serialise on each other's trip counts: rocprofv3 measured 11.5 % VALU lane utilisation
// (profiles/r1a_summary.md).  Here each lane carries an explicit `state`; every trip of the wave loop advances
// EVERY lane that is inside one of the hot loops by one iteration, whatever loop of whatever path vertex it is in:
//
//   TAP class   one evaluation point that needs |p|, p/|p|, the sphere UV map and one bilinear map tap
//               — a sphere-trace step (topography), a cloud-shell tracking step (cloud map), a normal tap
//   GAS class   one tracking step through the Rayleigh/Mie/ozone profiles
//
// Everything between loops (segment set-up, event resolution, shading, sample start/end) is "transition" code: it
// runs when enough lanes are waiting for it (or nothing else can run), in pipeline order, so a lane falls through
// consecutive transitions in one pass.  Lanes start their next sample immediately when a path ends (path
// regeneration), so short paths do not idle the lane.
//
// Results are IDENTICAL to render_kernel.hip and to the CPU oracle: each lane executes the same arithmetic in the
// same order on its own RNG stream; only the interleaving between lanes changes.
#pragma once
#include "de_kernels.h"

namespace v2 {

enum { EV_NULL = 0, EV_ABSORB = 1, EV_SCATTER = 2 };

// lane states.  Loop states first (class tests are range tests).
enum : int {
    // --- TAP class
    S_ST_P = 0,      // sphere trace, path ray                       pathtracer.py:37-44
    S_ST_S,          // sphere trace, shadow ray from the surface    pathtracer.py:422
    S_NRM0, S_NRM1, S_NRM2, S_NRM3,   // land_normal's four SDF evaluations   pathtracer.py:16-25
    S_DT_C,          // delta tracking, cloud shell                  pathtracer.py:198
    S_RT_C_V,        // ratio tracking, cloud shell, light sample of a volume vertex   pathtracer.py:231
    S_RT_C_S,        // same for a surface vertex
    // --- GAS class
    S_DT_G,          // delta tracking, gases                        pathtracer.py:186
    S_RT_G_V,        // ratio tracking, gases (volume vertex)        pathtracer.py:227
    S_RT_G_S,        // ratio tracking, gases (surface vertex)
    // --- transitions, in pipeline order
    S_AFTER_ST_P, S_AFTER_ST_S,
    S_CLOUD_SETUP_DT, S_CLOUD_SETUP_V, S_CLOUD_SETUP_S,
    S_AFTER_DT_C,
    S_SC_FINISH,
    S_SURF_SETUP,
    S_SURF_FINISH,
    S_END, S_END_MISS,
    S_NEW,
    S_VERTEX,
    S_DONE
};
DE_DEV bool is_tap(int s) { return s <= S_RT_C_S; }
DE_DEV bool is_gas(int s) { return s >= S_DT_G && s <= S_RT_G_S; }
DE_DEV bool is_cloud_tap(int s) { return s >= S_DT_C && s <= S_RT_C_S; }

struct Work2 { uint32_t taps_r8, taps_rgb, sphere_steps, tracking_steps, vertices, draws; };

// pathtracer.py:145-169
DE_DEV void intersect_cloud_limits(vec3 ray_pos, vec3 ray_dir, float land_isection, float* t_start_out, float* t_max_out) {
    float t_start = 0.0f, t_max = 0.0f;
    float elevation = length(ray_pos);
    vec2_ lower = rsi(ray_pos, ray_dir, DE_CLOUDS_LOWER);
    vec2_ upper = rsi(ray_pos, ray_dir, DE_CLOUDS_UPPER);
    if (elevation >= DE_CLOUDS_UPPER) {
        t_start = de_max(0.0f, upper.x);
        t_max = (lower.y >= 0.0f) ? lower.x : upper.y;
        if (upper.y < 0.0f) t_max = -1.0f;
    } else if (elevation >= DE_CLOUDS_LOWER) {
        t_start = 0.0f;
        t_max = (lower.y >= 0.0f) ? lower.x : upper.y;
    } else {
        t_start = lower.y;
        t_max = upper.y;
        if (land_isection > 0.0f) t_max = -1.0f;
    }
    *t_start_out = t_start; *t_max_out = t_max;
}
// pathtracer.py:145-168 up to (not including) the `if land_isection > 0.0: t_max = -1.0` of the below-cloud branch:
// everything that depends on the ray alone, computed when the ray is set up.  `below` tells the caller to apply that line.
DE_DEV void cloud_limits_of_ray(vec3 ray_pos, vec3 ray_dir, float* t_start_out, float* t_max_out, int* below_out) {
    float t_start = 0.0f, t_max = 0.0f;
    int below = 0;
    float elevation = length_nr(ray_pos);
    vec2_ lower = rsi(ray_pos, ray_dir, DE_CLOUDS_LOWER);
    vec2_ upper = rsi(ray_pos, ray_dir, DE_CLOUDS_UPPER);
    if (elevation >= DE_CLOUDS_UPPER) {
        t_start = de_max(0.0f, upper.x);
        t_max = (lower.y >= 0.0f) ? lower.x : upper.y;
        if (upper.y < 0.0f) t_max = -1.0f;
    } else if (elevation >= DE_CLOUDS_LOWER) {
        t_start = 0.0f;
        t_max = (lower.y >= 0.0f) ? lower.x : upper.y;
    } else {
        t_start = lower.y;
        t_max = upper.y;
        below = 1;
    }
    *t_start_out = t_start; *t_max_out = t_max; *below_out = below;
}
// pathtracer.py:235-247
DE_DEV float evaluate_phase(const FrameConsts& fc, vec3 ray_dir, vec3 light_dir, int id, bool reduce_peak) {
    float phase = 0.0f;
    float c = dot(ray_dir, light_dir);
    if (id == RAYLEIGH_ID) phase += rayleigh_phase(c);
    else if (id == MIE_ID) phase += klein_nishina_phase(c, DE_MIE_ASYMMETRY, fc.kn_log);
    else if (id == CLOUD_ID) phase += cloud_phase(fc.cloud, c, reduce_peak);
    else if (id == ISOTROPIC_CLOUD_ID) phase += (float)(1.0 / (4.0 * M_PI));
    return phase;
}
// pathtracer.py:249-261
DE_DEV vec3 sample_phase(const FrameConsts& fc, Rng& rng, vec3 ray_dir, int id, bool reduce_peak, float* phase_div_pdf) {
    *phase_div_pdf = 1.0f;
    if (id == RAYLEIGH_ID || id == ISOTROPIC_CLOUD_ID) {
        float r0 = rng_next(rng);
        float r1 = rng_next(rng);
        vec3 d = sample_sphere(r0, r1);
        *phase_div_pdf = evaluate_phase(fc, ray_dir, d, id, reduce_peak) * (float)(4.0 * M_PI);
        return d;
    } else if (id == MIE_ID) {
        return sample_klein_nishina_phase(rng, ray_dir, DE_MIE_ASYMMETRY);
    }
    return sample_cloud_phase(fc.cloud, rng, ray_dir, reduce_peak);
}
DE_DEV float srgb_to_spectrum(const LambdaNode& L, vec3 rgb) {
    return (L.s2s_valid != 0.0f) ? dot(rgb, v3(L.c0, L.c1, L.c2)) : 0.0f;
}
// lib/colour.py:21-39.  The first DE_V2_LDS_TREE_LEVELS levels of the bisection tree are read from the wave's LDS copy, the
// rest from global memory (measured per frame: 0 levels 232 ms, 5 levels 230 ms, 6 levels 228 ms, 7 levels 238 ms — the LDS
// they take is what limits the waves per CU; reading the last two levels with one 32-byte block read changed nothing).
#ifndef DE_V2_LDS_TREE_LEVELS
#define DE_V2_LDS_TREE_LEVELS 6
#endif
DE_DEV int spectrum_node(const float* lds_val, const float* node_val, float sample) {
    int n = 1;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        float val = (it < DE_V2_LDS_TREE_LEVELS) ? lds_val[n] : node_val[n];
        if (val < sample) n = 2 * n + 1;
        else if (val > sample) n = 2 * n;
        else break;
    }
    return n;
}
DE_DEV vec3 get_cast_dir(Rng& rng, const FrameConsts& fc, int H, int u, int v) {   // renderer.py:269-279
    float fov = fc.fov;
    float fu = (2.0f * fov * ((float)u + rng_next(rng)) / (float)H - fov * fc.aspect_ratio - 1e-5f) * fc.aspect_scale;
    float fv = 2.0f * fov * ((float)v + rng_next(rng)) / (float)H - fov - 1e-5f;
    return normalized(fc.d + fu * fc.du + fv * fc.dv);
}

}  // namespace v2

#ifndef DE_V2_CHUNK
#define DE_V2_CHUNK 128           // default work items a wave takes from the device-wide counter per atomic (RenderArgs::chunk, set by the host)
#endif
#ifndef DE_V2_BLOCK
// Threads per workgroup.  One wave per workgroup: the waves share nothing but a 1 KB table, and a workgroup's LDS and
// wave slots are released only when ALL its waves have ended — with 4-wave workgroups the last long paths of a launch
// kept 3 idle waves' worth of the machine from the next launch (frames in flight, bench.py).  20 x 7.25 KB of LDS per CU.
#define DE_V2_BLOCK 64
#endif
#ifndef DE_V2_MIN_WAVES
#define DE_V2_MIN_WAVES 5          // waves per SIMD the register allocator must leave room for (96 VGPRs)
#endif

// MODE 0: accumulate; 1: accumulate + work counters; 2: trace one sample per pixel into debug_out
template <bool CLAMP, int MODE>
#ifdef DE_V2_NUM_VGPR
__attribute__((amdgpu_num_vgpr(DE_V2_NUM_VGPR)))
#endif
__global__ void __launch_bounds__(DE_V2_BLOCK, DE_V2_MIN_WAVES) render_kernel_v2(RenderArgs a, const FrameConsts* __restrict__ fcp) {
    using namespace v2;
    // Persistent waves over WORK ITEMS = single samples.  Item w of the launch is sample (w / 64) % spp of lane-pixel
    // (w & 63) of owned tile a.tiles[w / (64 * spp)].  A wave takes chunks of DE_V2_CHUNK consecutive items from a
    // device-wide counter (one atomic per chunk) and hands them to its lanes one by one as paths end, so no lane idles
    // while work is left and the longest indivisible piece of work is ONE path — not one pixel's 64 samples.
    // A sample does not touch the HDR buffer: it writes (radiance, wavelength node) to a.contrib[w]; accumulate_kernel
    // then adds each pixel's samples in index order, ((c + s0) + s1) + ..., exactly like renderer.py:330 does over
    // successive launches, so the result is bit-identical to one-lane-per-pixel accumulation.
    const int lane = threadIdx.x & 63;
    const uint32_t items_per_tile = 64u * (uint32_t)a.spp_count;
    const bool resume = a.resume_rec != nullptr;        // wave-uniform: finisher of the wavefront pipeline (de_kernels.h)
    const uint32_t total_items = resume ? min(*a.resume_count, a.resume_capacity) : (uint32_t)a.n_tiles * items_per_tile;
    uint32_t w_lo = 0, w_hi = 0;          // wave-uniform: this wave's current chunk [w_lo, w_hi)
    uint32_t w = 0;                       // this lane's current item
    // the frame constants come through a top-level `const __restrict__` kernel argument: the compiler then knows that no
    // store of this kernel can change them and reads them with scalar loads (as a member of RenderArgs they were
    // re-read with vector loads, each with its own wait, all through the transition code)
    const FrameConsts& fc = *fcp;
    const float MAX_RAY_DIST = (float)(6371e3 * 10.0);
    const bool same_dims = (a.clouds.w == a.topo.w) && (a.clouds.h == a.topo.h);
    // frame constants of the hot loops, read once (left as fc.x they are re-loaded from memory, with a wait, on every trip)
    const float land_height_scale = fc.land_height_scale, normal_eps = fc.normal_eps;
    const float escape_r = DE_PLANET_R + de_max(land_height_scale, 0.0f) + 8100.0f;

    Work2 wk = {0, 0, 0, 0, 0, 0};
    uint32_t n_samples = 0, path_vertices = 0;

    // ---- per-lane path state
    Rng rng; rng.s0 = 1; rng.s1 = 0; rng.draws = 0;
    int node = 0;
    vec3 P = v3(0, 0, 0), D = v3(0, 0, 1), LD = v3(0, 0, 1), C = v3(0, 0, 0), N = v3(0, 0, 1);
    float thr = 1.0f, Ls = 0.0f;
    int sc = 0;
    float t = 0.0f, tmax = 0.0f, trn = 1.0f, trg = 1.0f;
    int cnt = 0;
    float atm_x = 0.0f, atm_y = 0.0f;     // rsi(P, W, atmosphere) of the ray being traced
    float isect = 0.0f, rmo_t = 0.0f;
    int rmo_event = 0, ev_id = 0;         // ev_id: interaction id of the resolved event
    float li = -1.0f;                     // land_isection argument of the light-sample transmittance
    float c_ts = 0.0f, c_tm = 0.0f;       // cloud-shell segment of the ray being traced (cloud_limits_of_ray)
    int c_below = 0;
    float d0 = 0.0f, alb = 0.0f, ocean = 0.0f, bathy = 0.0f;
    float ext_r = 0.0f, ext_m = 0.0f, ext_o = 0.0f, rmaxr = 0.0f;   // rmaxr = RN(1 / max_ext_rmo): every quotient by the majorant is a product with it (contract 2)
    int state = S_NEW;
    if (a.spp_count <= 0 || a.n_tiles <= 0) state = S_DONE;

    // ---- second path slot per lane ("stash", LDS).  The three expensive transitions — volume-vertex finish
    // (phase evaluation + sampling), surface set-up (material taps + grading) and surface finish (two BRDF
    // evaluations) — cost 600-1300 instructions each and a wave meets each of them less than once per trip: run
    // in place they execute for 1-2 lanes of 64.  Instead a path that reaches one of them is put aside in the
    // lane's stash and the lane starts (or resumes) another path; when >= a.heavy_thr lanes of the wave hold a
    // stashed path waiting for the same transition, those lanes swap the stashed path in, run the transition
    // together, and swap back.  The processed path waits in the stash (READY) until the lane's current path ends
    // or is put aside itself.  Per-lane slots only: no cross-lane sharing, no barriers.
    // LDS per wave = the stash (29 dwords per lane, 7.25 KB, in the production mode) + the first six levels of
    // spectrum_sample's bisection tree (256 B), so that 20 one-wave workgroups fit the CU's 160 KB (5 waves per SIMD, with
    // 96 VGPRs).  The tree's last two levels are read from global memory (L1-resident).
    __shared__ float s_node_val[1 << DE_V2_LDS_TREE_LEVELS];
    for (int i = lane; i < (1 << DE_V2_LDS_TREE_LEVELS); i += 64) s_node_val[i] = a.node_val[i];
    __syncthreads();
    constexpr int STASH_FIELDS = (MODE == 0) ? 29 : 31;
    __shared__ uint32_t s_stash[STASH_FIELDS][64];
    uint32_t* const slot = &s_stash[0][lane];
    int stash = 0;                        // 0 empty; 1/2/3 waiting for S_SC_FINISH / S_SURF_SETUP / S_SURF_FINISH; 4 ready to resume
#define DE_XF(i, var) { const uint32_t tmp_ = slot[(i) * 64]; slot[(i) * 64] = __builtin_bit_cast(uint32_t, var); var = __builtin_bit_cast(decltype(var), tmp_); }
    // the small integers travel in one dword: state (5 bits) | sc (5) | ev_id (3) | c_below (1) | cnt (8) | node (9)
#define DE_EXCHANGE_PATH()                                                                                              \
    { DE_XF(0, P.x) DE_XF(1, P.y) DE_XF(2, P.z) DE_XF(3, D.x) DE_XF(4, D.y) DE_XF(5, D.z) DE_XF(6, LD.x) DE_XF(7, LD.y)   \
      DE_XF(8, LD.z) DE_XF(9, N.x) DE_XF(10, N.y) DE_XF(11, N.z) DE_XF(12, thr) DE_XF(13, Ls) DE_XF(14, trg) DE_XF(15, trn) \
      DE_XF(16, alb) DE_XF(17, ocean) DE_XF(18, bathy) DE_XF(19, li) DE_XF(20, atm_x) DE_XF(21, atm_y) DE_XF(22, t)       \
      DE_XF(23, rng.s0) DE_XF(24, rng.s1) DE_XF(25, w) DE_XF(26, c_ts) DE_XF(27, c_tm)                                     \
      { uint32_t pk_ = (uint32_t)state | ((uint32_t)sc << 5) | ((uint32_t)ev_id << 10) | ((uint32_t)c_below << 13) |       \
                       ((uint32_t)cnt << 14) | ((uint32_t)node << 22);                                                     \
        DE_XF(28, pk_)                                                                                                     \
        state = (int)(pk_ & 31u); sc = (int)((pk_ >> 5) & 31u); ev_id = (int)((pk_ >> 10) & 7u);                            \
        c_below = (int)((pk_ >> 13) & 1u); cnt = (int)((pk_ >> 14) & 255u); node = (int)(pk_ >> 22); }                      \
      if (MODE != 0) { DE_XF(29, rng.draws) DE_XF(30, path_vertices) } }
#define DE_RELOAD_NODE()                                                                                                \
    { const LambdaNode& Ln_ = a.nodes[node]; ext_r = Ln_.ext_r; ext_m = Ln_.ext_m; ext_o = Ln_.ext_o; rmaxr = Ln_.inv_max_ext_rmo; }

    // ---- loop-to-loop glue, executed in place by the lane that finishes a loop (no wait for the transition pass)
    // enter the cloud-shell segment of the current ray: kind 0 = delta tracking of the path ray, 1 / 2 = light sample of a volume / surface vertex
    auto enter_cloud = [&](int kind) {
        const vec3 W = (kind == 0) ? D : LD;
        const float li_arg = (kind == 0) ? isect : li;
        const float t_start = c_ts;
        const float t_max = (c_below && li_arg > 0.0f) ? -1.0f : c_tm;              // pathtracer.py:166
        bool run = true;
        if (kind == 0) run = (rmo_event == EV_NULL) || (rmo_t > t_start);            // pathtracer.py:195
        t = t_start; tmax = t_max;
        C = P + t * W;
        trn = 1.0f;
        const bool enter = run && (t_start < t_max);
        if (kind == 0) state = enter ? S_DT_C : S_AFTER_DT_C;
        else if (kind == 1) state = enter ? S_RT_C_V : S_SC_FINISH;
        else state = enter ? S_RT_C_S : S_SURF_FINISH;
    };
    // after a sphere trace (pathtracer.py:46): its result, then the gas segment of the same ray (pathtracer.py:180-184, 219-225)
    auto after_sphere_trace = [&](bool primary) {
        const float hit = (t < MAX_RAY_DIST) ? t : -1.0f;
        float land_isection;
        if (primary) { isect = hit; land_isection = hit; }
        else { li = (hit < 0.0f) ? -1.0f : 0.0f; land_isection = li; }
        const float t_start = de_max(0.0f, atm_x);
        float t_max = (land_isection >= 0.0f) ? land_isection : atm_y;
        if (atm_y < 0.0f) t_max = -1.0f;
        const vec3 W = primary ? D : LD;
        t = t_start; tmax = t_max;
        C = P + t * W;
        trn = 1.0f;
        if (t_start < t_max) state = primary ? S_DT_G : S_RT_G_S;
        else if (primary) { rmo_event = EV_NULL; rmo_t = t; ev_id = 0; enter_cloud(0); }
        else { trg = 1.0f; enter_cloud(2); }
    };

    // MODE 1: per-wave scheduler statistics (de_debug_sched_stats; indices documented in tools/sched_stats.py)
    unsigned long long tm[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tlast = (MODE == 1) ? __builtin_amdgcn_s_memtime() : 0ull;
    uint32_t st[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) st[i] = 0;
#define DE_ST(i, v) { if (MODE == 1) st[i] += (uint32_t)(v); }
#define DE_ST_RUN(i, pred) { if (MODE == 1) { st[i] += 1u; st[(i) + 1] += (uint32_t)__popcll(__ballot(pred)); } }
    int n_wait = 0;
    for (;;) {
        const bool tapL = is_tap(state), gasL = is_gas(state);
        const unsigned long long m_tap = __ballot(tapL), m_gas = __ballot(gasL);
        if (MODE == 1) {
            const unsigned long long m_live = __ballot(state != S_DONE || stash != 0);
            st[0]++; st[2] += __popcll(m_tap); st[3] += __popcll(m_gas);
            st[4] += (m_tap != 0ull);
            st[29] += 64 - __popcll(m_live);
            st[30] += __popcll(__ballot(state >= S_SC_FINISH && state <= S_SURF_FINISH && stash >= 1 && stash <= 3));
            st[31] += __popcll(__ballot(state >= S_AFTER_ST_P && state != S_DONE));
        }

        if (MODE == 1) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tm[0] += now_ - tlast; tlast = now_; }
        // ============================================================ first half: advance, position, UV, issue the tap
        // Shared by the TAP class and (on the trips the GAS class runs) the GAS class: a tracking step of either kind starts
        // with the same free-flight draw, division by its majorant, advance and |position|.  The map tap — the long-latency
        // operation of a trip (L2 / HBM) — is issued here; the gas densities are evaluated while it is in flight and the
        // texel is consumed in the second half below.
        uint32_t tap_q = 0;
        float tap_fx = 0.0f, tap_fy = 0.0f, tap_len = 0.0f;
        bool tap_alive = false, tap_do = false, tap_escaped = false;
        float dens_r = 0.0f, dens_m = 0.0f, dens_o = 0.0f;   // gas densities at the step's point, fetched from the altitude table
        bool dens_oob = false;
        // the GAS class runs when enough lanes wait in it (a.gas_thr) or the TAP class is short of lanes
        const bool run_gas = (m_gas != 0ull) && (__popcll(m_gas) >= a.gas_thr || __popcll(m_tap) < 24);
        const bool gasR = gasL && run_gas;
        if (m_tap != 0ull || run_gas) {
            DE_ST_RUN(7, tapL || gasR)
            if (run_gas) DE_ST_RUN(5, gasR)
            if (tapL || gasR) {
                const bool cloud = is_cloud_tap(state);
                const bool track = cloud || gasR;
                const vec3 W = (state == S_ST_P || state == S_DT_C || state == S_DT_G) ? D : LD;
                vec3 Q;
                tap_alive = true;
                if (track) {
                    // free flight: -log(xi) * RN(1 / majorant) (contract 2: the quotient by the majorant as a product)
                    const float rmaxv = cloud ? ((sc > 9) ? (1.0f / (0.02f * DE_CLOUDS_DENSITY)) : (1.0f / (DE_CLOUDS_EXTINCT * DE_CLOUDS_DENSITY))) : rmaxr;
                    const float neg_log = -de_log_unit(rng_next(rng));
                    const float t_step = neg_log * rmaxv;
                    C = C + t_step * W;
                    t += t_step;
                    if (MODE) wk.tracking_steps++;
                    tap_alive = !(t >= tmax);
                    Q = C;
                } else if (state <= S_ST_S) {
                    Q = P + W * t;
                } else {
                    const float e = normal_eps;
                    Q = P;
                    if (state == S_NRM1) Q = P - v3(e, 0.0f, 0.0f);
                    if (state == S_NRM2) Q = P - v3(0.0f, e, 0.0f);
                    if (state == S_NRM3) Q = P - v3(0.0f, 0.0f, e);
                }
                if (tap_alive) {
                    tap_len = length_nr(Q);                        // positions: |Q|^2 ~ 4e13
#ifndef DE_V2_NO_DENS_TABLE
                    if (gasR) {
                        // get_density(get_elevation(pos)) from the altitude table: |Q| lies in [2^22, 2^23), where f32 has a
                        // spacing of 0.5 m, so h = |Q| - R is an exact multiple of 0.5 and 2h indexes a table that holds
                        // get_density(h) itself for EVERY possible h below 131 km (dens_table_kernel: the same device function,
                        // evaluated once per entry).  Issued here so that the load is in flight during the TAP lanes' UV mapping.
                        const float h2 = de_max(tap_len - DE_PLANET_R, 0.0f) * 2.0f;
                        const uint32_t di = (uint32_t)(int)h2;
                        dens_oob = !(h2 < (float)DE_DENS_TABLE_N);
                        const vec3 dtv = dens_table_read(a.dens_table, dens_oob ? 0u : di);
                        dens_r = dtv.x; dens_m = dtv.y; dens_o = dtv.z;
                    }
#else
                    dens_oob = gasR;
#endif
                    tap_do = tapL && (!cloud || (tap_len > DE_CLOUDS_LOWER && tap_len < DE_CLOUDS_UPPER));
                    // Escaped ray (generalises SURVEY Q2).  If the rest of the ray stays more than 8 km above every possible
                    // terrain height (its perigee, or its current point once it moves outward), every further SDF value is
                    // > 8 km > 1e-4 * max_ray_dist (6371 m): the `abs(dist) < ray_dist*1e-4` test can never fire again, each
                    // step advances by at least its height, and the trace leaves through ray_dist > max_ray_dist after at most
                    // 127 more steps (worst case: inward tangent ray over maximal terrain) — inside the 250-step budget while
                    // cnt <= 100.  The result is -1 whatever the steps in between: skip them.
                    if (state <= S_ST_S && cnt <= 100) {
                        const float b = dot(Q, W);
                        const float K = escape_r;
                        const float peri2 = (b < 0.0f) ? (tap_len * tap_len - b * b) : tap_len * tap_len;
                        if (peri2 > K * K) { tap_escaped = true; tap_do = false; }
                    }
                    if (tap_do) {
                        const float inv_len = de_rcp_nr(tap_len);      // RN(1 / len) in 3 operations; normalized() = v * RN(1 / |v|)
                        const vec3 n = Q * inv_len;
                        if (MODE) wk.taps_r8++;
                        TexR8 T = a.topo;   // per-lane map select (two maps only in this class)
                        T.data = cloud ? a.clouds.data : a.topo.data;
                        if (!same_dims) {   // wave-uniform: the reference's tiers give both maps one size (lib/textures.py:65-79)
                            T.w = cloud ? a.clouds.w : a.topo.w; T.h = cloud ? a.clouds.h : a.topo.h;
                            T.tiles_x = cloud ? a.clouds.tiles_x : a.topo.tiles_x;
                            T.wf = cloud ? a.clouds.wf : a.topo.wf; T.hf = cloud ? a.clouds.hf : a.topo.hf;
                        }
                        vec2_ uv = sphere_UV_map(n);
                        tap_q = tap_r8_issue<CLAMP>(T, fract_(uv.x * 1.0f), fract_(uv.y * 1.0f), &tap_fx, &tap_fy);
                    }
                }
            }
        }

        if (MODE == 1) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tm[1] += now_ - tlast; tlast = now_; }
        // ============================================================ GAS class: densities and collision / transmittance
        if (run_gas) {
            if (gasR) {
                auto gas_div = [&](float x) -> float { return x * rmaxr; };       // x / max_ext_rmo under contract 2
                if (!tap_alive) {                                  // t >= t_max: the segment ends without an event
                    if (state == S_DT_G) { rmo_event = EV_NULL; rmo_t = t; ev_id = 0; enter_cloud(0); }
                    else { trg = trn; enter_cloud(state == S_RT_G_V ? 1 : 2); }
                } else {
                    vec3 dens = v3(dens_r, dens_m, dens_o);           // volume.get_density(volume.get_elevation(pos)), tabulated
                    if (__ballot(dens_oob) != 0ull) {                  // above the table (never inside the atmosphere segment): evaluate
                        asm volatile("" ::: "memory");
                        if (dens_oob) dens = get_density(tap_len - DE_PLANET_R);
                    }
                    vec3 es = v3(ext_r, ext_m, ext_o) * dens;
                    float sum = (es.x + es.y) + es.z;
                    if (state == S_DT_G) {
                        float rand = rng_next(rng);
                        if (rand < gas_div(sum)) {
                            int id = 0;
                            float cmf = es.x;
                            if (!(rand < gas_div(cmf))) {
                                id = 1;
                                cmf += es.y;
                                if (!(rand < gas_div(cmf))) {
                                    id = 2;
                                    cmf += es.z;
                                    if (!(rand < gas_div(cmf))) id = 3;
                                }
                            }
                            const float albedo = (id == 0) ? 1.0f : ((id == 1) ? 0.95f : ((id == 2) ? 0.0f : 0.99f));
                            rmo_event = (rng_next(rng) < albedo) ? EV_SCATTER : EV_ABSORB;
                            rmo_t = t;
                            ev_id = id;
                            enter_cloud(0);
                        }
                    } else {
                        trn *= 1.0f - gas_div(sum);
                        if (trn < 1e-5f) { trg = trn; enter_cloud(state == S_RT_G_V ? 1 : 2); }
                    }
                    if (is_gas(state) && !(t < tmax)) {            // loop-head test, see the cloud class
                        if (state == S_DT_G) { rmo_event = EV_NULL; rmo_t = t; ev_id = 0; enter_cloud(0); }
                        else { trg = trn; enter_cloud(state == S_RT_G_V ? 1 : 2); }
                    }
                }
            }
        }

        if (MODE == 1) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tm[2] += now_ - tlast; tlast = now_; }
        // ============================================================ TAP class, second half: consume the texel
        if (m_tap != 0ull) {
            if (tapL) {
                const bool cloud = is_cloud_tap(state);
                if (tap_alive) {
                    const float len = tap_len;
                    float tex = 0.0f;
                    if (tap_do) tex = tap_r8_finish(tap_q, tap_fx, tap_fy);
                    if (cloud) {
                        const float ext_w = (sc > 9) ? 0.02f : DE_CLOUDS_EXTINCT;
                        const float rmaxc = (sc > 9) ? (1.0f / (0.02f * DE_CLOUDS_DENSITY)) : (1.0f / (DE_CLOUDS_EXTINCT * DE_CLOUDS_DENSITY));
                        // get_clouds_density, pathtracer.py:48-65
                        float density = 0.0f;
                        if (tap_do) {
                            float h = (len - DE_CLOUDS_LOWER) * (1.0f / DE_CLOUDS_THICKNESS);
                            const float split = 0.2f;
                            density = (h - split < tex * (float)(1.0 - 0.2) && split - h < tex * split) ? de_max(tex, 0.4f) : 0.0f;
                        }
                        float es = ext_w * (density * DE_CLOUDS_DENSITY);
                        if (state == S_DT_C) {
                            float rand = rng_next(rng);
                            if (rand < es * rmaxc) {
                                rmo_event |= ((rng_next(rng) < 0.99f) ? EV_SCATTER : EV_ABSORB) << 4;   // cloud event in bits 4-5
                                state = S_AFTER_DT_C;
                            }
                        } else {
                            trn *= 1.0f - es * rmaxc;
                            if (trn < 1e-5f) state = (state == S_RT_C_V) ? S_SC_FINISH : S_SURF_FINISH;
                        }
                        // `while t < t_max` at the loop head (pathtracer.py:91,130): implied by the test above unless t is NaN
                        // (a degenerate camera); without it a NaN path would iterate for ever
                        if (is_cloud_tap(state) && !(t < tmax)) {
                            if (state == S_DT_C) state = S_AFTER_DT_C;
                            else state = (state == S_RT_C_V) ? S_SC_FINISH : S_SURF_FINISH;
                        }
                    } else {
                        float sdf = len - DE_PLANET_R - land_height_scale * tex;
                        if (state <= S_ST_S && tap_escaped) {
                            t = MAX_RAY_DIST;                              // "ray_dist < max_ray_dist" fails -> -1
                            after_sphere_trace(state == S_ST_P);
                        } else if (state <= S_ST_S) {
                            t += sdf;
                            cnt++;
                            if (MODE) wk.sphere_steps++;
                            if (t > MAX_RAY_DIST || de_abs(sdf) < t * 0.0001f || cnt >= 250) after_sphere_trace(state == S_ST_P);
                        } else {
                            if (state == S_NRM0) d0 = sdf;
                            else if (state == S_NRM1) N.x = d0 - sdf;
                            else if (state == S_NRM2) N.y = d0 - sdf;
                            else N.z = d0 - sdf;
                            state = (state == S_NRM3) ? S_SURF_SETUP : state + 1;
                        }
                    }
                } else {
                    // t >= t_max: the tracking loop ends without an event
                    if (state == S_DT_C) state = S_AFTER_DT_C;
                    else state = (state == S_RT_C_V) ? S_SC_FINISH : S_SURF_FINISH;
                }
            }
        }

        if (MODE == 1) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tm[3] += now_ - tlast; tlast = now_; }
        // ============================================================ transitions
        const bool heavyA = (state >= S_SC_FINISH && state <= S_SURF_FINISH);
        const bool parked = heavyA && stash >= 1 && stash <= 3;          // both of the lane's paths wait for a batch
        const bool pendL = (state >= S_AFTER_ST_P && state != S_DONE && !parked) || (state == S_DONE && stash == 4);
        const int n_pend = __popcll(__ballot(pendL));
        const int n_loop = __popcll(__ballot(is_tap(state) || is_gas(state)));
        const bool flush_all = n_loop < a.tune_b;                        // too few lanes iterating: process everything that waits
        // n_wait: stashed paths waiting for a batch.  The stash changes only inside a pass, and a pass runs every batch that
        // has reached a.heavy_thr, so between passes no batch is due and the count taken at the end of the last pass holds.
        if (n_loop == 0 && n_pend == 0 && n_wait == 0) break;            // every lane is S_DONE with an empty stash
        if (!(n_pend >= a.pend_thr || (flush_all && (n_pend > 0 || n_wait > 0)))) continue;
        if (MODE == 1) { st[1]++; st[10] += n_pend; }

        if (MODE == 1) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tm[4] += now_ - tlast; tlast = now_; }
        // ---- event resolution, light sample, branch (pathtracer.py:200-205, 369-375, 404, 441-444)
        if (__ballot(state == S_AFTER_DT_C) != 0ull) {
            DE_ST_RUN(11, state == S_AFTER_DT_C)
            if (state == S_AFTER_DT_C) {
                const int cloud_event = rmo_event >> 4;
                const int gas_event = rmo_event & 15;
                int event = gas_event;
                float tt = rmo_t;
                int id = ev_id;
                // cloud_t is the current t when the cloud loop produced an event
                if (cloud_event > 0 && (t < rmo_t || gas_event == EV_NULL)) { tt = t; id = CLOUD_ID; event = cloud_event; }
                if (sc > 9 && id == CLOUD_ID) id = ISOTROPIC_CLOUD_ID;
                ev_id = id;
                {   // sample_cone_oriented (lib/sampling.py:25-28) with the light direction's basis taken from the frame constants
                    const vec3 cv = sample_cone(rng, fc.sun_cos_angle);
                    const vec3 bx = fc.sun_x, by = fc.sun_y, bn = fc.light_dir;
                    LD = v3((bx.x * cv.x + by.x * cv.y) + bn.x * cv.z, (bx.y * cv.x + by.y * cv.y) + bn.y * cv.z, (bx.z * cv.x + by.z * cv.y) + bn.z * cv.z);
                }
                if (event == EV_ABSORB) {
                    state = S_END;
                } else if (event == EV_SCATTER) {
                    P = P + tt * D;
                    const bool blocked = rsi(P, LD, DE_PLANET_R).y > 0.0f;
                    if (blocked) {
                        trg = 0.0f; trn = 1.0f;           // direct_transmittance stays 0
                        li = 1.0f;                        // marks "blocked" for S_SC_FINISH
                        state = S_SC_FINISH;
                    } else {
                        li = -1.0f;
                        vec2_ atmos = rsi(P, LD, DE_ATMOS_UPPER);
                        cloud_limits_of_ray(P, LD, &c_ts, &c_tm, &c_below);
                        float t_start = de_max(0.0f, atmos.x);
                        float t_max = atmos.y;            // land_isection = -1
                        if (atmos.y < 0.0f) t_max = -1.0f;
                        t = t_start; tmax = t_max;
                        C = P + t * LD;
                        trn = 1.0f;
                        if (t_start < t_max) state = S_RT_G_V;
                        else { trg = 1.0f; enter_cloud(1); }
                    }
                } else if (isect > 0.0f) {
                    P = P + D * isect;                    // land_pos
                    state = S_NRM0;
                } else {
                    state = (sc == 0) ? S_END_MISS : S_END;
                }
            }
        }
        if (MODE == 1) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tm[5] += now_ - tlast; tlast = now_; }
        // ---- end of path: miss epilogue, NaN guard, colour accumulation (pathtracer.py:455-469, renderer.py:329-330)
        if (__ballot(state == S_END || state == S_END_MISS) != 0ull) {
            DE_ST_RUN(13, state == S_END || state == S_END_MISS)
            if (state == S_END || state == S_END_MISS) {
                const LambdaNode& L = a.nodes[node];
                if (state == S_END_MISS) {
                    if (dot(fc.light_dir, D) > fc.sun_cos_angle) Ls += L.sun_power;     // D is still the primary direction
                    if (MODE) wk.taps_rgb += 1;
                    vec3 stars_srgb = sphere_tap_rgb<CLAMP>(a.stars, normalized(D));
                    float stars_power = srgb_to_spectrum(L, stars_srgb);
                    Ls += stars_power * L.sun_power * 0.0000001f;
                }
                if (__builtin_isinf(Ls) || Ls != Ls || Ls < 0.0f) Ls = 0.0f;
                if (MODE != 2) a.contrib[w] = make_uint2(__builtin_bit_cast(uint32_t, Ls), (uint32_t)node);
                if (MODE) wk.draws += rng.draws;
                if (MODE == 2) {
                    const uint32_t tile_ = a.tiles[w / items_per_tile];
                    const int sub_ = (int)((w % items_per_tile) & 63u);
                    const uint32_t pixel_ = (uint32_t)(((int)(tile_ / (uint32_t)a.tiles_x) * 8 + (sub_ >> 3)) * a.W + (int)(tile_ % (uint32_t)a.tiles_x) * 8 + (sub_ & 7));
                    float* q = a.debug_out + (size_t)pixel_ * 4;
                    q[0] = Ls; q[1] = L.wavelength; q[2] = (float)rng.draws; q[3] = (float)path_vertices;
                }
                n_samples++;
                state = S_NEW;
            }
        }
        if (MODE == 1) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tm[6] += now_ - tlast; tlast = now_; }
        // ---- a path reached an expensive transition: put it aside (or swap with the lane's resumable path)
        if (__ballot(state >= S_SC_FINISH && state <= S_SURF_FINISH) != 0ull) {
            DE_ST_RUN(15, state >= S_SC_FINISH && state <= S_SURF_FINISH && (stash == 0 || stash == 4))
            if (state >= S_SC_FINISH && state <= S_SURF_FINISH && (stash == 0 || stash == 4)) {
                const int kind = state - S_SC_FINISH + 1;
                const bool resume = (stash == 4);
                DE_EXCHANGE_PATH()
                stash = kind;
                if (resume) { DE_RELOAD_NODE() }      // `state` is now the resumed path's
                else state = S_NEW;                   // slot held nothing: start a new path
            }
        }
        // ---- batches of the expensive transitions on the stashed paths
        {
            const int c1 = __popcll(__ballot(stash == 1)), c2 = __popcll(__ballot(stash == 2)), c3 = __popcll(__ballot(stash == 3));
            const int n_loop2 = __popcll(__ballot(is_tap(state) || is_gas(state)));
            const bool flush = n_loop2 < a.tune_b;
            const bool run1 = c1 >= a.heavy_thr || (flush && c1 > 0);
            const bool run2 = c2 >= a.heavy_thr || (flush && c2 > 0);
            const bool run3 = c3 >= a.heavy_thr || (flush && c3 > 0);
            const bool in_batch = (stash == 1 && run1) || (stash == 2 && run2) || (stash == 3 && run3);
            if (run1 || run2 || run3) {
                if (in_batch) { DE_EXCHANGE_PATH() }  // stashed path in, current path out
                const bool do1 = run1 && state == S_SC_FINISH, do2 = run2 && state == S_SURF_SETUP, do3 = run3 && state == S_SURF_FINISH;
                // ---- volume vertex: in-scattering, phase sampling, roulette (pathtracer.py:394-401, 448-453)
                if (__ballot(do1) != 0ull) {
                    DE_ST_RUN(17, do1)
                    if (do1) {
                        const LambdaNode& L = a.nodes[node];
                        const float direct_tr = (li > 0.0f) ? 0.0f : trg * trn;
                        const float direct_phase = evaluate_phase(fc, D, LD, ev_id, sc > 0);
                        Ls += thr * direct_tr * L.sun_irradiance * direct_phase;
                        float phase_div_pdf;
                        vec3 nd = sample_phase(fc, rng, D, ev_id, sc > 0, &phase_div_pdf);
                        D = nd;
                        thr *= phase_div_pdf;
                        state = S_VERTEX;
                        if (sc > 3) {
                            float termination_p = de_max(0.05f, 1.0f - thr);
                            if (rng_next(rng) < termination_p) state = S_END;
                            else thr /= 1.0f - termination_p;
                        }
                        sc++;
                        if (state == S_VERTEX && sc >= 25) state = S_END;
                    }
                }
                // ---- surface vertex, first half: normal, material, emissive, shadow-ray set-up (pathtracer.py:406-422)
                if (__ballot(do2) != 0ull) {
                    DE_ST_RUN(19, do2)
                    if (do2) {
                        const LambdaNode& L = a.nodes[node];
                        N = normalized(N);
                        vec3 n = normalized(P);
                        vec2_ uv = sphere_UV_map(n);
                        float uu = fract_(uv.x * 1.0f), vv = fract_(uv.y * 1.0f);
                        ocean = tap_r8<CLAMP>(a.ocean, uu, vv);
                        vec3 tex = tap_rgb<CLAMP>(a.albedo, uu, vv);
                        vec3 land = mix3(lum3(tex), tex, 6.5f);
                        float greenery = sqr(land.y / lum(land));
                        greenery = smoothstep_(1.5f, 1.9f, greenery);
                        land = (1.0f * tex) / (greenery * 0.7f + 1.0f);
                        land = mix3(lum3(land), land, 1.4f - greenery * 0.45f);
                        land = mix3(land, (land * v3(255.0f, 128.0f, 64.0f)) / 255.0f, 0.2f * (1.0f - greenery));
                        vec3 ocean_albedo = mix3(lum3(tex), tex, 0.75f) * 0.9f;
                        vec3 albedo_srgb = mix3(land, ocean_albedo, ocean);
                        bathy = tap_r8<CLAMP>(a.bathy, uu, vv);
                        float emissive = tap_r8<CLAMP>(a.emissive, uu, vv);
                        if (MODE) { wk.taps_r8 += 3; wk.taps_rgb += 1; }
                        alb = srgb_to_spectrum(L, albedo_srgb);
                        Ls += thr * emissive * L.night_power;
                        P = P * fc.offset_scale;                                 // offset_pos
                        // intersect_land(offset_pos, light_dir) head
                        vec2_ atmos = rsi(P, LD, DE_ATMOS_UPPER);
                        atm_x = atmos.x; atm_y = atmos.y;
                        cloud_limits_of_ray(P, LD, &c_ts, &c_tm, &c_below);
                        t = (atmos.x > 0.0f) ? atmos.x : 0.0f;
                        cnt = 0;
                        state = S_ST_S;
                    }
                }
                // ---- surface vertex, second half: direct light, bounce, roulette (pathtracer.py:431-439, 448-453)
                if (__ballot(do3) != 0ull) {
                    DE_ST_RUN(21, do3)
                    if (do3) {
                        const LambdaNode& L = a.nodes[node];
                        const float direct_tr = trg * trn;
                        const bool visible = li < 0.0f;
                        float direct_ndl;
                        float direct_brdf = earth_brdf(alb, ocean, bathy, -D, N, LD, &direct_ndl);
                        Ls += thr * direct_tr * (visible ? 1.0f : 0.0f) * L.sun_irradiance * direct_brdf * direct_ndl;
                        vec3 view_dir = -D;
                        D = sample_hemisphere_cosine_weighted(rng, N);
                        float unused;
                        float brdf = earth_brdf(alb, ocean, bathy, view_dir, N, D, &unused);
                        thr *= brdf * (float)M_PI;
                        state = S_VERTEX;
                        if (sc > 3) {
                            float termination_p = de_max(0.05f, 1.0f - thr);
                            if (rng_next(rng) < termination_p) state = S_END;
                            else thr /= 1.0f - termination_p;
                        }
                        sc++;
                        if (state == S_VERTEX && sc >= 25) state = S_END;
                    }
                }

                if (in_batch) { DE_EXCHANGE_PATH() stash = 4; }   // processed path back to the slot, resumable
            }
        }
        if (MODE == 1) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tm[7] += now_ - tlast; tlast = now_; }
        // ---- resume a processed path where the lane has nothing else to run
        if (__ballot(stash == 4 && (state == S_NEW || state == S_DONE || (state >= S_SC_FINISH && state <= S_SURF_FINISH))) != 0ull) {
            DE_ST_RUN(23, stash == 4 && (state == S_NEW || state == S_DONE || (state >= S_SC_FINISH && state <= S_SURF_FINISH)))
            if (stash == 4 && (state == S_NEW || state == S_DONE)) {
                DE_EXCHANGE_PATH()                    // the slot's content is dead afterwards
                stash = 0;
                DE_RELOAD_NODE()
            } else if (stash == 4 && state >= S_SC_FINISH && state <= S_SURF_FINISH) {
                const int kind = state - S_SC_FINISH + 1;
                DE_EXCHANGE_PATH()
                stash = kind;
                DE_RELOAD_NODE()
            }
        }
        if (MODE == 1) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tm[8] += now_ - tlast; tlast = now_; }
        // ---- next sample of this pixel (renderer.py:305-314)
        if (__ballot(state == S_NEW) != 0ull) {
            DE_ST_RUN(25, state == S_NEW)
            // Hand out items of the wave's chunk; refill it (one device-wide atomic) when it runs dry.  Executed by ALL
            // lanes (wave-uniform branch) so that the chunk bounds w_lo / w_hi stay wave-uniform.
            const bool need = (state == S_NEW);
            bool got = false;
            for (;;) {
                const unsigned long long m_need = __ballot(need && !got);
                if (m_need == 0ull) break;
                const uint32_t avail = w_hi - w_lo;
                const int my = __popcll(m_need & ((1ull << lane) - 1ull));
                if (need && !got && (uint32_t)my < avail) { w = w_lo + (uint32_t)my; got = true; }
                w_lo += min((uint32_t)__popcll(m_need), avail);
                if (__ballot(need && !got) == 0ull) break;
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(a.work_counter, (uint32_t)a.chunk);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (base >= total_items) break;
                w_lo = base;
                w_hi = min(base + (uint32_t)a.chunk, total_items);
            }
            if (need) {
                if (!got) {
                    state = S_DONE;
                } else if (resume) {
                    // a queued path ray of the wavefront pipeline (wf::Path layout, render_kernel_v3.hip): position, direction, the ray
                    // head computed by its last vertex (atmosphere and cloud-shell limits), RNG state, node, vertex count
                    const uint4* r = a.resume_rec + (size_t)w * 4u;
                    const uint4 q0 = r[0];
                    if (q0.x != 0xffffffffu) {                     // not a hole left by a wave's bulk reservation
                        const uint4 q1 = r[1], q2 = r[2], q3 = r[3];
                        w = q0.x;                                  // from here on the path's own index (contribution slot)
                        rng.s0 = q0.z; rng.s1 = q0.w; rng.draws = 0;
                        node = (int)((q0.y >> 2) & 511u); sc = (int)((q0.y >> 11) & 31u); c_below = (int)((q0.y >> 16) & 1u);
                        P = v3(__builtin_bit_cast(float, q1.x), __builtin_bit_cast(float, q1.y), __builtin_bit_cast(float, q1.z));
                        D = v3(__builtin_bit_cast(float, q1.w), __builtin_bit_cast(float, q2.x), __builtin_bit_cast(float, q2.y));
                        t = __builtin_bit_cast(float, q2.z);
                        atm_x = __builtin_bit_cast(float, q2.w); atm_y = __builtin_bit_cast(float, q3.x);
                        c_ts = __builtin_bit_cast(float, q3.y); c_tm = __builtin_bit_cast(float, q3.z);
                        const float* cold = a.resume_cold + (size_t)w * 16u;
                        thr = cold[0]; Ls = cold[1];
                        DE_RELOAD_NODE()
                        cnt = 0; rmo_event = 0;
                        path_vertices = 0;
                        state = S_ST_P;
                    }
                } else {
                    // (w >> 6) / spp by multiplication (a.spp_magic = floor(2^40 / spp) + 1; exact while (w >> 6) * spp < 2^40)
                    const uint32_t wq = w >> 6;
                    const uint32_t tile_k = (uint32_t)(((unsigned long long)wq * a.spp_magic) >> 40);
                    const uint32_t tile = a.tiles_identity ? tile_k : a.tiles[tile_k];   // one rank owns every tile: no dependent load (-1.6 % per frame)
                    const int s_idx = (int)(wq - tile_k * (uint32_t)a.spp_count);
                    const int sub = (int)(w & 63u);
                    const int u = (int)(tile % (uint32_t)a.tiles_x) * 8 + (sub & 7);
                    const int v = (int)(tile / (uint32_t)a.tiles_x) * 8 + (sub >> 3);
                    const uint32_t pixel = (uint32_t)(v * a.W + u);
                    rng_seed(rng, a.seed_lo, a.seed_hi, pixel, (uint32_t)(a.spp_begin + s_idx * a.spp_stride));
                    node = 0;
                    if (!a.fixed_wavelength) node = v2::spectrum_node(s_node_val, a.node_val, rng_next(rng));
                    const LambdaNode& L = a.nodes[node];
                    ext_r = L.ext_r; ext_m = L.ext_m; ext_o = L.ext_o; rmaxr = L.inv_max_ext_rmo;
                    D = v2::get_cast_dir(rng, fc, a.H, u, v);
                    P = fc.cam_pos;
                    thr = 1.0f; Ls = 0.0f; sc = 0;
                    path_vertices = 0;
                    state = S_VERTEX;
                }
            }
        }
        if (MODE == 1) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tm[9] += now_ - tlast; tlast = now_; }
        // ---- start of a vertex: intersect_land head (pathtracer.py:29-35)
        if (__ballot(state == S_VERTEX) != 0ull) {
            DE_ST_RUN(27, state == S_VERTEX)
            if (state == S_VERTEX) {
                if (MODE) { wk.vertices++; path_vertices++; }
                vec2_ atmos = rsi(P, D, DE_ATMOS_UPPER);
                atm_x = atmos.x; atm_y = atmos.y;
                cloud_limits_of_ray(P, D, &c_ts, &c_tm, &c_below);
                t = (atmos.x > 0.0f) ? atmos.x : 0.0f;
                cnt = 0;
                rmo_event = 0;
                state = S_ST_P;
            }
        }
        n_wait = __popcll(__ballot(stash >= 1 && stash <= 3));
        if (MODE == 1) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tm[10] += now_ - tlast; tlast = now_; }
    }

    if (MODE == 1) {
        atomicAdd(&a.counters[0], (unsigned long long)n_samples);
        atomicAdd(&a.counters[1], (unsigned long long)wk.taps_r8);
        atomicAdd(&a.counters[2], (unsigned long long)wk.taps_rgb);
        atomicAdd(&a.counters[3], (unsigned long long)wk.sphere_steps);
        atomicAdd(&a.counters[4], (unsigned long long)wk.tracking_steps);
        atomicAdd(&a.counters[5], (unsigned long long)wk.vertices);
        atomicAdd(&a.counters[6], (unsigned long long)wk.draws);
        if (lane == 0) {   // per-wave scheduler statistics: trips, passes, section cycle counters (s_memtime)
            atomicAdd(&a.counters[7], (unsigned long long)st[0]);
            atomicAdd(&a.counters[8], (unsigned long long)st[1]);
#pragma unroll
            for (int i = 0; i < 32; ++i) atomicAdd(&a.counters[16 + i], (unsigned long long)st[i]);
#pragma unroll
            for (int i = 0; i < 12; ++i) atomicAdd(&a.counters[48 + i], tm[i]);
        }
    }
}
