// de_stages.h — path_tracer (pathtracer.py:316-469) cut into STAGES: what a path does between two points where it can change its work class.
//
//     ST (sphere trace)  ->  GAS (tracking through Rayleigh/Mie/ozone)  ->  CLOUD (tracking through the cloud shell)
//        ->  EVENT (event resolution, light sample)  ->  [SURF (normal + material)  ->  ST (shadow ray)]
//        ->  GAS -> CLOUD (ratio tracking towards the sun)  ->  SCFIN / SURFFIN (shading, next direction, roulette)  ->  ST ...
//
// A path between stages is a RECORD (wf::Path: position, traced direction, the stage's scalars, the 8 bytes of generator state, the small
// integers in one word); what the loops do not need (throughput, radiance so far, incoming direction, normal, material) is its COLD record
// (wf::Cold).  The three loop stages (StStage, GasStage, CloudStage) are structs with begin / step / finish (+ suspend / restore: the loop's
// state between two steps, bit for bit); the four straight-line stages are functions from a record to the stage it goes to next.
// Written once, used by render_kernel_v6.hip (the product: one persistent launch, stage queues in LDS) and by the two legacy statements of
// the same integrator (legacy/render_kernel_v3.hip: level-synchronous kernels over HBM queues; legacy/render_kernel_v5.hip: one persistent
// launch over HBM queues).  Every path executes the same arithmetic in the same order on its own RNG stream whichever of them runs it.
// (Until round 5 this text lived in render_kernel_v3.hip.)
#pragma once
#include "de_kernels.h"
#include "render_kernel_v2.hip"      // namespace v2: the leaf helpers the stage bodies share with the state machine (cloud limits, camera ray, phase evaluation / sampling)

namespace wf {

// a pointer the optimiser knows nothing about: loads through it are neither hoisted above this point nor merged with earlier ones (the persistent
// kernels read the launch's RenderArgs per stage through one: as kernel arguments they were hoisted to the kernel's entry and spilled)
template <typename T> DE_DEV const T* opaque_ptr(const T* p) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)p), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uintptr_t)p >> 32));
    asm volatile("" : "+s"(lo), "+s"(hi));
    return reinterpret_cast<const T*>((uintptr_t)lo | ((uintptr_t)hi << 32));
}

enum { Q_ST = 0, Q_GAS, Q_CLOUD, Q_EVENT, Q_SURF, Q_SCFIN, Q_SURFFIN, NQ };
enum { EV_NULL = 0, EV_ABSORB = 1, EV_SCATTER = 2 };

// per-path state outside the loops (64 bytes, one sector per access)
struct Cold { float thr, Ls, dx, dy, dz, nx, ny, nz, alb, ocean, bathy, pad0, pad1, pad2, pad3, pad4; };

// A path inside a stage.  Record layout: v[0] = (pid, meta, rng.s0, rng.s1), v[1] = (P.xyz, W.x), v[2] = (W.yz, x0, x1),
// v[3] = (x2, x3, x4, x5).  W is the direction being traced: D for kind 0, the light direction LD otherwise.
// meta: kind (2 bits: 0 path ray, 1 light ray of a volume vertex, 2 shadow / light ray of a surface vertex) | node << 2 (9) |
//       sc << 11 (5) | c_below << 16 | ev_id << 17 (3) | rmo_event << 20 (6: gas event, cloud event << 4)
struct Path {
    uint32_t pid, meta;
    Rng rng;
    vec3 P, W;
    float x[6];
};
DE_DEV int m_kind(uint32_t m) { return (int)(m & 3u); }
DE_DEV int m_node(uint32_t m) { return (int)((m >> 2) & 511u); }
DE_DEV int m_sc(uint32_t m) { return (int)((m >> 11) & 31u); }
DE_DEV int m_below(uint32_t m) { return (int)((m >> 16) & 1u); }
DE_DEV int m_evid(uint32_t m) { return (int)((m >> 17) & 7u); }
DE_DEV int m_event(uint32_t m) { return (int)((m >> 20) & 63u); }
DE_DEV uint32_t m_make(int kind, int node, int sc, int below, int ev_id, int event) {
    return (uint32_t)kind | ((uint32_t)node << 2) | ((uint32_t)sc << 11) | ((uint32_t)below << 16) | ((uint32_t)ev_id << 17) | ((uint32_t)event << 20);
}
DE_DEV float asf(uint32_t u) { return __builtin_bit_cast(float, u); }
DE_DEV uint32_t asu(float f) { return __builtin_bit_cast(uint32_t, f); }

// pathtracer.py:145-168 without the `land_isection` line (the caller applies it through c_below): shared with v2
DE_DEV void cloud_limits_of_ray(vec3 ray_pos, vec3 ray_dir, float* t_start_out, float* t_max_out, int* below_out) {
    v2::cloud_limits_of_ray(ray_pos, ray_dir, t_start_out, t_max_out, below_out);
}

// ---- stage glue (pathtracer.py:180-205, 219-232): the same statements v2 executes "in place" when a loop ends
// enter the cloud-shell segment of the ray in p (kind in meta).  aux = rmo_t (kind 0) or trg (kinds 1, 2); X = isect / li.
// Returns the queue the record goes to, with p.x laid out for it.
DE_DEV int enter_cloud(Path& p, float c_ts, float c_tm, float aux, float X) {
    const int kind = m_kind(p.meta);
    const float t_start = c_ts;
    const float t_max = (m_below(p.meta) && X > 0.0f) ? -1.0f : c_tm;                      // pathtracer.py:166
    bool run = true;
    if (kind == 0) run = ((m_event(p.meta) & 15) == EV_NULL) || (aux > t_start);          // pathtracer.py:195
    const bool enter = run && (t_start < t_max);
    if (enter) { p.x[0] = t_start; p.x[1] = t_max; p.x[2] = aux; p.x[3] = X; return Q_CLOUD; }
    if (kind == 0) { p.x[0] = t_start; p.x[1] = aux; p.x[2] = X; return Q_EVENT; }        // EVENT: t, rmo_t, isect
    p.x[0] = aux; p.x[1] = 1.0f; p.x[2] = X;                                              // SCFIN / SURFFIN: trg, trn, li
    return kind == 1 ? Q_SCFIN : Q_SURFFIN;
}
// after a sphere trace (pathtracer.py:46, then :180-184 / :219-225).  p.x = (t0, atm_x, atm_y, c_ts, c_tm); t = the trace's result
DE_DEV int after_sphere_trace(Path& p, float t, float max_ray_dist) {
    const int kind = m_kind(p.meta);
    const float atm_x = p.x[1], atm_y = p.x[2], c_ts = p.x[3], c_tm = p.x[4];
    const float hit = (t < max_ray_dist) ? t : -1.0f;
    const float X = (kind == 0) ? hit : ((hit < 0.0f) ? -1.0f : 0.0f);                     // isect / li
    const float t_start = de_max(0.0f, atm_x);
    float t_max = (X >= 0.0f) ? X : atm_y;
    if (atm_y < 0.0f) t_max = -1.0f;
    if (t_start < t_max) { p.x[0] = t_start; p.x[1] = t_max; p.x[2] = c_ts; p.x[3] = c_tm; p.x[4] = X; return Q_GAS; }
    // no gas segment: rmo_event = NULL, rmo_t = t (kind 0) / trg = 1 (shadow ray)
    return enter_cloud(p, c_ts, c_tm, (kind == 0) ? t_start : 1.0f, X);
}
// the head of a ray (pathtracer.py:29-35 + the ray-only parts of :173-178 and :145-168): fills p.x for the ST queue
DE_DEV void ray_head(Path& p) {
    const vec2_ atmos = rsi(p.P, p.W, DE_ATMOS_UPPER);
    float c_ts, c_tm; int below;
    cloud_limits_of_ray(p.P, p.W, &c_ts, &c_tm, &below);
    p.x[0] = (atmos.x > 0.0f) ? atmos.x : 0.0f;
    p.x[1] = atmos.x; p.x[2] = atmos.y; p.x[3] = c_ts; p.x[4] = c_tm;
    p.meta = (p.meta & ~(1u << 16)) | ((uint32_t)below << 16);
}

// ---- the loops
// intersect_land's loop (pathtracer.py:37-44) from t = p.x[0]; returns ray_dist (max_ray_dist for an escaped ray, v2's early-out)
template <bool CLAMP>
DE_DEV float sphere_trace(const RenderArgs& a, const Path& p, float land_height_scale, float escape_r, float max_ray_dist) {
    float t = p.x[0];
    int cnt = 0;
    for (;;) {
        const vec3 Q = p.P + p.W * t;
        const float len = length_nr(Q);
        if (cnt <= 100) {
            const float b = dot(Q, p.W);
            const float peri2 = (b < 0.0f) ? (len * len - b * b) : len * len;
            if (peri2 > escape_r * escape_r) { t = max_ray_dist; break; }
        }
        const vec3 n = Q * de_rcp_nr(len);
        const vec2_ uv = sphere_UV_map(n);
        const float tex = tap_r8<CLAMP>(a.topo, fract_(uv.x * 1.0f), fract_(uv.y * 1.0f));
        const float sdf = len - DE_PLANET_R - land_height_scale * tex;
        t += sdf;
        cnt++;
        if (t > max_ray_dist || de_abs(sdf) < t * 0.0001f || cnt >= 250) break;
    }
    return t;
}

DE_DEV void write_contrib(const RenderArgs& a, uint32_t pid, float Ls, int node) {
    if (__builtin_isinf(Ls) || Ls != Ls || Ls < 0.0f) Ls = 0.0f;                          // pathtracer.py:466-467
    a.contrib[pid] = make_uint2(asu(Ls), (uint32_t)node);
}


// renderer.py:305-314 + the head of the primary ray: work item w -> (pixel, sample) exactly as render_kernel_v2 maps them, the
// path's generator, wavelength node (lib/colour.py:21-39; the first six tree levels from the wave's LDS copy) and camera ray
DE_DEV void new_path(const RenderArgs& a, const FrameConsts& fc, const float* s_node_val, uint32_t w, Path& p) {
    const uint32_t wq = w >> 6;
    const uint32_t tile_k = (uint32_t)(((unsigned long long)wq * a.spp_magic) >> 40);
    const uint32_t tile = a.tiles_identity ? tile_k : a.tiles[tile_k];
    const int s_idx = (int)(wq - tile_k * (uint32_t)a.spp_count);
    const int sub = (int)(w & 63u);
    const int u = (int)(tile % (uint32_t)a.tiles_x) * 8 + (sub & 7);
    const int v = (int)(tile / (uint32_t)a.tiles_x) * 8 + (sub >> 3);
    const uint32_t pixel = (uint32_t)(v * a.W + u);
    rng_seed(p.rng, a.seed_lo, a.seed_hi, pixel, (uint32_t)(a.spp_begin + s_idx * a.spp_stride));
    int node = 0;
    if (!a.fixed_wavelength) {
        const float sample = rng_next(p.rng);
        node = 1;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const float val = (it < 6) ? s_node_val[node] : a.node_val[node];
            if (val < sample) node = 2 * node + 1;
            else if (val > sample) node = 2 * node;
            else break;
        }
    }
    p.pid = w;
    p.W = v2::get_cast_dir(p.rng, fc, a.H, u, v);
    p.P = fc.cam_pos;
    p.meta = m_make(0, node, 0, 0, 0, 0);
    p.x[5] = 0.0f;
    ray_head(p);
}

// ---- the straight-line stages' bodies: one record in, the queue it goes to (or -1: the path has ended and its contribution is written)

// EVENT: event resolution, light sample, branch — pathtracer.py:200-205, 369-393, 404, 441-444, and the miss epilogue :455-469.
// p.x = (cloud_t, rmo_t, isect); W = D
template <bool CLAMP, typename ColdT>
DE_DEV int event_body(const RenderArgs& a, const FrameConsts& fc, const ColdT& cold, Path& p) {
    int target = -1;
    const int sc = m_sc(p.meta), node = m_node(p.meta);
    const int rmo_event = m_event(p.meta);
    const int cloud_event = rmo_event >> 4, gas_event = rmo_event & 15;
    const float t = p.x[0], rmo_t = p.x[1], isect = p.x[2];
    const vec3 D = p.W;
    int event = gas_event;
    float tt = rmo_t;
    int id = m_evid(p.meta);
    if (cloud_event > 0 && (t < rmo_t || gas_event == EV_NULL)) { tt = t; id = CLOUD_ID; event = cloud_event; }
    if (sc > 9 && id == CLOUD_ID) id = ISOTROPIC_CLOUD_ID;
    vec3 LD;
    {   // sample_cone_oriented (lib/sampling.py:25-28), the light direction's basis from the frame constants
        const vec3 cv = sample_cone(p.rng, fc.sun_cos_angle);
        const vec3 bx = fc.sun_x, by = fc.sun_y, bn = fc.light_dir;
        LD = v3((bx.x * cv.x + by.x * cv.y) + bn.x * cv.z, (bx.y * cv.x + by.y * cv.y) + bn.y * cv.z, (bx.z * cv.x + by.z * cv.y) + bn.z * cv.z);
    }
    bool end = false, miss = false;
    if (event == EV_ABSORB) {
        end = true;
    } else if (event == EV_SCATTER) {
        p.P = p.P + tt * D;
        if (sc == 0) cold.set_thr_ls(p.pid, 1.0f, 0.0f);
        cold.set_dir(p.pid, D);
        p.W = LD;
        const bool blocked = rsi(p.P, LD, DE_PLANET_R).y > 0.0f;
        if (blocked) {
            p.meta = m_make(1, node, sc, 0, id, 0);
            p.x[0] = 0.0f; p.x[1] = 1.0f; p.x[2] = 1.0f;            // trg = 0 (direct_transmittance stays 0), trn, li = 1 marks "blocked"
            target = Q_SCFIN;
        } else {
            const vec2_ atmos = rsi(p.P, LD, DE_ATMOS_UPPER);
            float c_ts, c_tm; int below;
            cloud_limits_of_ray(p.P, LD, &c_ts, &c_tm, &below);
            p.meta = m_make(1, node, sc, below, id, 0);
            const float t_start = de_max(0.0f, atmos.x);
            float t_max = atmos.y;                                  // land_isection = -1
            if (atmos.y < 0.0f) t_max = -1.0f;
            if (t_start < t_max) {
                p.x[0] = t_start; p.x[1] = t_max; p.x[2] = c_ts; p.x[3] = c_tm; p.x[4] = -1.0f;
                target = Q_GAS;
            } else {
                target = enter_cloud(p, c_ts, c_tm, 1.0f, -1.0f);  // trg = 1
            }
        }
    } else if (isect > 0.0f) {
        p.P = p.P + D * isect;                                       // land_pos
        if (sc == 0) cold.set_thr_ls(p.pid, 1.0f, 0.0f);
        cold.set_dir(p.pid, D);
        p.W = LD;
        p.meta = m_make(2, node, sc, 0, id, 0);
        target = Q_SURF;
    } else {
        end = true;
        miss = (sc == 0);
    }
    if (end) {
        const LambdaNode& L = a.nodes[node];
        float Ls = (sc == 0) ? 0.0f : cold.get_ls(p.pid);
        if (miss) {                                                  // pathtracer.py:455-463; D is still the primary direction
            if (dot(fc.light_dir, D) > fc.sun_cos_angle) Ls += L.sun_power;
            const vec3 stars_srgb = sphere_tap_rgb<CLAMP>(a.stars, normalized(D));
            const float stars_power = v2::srgb_to_spectrum(L, stars_srgb);
            Ls += stars_power * L.sun_power * 0.0000001f;
        }
        write_contrib(a, p.pid, Ls, node);
    }
    return target;
}

// SURF: land_normal (pathtracer.py:16-25), get_land_material (:276-312), emissive term, offset position (:406-422), head of the
// shadow ray.  P = land_pos, W = LD.  Always continues to Q_ST.
template <bool CLAMP, typename ColdT>
DE_DEV int surf_body(const RenderArgs& a, const FrameConsts& fc, const ColdT& cold, Path& p) {
    const float land_height_scale = fc.land_height_scale, e = fc.normal_eps;
    const LambdaNode& L = a.nodes[m_node(p.meta)];
    auto sdf_at = [&](vec3 Q) -> float {
        const float len = length_nr(Q);
        const vec3 nn = Q * de_rcp_nr(len);
        const vec2_ uv = sphere_UV_map(nn);
        const float tex = tap_r8<CLAMP>(a.topo, fract_(uv.x * 1.0f), fract_(uv.y * 1.0f));
        return len - DE_PLANET_R - land_height_scale * tex;
    };
    const float d0 = sdf_at(p.P);
    vec3 N = v3(d0 - sdf_at(p.P - v3(e, 0.0f, 0.0f)), d0 - sdf_at(p.P - v3(0.0f, e, 0.0f)), d0 - sdf_at(p.P - v3(0.0f, 0.0f, e)));
    N = normalized(N);
    const vec3 nrm = normalized(p.P);
    const vec2_ uv = sphere_UV_map(nrm);
    const float uu = fract_(uv.x * 1.0f), vv = fract_(uv.y * 1.0f);
    const float ocean = tap_r8<CLAMP>(a.ocean, uu, vv);
    const vec3 tex = tap_rgb<CLAMP>(a.albedo, uu, vv);
    vec3 land = mix3(lum3(tex), tex, 6.5f);
    float greenery = sqr(land.y / lum(land));
    greenery = smoothstep_(1.5f, 1.9f, greenery);
    land = (1.0f * tex) / (greenery * 0.7f + 1.0f);
    land = mix3(lum3(land), land, 1.4f - greenery * 0.45f);
    land = mix3(land, (land * v3(255.0f, 128.0f, 64.0f)) / 255.0f, 0.2f * (1.0f - greenery));
    const vec3 ocean_albedo = mix3(lum3(tex), tex, 0.75f) * 0.9f;
    const vec3 albedo_srgb = mix3(land, ocean_albedo, ocean);
    const float bathy = tap_r8<CLAMP>(a.bathy, uu, vv);
    const float emissive = tap_r8<CLAMP>(a.emissive, uu, vv);
    const float alb = v2::srgb_to_spectrum(L, albedo_srgb);
    float thr, Ls;
    cold.get_thr_ls(p.pid, thr, Ls);
    cold.set_thr_ls(p.pid, thr, Ls + thr * emissive * L.night_power);
    cold.set_surf(p.pid, N, alb, ocean, bathy);
    p.P = p.P * fc.offset_scale;                                     // offset_pos
    ray_head(p);                                                     // intersect_land(offset_pos, light_dir) head
    return Q_ST;
}

// the end of a vertex, shared by SCFIN and SURFFIN: roulette (pathtracer.py:448-453), vertex limit, then either the contribution
// or the next vertex's ray
template <typename ColdT>
DE_DEV int vertex_end(const RenderArgs& a, const ColdT& cold, Path& p, int node, int sc, float thr, float Ls, vec3 nd) {
    bool end = false;
    if (sc > 3) {
        const float termination_p = de_max(0.05f, 1.0f - thr);
        if (rng_next(p.rng) < termination_p) end = true;
        else thr /= 1.0f - termination_p;
    }
    sc++;
    if (!end && sc >= 25) end = true;
    if (end) {
        write_contrib(a, p.pid, Ls, node);
        return -1;
    }
    cold.set_thr_ls(p.pid, thr, Ls);
    p.W = nd;
    p.meta = m_make(0, node, sc, 0, 0, 0);
    ray_head(p);
    return Q_ST;
}

// SCFIN: volume vertex — in-scattering, phase sampling (pathtracer.py:394-401).  p.x = (trg, trn, li); W = LD
template <typename ColdT>
DE_DEV int scfin_body(const RenderArgs& a, const FrameConsts& fc, const ColdT& cold, Path& p) {
    const int node = m_node(p.meta), ev_id = m_evid(p.meta);
    const int sc = m_sc(p.meta);
    const LambdaNode& L = a.nodes[node];
    float thr, Ls;
    cold.get_thr_ls(p.pid, thr, Ls);
    const vec3 D = cold.get_dir(p.pid), LD = p.W;
    const float trg = p.x[0], trn = p.x[1], li = p.x[2];
    const float direct_tr = (li > 0.0f) ? 0.0f : trg * trn;
    const float direct_phase = v2::evaluate_phase(fc, D, LD, ev_id, sc > 0);
    Ls += thr * direct_tr * L.sun_irradiance * direct_phase;
    float phase_div_pdf;
    const vec3 nd = v2::sample_phase(fc, p.rng, D, ev_id, sc > 0, &phase_div_pdf);
    thr *= phase_div_pdf;
    return vertex_end(a, cold, p, node, sc, thr, Ls, nd);
}

// SURFFIN: surface vertex — direct light, cosine-weighted bounce (pathtracer.py:431-439).  p.x = (trg, trn, li); P = offset_pos, W = LD
template <typename ColdT>
DE_DEV int surffin_body(const RenderArgs& a, const ColdT& cold, Path& p) {
    const int node = m_node(p.meta);
    const int sc = m_sc(p.meta);
    const LambdaNode& L = a.nodes[node];
    float thr, Ls;
    cold.get_thr_ls(p.pid, thr, Ls);
    const vec3 D = cold.get_dir(p.pid), LD = p.W;
    vec3 N; float alb, ocean, bathy;
    cold.get_surf(p.pid, N, alb, ocean, bathy);
    const float trg = p.x[0], trn = p.x[1], li = p.x[2];
    const float direct_tr = trg * trn;
    const bool visible = li < 0.0f;
    float direct_ndl;
    const float direct_brdf = earth_brdf(alb, ocean, bathy, -D, N, LD, &direct_ndl);
    Ls += thr * direct_tr * (visible ? 1.0f : 0.0f) * L.sun_irradiance * direct_brdf * direct_ndl;
    const vec3 view_dir = -D;
    const vec3 nd = sample_hemisphere_cosine_weighted(p.rng, N);
    float unused;
    const float brdf = earth_brdf(alb, ocean, bathy, view_dir, N, nd, &unused);
    thr *= brdf * (float)M_PI;
    return vertex_end(a, cold, p, node, sc, thr, Ls, nd);
}

}  // namespace wf

// ------------------------------------------------------------------------------------------------ ST: sphere trace of queued rays
// intersect_land's loop (pathtracer.py:37-44), one step per trip; p.x = (t0, atm_x, atm_y, c_ts, c_tm)
template <bool CLAMP>
struct StStage {
    const TexR8 topo;                        // by value: what the loop reads stays in scalar registers for the stage's duration only
    float land_height_scale, escape_r, max_ray_dist;
    float t; int cnt;
    DE_DEV void begin(wf::Path& p) { t = p.x[0]; cnt = 0; }
    // A step in two halves — `issue` up to the map tap's load, `resolve` from the texel on — so that a caller holding two records per lane
    // can have both loads in flight before it consumes either (render_kernel_v5.hip: run_loop2).  step() = the two in a row.
    struct Tok { uint32_t q; float fx, fy, len; };
    DE_DEV bool issue(wf::Path& p, Tok& k) {                                 // true: the record ended in this half (escaped ray)
        const vec3 Q = p.P + p.W * t;
        const float len = length_nr(Q);
        if (cnt <= 100) {                                                    // escaped ray: see render_kernel_v2.hip
            const float b = dot(Q, p.W);
            const float peri2 = (b < 0.0f) ? (len * len - b * b) : len * len;
            if (peri2 > escape_r * escape_r) { t = max_ray_dist; return true; }
        }
        const vec3 n = Q * de_rcp_nr(len);
        const vec2_ uv = sphere_UV_map(n);
        k.q = tap_r8_issue<CLAMP>(topo, fract_(uv.x * 1.0f), fract_(uv.y * 1.0f), &k.fx, &k.fy);
        k.len = len;
        return false;
    }
    DE_DEV bool resolve(wf::Path&, const Tok& k) {
        const float tex = tap_r8_finish(k.q, k.fx, k.fy);
        const float sdf = k.len - DE_PLANET_R - land_height_scale * tex;
        t += sdf;
        cnt++;
        return t > max_ray_dist || de_abs(sdf) < t * 0.0001f || cnt >= 250;
    }
    DE_DEV bool step(wf::Path& p) { Tok k; return issue(p, k) || resolve(p, k); }
    DE_DEV int finish(wf::Path& p) { return wf::after_sphere_trace(p, t, max_ray_dist); }
    // the loop's state between two steps: t goes where begin() reads it (p.x[0]; nothing after the loop reads that word), the rest to spare
    // words of the path's cold record (render_kernel_v6.hip: a wave that is left with a few long records hands them back to the queue;
    // begin() runs first on the resuming lane, then restore()).  pad0 is not used here: render_kernel_v6 keeps the path's id there.
    DE_DEV void suspend(wf::Path& p, wf::Cold* c) const { p.x[0] = t; c->pad1 = wf::asf((uint32_t)cnt); }
    DE_DEV void restore(const wf::Cold* c) { cnt = (int)wf::asu(c->pad1); }
};

// ------------------------------------------------------------------------------------------------ GAS: delta / ratio tracking
// pathtracer.py:91-112 (delta) and :130-141 (ratio) through the Rayleigh / Mie / ozone profiles; p.x = (t, tmax, c_ts, c_tm, X)
struct GasStage {
    const LambdaNode* nodes;
    const float* dens_table;
    float ext_r, ext_m, ext_o, rmaxr, t, tmax, trn;
    vec3 C;
    int kind, event, ev_id;
    DE_DEV void begin(wf::Path& p) {
        const LambdaNode& L = nodes[wf::m_node(p.meta)];
        ext_r = L.ext_r; ext_m = L.ext_m; ext_o = L.ext_o; rmaxr = L.inv_max_ext_rmo;
        kind = wf::m_kind(p.meta);
        t = p.x[0]; tmax = p.x[1];
        C = p.P + t * p.W;
        trn = 1.0f; event = wf::EV_NULL; ev_id = 0;
    }
    DE_DEV bool step(wf::Path& p) {
        const float neg_log = -de_log_unit(rng_next(p.rng));
        const float t_step = neg_log * rmaxr;
        C = C + t_step * p.W;
        t += t_step;
        if (t >= tmax) return true;                                          // the segment ends without an event
        const float len = length_nr(C);
        const float h2 = de_max(len - DE_PLANET_R, 0.0f) * 2.0f;
        vec3 dens;
#ifndef DE_GAS_TABLE
        // The profiles evaluated — what the altitude table holds, entry by entry (aux_kernels.hip: dens_table_kernel; h2 / 2 is the
        // entry's altitude exactly), so the bits are the table's.  The state machine reads the table because it is issue-bound; a
        // single-class gas wave is latency-bound on that read (one dependent load per 60 instructions) and four exponentials cost it
        // less than the wait: the persistent scheduler 220 -> 194 ms per frame, the pipeline's gas kernel -1 ms (round 4).
        (void)h2;
        dens = get_density(de_max(len - DE_PLANET_R, 0.0f));
#else
        if (h2 < (float)DE_DENS_TABLE_N) {
            dens = dens_table_read(dens_table, (uint32_t)(int)h2);
        } else {
            dens = get_density(len - DE_PLANET_R);
        }
#endif
        const vec3 es = v3(ext_r, ext_m, ext_o) * dens;
        const float sum = (es.x + es.y) + es.z;
        // delta tracking (kind 0) draws, ratio tracking (kinds 1, 2) multiplies — as selects, for the reason given in CloudStage::resolve: the draw is made
        // on a copy of the generator and committed for kind-0 lanes only; the rare collision (species walk, scatter / absorb draw) keeps its branch
        const float pr = sum * rmaxr;
        const bool delta = kind == 0;
        Rng g = p.rng;
        const float rand = rng_next(g);
        p.rng.s0 = delta ? g.s0 : p.rng.s0; p.rng.s1 = delta ? g.s1 : p.rng.s1;
        const float trn_r = trn * (1.0f - pr);
        trn = delta ? trn : trn_r;
        bool done = !delta & (trn_r < 1e-5f);
        if (delta & (rand < pr)) {
            int id = 0;
            float cmf = es.x;
            if (!(rand < cmf * rmaxr)) {
                id = 1;
                cmf += es.y;
                if (!(rand < cmf * rmaxr)) {
                    id = 2;
                    cmf += es.z;
                    if (!(rand < cmf * rmaxr)) id = 3;
                }
            }
            const float albedo = (id == 0) ? 1.0f : ((id == 1) ? 0.95f : ((id == 2) ? 0.0f : 0.99f));
            event = (rng_next(p.rng) < albedo) ? wf::EV_SCATTER : wf::EV_ABSORB;
            ev_id = id;
            done = true;
        }
        return done | !(t < tmax);                                           // loop-head test (differs from "no event" only for NaN)
    }
    DE_DEV int finish(wf::Path& p) {
        // rmo_event / rmo_t / ev_id (delta) or trg = trn (ratio), then the cloud-shell segment of the same ray
        if (kind == 0) p.meta = (p.meta & ~((7u << 17) | (63u << 20))) | ((uint32_t)ev_id << 17) | ((uint32_t)event << 20);
        return wf::enter_cloud(p, p.x[2], p.x[3], (kind == 0) ? t : trn, p.x[4]);
    }
    // as StStage's: the running point C is accumulated step by step, so it travels with t and trn (event / ev_id are set by the last step only)
    DE_DEV void suspend(wf::Path& p, wf::Cold* c) const { p.x[0] = t; c->pad1 = trn; c->pad2 = C.x; c->pad3 = C.y; c->pad4 = C.z; }
    DE_DEV void restore(const wf::Cold* c) { trn = c->pad1; C = v3(c->pad2, c->pad3, c->pad4); }
};

// ------------------------------------------------------------------------------------------------ CLOUD: tracking through the cloud shell
// the same loops with get_clouds_density (pathtracer.py:48-65); p.x = (t, tmax, aux = rmo_t | trg, X = isect | li)
template <bool CLAMP>
struct CloudStage {
    const TexR8 clouds;
    float ext_w, rmaxc, t, tmax, trn;
    vec3 C;
    int kind, cloud_event;
#ifndef DE_NO_TAP_REUSE
    uint32_t last_idx = 0xffffffffu, last_q = 0u;      // the lane's previous cloud-map footprint (de_device.h: tap_r8_issue_cached); outlives the record
#endif
    DE_DEV void begin(wf::Path& p) {
        const int sc = wf::m_sc(p.meta);
        ext_w = (sc > 9) ? 0.02f : DE_CLOUDS_EXTINCT;
        rmaxc = (sc > 9) ? (1.0f / (0.02f * DE_CLOUDS_DENSITY)) : (1.0f / (DE_CLOUDS_EXTINCT * DE_CLOUDS_DENSITY));
        kind = wf::m_kind(p.meta);
        t = p.x[0]; tmax = p.x[1];
        C = p.P + t * p.W;
        trn = 1.0f; cloud_event = 0;
    }
    // issue / resolve: as in StStage (the draws keep their order: free flight in `issue`; collision test and event in `resolve`)
    struct Tok { uint32_t q; float fx, fy, h; bool shell; };
    DE_DEV bool issue(wf::Path& p, Tok& k) {                                 // true: the segment ended without a further point
        const float neg_log = -de_log_unit(rng_next(p.rng));
        const float t_step = neg_log * rmaxc;
        C = C + t_step * p.W;
        t += t_step;
        if (t >= tmax) return true;
        const float len = length_nr(C);
        k.shell = len > DE_CLOUDS_LOWER && len < DE_CLOUDS_UPPER;
        if (k.shell) {
            const vec3 nrm = C * de_rcp_nr(len);
            const vec2_ uv = sphere_UV_map(nrm);
#ifndef DE_NO_TAP_REUSE
            k.q = tap_r8_issue_cached<CLAMP>(clouds, fract_(uv.x * 1.0f), fract_(uv.y * 1.0f), &k.fx, &k.fy, last_idx, last_q);
#else
            k.q = tap_r8_issue<CLAMP>(clouds, fract_(uv.x * 1.0f), fract_(uv.y * 1.0f), &k.fx, &k.fy);
#endif
            k.h = (len - DE_CLOUDS_LOWER) * (1.0f / DE_CLOUDS_THICKNESS);
        }
        return false;
    }
    DE_DEV bool resolve(wf::Path& p, const Tok& k) {
        float density = 0.0f;
        if (k.shell) {
            const float tex = tap_r8_finish(k.q, k.fx, k.fy);
            const float h = k.h;
            const float split = 0.2f;
            density = ((h - split < tex * (float)(1.0 - 0.2)) & (split - h < tex * split)) ? de_max(tex, 0.4f) : 0.0f;      // both tests always: no short-circuit branch
        }
        const float es = ext_w * (density * DE_CLOUDS_DENSITY);
        // delta tracking (kind 0) draws, ratio tracking (kinds 1, 2) multiplies: the same statements as a literal `if (kind == 0) ... else ...`, written as
        // selects — a wave of this stage holds both kinds, so both sides run either way, and the divergent form pays a dozen scalar instructions of
        // exec-mask bookkeeping per step on top (this kernel pays for instructions, scalar ones included).  The draw is made on a copy of the generator
        // and committed for kind-0 lanes only; the rare collision keeps its branch.
        const float pr = es * rmaxc;
        const bool delta = kind == 0;
        Rng g = p.rng;
        const float rand = rng_next(g);
        p.rng.s0 = delta ? g.s0 : p.rng.s0; p.rng.s1 = delta ? g.s1 : p.rng.s1;
        const float trn_r = trn * (1.0f - pr);
        trn = delta ? trn : trn_r;
        bool done = !delta & (trn_r < 1e-5f);
        if (delta & (rand < pr)) { cloud_event = (rng_next(p.rng) < 0.99f) ? wf::EV_SCATTER : wf::EV_ABSORB; done = true; }
        return done | !(t < tmax);
    }
    DE_DEV bool step(wf::Path& p) { Tok k; return issue(p, k) || resolve(p, k); }
    DE_DEV int finish(wf::Path& p) {
        const float aux = p.x[2], X = p.x[3];
        if (kind == 0) {
            p.meta |= ((uint32_t)cloud_event << 4) << 20;
            p.x[0] = t; p.x[1] = aux; p.x[2] = X;                            // EVENT: cloud_t, rmo_t, isect
            return wf::Q_EVENT;
        }
        p.x[0] = aux; p.x[1] = trn; p.x[2] = X;                              // SCFIN / SURFFIN: trg, trn, li
        return (kind == 1) ? wf::Q_SCFIN : wf::Q_SURFFIN;
    }
    DE_DEV void suspend(wf::Path& p, wf::Cold* c) const { p.x[0] = t; c->pad1 = trn; c->pad2 = C.x; c->pad3 = C.y; c->pad4 = C.z; }
    DE_DEV void restore(const wf::Cold* c) { trn = c->pad1; C = v3(c->pad2, c->pad3, c->pad4); }
};
