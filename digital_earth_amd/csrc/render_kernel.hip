// render_kernel.hip — the hot path: Renderer.render (renderer.py:283-330) with pathtracer.path_tracer
// (pathtracer.py:316-469) and its alternative ray_marcher (pathtracer.py:544-685), for gfx950.
//
// Work decomposition: one 64-lane wavefront owns one 8x8-pixel tile, one lane owns one pixel and traces that
// pixel's samples in order, so `color_buffer[u, v] += ...` (renderer.py:330) keeps the reference's association
// ((c + s0) + s1) + ... whatever the launch shape or the number of GPUs.  The RGB sum lives in registers for the
// whole launch: one read-modify-write of the HDR buffer per pixel per launch instead of one per sample.
//
// Differences from a literal transcription, all value-preserving under the arithmetic contract:
//  * everything that depends only on the wavelength comes from the LambdaNode table (setup_kernel);
//  * tracking through Rayleigh/Mie/ozone does not tap the cloud map and tracking through the cloud shell does not
//    evaluate the gas profiles: the reference multiplies those terms by an extinction of exactly 0
//    (pathtracer.py:185,197), and x + 0*finite == x bit for bit;
//  * |pos| is computed once per evaluation point and shared by the SDF, the normalisation and the elevation.
#include "de_kernels.h"

namespace {

enum { NULL_EVENT = 0, ABSORB_EVENT = 1, SCATTER_EVENT = 2 };

struct Work { uint32_t taps_r8, taps_rgb, sphere_steps, tracking_steps, vertices; };

// TABLE: gas densities from the altitude table (de_kernels.h DE_DENS_TABLE_N) instead of four exponentials per point —
// the ray marcher evaluates 64 x 17 points per bounce.  Same values by construction (the table holds get_density itself).
template <bool CLAMP, bool TABLE = false>
struct Tracer {
    const RenderArgs& a;
    const FrameConsts& fc;
    Work& wk;
    __device__ Tracer(const RenderArgs& a_, const FrameConsts& fc_, Work& wk_) : a(a_), fc(fc_), wk(wk_) {}

    // pathtracer.py:11-14
    DE_DEV float land_sdf(vec3 pos) {
        float len = length(pos);
        vec3 n = pos * (1.0f / len);            // normalized() under contract 2
        wk.taps_r8++;
        return len - DE_PLANET_R - fc.land_height_scale * sphere_tap_r8<CLAMP>(a.topo, n);
    }
    // pathtracer.py:16-25
    DE_DEV vec3 land_normal(vec3 pos) {
        float d = land_sdf(pos);
        float e = fc.normal_eps;
        vec3 n = v3(d - land_sdf(pos - v3(e, 0.0f, 0.0f)), d - land_sdf(pos - v3(0.0f, e, 0.0f)), d - land_sdf(pos - v3(0.0f, 0.0f, e)));
        return normalized(n);
    }
    // pathtracer.py:27-46
    DE_DEV float intersect_land(vec3 pos, vec3 dir) {
        float ray_dist = 0.0f;
        const float max_ray_dist = (float)(6371e3 * 10.0);
        vec2_ rsi_dist = rsi(pos, dir, DE_ATMOS_UPPER);
        if (rsi_dist.x > 0.0f) ray_dist = rsi_dist.x;
        for (int i = 0; i < 250; ++i) {
            vec3 ro = pos + dir * ray_dist;
            float dist = land_sdf(ro);
            ray_dist += dist;
            wk.sphere_steps++;
            if (ray_dist > max_ray_dist || de_abs(dist) < ray_dist * 0.0001f) break;
        }
        return (ray_dist < max_ray_dist) ? ray_dist : -1.0f;
    }
    // pathtracer.py:48-65
    DE_DEV float get_clouds_density(vec3 pos) {
        float r = length(pos);
        float density = 0.0f;
        if (r > DE_CLOUDS_LOWER && r < DE_CLOUDS_UPPER) {
            float h = (r - DE_CLOUDS_LOWER) * (1.0f / DE_CLOUDS_THICKNESS);
            wk.taps_r8++;
            float cloud_texture = sphere_tap_r8<CLAMP>(a.clouds, pos * (1.0f / r));
            float column_height = cloud_texture;
            const float split = 0.2f;
            density = (h - split < column_height * (float)(1.0 - 0.2) && split - h < column_height * split) ? de_max(cloud_texture, 0.4f) : 0.0f;
        }
        return density * DE_CLOUDS_DENSITY;
    }
    DE_DEV vec3 gas_density(vec3 pos) {   // volume.get_density(volume.get_elevation(pos))
        if (TABLE) {
            // |pos| in [2^22, 2^23) m has a spacing of 0.5 m: h is an exact multiple of 0.5 and 2h is the table index
            const float len = de_sqrt_nr((pos.x * pos.x + pos.y * pos.y) + pos.z * pos.z);
            const float h2 = de_max(len - DE_PLANET_R, 0.0f) * 2.0f;
            if (h2 < (float)DE_DENS_TABLE_N) {
                return dens_table_read(a.dens_table, (uint32_t)(int)h2);
            }
            return get_density(len - DE_PLANET_R);
        }
        return get_density(de_sqrt((pos.x * pos.x + pos.y * pos.y) + pos.z * pos.z) - DE_PLANET_R);
    }
    // pathtracer.py:77-115 with extinctions = (r, m, o, 0)
    DE_DEV void delta_tracking_rmo(Rng& rng, vec3 ray_pos, vec3 ray_dir, float t_start, float t_max, vec3 ext, float max_ext,
                                   int* event_out, float* t_out, int* id_out) {
        float t = t_start;
        const float inv_max = 1.0f / max_ext;      // contract 2: quotients by the majorant are products with RN(1 / majorant)
        ray_pos = ray_pos + t * ray_dir;
        int id = 0;
        int event = NULL_EVENT;
        while (t < t_max) {
            float t_step = -de_log_unit(rng_next(rng)) * inv_max;
            ray_pos = ray_pos + t_step * ray_dir;
            t += t_step;
            wk.tracking_steps++;
            if (t >= t_max) break;
            vec3 es = ext * gas_density(ray_pos);
            float rand = rng_next(rng);
            float sum = (es.x + es.y) + es.z;
            if (rand < sum * inv_max) {
                float cmf = es.x;
                if (!(rand < cmf * inv_max)) {
                    id = 1;
                    cmf += es.y;
                    if (!(rand < cmf * inv_max)) {
                        id = 2;
                        cmf += es.z;
                        if (!(rand < cmf * inv_max)) id = 3;
                    }
                }
                const float albedo = (id == 0) ? 1.0f : ((id == 1) ? 0.95f : ((id == 2) ? 0.0f : 0.99f));   // pathtracer.py:263-270
                event = (rng_next(rng) < albedo) ? SCATTER_EVENT : ABSORB_EVENT;
                break;
            }
        }
        *event_out = event; *t_out = t; *id_out = id;
    }
    // pathtracer.py:77-115 with extinctions = (0, 0, 0, w): the species walk always ends at id 3
    DE_DEV void delta_tracking_cloud(Rng& rng, vec3 ray_pos, vec3 ray_dir, float t_start, float t_max, float ext_w, float max_ext,
                                     int* event_out, float* t_out) {
        float t = t_start;
        const float inv_max = 1.0f / max_ext;      // contract 2: quotients by the majorant are products with RN(1 / majorant)
        ray_pos = ray_pos + t * ray_dir;
        int event = NULL_EVENT;
        while (t < t_max) {
            float t_step = -de_log_unit(rng_next(rng)) * inv_max;
            ray_pos = ray_pos + t_step * ray_dir;
            t += t_step;
            wk.tracking_steps++;
            if (t >= t_max) break;
            float es = ext_w * get_clouds_density(ray_pos);
            float rand = rng_next(rng);
            if (rand < es * inv_max) {
                event = (rng_next(rng) < 0.99f) ? SCATTER_EVENT : ABSORB_EVENT;
                break;
            }
        }
        *event_out = event; *t_out = t;
    }
    // pathtracer.py:117-143, gas part
    DE_DEV float ratio_tracking_rmo(Rng& rng, vec3 ray_pos, vec3 ray_dir, float t_start, float t_max, vec3 ext, float max_ext) {
        float t = t_start;
        const float inv_max = 1.0f / max_ext;      // contract 2: quotients by the majorant are products with RN(1 / majorant)
        ray_pos = ray_pos + t * ray_dir;
        float tr = 1.0f;
        while (t < t_max) {
            float t_step = -de_log_unit(rng_next(rng)) * inv_max;
            ray_pos = ray_pos + t_step * ray_dir;
            t += t_step;
            wk.tracking_steps++;
            if (t >= t_max) break;
            vec3 es = ext * gas_density(ray_pos);
            tr *= 1.0f - ((es.x + es.y) + es.z) * inv_max;
            if (tr < 1e-5f) break;
        }
        return tr;
    }
    // pathtracer.py:117-143, cloud part
    DE_DEV float ratio_tracking_cloud(Rng& rng, vec3 ray_pos, vec3 ray_dir, float t_start, float t_max, float ext_w, float max_ext) {
        float t = t_start;
        const float inv_max = 1.0f / max_ext;      // contract 2: quotients by the majorant are products with RN(1 / majorant)
        ray_pos = ray_pos + t * ray_dir;
        float tr = 1.0f;
        while (t < t_max) {
            float t_step = -de_log_unit(rng_next(rng)) * inv_max;
            ray_pos = ray_pos + t_step * ray_dir;
            t += t_step;
            wk.tracking_steps++;
            if (t >= t_max) break;
            float es = ext_w * get_clouds_density(ray_pos);
            tr *= 1.0f - es * inv_max;
            if (tr < 1e-5f) break;
        }
        return tr;
    }
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
    // pathtracer.py:172-207
    DE_DEV void sample_interaction(Rng& rng, vec3 ray_pos, vec3 ray_dir, float land_isection, vec3 ext, float ext_w,
                                   float max_ext_rmo, float max_ext_cloud, int* event_out, float* t_out, int* id_out) {
        vec2_ atmos = rsi(ray_pos, ray_dir, DE_ATMOS_UPPER);
        float t_start = de_max(0.0f, atmos.x);
        float t_max = (land_isection >= 0.0f) ? land_isection : atmos.y;
        if (atmos.y < 0.0f) t_max = -1.0f;
        int rmo_event, rmo_id;
        float rmo_t;
        delta_tracking_rmo(rng, ray_pos, ray_dir, t_start, t_max, ext, max_ext_rmo, &rmo_event, &rmo_t, &rmo_id);
        intersect_cloud_limits(ray_pos, ray_dir, land_isection, &t_start, &t_max);
        int event = rmo_event;
        float t = rmo_t;
        int id = rmo_id;
        if (rmo_event == NULL_EVENT || rmo_t > t_start) {
            int cloud_event;
            float cloud_t;
            delta_tracking_cloud(rng, ray_pos, ray_dir, t_start, t_max, ext_w, max_ext_cloud, &cloud_event, &cloud_t);
            if (cloud_event > 0 && (cloud_t < rmo_t || rmo_event == NULL_EVENT)) {
                t = cloud_t;
                id = CLOUD_ID;
                event = cloud_event;
            }
        }
        *event_out = event; *t_out = t; *id_out = id;
    }
    // pathtracer.py:211-232
    DE_DEV float sample_transmittance(Rng& rng, vec3 ray_pos, vec3 ray_dir, float land_isection, vec3 ext, float ext_w,
                                      float max_ext_rmo, float max_ext_cloud) {
        vec2_ atmos = rsi(ray_pos, ray_dir, DE_ATMOS_UPPER);
        float t_start = de_max(0.0f, atmos.x);
        float t_max = (land_isection >= 0.0f) ? land_isection : atmos.y;
        if (atmos.y < 0.0f) t_max = -1.0f;
        float tr = ratio_tracking_rmo(rng, ray_pos, ray_dir, t_start, t_max, ext, max_ext_rmo);
        intersect_cloud_limits(ray_pos, ray_dir, land_isection, &t_start, &t_max);
        tr *= ratio_tracking_cloud(rng, ray_pos, ray_dir, t_start, t_max, ext_w, max_ext_cloud);
        return tr;
    }
    // pathtracer.py:235-247
    DE_DEV float evaluate_phase(vec3 ray_dir, vec3 light_dir, int id, bool reduce_peak) {
        float phase = 0.0f;
        float c = dot(ray_dir, light_dir);
        if (id == RAYLEIGH_ID) phase += rayleigh_phase(c);
        else if (id == MIE_ID) phase += klein_nishina_phase(c, DE_MIE_ASYMMETRY, fc.kn_log);
        else if (id == CLOUD_ID) phase += cloud_phase(fc.cloud, c, reduce_peak);
        else if (id == ISOTROPIC_CLOUD_ID) phase += (float)(1.0 / (4.0 * M_PI));
        return phase;
    }
    // pathtracer.py:249-261
    DE_DEV vec3 sample_phase(Rng& rng, vec3 ray_dir, int id, bool reduce_peak, float* phase_div_pdf) {
        *phase_div_pdf = 1.0f;
        if (id == RAYLEIGH_ID || id == ISOTROPIC_CLOUD_ID) {
            float r0 = rng_next(rng);
            float r1 = rng_next(rng);
            vec3 d = sample_sphere(r0, r1);
            *phase_div_pdf = evaluate_phase(ray_dir, d, id, reduce_peak) * (float)(4.0 * M_PI);
            return d;
        } else if (id == MIE_ID) {
            return sample_klein_nishina_phase(rng, ray_dir, DE_MIE_ASYMMETRY);
        }
        return sample_cloud_phase(fc.cloud, rng, ray_dir, reduce_peak);
    }
    DE_DEV float srgb_to_spectrum(const LambdaNode& L, vec3 rgb) {   // lib/colour.py:62-71 with the per-wavelength part tabulated
        return (L.s2s_valid != 0.0f) ? dot(rgb, v3(L.c0, L.c1, L.c2)) : 0.0f;
    }
    // pathtracer.py:284-313
    DE_DEV void get_land_material(vec3 pos, vec3* albedo_srgb, float* ocean_out, float* bathy_out, float* emissive_out) {
        vec3 n = normalized(pos);
        vec2_ uv = sphere_UV_map(n);
        float u = fract_(uv.x * 1.0f), v = fract_(uv.y * 1.0f);
        float ocean = tap_r8<CLAMP>(a.ocean, u, v);
        vec3 tex = tap_rgb<CLAMP>(a.albedo, u, v);
        vec3 land = mix3(lum3(tex), tex, 6.5f);
        float greenery = sqr(land.y / lum(land));
        greenery = smoothstep_(1.5f, 1.9f, greenery);
        land = (1.0f * tex) / (greenery * 0.7f + 1.0f);
        land = mix3(lum3(land), land, 1.4f - greenery * 0.45f);
        land = mix3(land, (land * v3(255.0f, 128.0f, 64.0f)) / 255.0f, 0.2f * (1.0f - greenery));
        vec3 ocean_albedo = mix3(lum3(tex), tex, 0.75f) * 0.9f;
        *albedo_srgb = mix3(land, ocean_albedo, ocean);
        *ocean_out = ocean;
        *bathy_out = tap_r8<CLAMP>(a.bathy, u, v);
        *emissive_out = tap_r8<CLAMP>(a.emissive, u, v);
        wk.taps_r8 += 3;
        wk.taps_rgb += 1;
    }
    // pathtracer.py:316-469
    DE_DEV float path_tracer(Rng& rng, const LambdaNode& L, vec3 ray_pos, vec3 ray_dir) {
        const vec3 primary_dir = ray_dir;
        vec3 ext = v3(L.ext_r, L.ext_m, L.ext_o);
        float ext_w = DE_CLOUDS_EXTINCT;
        bool primary_miss = false;
        float in_scattering = 0.0f;
        float throughput = 1.0f;
        for (int scatter_count = 0; scatter_count < 25; ++scatter_count) {
            wk.vertices++;
            if (scatter_count > 9) ext_w = 0.02f;
            float max_ext_rmo = L.max_ext_rmo;
            float max_ext_cloud = ext_w * DE_CLOUDS_DENSITY;
            float earth_intersection = intersect_land(ray_pos, ray_dir);
            int event, id;
            float interaction_dist;
            sample_interaction(rng, ray_pos, ray_dir, earth_intersection, ext, ext_w, max_ext_rmo, max_ext_cloud, &event, &interaction_dist, &id);
            if (scatter_count > 9 && id == CLOUD_ID) id = ISOTROPIC_CLOUD_ID;
            vec3 light_dir = tangent_space_apply(fc.light_dir, sample_cone(rng, fc.sun_cos_angle));
            if (event == ABSORB_EVENT) {
                break;
            } else if (event == SCATTER_EVENT) {
                vec3 ipos = ray_pos + interaction_dist * ray_dir;
                bool blocked = rsi(ipos, light_dir, DE_PLANET_R).y > 0.0f;
                float direct_tr = 0.0f;
                if (!blocked) direct_tr = sample_transmittance(rng, ipos, light_dir, -1.0f, ext, ext_w, max_ext_rmo, max_ext_cloud);
                float direct_phase = evaluate_phase(ray_dir, light_dir, id, scatter_count > 0);
                in_scattering += throughput * direct_tr * L.sun_irradiance * direct_phase;
                float phase_div_pdf;
                vec3 scatter_dir = sample_phase(rng, ray_dir, id, scatter_count > 0, &phase_div_pdf);
                ray_dir = scatter_dir;
                ray_pos = ipos;
                throughput *= phase_div_pdf;
            } else if (earth_intersection > 0.0f) {
                vec3 land_pos = ray_pos + ray_dir * earth_intersection;
                vec3 land_n = land_normal(land_pos);
                vec3 albedo_srgb;
                float ocean, bathy, emissive;
                get_land_material(land_pos, &albedo_srgb, &ocean, &bathy, &emissive);
                float albedo = srgb_to_spectrum(L, albedo_srgb);
                in_scattering += throughput * emissive * L.night_power;
                vec3 offset_pos = land_pos * fc.offset_scale;
                bool visible = intersect_land(offset_pos, light_dir) < 0.0f;
                float direct_tr = sample_transmittance(rng, offset_pos, light_dir, visible ? -1.0f : 0.0f, ext, ext_w, max_ext_rmo, max_ext_cloud);
                float direct_ndl;
                float direct_brdf = earth_brdf(albedo, ocean, bathy, -ray_dir, land_n, light_dir, &direct_ndl);
                in_scattering += throughput * direct_tr * (visible ? 1.0f : 0.0f) * L.sun_irradiance * direct_brdf * direct_ndl;
                vec3 view_dir = -ray_dir;
                ray_dir = sample_hemisphere_cosine_weighted(rng, land_n);
                ray_pos = offset_pos;
                float unused;
                float brdf = earth_brdf(albedo, ocean, bathy, view_dir, land_n, ray_dir, &unused);
                throughput *= brdf * (float)M_PI;
            } else {
                if (scatter_count == 0) primary_miss = true;
                break;
            }
            if (scatter_count > 3) {
                float termination_p = de_max(0.05f, 1.0f - throughput);
                if (rng_next(rng) < termination_p) break;
                throughput /= 1.0f - termination_p;
            }
        }
        if (primary_miss) {
            if (dot(fc.light_dir, primary_dir) > fc.sun_cos_angle) in_scattering += L.sun_power;
            wk.taps_rgb += 1;
            vec3 stars_srgb = sphere_tap_rgb<CLAMP>(a.stars, normalized(primary_dir));
            float stars_power = srgb_to_spectrum(L, stars_srgb);
            in_scattering += stars_power * L.sun_power * 0.0000001f;
        }
        if (__builtin_isinf(in_scattering) || in_scattering != in_scattering || in_scattering < 0.0f) in_scattering = 0.0f;
        return in_scattering;
    }

    // ---------------------------------------------------------------- ray marcher, pathtracer.py:471-685
    DE_DEV float ray_march_transmittance(vec3 ray_pos, vec3 ray_dir, vec3 rmo_ext) {
        const int steps = 16;
        const float r_steps = 1.0f / (float)steps;
        float tr = 0.0f;
        bool blocked = rsi(ray_pos, ray_dir, DE_PLANET_R).y > 0.0f;
        if (!blocked) {
            vec2_ atmos = rsi(ray_pos, ray_dir, DE_ATMOS_UPPER);
            float t_max = atmos.y;
            if (atmos.y < 0.0f) t_max = -1.0f;
            float dd = t_max * r_steps;
            vec3 ray_step = ray_dir * dd;
            vec3 od = v3(0.0f, 0.0f, 0.0f);
            for (int i = 0; i < steps; ++i) {
                vec3 density = gas_density(ray_pos);
                od = od + density * dd;
                ray_pos = ray_pos + ray_step;
            }
            tr = de_exp(-dot(rmo_ext, od));
        }
        return tr;
    }
    DE_DEV void ray_march_atmos(vec3 ray_pos, vec3 ray_dir, float t_start, float t_max, vec3 sun_dir, vec3 rmo_ext, float sc_r, float sc_m,
                                float* in_scatter_out, float* tr_out) {
        const int steps = 64;
        const float r_steps = 1.0f / (float)steps;
        float dd = (t_max - t_start) * r_steps;
        vec3 ray_step = ray_dir * dd;
        ray_pos = ray_pos + ray_dir * t_start;
        float c = dot(ray_dir, sun_dir);
        float ph_r = rayleigh_phase(c), ph_m = klein_nishina_phase(c, DE_MIE_ASYMMETRY, fc.kn_log);
        float tr = 1.0f, in_scatter = 0.0f;
        for (int i = 0; i < steps; ++i) {
            vec3 density = gas_density(ray_pos);
            float step_od = dot(rmo_ext, density * dd);
            float step_tr = de_saturate(de_exp(-step_od));
            float step_integral = de_saturate((1.0f - step_tr) / step_od);
            float visible = tr * step_integral;
            float sun_tr = ray_march_transmittance(ray_pos, sun_dir, rmo_ext);
            float step_scattering = sc_r * (density.x * ph_r) + sc_m * (density.y * ph_m);
            in_scatter += step_scattering * sun_tr * visible * dd;
            tr *= step_tr;
            ray_pos = ray_pos + ray_step;
        }
        *in_scatter_out = in_scatter; *tr_out = tr;
    }
    DE_DEV float ray_marcher(Rng& rng, const LambdaNode& L, vec3 ray_pos, vec3 ray_dir) {
        const vec3 primary_dir = ray_dir;
        vec3 ext = v3(L.ext_r, L.ext_m, L.ext_o);
        float sc_r = ext.x * 1.0f, sc_m = ext.y * 0.95f;
        bool primary_miss = false;
        float accum = 0.0f, throughput = 1.0f;
        for (int scatter_count = 0; scatter_count < 3; ++scatter_count) {
            wk.vertices++;
            float earth_intersection = intersect_land(ray_pos, ray_dir);
            vec2_ atmos = rsi(ray_pos, ray_dir, DE_ATMOS_UPPER);
            float t_start = de_max(0.0f, atmos.x);
            float t_max = (earth_intersection > 0.0f) ? earth_intersection : atmos.y;
            if (atmos.y < 0.0f) {
                primary_miss = (scatter_count == 0);
                break;
            }
            vec3 light_dir = tangent_space_apply(fc.light_dir, sample_cone(rng, fc.sun_cos_angle));
            float in_scatter, tr;
            ray_march_atmos(ray_pos, ray_dir, t_start, t_max, light_dir, ext, sc_r, sc_m, &in_scatter, &tr);
            accum += throughput * in_scatter;
            throughput *= tr;
            if (earth_intersection > 0.0f) {
                vec3 land_pos = ray_pos + ray_dir * earth_intersection;
                vec3 land_n = land_normal(land_pos);
                vec3 albedo_srgb;
                float ocean, bathy, emissive;
                get_land_material(land_pos, &albedo_srgb, &ocean, &bathy, &emissive);
                float albedo = srgb_to_spectrum(L, albedo_srgb);
                accum += throughput * emissive * L.night_power;
                vec3 offset_pos = land_pos * fc.offset_scale;
                bool visible = intersect_land(offset_pos, light_dir) < 0.0f;
                float direct_tr = 1.0f;
                float direct_ndl;
                float direct_brdf = earth_brdf(albedo, ocean, bathy, -ray_dir, land_n, light_dir, &direct_ndl);
                accum += throughput * direct_tr * (visible ? 1.0f : 0.0f) * L.sun_irradiance * direct_brdf * direct_ndl;
                vec3 view_dir = -ray_dir;
                ray_dir = sample_hemisphere_cosine_weighted(rng, land_n);
                ray_pos = offset_pos;
                float unused;
                float brdf = earth_brdf(albedo, ocean, bathy, view_dir, land_n, ray_dir, &unused);
                throughput *= brdf * (float)M_PI;
            }
        }
        if (primary_miss) {
            if (dot(fc.light_dir, primary_dir) > fc.sun_cos_angle) accum += L.sun_power;
            wk.taps_rgb += 1;
            vec3 stars_srgb = sphere_tap_rgb<CLAMP>(a.stars, normalized(primary_dir));
            float stars_power = srgb_to_spectrum(L, stars_srgb);
            accum += stars_power * L.sun_power * 0.0000001f;
        }
        if (__builtin_isinf(accum) || accum != accum || accum < 0.0f) accum = 0.0f;
        return accum;
    }
};

// spectrum_sample's bisection (lib/colour.py:21-39) on the tabulated node values; returns the heap index of the node
DE_DEV int spectrum_node(const float* node_val, float sample) {
    int n = 1;
    for (int it = 0; it < 8; ++it) {
        float val = node_val[n];
        if (val < sample) n = 2 * n + 1;
        else if (val > sample) n = 2 * n;
        else break;
    }
    return n;
}

// renderer.py:269-279
DE_DEV vec3 get_cast_dir(Rng& rng, const FrameConsts& fc, int H, int u, int v) {
    float fov = fc.fov;
    float fu = (2.0f * fov * ((float)u + rng_next(rng)) / (float)H - fov * fc.aspect_ratio - 1e-5f) * fc.aspect_scale;
    float fv = 2.0f * fov * ((float)v + rng_next(rng)) / (float)H - fov - 1e-5f;
    return normalized(fc.d + fu * fc.du + fv * fc.dv);
}

}  // namespace

// MODE 0: accumulate; 1: accumulate + work counters; 2: trace one sample per pixel into debug_out (no accumulation)
template <bool CLAMP, bool MARCH, int MODE>
__global__ void __launch_bounds__(256) render_kernel(RenderArgs a) {
    const int wave = (int)(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (wave >= a.n_tiles) return;
    const uint32_t tile = a.tiles[wave];
    const int lane = threadIdx.x & 63;
    const int u = (int)(tile % (uint32_t)a.tiles_x) * 8 + (lane & 7);
    const int v = (int)(tile / (uint32_t)a.tiles_x) * 8 + (lane >> 3);
    const uint32_t pixel = (uint32_t)(v * a.W + u);
    const FrameConsts& fc = *a.fc;
    Work wk = {0, 0, 0, 0, 0};
    Tracer<CLAMP> tr(a, fc, wk);
    float* px = a.hdr + (size_t)pixel * 3;
    float acc_r = 0.0f, acc_g = 0.0f, acc_b = 0.0f;
    if (MODE != 2) { acc_r = px[0]; acc_g = px[1]; acc_b = px[2]; }
    uint32_t draws = 0;
    for (int s = 0; s < a.spp_count; ++s) {
        Rng rng;
        rng_seed(rng, a.seed_lo, a.seed_hi, pixel, (uint32_t)(a.spp_begin + s * a.spp_stride));
        int node = 0;
        if (!a.fixed_wavelength) node = spectrum_node(a.node_val, rng_next(rng));
        const LambdaNode L = a.nodes[node];
        vec3 ray_dir = get_cast_dir(rng, fc, a.H, u, v);
        float sample = MARCH ? tr.ray_marcher(rng, L, fc.cam_pos, ray_dir) : tr.path_tracer(rng, L, fc.cam_pos, ray_dir);
        vec3 xyz = (sample * v3(L.rx, L.ry, L.rz)) * L.rcp_pdf;
        vec3 rgb = xyz_to_rgb_d65(xyz);
        acc_r += rgb.x; acc_g += rgb.y; acc_b += rgb.z;
        draws += rng.draws;
        if (MODE == 2) {
            float* q = a.debug_out + (size_t)pixel * 4;
            q[0] = sample; q[1] = L.wavelength; q[2] = (float)rng.draws; q[3] = (float)wk.vertices;
        }
    }
    if (MODE != 2) { px[0] = acc_r; px[1] = acc_g; px[2] = acc_b; }
    if (MODE == 1) {
        atomicAdd(&a.counters[0], (unsigned long long)a.spp_count);
        atomicAdd(&a.counters[1], (unsigned long long)wk.taps_r8);
        atomicAdd(&a.counters[2], (unsigned long long)wk.taps_rgb);
        atomicAdd(&a.counters[3], (unsigned long long)wk.sphere_steps);
        atomicAdd(&a.counters[4], (unsigned long long)wk.tracking_steps);
        atomicAdd(&a.counters[5], (unsigned long long)wk.vertices);
        atomicAdd(&a.counters[6], (unsigned long long)draws);
    }
}


// ray_marcher (pathtracer.py:471-685) as its own kernel: the deterministic 64 x 16 march has fixed trip counts, so one
// lane = one pixel keeps the wave coherent through the march (only the sphere trace of intersect_land diverges); the
// gas densities come from the altitude table, the per-pixel RGB sum stays in registers over the launch's samples.
template <bool CLAMP, int MODE>
__global__ void __launch_bounds__(256) ray_march_kernel(RenderArgs a) {
    const int wave = (int)(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (wave >= a.n_tiles) return;
    const uint32_t tile = a.tiles[wave];
    const int lane = threadIdx.x & 63;
    const int u = (int)(tile % (uint32_t)a.tiles_x) * 8 + (lane & 7);
    const int v = (int)(tile / (uint32_t)a.tiles_x) * 8 + (lane >> 3);
    const uint32_t pixel = (uint32_t)(v * a.W + u);
    const FrameConsts& fc = *a.fc;
    Work wk = {0, 0, 0, 0, 0};
    Tracer<CLAMP, true> tr(a, fc, wk);
    float* px = a.hdr + (size_t)pixel * 3;
    float acc_r = 0.0f, acc_g = 0.0f, acc_b = 0.0f;
    if (MODE != 2) { acc_r = px[0]; acc_g = px[1]; acc_b = px[2]; }
    uint32_t draws = 0;
    for (int s = 0; s < a.spp_count; ++s) {
        Rng rng;
        rng_seed(rng, a.seed_lo, a.seed_hi, pixel, (uint32_t)(a.spp_begin + s * a.spp_stride));
        int node = 0;
        if (!a.fixed_wavelength) node = spectrum_node(a.node_val, rng_next(rng));
        const LambdaNode L = a.nodes[node];
        vec3 ray_dir = get_cast_dir(rng, fc, a.H, u, v);
        float sample = tr.ray_marcher(rng, L, fc.cam_pos, ray_dir);
        vec3 xyz = (sample * v3(L.rx, L.ry, L.rz)) * L.rcp_pdf;
        vec3 rgb = xyz_to_rgb_d65(xyz);
        acc_r += rgb.x; acc_g += rgb.y; acc_b += rgb.z;
        draws += rng.draws;
        if (MODE == 2) {
            float* q = a.debug_out + (size_t)pixel * 4;
            q[0] = sample; q[1] = L.wavelength; q[2] = (float)rng.draws; q[3] = (float)wk.vertices;
        }
    }
    if (MODE != 2) { px[0] = acc_r; px[1] = acc_g; px[2] = acc_b; }
    if (MODE == 1) {
        atomicAdd(&a.counters[0], (unsigned long long)a.spp_count);
        atomicAdd(&a.counters[1], (unsigned long long)wk.taps_r8);
        atomicAdd(&a.counters[2], (unsigned long long)wk.taps_rgb);
        atomicAdd(&a.counters[3], (unsigned long long)wk.sphere_steps);
        atomicAdd(&a.counters[4], (unsigned long long)wk.tracking_steps);
        atomicAdd(&a.counters[5], (unsigned long long)wk.vertices);
        atomicAdd(&a.counters[6], (unsigned long long)draws);
    }
}
