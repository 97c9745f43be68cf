// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// oracle_pathtracer.h — restatement of /root/reference/pathtracer.py:11-469 (path_tracer, live) and
// :471-685 (ray_marcher, dead in the reference but named by BASELINE.json's north_star).
// Statement order, expression order and RNG draw order follow the reference.
#pragma once
#include "oracle_lib.h"

namespace deo {

struct PathParameters { float wavelength; vec3 ray_dir; vec3 ray_pos; };                  // lib/parameters.py:4-8
struct SceneParameters { vec3 light_direction; float sun_cos_angle, sun_angular_radius, land_height_scale; };  // :10-15

struct Scene {
    const Texture* albedo; const Texture* height; const Texture* ocean; const Texture* clouds;
    const Texture* bathymetry; const Texture* emissive; const Texture* stars;
    const float* srgb_to_spectrum;  // 300 x 3
    const float* o3_crossec;        // 441
    int topo_res;                   // TOPOGRAPHY_TEX_RES[0]
};
struct WorkCounters { uint64_t sphere_steps = 0, tracking_steps = 0, vertices = 0; };
static thread_local WorkCounters g_work;

// pathtracer.py:11-14
static inline float land_sdf(const Texture& heightmap, vec3 pos, float scale) {
    return length(pos) - volume::planet_r - scale * sample_sphere_texture_r(heightmap, pos);
}
// :16-25
static inline vec3 land_normal(const Texture& heightmap, int topo_res, vec3 pos, float scale) {
    float d = land_sdf(heightmap, pos, scale);
    float e = (float)(M_PI * 6371e3 / (double)topo_res);
    vec3 n(d - land_sdf(heightmap, pos - vec3(e, 0.0f, 0.0f), scale),
           d - land_sdf(heightmap, pos - vec3(0.0f, e, 0.0f), scale),
           d - land_sdf(heightmap, pos - vec3(0.0f, 0.0f, e), scale));
    return normalized(n);
}
// :27-46
static inline float intersect_land(const Texture& heightmap, vec3 pos, vec3 dir, float height_scale) {
    float ray_dist = 0.0f;
    const float max_ray_dist = (float)(6371e3 * 10.0);
    vec2 rsi_dist = rsi(pos, dir, volume::atmos_upper_limit);
    if (rsi_dist.x > 0.0f) ray_dist = rsi_dist.x;
    for (int i = 0; i < 250; ++i) {
        vec3 ro = pos + dir * ray_dist;
        float dist = land_sdf(heightmap, ro, height_scale);
        ray_dist += dist;
        g_work.sphere_steps++;
        if (ray_dist > max_ray_dist || fabsf(dist) < ray_dist * 0.0001f) break;
    }
    return (ray_dist < max_ray_dist) ? ray_dist : -1.0f;
}
// :48-65
static inline float get_clouds_density(const Texture& clouds, vec3 pos) {
    float r = length(pos);
    float density = 0.0f;
    if (r > volume::clouds_lower_limit && r < volume::clouds_upper_limit) {
        float h = (r - volume::clouds_lower_limit) * (1.0f / volume::clouds_thickness);   // contract 2: product with RN(1 / thickness)
        float cloud_texture = sample_sphere_texture_r(clouds, pos);
        float column_height = cloud_texture;
        const float split = 0.2f;
        density = (h - split < column_height * (float)(1.0 - 0.2) && split - h < column_height * split)
                      ? max_(cloud_texture, 0.4f) : 0.0f;
    }
    return density * volume::clouds_density;
}
// :67-71
static inline vec4 get_atmos_density(vec3 pos, const Texture& clouds) {
    vec3 rmo = volume::get_density(volume::get_elevation(pos));
    float c = get_clouds_density(clouds, pos);
    return vec4{rmo.x, rmo.y, rmo.z, c};
}
enum { NULL_EVENT = 0, ABSORB_EVENT = 1, SCATTER_EVENT = 2 };                              // :73-75

// :263-270
static inline bool sample_scatter_event(Rng& rng, int interaction_id) {
    if (interaction_id == volume::ISOTROPIC_CLOUD_ID) interaction_id = volume::CLOUD_ID;
    const float albedos[4] = {volume::rayleigh_albedo, volume::aerosol_albedo, volume::ozone_albedo, volume::cloud_albedo};
    return rng.next_f32() < albedos[interaction_id];
}
static inline vec4 mul4(vec4 a, vec4 b) { return vec4{a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
static inline float sum4(vec4 a) { return ((a.x + a.y) + a.z) + a.w; }
static inline float comp4(vec4 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : (i == 2 ? a.z : a.w)); }

// :77-115
static inline void sample_interaction_delta_tracking(Rng& rng, vec3 ray_pos, vec3 ray_dir, float t_start, float t_max,
                                                     vec4 extinctions, float max_extinction, const Texture& clouds,
                                                     int* event_out, float* t_out, int* id_out) {
    float t = t_start;
    const float rcp_majorant = 1.0f / max_extinction;   // contract 2: "/ max_extinction" is "* RN(1 / max_extinction)" throughout
    ray_pos = ray_pos + t * ray_dir;
    int interaction_id = 0;
    int event = NULL_EVENT;
    while (t < t_max) {
        float t_step = -log_unit_(rng.next_f32()) * rcp_majorant;
        ray_pos = ray_pos + t_step * ray_dir;
        t += t_step;
        g_work.tracking_steps++;
        if (t >= t_max) break;
        vec4 extinction_sample = mul4(extinctions, get_atmos_density(ray_pos, clouds));
        float rand = rng.next_f32();
        if (rand < sum4(extinction_sample) * rcp_majorant) {
            float cmf = 0.0f;
            while (interaction_id < 3) {
                cmf += comp4(extinction_sample, interaction_id);
                if (rand < cmf * rcp_majorant) break;
                interaction_id += 1;
            }
            event = sample_scatter_event(rng, interaction_id) ? SCATTER_EVENT : ABSORB_EVENT;
            break;
        }
    }
    *event_out = event; *t_out = t; *id_out = interaction_id;
}
// :117-143
static inline float transmittance_ratio_tracking(Rng& rng, vec3 ray_pos, vec3 ray_dir, float t_start, float t_max,
                                                 vec4 extinctions, float max_extinction, const Texture& clouds) {
    float t = t_start;
    const float rcp_majorant = 1.0f / max_extinction;
    ray_pos = ray_pos + t * ray_dir;
    float transmittance = 1.0f;
    while (t < t_max) {
        float t_step = -log_unit_(rng.next_f32()) * rcp_majorant;
        ray_pos = ray_pos + t_step * ray_dir;
        t += t_step;
        g_work.tracking_steps++;
        if (t >= t_max) break;
        vec4 extinction_sample = mul4(extinctions, get_atmos_density(ray_pos, clouds));
        transmittance *= 1.0f - sum4(extinction_sample) * rcp_majorant;
        if (transmittance < 1e-5f) break;
    }
    return transmittance;
}
// :145-169
static inline void intersect_cloud_limits(vec3 ray_pos, vec3 ray_dir, float land_isection, float* t_start_out, float* t_max_out) {
    float t_start = 0.0f, t_max = 0.0f;
    float elevation = length(ray_pos);
    vec2 lower = rsi(ray_pos, ray_dir, volume::clouds_lower_limit);
    vec2 upper = rsi(ray_pos, ray_dir, volume::clouds_upper_limit);
    if (elevation >= volume::clouds_upper_limit) {
        t_start = max_(0.0f, upper.x);
        t_max = (lower.y >= 0.0f) ? lower.x : upper.y;
        if (upper.y < 0.0f) t_max = -1.0f;
    } else if (elevation >= volume::clouds_lower_limit) {
        t_start = 0.0f;
        t_max = (lower.y >= 0.0f) ? lower.x : upper.y;
    } else {
        t_start = lower.y;
        t_max = upper.y;
        if (land_isection > 0.0f) t_max = -1.0f;
    }
    *t_start_out = t_start; *t_max_out = t_max;
}
// :172-207
static inline void sample_interaction(Rng& rng, vec3 ray_pos, vec3 ray_dir, float land_isection, vec4 extinctions,
                                      float max_extinction_rmo, float max_extinction_cloud, const Texture& clouds,
                                      int* event_out, float* t_out, int* id_out) {
    vec2 atmos = rsi(ray_pos, ray_dir, volume::atmos_upper_limit);
    float t_start = max_(0.0f, atmos.x);
    float t_max = (land_isection >= 0.0f) ? land_isection : atmos.y;
    if (atmos.y < 0.0f) t_max = -1.0f;
    vec4 rmo_ext{extinctions.x, extinctions.y, extinctions.z, 0.0f};
    int rmo_event, rmo_id; float rmo_t;
    sample_interaction_delta_tracking(rng, ray_pos, ray_dir, t_start, t_max, rmo_ext, max_extinction_rmo, clouds, &rmo_event, &rmo_t, &rmo_id);
    intersect_cloud_limits(ray_pos, ray_dir, land_isection, &t_start, &t_max);
    int event = rmo_event; float t = rmo_t; int interaction_id = rmo_id;
    if (rmo_event == NULL_EVENT || rmo_t > t_start) {
        vec4 cloud_ext{0.0f, 0.0f, 0.0f, extinctions.w};
        int cloud_event, cloud_id; float cloud_t;
        sample_interaction_delta_tracking(rng, ray_pos, ray_dir, t_start, t_max, cloud_ext, max_extinction_cloud, clouds, &cloud_event, &cloud_t, &cloud_id);
        if (cloud_event > 0 && (cloud_t < rmo_t || rmo_event == NULL_EVENT)) {
            t = cloud_t;
            interaction_id = volume::CLOUD_ID;
            event = cloud_event;
        }
    }
    *event_out = event; *t_out = t; *id_out = interaction_id;
}
// :211-232
static inline float sample_transmittance(Rng& rng, vec3 ray_pos, vec3 ray_dir, float land_isection, vec4 extinctions,
                                         float max_extinction_rmo, float max_extinction_cloud, const Texture& clouds) {
    vec2 atmos = rsi(ray_pos, ray_dir, volume::atmos_upper_limit);
    float t_start = max_(0.0f, atmos.x);
    float t_max = (land_isection >= 0.0f) ? land_isection : atmos.y;
    if (atmos.y < 0.0f) t_max = -1.0f;
    vec4 rmo_ext{extinctions.x, extinctions.y, extinctions.z, 0.0f};
    float transmittance = transmittance_ratio_tracking(rng, ray_pos, ray_dir, t_start, t_max, rmo_ext, max_extinction_rmo, clouds);
    intersect_cloud_limits(ray_pos, ray_dir, land_isection, &t_start, &t_max);
    vec4 cloud_ext{0.0f, 0.0f, 0.0f, extinctions.w};
    transmittance *= transmittance_ratio_tracking(rng, ray_pos, ray_dir, t_start, t_max, cloud_ext, max_extinction_cloud, clouds);
    return transmittance;
}
// :235-247
static inline float evaluate_phase(vec3 ray_dir, vec3 light_dir, int interaction_id, bool reduce_peak) {
    float phase = 0.0f;
    float cos_theta = dot(ray_dir, light_dir);
    if (interaction_id == volume::RAYLEIGH_ID) phase += volume::rayleigh_phase(cos_theta);
    else if (interaction_id == volume::MIE_ID) phase += volume::mie_phase(cos_theta);
    else if (interaction_id == volume::CLOUD_ID) phase += volume::cloud_phase(cos_theta, reduce_peak);
    else if (interaction_id == volume::ISOTROPIC_CLOUD_ID) phase += (float)(1.0 / (4.0 * M_PI));
    return phase;
}
// :249-261
static inline vec3 sample_phase(Rng& rng, vec3 ray_dir, int interaction_id, bool reduce_peak, float* phase_div_pdf) {
    *phase_div_pdf = 1.0f;
    if (interaction_id == volume::RAYLEIGH_ID || interaction_id == volume::ISOTROPIC_CLOUD_ID) {
        float r0 = rng.next_f32();
        float r1 = rng.next_f32();
        vec3 d = sample_sphere(r0, r1);
        *phase_div_pdf = evaluate_phase(ray_dir, d, interaction_id, reduce_peak) * (float)(4.0 * M_PI);
        return d;
    } else if (interaction_id == volume::MIE_ID) {
        return volume::sample_mie_phase(rng, ray_dir);
    }
    return volume::sample_cloud_phase(rng, ray_dir, reduce_peak);
}
// :284-313
struct LandMaterial { vec3 albedo_srgb; float ocean, bathymetry, emissive; };
// the colour grade of get_land_material (pathtracer.py:292-306) for an albedo texel and an ocean-mask value
static inline vec3 grade_land_albedo(vec3 tex, float ocean) {
    vec3 land = mix(lum3(tex), tex, 6.5f);
    float greenery = sqr(land.y / lum(land));
    greenery = smoothstep(1.5f, 1.9f, greenery);
    land = (1.0f * tex) / (greenery * 0.7f + 1.0f);
    land = mix(lum3(land), land, 1.4f - greenery * 0.45f);
    land = mix(land, (land * vec3(255.0f, 128.0f, 64.0f)) / 255.0f, 0.2f * (1.0f - greenery));   // (a * vec3(..)) / 255.0
    vec3 ocean_albedo = mix(lum3(tex), tex, 0.75f) * 0.9f;
    return mix(land, ocean_albedo, ocean);
}
static inline LandMaterial get_land_material(const Scene& sc, vec3 pos) {
    LandMaterial m;
    m.ocean = sample_sphere_texture_r(*sc.ocean, pos);
    float a3[3]; sample_sphere_texture(*sc.albedo, pos, a3, 3);
    m.albedo_srgb = grade_land_albedo(vec3(a3[0], a3[1], a3[2]), m.ocean);
    m.bathymetry = sample_sphere_texture_r(*sc.bathymetry, pos);
    m.emissive = sample_sphere_texture_r(*sc.emissive, pos);
    return m;
}

// :316-469
static inline float path_tracer(Rng& rng, const PathParameters& path, const SceneParameters& scene, const Scene& sc) {
    vec3 ray_pos = path.ray_pos;
    vec3 ray_dir = path.ray_dir;
    float sun_power = plancks(5778.0f, path.wavelength);
    float nightlights_power = plancks(2700.0f, path.wavelength) * 0.0001f;
    float sun_irradiance = sun_power * cone_angle_to_solid_angle(scene.sun_angular_radius);
    vec3 d0 = volume::get_density(0.0f);
    vec3 max_densities_rmo(d0.x, d0.y, volume::get_ozone_density(volume::ozone_peak_height));
    float max_density_cloud = volume::clouds_density;
    vec4 extinctions{volume::spectra_extinction_rayleigh(path.wavelength), volume::spectra_extinction_mie(path.wavelength),
                     volume::spectra_extinction_ozone(path.wavelength, sc.o3_crossec), volume::clouds_extinct};
    bool primary_ray_did_not_intersect = false;
    float in_scattering = 0.0f;
    float throughput = 1.0f;
    for (int scatter_count = 0; scatter_count < 25; ++scatter_count) {
        g_work.vertices++;
        if (scatter_count > 9) extinctions.w = 0.02f;
        float max_extinction_rmo = (extinctions.x * max_densities_rmo.x + extinctions.y * max_densities_rmo.y) + extinctions.z * max_densities_rmo.z;
        float max_extinction_cloud = extinctions.w * max_density_cloud;
        float earth_intersection = intersect_land(*sc.height, ray_pos, ray_dir, scene.land_height_scale);
        int event, interaction_id; float interaction_dist;
        sample_interaction(rng, ray_pos, ray_dir, earth_intersection, extinctions, max_extinction_rmo, max_extinction_cloud,
                           *sc.clouds, &event, &interaction_dist, &interaction_id);
        if (scatter_count > 9 && interaction_id == volume::CLOUD_ID) interaction_id = volume::ISOTROPIC_CLOUD_ID;
        vec3 light_dir = sample_cone_oriented(rng, scene.sun_cos_angle, scene.light_direction);
        if (event == ABSORB_EVENT) {
            break;
        } else if (event == SCATTER_EVENT) {
            vec3 interaction_pos = ray_pos + interaction_dist * ray_dir;
            bool direct_visibility = rsi(interaction_pos, light_dir, volume::planet_r).y > 0.0f;   // true = planet blocks (Q5)
            float direct_transmittance = 0.0f;
            if (!direct_visibility)
                direct_transmittance = sample_transmittance(rng, interaction_pos, light_dir, -1.0f, extinctions,
                                                            max_extinction_rmo, max_extinction_cloud, *sc.clouds);
            float direct_phase = evaluate_phase(ray_dir, light_dir, interaction_id, scatter_count > 0);
            in_scattering += throughput * direct_transmittance * sun_irradiance * direct_phase;
            float phase_div_pdf;
            vec3 scatter_dir = sample_phase(rng, ray_dir, interaction_id, scatter_count > 0, &phase_div_pdf);
            ray_dir = scatter_dir;
            ray_pos = interaction_pos;
            throughput *= phase_div_pdf;
        } else if (earth_intersection > 0.0f) {
            vec3 land_pos = ray_pos + ray_dir * earth_intersection;
            vec3 land_n = land_normal(*sc.height, sc.topo_res, land_pos, scene.land_height_scale);
            LandMaterial mat = get_land_material(sc, land_pos);
            float albedo = srgb_to_spectrum(sc.srgb_to_spectrum, mat.albedo_srgb, path.wavelength);
            in_scattering += throughput * mat.emissive * nightlights_power;
            vec3 offset_pos = land_pos * (1.0f + 0.0001f * scene.land_height_scale / 12000.0f);
            bool direct_visibility = intersect_land(*sc.height, offset_pos, light_dir, scene.land_height_scale) < 0.0f;
            float direct_transmittance = sample_transmittance(rng, offset_pos, light_dir, direct_visibility ? -1.0f : 0.0f, extinctions,
                                                              max_extinction_rmo, max_extinction_cloud, *sc.clouds);
            float direct_n_dot_l;
            float direct_brdf = surface::earth_brdf(albedo, mat.ocean, mat.bathymetry, -ray_dir, land_n, light_dir, &direct_n_dot_l);
            in_scattering += throughput * direct_transmittance * (direct_visibility ? 1.0f : 0.0f) * sun_irradiance * direct_brdf * direct_n_dot_l;
            vec3 view_dir = -ray_dir;
            ray_dir = sample_hemisphere_cosine_weighted(rng, land_n);
            ray_pos = offset_pos;
            float unused;
            float brdf = surface::earth_brdf(albedo, mat.ocean, mat.bathymetry, view_dir, land_n, ray_dir, &unused);
            throughput *= brdf * (float)M_PI;
        } else {
            if (scatter_count == 0) primary_ray_did_not_intersect = true;
            break;
        }
        if (scatter_count > 3) {
            float termination_p = max_(0.05f, 1.0f - throughput);
            if (rng.next_f32() < termination_p) break;
            throughput /= 1.0f - termination_p;
        }
    }
    if (primary_ray_did_not_intersect) {
        if (dot(scene.light_direction, path.ray_dir) > scene.sun_cos_angle) in_scattering += sun_power;
        float s3[3]; sample_sphere_texture(*sc.stars, path.ray_dir, s3, 3);
        float stars_power = srgb_to_spectrum(sc.srgb_to_spectrum, vec3(s3[0], s3[1], s3[2]), path.wavelength);
        in_scattering += stars_power * sun_power * 0.0000001f;
    }
    if (isinf(in_scattering) || isnan(in_scattering) || in_scattering < 0.0f) in_scattering = 0.0f;
    return in_scattering;
}

// ---------------------------------------------------------------- ray marcher (dead alternative path)
// :471-500
static inline float ray_march_transmittance(vec3 ray_pos, vec3 ray_dir, vec3 rmo_extinction) {
    const int steps = 16;
    const float r_steps = 1.0f / (float)steps;
    float transmittance = 0.0f;
    bool visibility = rsi(ray_pos, ray_dir, volume::planet_r).y > 0.0f;
    if (!visibility) {
        vec2 atmos = rsi(ray_pos, ray_dir, volume::atmos_upper_limit);
        float t_max = atmos.y;
        if (atmos.y < 0.0f) t_max = -1.0f;
        float dd = t_max * r_steps;
        vec3 ray_step = ray_dir * dd;
        vec3 od(0.0f, 0.0f, 0.0f);
        for (int i = 0; i < steps; ++i) {
            vec3 density = volume::get_density(volume::get_elevation(ray_pos));
            od = od + density * dd;
            ray_pos = ray_pos + ray_step;
        }
        transmittance = exp_(-dot(rmo_extinction, od));
    }
    return transmittance;
}
// :502-542
static inline void ray_marh_atmos(vec3 ray_pos, vec3 ray_dir, float t_start, float t_max, vec3 sun_dir, vec3 rmo_extinction,
                                  vec2 rm_scattering, float* in_scatter_out, float* transmittance_out) {
    const int steps = 64;
    const float r_steps = 1.0f / (float)steps;
    float dd = (t_max - t_start) * r_steps;
    vec3 ray_step = ray_dir * dd;
    ray_pos = ray_pos + ray_dir * t_start;
    float cos_theta = dot(ray_dir, sun_dir);
    vec2 phase{volume::rayleigh_phase(cos_theta), volume::mie_phase(cos_theta)};
    float transmittance = 1.0f;
    float in_scatter = 0.0f;
    for (int i = 0; i < steps; ++i) {
        float h = volume::get_elevation(ray_pos);
        vec3 density = volume::get_density(h);
        float step_optical_depth = dot(rmo_extinction, density * dd);
        float step_transmittance = saturate_(exp_(-step_optical_depth));
        float step_integral = saturate_((1.0f - step_transmittance) / step_optical_depth);
        float visible_scattering = transmittance * step_integral;
        float sun_transmittance = ray_march_transmittance(ray_pos, sun_dir, rmo_extinction);
        float step_scattering = rm_scattering.x * (density.x * phase.x) + rm_scattering.y * (density.y * phase.y);
        in_scatter += step_scattering * sun_transmittance * visible_scattering * dd;
        transmittance *= step_transmittance;
        ray_pos = ray_pos + ray_step;
    }
    *in_scatter_out = in_scatter; *transmittance_out = transmittance;
}
// :544-685
static inline float ray_marcher(Rng& rng, const PathParameters& path, const SceneParameters& scene, const Scene& sc) {
    vec3 ray_pos = path.ray_pos;
    vec3 ray_dir = path.ray_dir;
    float sun_power = plancks(5778.0f, path.wavelength);
    float nightlights_power = plancks(2700.0f, path.wavelength) * 0.0001f;
    float sun_irradiance = sun_power * cone_angle_to_solid_angle(scene.sun_angular_radius);
    vec3 ext(volume::spectra_extinction_rayleigh(path.wavelength), volume::spectra_extinction_mie(path.wavelength),
             volume::spectra_extinction_ozone(path.wavelength, sc.o3_crossec));
    vec2 scattering{ext.x * volume::rayleigh_albedo, ext.y * volume::aerosol_albedo};
    bool primary_ray_did_not_intersect = false;
    float accum = 0.0f;
    float throughput = 1.0f;
    for (int scatter_count = 0; scatter_count < 3; ++scatter_count) {
        g_work.vertices++;
        float earth_intersection = intersect_land(*sc.height, ray_pos, ray_dir, scene.land_height_scale);
        vec2 atmos = rsi(ray_pos, ray_dir, volume::atmos_upper_limit);
        float t_start = max_(0.0f, atmos.x);
        float t_max = (earth_intersection > 0.0f) ? earth_intersection : atmos.y;
        if (atmos.y < 0.0f) {
            primary_ray_did_not_intersect = (scatter_count == 0);
            break;
        }
        vec3 light_dir = sample_cone_oriented(rng, scene.sun_cos_angle, scene.light_direction);
        float in_scatter, transmittance;
        ray_marh_atmos(ray_pos, ray_dir, t_start, t_max, light_dir, ext, scattering, &in_scatter, &transmittance);
        accum += throughput * in_scatter;
        throughput *= transmittance;
        if (earth_intersection > 0.0f) {
            vec3 land_pos = ray_pos + ray_dir * earth_intersection;
            vec3 land_n = land_normal(*sc.height, sc.topo_res, land_pos, scene.land_height_scale);
            LandMaterial mat = get_land_material(sc, land_pos);
            float albedo = srgb_to_spectrum(sc.srgb_to_spectrum, mat.albedo_srgb, path.wavelength);
            accum += throughput * mat.emissive * nightlights_power;
            vec3 offset_pos = land_pos * (1.0f + 0.0001f * scene.land_height_scale / 12000.0f);
            bool direct_visibility = intersect_land(*sc.height, offset_pos, light_dir, scene.land_height_scale) < 0.0f;
            float direct_transmittance = 1.0f;
            float direct_n_dot_l;
            float direct_brdf = surface::earth_brdf(albedo, mat.ocean, mat.bathymetry, -ray_dir, land_n, light_dir, &direct_n_dot_l);
            accum += throughput * direct_transmittance * (direct_visibility ? 1.0f : 0.0f) * sun_irradiance * direct_brdf * direct_n_dot_l;
            vec3 view_dir = -ray_dir;
            ray_dir = sample_hemisphere_cosine_weighted(rng, land_n);
            ray_pos = offset_pos;
            float unused;
            float brdf = surface::earth_brdf(albedo, mat.ocean, mat.bathymetry, view_dir, land_n, ray_dir, &unused);
            throughput *= brdf * (float)M_PI;
        }
    }
    if (primary_ray_did_not_intersect) {
        if (dot(scene.light_direction, path.ray_dir) > scene.sun_cos_angle) accum += sun_power;
        float s3[3]; sample_sphere_texture(*sc.stars, path.ray_dir, s3, 3);
        float stars_power = srgb_to_spectrum(sc.srgb_to_spectrum, vec3(s3[0], s3[1], s3[2]), path.wavelength);
        accum += stars_power * sun_power * 0.0000001f;
    }
    if (isinf(accum) || isnan(accum) || accum < 0.0f) accum = 0.0f;
    return accum;
}

}  // namespace deo
