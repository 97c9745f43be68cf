// render_kernel_v3.hip — path_tracer (pathtracer.py:316-469) as a WAVEFRONT PIPELINE through HBM-resident stage queues.
//
// Why.  render_kernel_v2 keeps every path in a lane of a persistent wave and lets the wave's lanes be in different loops of
// different vertices; its lanes are 41 % busy because a wave of 64 always holds a mix of work classes (sphere-trace steps,
// cloud steps, gas steps, five kinds of transition).  It uses 4 % of the chip's 8 TB/s.  This kernel family spends that
// idle bandwidth to buy coherence: a path is a 64-byte RECORD that travels through seven queues in HBM, one per work class,
//
//     ST (sphere trace)  ->  GAS (tracking through Rayleigh/Mie/ozone)  ->  CLOUD (tracking through the cloud shell)
//        ->  EVENT (event resolution, light sample)  ->  [SURF (normal + material)  ->  ST (shadow ray)]
//        ->  GAS -> CLOUD (ratio tracking towards the sun)  ->  SCFIN / SURFFIN (shading, next direction, roulette)  ->  ST ...
//
// and every kernel below drains ONE queue: all 64 lanes of a wave run the same loop of the same stage, a lane = one record
// from start to finish of that stage, and the results are appended (one wave-level atomic per push) to the queues of the next
// stages.  The frame is level-synchronous: ten launches per path vertex (de_api.hip: launch_v3), 25 vertices at most.
// What a path does not need inside the loops (throughput, radiance so far, normal, material, incoming direction) lives in a
// 64-byte "cold" record per path, touched only by the four shading stages.
//
// Results are IDENTICAL to render_kernel_v2 / render_kernel.hip / the CPU oracle: every path executes the same arithmetic in
// the same order on its own RNG stream (the 8 bytes of generator state travel in the record); only WHERE and WHEN changes.
// Traffic: ~8 hops x 128 B + ~300 B of cold state per vertex ~ 3 KB per sample — two orders above the taps' algorithmic bytes,
// by design: HBM was the idle resource.
#include "de_kernels.h"

namespace wf {

enum { Q_ST = 0, Q_GAS, Q_CLOUD, Q_EVENT, Q_SURF, Q_SCFIN, Q_SURFFIN, NQ };
enum { EV_NULL = 0, EV_ABSORB = 1, EV_SCATTER = 2 };

struct Rec { uint4 v[4]; };                      // 64 bytes
struct Queue {
    Rec* rec;
    uint32_t* ctl;                                // control words of the CURRENT filling: [0] entries, [1] chunk cursor of the draining kernel
    uint32_t* ctl_next;                           // ... of the next filling: zeroed by the kernel that drains the current one
};
// per-path state outside the loops (64 bytes, one sector per access)
struct Cold { float thr, Ls, dx, dy, dz, nx, ny, nz, alb, ocean, bathy, pad0, pad1, pad2, pad3, pad4; };

struct Args {
    RenderArgs a;
    Queue q[NQ];
    Cold* cold;
    uint32_t n_items;                             // paths of this launch
    uint32_t capacity;                            // records each queue holds (paths + room for holes): no kernel reads past it
};

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

DE_DEV void load_path(const Rec* r, Path& p) {
    const uint4 a = r->v[0], b = r->v[1], c = r->v[2], d = r->v[3];
    p.pid = a.x; p.meta = a.y; p.rng.s0 = a.z; p.rng.s1 = a.w; p.rng.draws = 0;
    p.P = v3(asf(b.x), asf(b.y), asf(b.z)); p.W = v3(asf(b.w), asf(c.x), asf(c.y));
    p.x[0] = asf(c.z); p.x[1] = asf(c.w); p.x[2] = asf(d.x); p.x[3] = asf(d.y); p.x[4] = asf(d.z); p.x[5] = asf(d.w);
}
DE_DEV void store_path(Rec* r, const Path& p) {
    r->v[0] = make_uint4(p.pid, p.meta, p.rng.s0, p.rng.s1);
    r->v[1] = make_uint4(asu(p.P.x), asu(p.P.y), asu(p.P.z), asu(p.W.x));
    r->v[2] = make_uint4(asu(p.W.y), asu(p.W.z), asu(p.x[0]), asu(p.x[1]));
    r->v[3] = make_uint4(asu(p.x[2]), asu(p.x[3]), asu(p.x[4]), asu(p.x[5]));
}
// Appending.  A wave RESERVES slots of a target queue in bulk (one atomic per DE_V3_BULK records — millions of waves bumping one
// counter once per push serialise in the L2: measured 10x the kernels' own time) and fills them from its pushes; what is left of
// its last reservation when the kernel ends is marked as holes (pid = DE_V3_HOLE), which the draining kernel skips.
#ifndef DE_V3_BULK
#define DE_V3_BULK 256u
#endif
#define DE_V3_HOLE 0xffffffffu
struct Out { uint32_t lo, hi; };                  // wave-uniform: the unfilled part of this wave's reservation in one target queue
DE_DEV void push(const Queue& q, Out& o, const Path& p, bool pred) {
    const unsigned long long mask = __ballot(pred);
    if (mask == 0ull) return;
    const int lane = (int)(threadIdx.x & 63u);
    const uint32_t n = (uint32_t)__popcll(mask), rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    const uint32_t rem = o.hi - o.lo;
    uint32_t slot = o.lo + rank;
    if (n <= rem) {
        o.lo += n;
    } else {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&q.ctl[0], DE_V3_BULK);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (rank >= rem) slot = base + (rank - rem);
        o.lo = base + (n - rem);
        o.hi = base + DE_V3_BULK;
    }
    if (pred) store_path(&q.rec[slot], p);
}
DE_DEV void close_out(const Queue& q, const Out& o) {
    for (uint32_t i = o.lo + (threadIdx.x & 63u); i < o.hi; i += 64u) q.rec[i].v[0] = make_uint4(DE_V3_HOLE, 0u, 0u, 0u);
}
// Draining.  The launch sequence is static, so the host knows which of a queue's two sets of control words is being filled and
// which is free: the kernel that drains a queue zeroes the OTHER set first (nothing touches it during that launch — no kernel
// pushes into the queue it drains) and the host flips the sets afterwards.  No per-block tickets, no host round trips.
// Work is handed out in chunks of 64 records: chunk b to block b, further chunks from an atomic cursor — a launch whose queue
// is shorter than the grid issues no atomics at all (the deep vertices: a few thousand paths, 8192 blocks).
DE_DEV void begin_drain(const Queue& q) {
    if (blockIdx.x == 0 && (threadIdx.x & 63u) == 0u) { q.ctl_next[0] = 0u; q.ctl_next[1] = 0u; }
}
// one-wave blocks, one record per lane, chunks of 64 by block index and stride: the straight-line stages (uniform work per record)
DE_DEV bool next_chunk(uint32_t n, uint32_t* chunk, uint32_t* idx, bool* active) {
    const uint32_t c = (*chunk == 0xffffffffu) ? blockIdx.x : *chunk + gridDim.x;
    *chunk = c;
    if ((unsigned long long)c * 64ull >= (unsigned long long)n) return false;
    *idx = c * 64u + (threadIdx.x & 63u);
    *active = *idx < n;
    return true;
}

// ---- persistent lanes: the loop stages (ST, GAS, CLOUD).  A lane takes a record, runs the stage's loop on it one step per trip
// of the wave loop, and when it ends keeps the finished record until the wave's next SERVICE: all finished records are pushed
// (one atomic per target queue) and the idle lanes take new records.  A service runs when >= SERVICE lanes are idle (or nothing
// is running), so its ~150 instructions and its wait for the new records' loads are shared by that many records: with one record
// per lane and no refill, a wave would idle behind its longest record (cloud segments have a heavy tail: lane utilisation 0.1).
// The threshold is per stage (A/B on three cameras, profiles/experiments/r3_variant4_sweeps.txt): cloud segments are long and uneven
// — refill early (12); gas segments are short and a gas step is 60 instructions around one load — a service costs several steps,
// so wait for most of the wave (48); the sphere trace sits between (32).
#ifndef DE_V3_SVC_ST
#define DE_V3_SVC_ST 32
#endif
#ifndef DE_V3_SVC_GAS
#define DE_V3_SVC_GAS 48
#endif
#ifndef DE_V3_SVC_CLOUD
#define DE_V3_SVC_CLOUD 12
#endif
#ifndef DE_V3_IN_CHUNK
#define DE_V3_IN_CHUNK 256u
#endif
template <int SERVICE, typename Stage, int NT>
DE_DEV void run_stage(const Args& A, const Queue& q, Stage& st, const int (&targets)[NT]) {
    const uint32_t n = min(q.ctl[0], A.capacity);
    const int lane = (int)(threadIdx.x & 63u);
    begin_drain(q);
    // this wave's current chunk of the queue (wave-uniform): chunk b to block b, then from the atomic cursor
    uint32_t w_lo = min(blockIdx.x * DE_V3_IN_CHUNK, n), w_hi = min(w_lo + DE_V3_IN_CHUNK, n);
    bool exhausted = false;                                   // wave-uniform: the queue has no chunk left for this wave
    Out out[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) { out[k].lo = 0u; out[k].hi = 0u; }
    Path p;
    bool busy = false, fin = false;
    int target = -1;
    for (;;) {
        const unsigned long long m_busy = __ballot(busy);
        if (m_busy == 0ull || (!exhausted && 64 - __popcll(m_busy) >= SERVICE)) {
#pragma unroll
            for (int k = 0; k < NT; ++k) push(A.q[targets[k]], out[k], p, fin && target == targets[k]);
            fin = false;
            bool got = false;
            uint32_t idx = 0;
            while (!exhausted) {
                const unsigned long long m_need = __ballot(!busy && !got);
                if (m_need == 0ull) break;
                const uint32_t avail = w_hi - w_lo;
                const int my = __popcll(m_need & ((1ull << lane) - 1ull));
                if (!busy && !got && (uint32_t)my < avail) { idx = w_lo + (uint32_t)my; got = true; }
                w_lo += min((uint32_t)__popcll(m_need), avail);
                if (__ballot(!busy && !got) == 0ull) break;
                uint32_t c = 0;
                if (lane == 0) c = atomicAdd(&q.ctl[1], 1u);
                c = gridDim.x + (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
                if ((unsigned long long)c * DE_V3_IN_CHUNK >= (unsigned long long)n) { exhausted = true; break; }
                w_lo = c * DE_V3_IN_CHUNK;
                w_hi = min(w_lo + DE_V3_IN_CHUNK, n);
            }
            if (got) {
                load_path(&q.rec[idx], p);
                if (p.pid != DE_V3_HOLE) { st.begin(p); busy = true; }
            }
            if (__ballot(busy) == 0ull) { if (exhausted) break; else continue; }
        }
        if (busy) {
            if (st.step(p)) { target = st.finish(p); busy = false; fin = true; }
        }
    }
#pragma unroll
    for (int k = 0; k < NT; ++k) close_out(A.q[targets[k]], out[k]);
}

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

#ifdef DE_V3_LDS_TILE
// NS-1 experiment (BASELINE north star: "coalesced loads of the maps staged through LDS tiles"; profiles/experiments/README.md).
// The primary rays of a wave are the 64 pixels of one 8x8 screen tile: the only spatially coherent taps of the whole integrator.
// Per sphere-trace step the wave computes the bounding box of its lanes' footprints in units of 128-byte map tiles (8 x 4
// footprint-packed entries); if it holds at most DE_V3_LDS_TILE tiles they are staged in LDS with coalesced loads (whole lines, 32
// dwords each) and every lane reads its dword from LDS; otherwise (a wave across the map seam, grazing rays) the lanes tap the map
// directly.  Same dword either way: results are unchanged.  st[0..3]: wave-steps, staged wave-steps, tiles staged, lanes served.
template <bool CLAMP>
DE_DEV float sphere_trace_lds(const RenderArgs& a, const Path& p, bool active, float land_height_scale, float escape_r, float max_ray_dist,
                              uint32_t* s_tile, unsigned long long* st) {
    float t = p.x[0];
    int cnt = 0;
    bool run = active;
    const int lane = (int)(threadIdx.x & 63u);
    const TexR8& T = a.topo;
    while (__ballot(run) != 0ull) {
        float len = 0.0f, fx = 0.0f, fy = 0.0f;
        int i0 = 0, j0 = 0;
        bool tap = false;
        if (run) {
            const vec3 Q = p.P + p.W * t;
            len = length_nr(Q);
            bool escaped = false;
            if (cnt <= 100) {
                const float b = dot(Q, p.W);
                const float peri2 = (b < 0.0f) ? (len * len - b * b) : len * len;
                if (peri2 > escape_r * escape_r) { t = max_ray_dist; run = false; escaped = true; }
            }
            if (!escaped) {
                const vec3 n = Q * de_rcp_nr(len);
                const vec2_ uv = sphere_UV_map(n);
                const float x = fract_(uv.x * 1.0f) * T.wf - 0.5f, y = fract_(uv.y * 1.0f) * T.hf - 0.5f;
                const float x0f = de_floor(x), y0f = de_floor(y);
                fx = x - x0f; fy = y - y0f;
                i0 = (int)x0f; j0 = (int)y0f;
                if (CLAMP) { if (i0 < 0) { i0 = 0; fx = 0.0f; } if (j0 < 0) { j0 = 0; fy = 0.0f; } }
                else { if (i0 < 0) i0 = T.w - 1; if (j0 < 0) j0 = T.h - 1; }
                tap = true;
            }
        }
        // bounding box of the tapping lanes, in map tiles
        int tx = i0 >> 3, ty = j0 >> 2;
        int xmin = tap ? tx : 0x7fffffff, xmax = tap ? tx : -1, ymin = tap ? ty : 0x7fffffff, ymax = tap ? ty : -1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            xmin = min(xmin, __shfl_xor(xmin, d)); xmax = max(xmax, __shfl_xor(xmax, d));
            ymin = min(ymin, __shfl_xor(ymin, d)); ymax = max(ymax, __shfl_xor(ymax, d));
        }
        const unsigned long long m_tap = __ballot(tap);
        uint32_t q = 0;
        if (m_tap != 0ull) {
            const int bw = xmax - xmin + 1, bh = ymax - ymin + 1;
            const int n_tiles = bw * bh;
            st[0] += 1ull;
            if (n_tiles <= DE_V3_LDS_TILE) {
                __syncthreads();                                   // the previous step's readers are done with the staging area
                for (int k = lane; k < n_tiles * 32; k += 64) {
                    const int tile = k >> 5, e = k & 31;
                    const int lx = xmin + tile % bw, ly = ymin + tile / bw;
                    s_tile[k] = T.data[((uint32_t)ly * (uint32_t)T.tiles_x + (uint32_t)lx) * 32u + (uint32_t)e];
                }
                __syncthreads();
                if (tap) q = s_tile[((ty - ymin) * bw + (tx - xmin)) * 32 + (((j0 & 3) << 3) | (i0 & 7))];
                st[1] += 1ull; st[2] += (unsigned long long)n_tiles; st[3] += (unsigned long long)__popcll(m_tap);
            } else if (tap) {
                q = T.data[((uint32_t)ty * (uint32_t)T.tiles_x + (uint32_t)tx) * 32u + (uint32_t)(((j0 & 3) << 3) | (i0 & 7))];
            }
        }
        if (tap) {
            const float tex = tap_r8_finish(q, fx, fy);
            const float sdf = len - DE_PLANET_R - land_height_scale * tex;
            t += sdf;
            cnt++;
            if (t > max_ray_dist || de_abs(sdf) < t * 0.0001f || cnt >= 250) run = false;
        }
    }
    return t;
}
#endif

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

// ---- access to the cold records.  The stage bodies below are written once and shared by this file's level-synchronous kernels and
// by the persistent scheduler kernel (render_kernel_v5.hip); what differs is HOW a cold record travels between the workgroups that
// touch it: here a kernel boundary lies between any two of them and plain loads / stores do; there the accessor writes through.
struct ColdPlain {
    Cold* base;
    DE_DEV void get_thr_ls(uint32_t pid, float& thr, float& Ls) const { thr = base[pid].thr; Ls = base[pid].Ls; }
    DE_DEV float get_ls(uint32_t pid) const { return base[pid].Ls; }
    DE_DEV void set_thr_ls(uint32_t pid, float thr, float Ls) const { base[pid].thr = thr; base[pid].Ls = Ls; }
    DE_DEV vec3 get_dir(uint32_t pid) const { return v3(base[pid].dx, base[pid].dy, base[pid].dz); }
    DE_DEV void set_dir(uint32_t pid, vec3 D) const { base[pid].dx = D.x; base[pid].dy = D.y; base[pid].dz = D.z; }
    DE_DEV void set_surf(uint32_t pid, vec3 N, float alb, float ocean, float bathy) const {
        Cold* c = &base[pid];
        c->nx = N.x; c->ny = N.y; c->nz = N.z; c->alb = alb; c->ocean = ocean; c->bathy = bathy;
    }
    DE_DEV void get_surf(uint32_t pid, vec3& N, float& alb, float& ocean, float& bathy) const {
        const Cold* c = &base[pid];
        N = v3(c->nx, c->ny, c->nz); alb = c->alb; ocean = c->ocean; bathy = c->bathy;
    }
};

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

#define DE_V3_BLOCK 64

// ------------------------------------------------------------------------------------------------ depth 0: new paths + primary ray
// renderer.py:305-314 + pathtracer.py:29-46 for every sample of the launch: lane = work item (same item <-> pixel/sample mapping
// as v2), coherent rays of one 8x8 tile per wave.
template <bool CLAMP>
__global__ void __launch_bounds__(DE_V3_BLOCK) v3_primary_kernel(wf::Args A, const FrameConsts* __restrict__ fcp) {
    using namespace wf;
    const RenderArgs& a = A.a;
    const FrameConsts& fc = *fcp;
    __shared__ float s_node_val[64];
    s_node_val[threadIdx.x & 63u] = a.node_val[threadIdx.x & 63u];
    __syncthreads();
#ifdef DE_V3_LDS_TILE
    __shared__ uint32_t s_tile[DE_V3_LDS_TILE * 32];
    unsigned long long lds_st[4] = {0ull, 0ull, 0ull, 0ull};
#endif
    const float MAX_RAY_DIST = (float)(6371e3 * 10.0);
    const float land_height_scale = fc.land_height_scale;
    const float escape_r = DE_PLANET_R + de_max(land_height_scale, 0.0f) + 8100.0f;
    const uint32_t stride = gridDim.x * DE_V3_BLOCK;
    Out o_gas = {0u, 0u}, o_cloud = {0u, 0u}, o_event = {0u, 0u};
    for (uint32_t w0 = blockIdx.x * DE_V3_BLOCK; w0 < A.n_items; w0 += stride) {
        const uint32_t w = w0 + threadIdx.x;
        const bool active = w < A.n_items;
        Path p;
        int target = -1;
        if (active) {
            new_path(a, fc, s_node_val, w, p);
#ifndef DE_V3_LDS_TILE
            const float t = sphere_trace<CLAMP>(a, p, land_height_scale, escape_r, MAX_RAY_DIST);
            target = after_sphere_trace(p, t, MAX_RAY_DIST);
#endif
        }
#ifdef DE_V3_LDS_TILE
        {
            const float t = sphere_trace_lds<CLAMP>(a, p, active, land_height_scale, escape_r, MAX_RAY_DIST, s_tile, lds_st);
            if (active) target = after_sphere_trace(p, t, MAX_RAY_DIST);
        }
#endif
        push(A.q[Q_GAS], o_gas, p, target == Q_GAS);
        push(A.q[Q_CLOUD], o_cloud, p, target == Q_CLOUD);
        push(A.q[Q_EVENT], o_event, p, target == Q_EVENT);
    }
    close_out(A.q[Q_GAS], o_gas); close_out(A.q[Q_CLOUD], o_cloud); close_out(A.q[Q_EVENT], o_event);
#ifdef DE_V3_LDS_TILE
    if ((threadIdx.x & 63u) == 0u) for (int k = 0; k < 4; ++k) atomicAdd(&a.counters[56 + k], lds_st[k]);
#endif
}

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
template <bool CLAMP>
__global__ void __launch_bounds__(DE_V3_BLOCK) v3_st_kernel(wf::Args A, const FrameConsts* __restrict__ fcp) {
    using namespace wf;
    const FrameConsts& fc = *fcp;
    const float MAX_RAY_DIST = (float)(6371e3 * 10.0);
    StStage<CLAMP> st{A.a.topo, fc.land_height_scale, DE_PLANET_R + de_max(fc.land_height_scale, 0.0f) + 8100.0f, MAX_RAY_DIST, 0.0f, 0};
    const int targets[4] = {Q_GAS, Q_CLOUD, Q_EVENT, Q_SURFFIN};
    run_stage<DE_V3_SVC_ST>(A, A.q[Q_ST], st, targets);
}

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
        if (kind == 0) {
            const float rand = rng_next(p.rng);
            if (rand < sum * rmaxr) {
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
                return true;
            }
        } else {
            trn *= 1.0f - sum * rmaxr;
            if (trn < 1e-5f) return true;
        }
        return !(t < tmax);                                                  // loop-head test (differs from the above only for NaN)
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
__global__ void __launch_bounds__(DE_V3_BLOCK) v3_gas_kernel(wf::Args A) {
    using namespace wf;
    GasStage st{A.a.nodes, A.a.dens_table};
    const int targets[4] = {Q_CLOUD, Q_EVENT, Q_SCFIN, Q_SURFFIN};
    run_stage<DE_V3_SVC_GAS>(A, A.q[Q_GAS], st, targets);
}

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
            density = (h - split < tex * (float)(1.0 - 0.2) && split - h < tex * split) ? de_max(tex, 0.4f) : 0.0f;
        }
        const float es = ext_w * (density * DE_CLOUDS_DENSITY);
        if (kind == 0) {
            const float rand = rng_next(p.rng);
            if (rand < es * rmaxc) { cloud_event = (rng_next(p.rng) < 0.99f) ? wf::EV_SCATTER : wf::EV_ABSORB; return true; }
        } else {
            trn *= 1.0f - es * rmaxc;
            if (trn < 1e-5f) return true;
        }
        return !(t < tmax);
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
// ------------------------------------------------------------------------------------------------ TRACK: gases, then the cloud shell, in one lane
// A ray's tracking through the gases is followed by its tracking through the cloud shell (pathtracer.py:186 -> :198, :227 -> :231) for 85 %
// of the gas segments.  Here the lane that finishes the gas segment goes on with the cloud segment: GasStage::finish's glue runs in place,
// the record never travels through the cloud queue (a 128-byte hop), and a wave holds lanes of both classes — an altitude-table load and a
// map tap in flight per trip, which is what the gas stage alone lacked (one dependent load per 60 instructions: 22 % of the issue ceiling).
// Same statements on the same generator in the same order: the bits do not move.
template <bool CLAMP>
struct TrackStage {
    GasStage g;
    CloudStage<CLAMP> c;
    bool in_cloud;
    int target;
    DE_DEV void begin(wf::Path& p) { in_cloud = false; g.begin(p); }
    DE_DEV bool step(wf::Path& p) {
        if (!in_cloud) {
            if (g.step(p)) {
                target = g.finish(p);
                if (target != wf::Q_CLOUD) return true;
                c.begin(p);
                in_cloud = true;
            }
            return false;
        }
        if (c.step(p)) { target = c.finish(p); return true; }
        return false;
    }
    DE_DEV int finish(wf::Path&) { return target; }
};
#ifndef DE_V3_SVC_TRACK
#define DE_V3_SVC_TRACK 16
#endif
// drains the gas queue with TrackStage, then what the sphere trace and the event stage queued for the cloud shell directly
template <bool CLAMP>
__global__ void __launch_bounds__(DE_V3_BLOCK) v3_track_kernel(wf::Args A) {
    using namespace wf;
    const int targets[3] = {Q_EVENT, Q_SCFIN, Q_SURFFIN};
    {
        TrackStage<CLAMP> st{GasStage{A.a.nodes, A.a.dens_table}, CloudStage<CLAMP>{A.a.clouds}, false, -1};
        run_stage<DE_V3_SVC_TRACK>(A, A.q[Q_GAS], st, targets);
    }
    {
        CloudStage<CLAMP> st{A.a.clouds};
        run_stage<DE_V3_SVC_CLOUD>(A, A.q[Q_CLOUD], st, targets);
    }
}

template <bool CLAMP>
__global__ void __launch_bounds__(DE_V3_BLOCK) v3_cloud_kernel(wf::Args A) {
    using namespace wf;
    CloudStage<CLAMP> st{A.a.clouds};
    const int targets[3] = {Q_EVENT, Q_SCFIN, Q_SURFFIN};
    run_stage<DE_V3_SVC_CLOUD>(A, A.q[Q_CLOUD], st, targets);
}

// ------------------------------------------------------------------------------------------------ EVENT: event resolution, light sample, branch
template <bool CLAMP>
__global__ void __launch_bounds__(DE_V3_BLOCK) v3_event_kernel(wf::Args A, const FrameConsts* __restrict__ fcp) {
    using namespace wf;
    const RenderArgs& a = A.a;
    const FrameConsts& fc = *fcp;
    const ColdPlain cold{A.cold};
    const Queue q = A.q[Q_EVENT];
    const uint32_t n = min(q.ctl[0], A.capacity);
    Out o_gas = {0u, 0u}, o_cloud = {0u, 0u}, o_scfin = {0u, 0u}, o_surf = {0u, 0u};
    begin_drain(q);
    uint32_t chunk = 0xffffffffu, idx; bool active;
    while (next_chunk(n, &chunk, &idx, &active)) {
        Path p;
        int target = -1;
        if (active) load_path(&q.rec[idx], p);
        if (active && p.pid == DE_V3_HOLE) active = false;
        if (active) target = event_body<CLAMP>(a, fc, cold, p);
        push(A.q[Q_GAS], o_gas, p, target == Q_GAS);
        push(A.q[Q_CLOUD], o_cloud, p, target == Q_CLOUD);
        push(A.q[Q_SCFIN], o_scfin, p, target == Q_SCFIN);
        push(A.q[Q_SURF], o_surf, p, target == Q_SURF);
    }
    close_out(A.q[Q_GAS], o_gas); close_out(A.q[Q_CLOUD], o_cloud); close_out(A.q[Q_SCFIN], o_scfin); close_out(A.q[Q_SURF], o_surf);
}

// ------------------------------------------------------------------------------------------------ SURF: normal, material, shadow-ray set-up
template <bool CLAMP>
__global__ void __launch_bounds__(DE_V3_BLOCK) v3_surf_kernel(wf::Args A, const FrameConsts* __restrict__ fcp) {
    using namespace wf;
    const RenderArgs& a = A.a;
    const FrameConsts& fc = *fcp;
    const ColdPlain cold{A.cold};
    const Queue q = A.q[Q_SURF];
    const uint32_t n = min(q.ctl[0], A.capacity);
    Out o_st = {0u, 0u};
    begin_drain(q);
    uint32_t chunk = 0xffffffffu, idx; bool active;
    while (next_chunk(n, &chunk, &idx, &active)) {
        Path p;
        if (active) load_path(&q.rec[idx], p);
        if (active && p.pid == DE_V3_HOLE) active = false;
        if (active) surf_body<CLAMP>(a, fc, cold, p);
        push(A.q[Q_ST], o_st, p, active);
    }
    close_out(A.q[Q_ST], o_st);
}

// ------------------------------------------------------------------------------------------------ SCFIN: volume vertex
__global__ void __launch_bounds__(DE_V3_BLOCK) v3_scfin_kernel(wf::Args A, const FrameConsts* __restrict__ fcp) {
    using namespace wf;
    const RenderArgs& a = A.a;
    const FrameConsts& fc = *fcp;
    const ColdPlain cold{A.cold};
    const Queue q = A.q[Q_SCFIN];
    const uint32_t n = min(q.ctl[0], A.capacity);
    Out o_st = {0u, 0u};
    begin_drain(q);
    uint32_t chunk = 0xffffffffu, idx; bool active;
    while (next_chunk(n, &chunk, &idx, &active)) {
        Path p;
        bool go = false;
        if (active) load_path(&q.rec[idx], p);
        if (active && p.pid == DE_V3_HOLE) active = false;
        if (active) go = scfin_body(a, fc, cold, p) == Q_ST;
        push(A.q[Q_ST], o_st, p, go);
    }
    close_out(A.q[Q_ST], o_st);
}

// ------------------------------------------------------------------------------------------------ SURFFIN: surface vertex
__global__ void __launch_bounds__(DE_V3_BLOCK) v3_surffin_kernel(wf::Args A, const FrameConsts* __restrict__ fcp) {
    using namespace wf;
    const RenderArgs& a = A.a;
    const ColdPlain cold{A.cold};
    const Queue q = A.q[Q_SURFFIN];
    const uint32_t n = min(q.ctl[0], A.capacity);
    Out o_st = {0u, 0u};
    begin_drain(q);
    uint32_t chunk = 0xffffffffu, idx; bool active;
    while (next_chunk(n, &chunk, &idx, &active)) {
        Path p;
        bool go = false;
        if (active) load_path(&q.rec[idx], p);
        if (active && p.pid == DE_V3_HOLE) active = false;
        if (active) go = surffin_body(a, cold, p) == Q_ST;
        push(A.q[Q_ST], o_st, p, go);
    }
    close_out(A.q[Q_ST], o_st);
}
