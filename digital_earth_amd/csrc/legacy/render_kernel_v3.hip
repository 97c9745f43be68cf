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
#include "../de_stages.h"

namespace wf {

struct Rec { uint4 v[4]; };                      // 64 bytes
struct Queue {
    Rec* rec;
    uint32_t* ctl;                                // control words of the CURRENT filling: [0] entries, [1] chunk cursor of the draining kernel
    uint32_t* ctl_next;                           // ... of the next filling: zeroed by the kernel that drains the current one
};

struct Args {
    RenderArgs a;
    Queue q[NQ];
    Cold* cold;
    uint32_t n_items;                             // paths of this launch
    uint32_t capacity;                            // records each queue holds (paths + room for holes): no kernel reads past it
};

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

template <bool CLAMP>
__global__ void __launch_bounds__(DE_V3_BLOCK) v3_st_kernel(wf::Args A, const FrameConsts* __restrict__ fcp) {
    using namespace wf;
    const FrameConsts& fc = *fcp;
    const float MAX_RAY_DIST = (float)(6371e3 * 10.0);
    StStage<CLAMP> st{A.a.topo, fc.land_height_scale, DE_PLANET_R + de_max(fc.land_height_scale, 0.0f) + 8100.0f, MAX_RAY_DIST, 0.0f, 0};
    const int targets[4] = {Q_GAS, Q_CLOUD, Q_EVENT, Q_SURFFIN};
    run_stage<DE_V3_SVC_ST>(A, A.q[Q_ST], st, targets);
}

__global__ void __launch_bounds__(DE_V3_BLOCK) v3_gas_kernel(wf::Args A) {
    using namespace wf;
    GasStage st{A.a.nodes, A.a.dens_table};
    const int targets[4] = {Q_CLOUD, Q_EVENT, Q_SCFIN, Q_SURFFIN};
    run_stage<DE_V3_SVC_GAS>(A, A.q[Q_GAS], st, targets);
}

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
