// render_kernel_v6.hip — path_tracer (pathtracer.py:316-469) as ONE PERSISTENT KERNEL whose stage queues live in LDS: one stage
// scheduler per workgroup (two workgroups of twelve waves per CU).
//
// Why.  render_kernel_v5 showed that the seven stage bodies of render_kernel_v3 can run inside one launch with coherent waves (lane
// utilisation 0.67) — and that handing 64-byte records from wave to wave through HBM chunks and device-wide rings costs what the
// coherence wins: every hop is a write-through, a ticket on a contended control word and a cold L2 read, and four single-class
// waves per SIMD cannot cover those latencies.  Here a path never leaves the CU that started it.  A workgroup owns DE_V6_P record
// slots in LDS (its half of the CU's 160 KB); a path keeps ONE slot from its first ray to its contribution, and what travels between
// the stages is the 16-bit slot number, through eight small rings (seven stages + the free slots) that also live in LDS.  A wave
// asks the scheduler for the fullest ring, claims up to 64 slot numbers with one LDS compare-and-swap, runs the stage body — the
// loop stages with persistent lanes that refill from the ring, as in v3 / v5 — writes the records back in place and appends the
// slot numbers to the rings of the next stages.  No HBM queue traffic at all: the only global memory the scheduler touches is the
// work counter (one fetch-add per 256 new paths), the paths' cold records (64 bytes per SLOT: 37 MB for the whole GPU, L2 / MALL
// resident) and the contribution of a path when it ends.
//
// Results are IDENTICAL to render_kernel_v2 / _v3 / _v5 / the CPU oracle: every path runs the same arithmetic in the same order on
// its own RNG stream; only where and when changes.
//
// Visibility.  All waves of a workgroup share the CU's LDS and its L1: LDS operations of a wave execute in order, a record is
// written (ds_write) before its slot number is (after a workgroup-scope release fence, which also drains the cold-record stores to
// the L1/L2 both waves use); a ring entry is taken only after it has been seen non-empty (acquire).  A producer takes a ticket
// (fetch-add on the ring's tail) and then stores the entries; the consumer that claimed a ticket early waits for the entry — a few
// LDS round trips.  Every wait is bounded: a wave that waits too long raises the abort word and the host reports DE_ERR_HIP.
//
// What makes it fast (profiles/experiments/README.md): (1) YIELD — a loop-stage wave whose ring is dry and that is left
// with few busy lanes suspends its records (the loop's state travels in the record and in spare words of the cold record, bit for bit) and
// hands them back to its own ring, where they fill the idle lanes of another wave; (2) every stage is a separate, non-inlined device
// function: 76-80 VGPRs each instead of 128 + spills for the seven inlined, so six waves per SIMD fit; (3) 56-byte records: the LDS holds
// 1.46 record slots per lane at that occupancy, and slots per lane is what bounds the design.
#include "de_kernels.h"

namespace bs {

using wf::Path;
using wf::NQ;
using wf::Q_ST; using wf::Q_GAS; using wf::Q_CLOUD; using wf::Q_EVENT; using wf::Q_SURF; using wf::Q_SCFIN; using wf::Q_SURFFIN;

enum { Q_FREE = NQ, NR = NQ + 1, Q_PRIMARY = NQ + 1, Q_EXIT = NQ + 2 };
// Geometry.  The stage functions are separate (non-inlined) device functions: each gets the registers ITS body needs (36 ... 78, as the
// stage kernels of render_kernel_v3.hip) instead of the 128 + spills the register allocator gives the seven stages inlined into one
// function, so the kernel fits 80 VGPRs = 6 waves per SIMD.  24 waves per CU = two workgroups of twelve waves, 80 KB of LDS each.
// (One workgroup of sixteen waves with 1984 slots at 128 VGPRs: 138 ms per frame; twelve of those waves: 177 ms — the kernel is bound by
// VALU issue and latency hiding, both of which are waves per SIMD.)
#ifndef DE_V6_WAVES
#define DE_V6_WAVES 12                     // waves per workgroup
#endif
#ifndef DE_V6_WG_PER_CU
#define DE_V6_WG_PER_CU 2
#endif
#ifndef DE_V6_WPE
#define DE_V6_WPE 6                        // waves per SIMD the kernel is compiled for (DE_V6_WAVES x DE_V6_WG_PER_CU / 4)
#endif
#ifndef DE_V6_P
#define DE_V6_P 1120u                      // record slots per workgroup, 56 B each (61 KB): with the rings and the rest the 80 KB a workgroup can have (1008 slots of 64 B: +5.6 % time; 960: +7 %; 768: +19 %)
#endif
#ifndef DE_V6_CAP
#define DE_V6_CAP 1128u                    // entries per ring (> DE_V6_P: an entry is never overwritten before it is taken); indices are taken modulo this
#endif
#define DE_V6_EMPTY 0xffffu
#define DE_V6_CTL_STRIDE 32u
// global control words (index x DE_V6_CTL_STRIDE dwords): 0 work counter, 8.. statistics (64-bit)
enum { G_WORK = 0, G_STAT0 = 8, G_WORDS = 8 + 64 };
// statistics: 0-7 records taken per stage (7 = new paths), 8-15 wave time per stage (100 MHz ticks; 15 = new paths), 16 idle rounds,
// 17 stage entries, 18 failed claims, 19 wave time in the scheduler, 20 yields of a loop stage, 21 records handed back by them, 22 wave time between the end of the work items and the wave's exit, 32-38 trips of the loop stages / passes of the straight-line stages, 40-46 busy lanes summed over them
enum { ST_RECS = 0, ST_CYC = 8, ST_IDLE = 16, ST_ENTRIES = 17, ST_CLAIMFAIL = 18, ST_SCHED = 19, ST_YIELD = 20, ST_YIELDED = 21, ST_DRAIN = 22, ST_TRIPS = 32, ST_LANETRIPS = 40, ST_N = 48 };
enum { ERR_ENTRY_TIMEOUT = 1, ERR_IDLE_WATCHDOG = 3 };

struct Args {
    const RenderArgs* ra;       // the launch's RenderArgs in device memory (written by v6_init_kernel)
    wf::Cold* cold;             // [workgroups x DE_V6_P] cold records, one per record slot
    uint32_t* ctl;              // G_WORDS x DE_V6_CTL_STRIDE dwords
    uint32_t* status;           // host-visible: [0] abort code (0 = clean)
    uint32_t n_items;
    uint32_t idle_limit;        // watchdog: scheduler rounds without work before a wave raises the abort word
    // tuning (de_api.hip: DE_V6_* environment)
    uint32_t svc[3];            // idle lanes at which a loop stage (sphere trace, gas, cloud) services: pushes finished records, refills
    uint32_t yield_max;         // a loop-stage wave whose ring is dry hands its records back when at most this many lanes are busy ...
    uint32_t elsewhere_min;     // ... and another ring (or the free slots, while paths are left to start) holds at least this many entries
    uint32_t retry;             // trips between two looks at a dry ring
    uint32_t enter_min;         // the scheduler prefers rings with at least this many entries (a loop stage entered with fewer runs part-filled)
    uint32_t flat_min;          // a straight-line stage is entered with fewer entries than this only when the loop stages' rings are empty (0: off; measured flat)
};
template <int... T> struct Targets {};

// what every stage function needs of the launch: written to LDS once per workgroup, read (wave-uniform) at each stage entry
struct Cfg {
    const RenderArgs* ra;
    const FrameConsts* fc;
    wf::Cold* cold;             // this workgroup's DE_V6_P cold records
    uint32_t* ctl;
    uint32_t* status;
    uint32_t n_items, idle_limit;
    uint32_t svc[3];
    uint32_t yield_max, elsewhere_min, retry, enter_min;
    uint32_t flat_min;
};
static_assert(sizeof(Cfg) % 4 == 0 && sizeof(Cfg) <= 128, "Cfg is copied dword by dword");

__shared__ uint4 s_rec[3u * DE_V6_P];                // the record slots: three 16-byte parts, part-major ...
__shared__ uint2 s_rec2[DE_V6_P];                    // ... and one of 8 bytes
__shared__ uint16_t s_rings[(NQ + 1) * DE_V6_CAP];   // NR rings of slot numbers
__shared__ uint32_t s_cw[32];                        // head[NR] at 0, tail[NR] at 8, abort at 16
__shared__ float s_node_val[64];
__shared__ unsigned long long s_stats[64];
__shared__ uint32_t s_cfg[32];

DE_DEV int lane_id() { return (int)(threadIdx.x & 63u); }
DE_DEV uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
DE_DEV uint32_t ld_wg(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
DE_DEV uint32_t add_wg(uint32_t* p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
DE_DEV uint32_t ld_e(const uint16_t* p) { return (uint32_t)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
DE_DEV void st_e(uint16_t* p, uint32_t v) { __hip_atomic_store(p, (uint16_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
DE_DEV void release_wg() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
DE_DEV void acquire_wg() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }

// records in LDS: 56 bytes — meta, the generator, P, W, x[0..4] (x[5] is never read; the path's id lives in its cold record: a stage
// that can end a path reads that line anyway) — as three 16-byte parts and one of 8, part-major
DE_DEV void load_rec(const uint4* rec, uint32_t s, Path& p) {
    const uint4 a = rec[s], b = rec[DE_V6_P + s], c = rec[2u * DE_V6_P + s];
    const uint2 d = s_rec2[s];
    p.pid = 0u; p.meta = a.x; p.rng.s0 = a.y; p.rng.s1 = a.z; p.rng.draws = 0;
    p.P = v3(wf::asf(a.w), wf::asf(b.x), wf::asf(b.y)); p.W = v3(wf::asf(b.z), wf::asf(b.w), wf::asf(c.x));
    p.x[0] = wf::asf(c.y); p.x[1] = wf::asf(c.z); p.x[2] = wf::asf(c.w); p.x[3] = wf::asf(d.x); p.x[4] = wf::asf(d.y); p.x[5] = 0.0f;
}
DE_DEV void store_rec(uint4* rec, uint32_t s, const Path& p) {
    rec[s] = make_uint4(p.meta, p.rng.s0, p.rng.s1, wf::asu(p.P.x));
    rec[DE_V6_P + s] = make_uint4(wf::asu(p.P.y), wf::asu(p.P.z), wf::asu(p.W.x), wf::asu(p.W.y));
    rec[2u * DE_V6_P + s] = make_uint4(wf::asu(p.W.z), wf::asu(p.x[0]), wf::asu(p.x[1]), wf::asu(p.x[2]));
    s_rec2[s] = make_uint2(wf::asu(p.x[3]), wf::asu(p.x[4]));
}

// the cold record of the path in THIS lane's slot (the stage bodies index by path id; here the slot decides)
struct ColdAt {
    wf::Cold* c;
    DE_DEV void get_thr_ls(uint32_t, float& thr, float& Ls) const { thr = c->thr; Ls = c->Ls; }
    DE_DEV float get_ls(uint32_t) const { return c->Ls; }
    DE_DEV void set_thr_ls(uint32_t, float thr, float Ls) const { c->thr = thr; c->Ls = Ls; }
    DE_DEV vec3 get_dir(uint32_t) const { return v3(c->dx, c->dy, c->dz); }
    DE_DEV void set_dir(uint32_t, vec3 D) const { c->dx = D.x; c->dy = D.y; c->dz = D.z; }
    DE_DEV void set_surf(uint32_t, vec3 N, float alb, float ocean, float bathy) const { c->nx = N.x; c->ny = N.y; c->nz = N.z; c->alb = alb; c->ocean = ocean; c->bathy = bathy; }
    DE_DEV void get_surf(uint32_t, vec3& N, float& alb, float& ocean, float& bathy) const { N = v3(c->nx, c->ny, c->nz); alb = c->alb; ocean = c->ocean; bathy = c->bathy; }
};

// Everything a wave carries between stages: wave-uniform.
template <bool CLAMP, bool STATS>
struct Wave {
    Cfg A;                      // wave-uniform copy of the workgroup's configuration
    uint4* rec;                 // LDS: the workgroup's record slots
    uint16_t* rings;            // LDS: NR rings of DE_V6_CAP slot numbers
    uint32_t* cw;               // LDS: head[NR] at 0, tail[NR] at 8, abort at 16
    wf::Cold* cold;             // the workgroup's cold records
    unsigned long long* s_st;   // LDS: statistics
    uint32_t n_work, w_lo, w_hi;
    bool primary_left, aborted;
    unsigned long long t_mark, t_drain;

    // a stage function's view of the wave: the configuration from LDS (scalar registers after the broadcast), the wave's own flag
    DE_DEV static Wave enter(bool primary_left) {
        Wave W;
        uint32_t* dst = reinterpret_cast<uint32_t*>(&W.A);
#pragma unroll
        for (uint32_t k = 0; k < sizeof(Cfg) / 4u; ++k) dst[k] = uni(s_cfg[k]);
        W.rec = s_rec; W.rings = s_rings; W.cw = s_cw; W.cold = W.A.cold; W.s_st = s_stats;
        W.n_work = W.A.n_items; W.w_lo = W.w_hi = 0u; W.primary_left = primary_left; W.aborted = false; W.t_mark = 0ull; W.t_drain = 0ull;
        return W;
    }

    DE_DEV void stat(int k, unsigned long long v) { if (STATS && lane_id() == 0) __hip_atomic_fetch_add(s_st + k, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    template <int STAGE>
    DE_DEV void mark() {
        if (!STATS) return;
        const unsigned long long t = __builtin_amdgcn_s_memrealtime();
        stat(STAGE >= 0 ? ST_CYC + STAGE : ST_SCHED, t - t_mark);
        t_mark = t;
    }
    DE_DEV uint32_t* head(uint32_t q) const { return cw + q; }
    DE_DEV uint32_t* tail(uint32_t q) const { return cw + 8u + q; }
    DE_DEV void raise(uint32_t code) {
        if (lane_id() == 0) {
            __hip_atomic_store(cw + 16, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(A.status, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        aborted = true;
    }
    DE_DEV bool abort_seen() {      // the workgroup's own word: the workgroups share nothing but the work counter, the others finish their paths
        if (!aborted && uni(ld_wg(cw + 16)) != 0u) aborted = true;
        return aborted;
    }

    // claim between kmin and kmax entries of ring q: a compare-and-swap on its head, never past its tail
    DE_DEV uint32_t claim(uint32_t q, uint32_t kmin, uint32_t kmax, uint32_t* first) {
        uint32_t got = 0, hh = 0;
        if (lane_id() == 0) {
            uint32_t h = ld_wg(head(q));
#pragma nounroll
            for (int tries = 0; tries < 16; ++tries) {
                const uint32_t t = ld_wg(tail(q));
                const int avail = (int)(t - h);
                if (avail < (int)kmin) break;
                const uint32_t k = (uint32_t)avail < kmax ? (uint32_t)avail : kmax;
                uint32_t expected = h;
                if (__hip_atomic_compare_exchange_strong(head(q), &expected, h + k, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) { got = k; hh = h; break; }
                h = expected;
            }
        }
        *first = uni(hh);
        return uni(got);
    }
    // the slot number in entry idx of ring q (claimed by this wave): wait until its producer has stored it, leave the entry empty
    DE_DEV uint32_t take(uint32_t q, uint32_t idx) {
        uint16_t* e = rings + q * DE_V6_CAP + idx % DE_V6_CAP;
        uint32_t v = DE_V6_EMPTY;
#pragma nounroll
        for (uint32_t spins = 0; spins < (1u << 20); ++spins) {
            v = ld_e(e);
            if (v != DE_V6_EMPTY) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (v != DE_V6_EMPTY) st_e(e, DE_V6_EMPTY);
        return v;
    }
    // append the slot numbers of the lanes with pred to ring Q (their records are written and released)
    template <int Q>
    DE_DEV void push(uint32_t slot, bool pred) {
        const unsigned long long mask = __ballot(pred);
        if (mask == 0ull) return;
        const uint32_t n = (uint32_t)__popcll(mask), rank = (uint32_t)__popcll(mask & ((1ull << lane_id()) - 1ull));
        uint32_t base = 0;
        if (lane_id() == 0) base = add_wg(tail(Q), n);
        base = uni(base);
        if (pred) st_e(rings + Q * DE_V6_CAP + (base + rank) % DE_V6_CAP, slot);
    }

    // ---- the scheduler: which stage this wave runs next — a queue id, Q_PRIMARY, or Q_EXIT
    DE_DEV int schedule() {
        uint32_t idle = 0;
        for (;;) {
            if (abort_seen()) return Q_EXIT;
            int ready = 0;
            const uint32_t l = (uint32_t)lane_id();
            if (l < (uint32_t)NR) ready = (int)(ld_wg(tail(l)) - ld_wg(head(l)));
            int best_q = -1, best = 0, low_q = -1, low = 0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int r = __builtin_amdgcn_readlane(ready, q);
                const bool shallow_flat = q > Q_CLOUD && r < (int)A.flat_min;      // the straight-line stages wait for a fuller pass while the loop stages have work
                if (!shallow_flat && r > best) { best = r; best_q = q; }
                if (shallow_flat && r > low) { low = r; low_q = q; }
            }
            if (best_q < 0) { best_q = low_q; best = low; }
            // a shallow ring is left alone for a few rounds (its records fill the idle lanes of the waves already in that stage, or it
            // deepens) unless nothing else is there
            if (best_q >= 0 && best < (int)A.enter_min && idle < 4u) best_q = -1;
            const int free_slots = __builtin_amdgcn_readlane(ready, Q_FREE);
            // new paths as soon as a wave-load of slots is free: the population stays at the pool's size, the rings stay deep
            if (primary_left && free_slots >= 64) return Q_PRIMARY;
            if (best_q >= 0) { stat(ST_ENTRIES, 1); return best_q; }
            if (!primary_left && free_slots == (int)DE_V6_P) return Q_EXIT;      // every slot is home and nothing is left to start
            stat(ST_IDLE, 1);
            if (++idle > A.idle_limit) { raise(ERR_IDLE_WATCHDOG); return Q_EXIT; }
            if (idle < 16u) __builtin_amdgcn_s_sleep(2); else __builtin_amdgcn_s_sleep(16);
        }
    }

    // ---- a loop stage: persistent lanes.  A lane takes a slot, loads its record, runs the stage's loop on it one step per trip of the
    // wave loop, keeps the finished record until the wave's next service; a service writes the finished records back, appends their
    // slots to the next stages' rings and refills the idle lanes from this stage's ring.
    template <int Q, typename Stage, int... TARGETS>
    DE_DEV void run_loop(Stage& st, Targets<TARGETS...>) {
        const int SERVICE = (int)A.svc[Q];
        Path p;
        uint32_t slot = 0;
        bool busy = false, fin = false;
        int target = -1;
        bool dry = false;
        uint32_t trips = 0, retry_at = 0;
        for (;;) {
            const int nb = __popcll(__ballot(busy));
            const bool may_refill = !dry || trips >= retry_at;
            if (nb == 0 || (64 - nb >= SERVICE && (may_refill || __ballot(fin) != 0ull))) {
                if (__ballot(fin) != 0ull) {
                    // the stage's epilogue for all the lanes that finished since the last service at once (run where a lane finishes it
                    // would execute for two or three lanes of 64 in most trips)
                    if (fin) { target = st.finish(p); store_rec(rec, slot, p); }
                    release_wg();
                    (push<TARGETS>(slot, fin && target == TARGETS), ...);
                    fin = false;
                }
                if (may_refill || nb == 0) {
                    dry = false;
                    const unsigned long long m_need = __ballot(!busy);
                    uint32_t first = 0;
                    const uint32_t got = claim(Q, 1u, (uint32_t)__popcll(m_need), &first);
                    if (got == 0u) {
                        dry = true; retry_at = trips + A.retry;
                        // Nothing to refill from, a few long records left, and other work waits: hand the records back — the loop's
                        // state goes to the spare words of their cold records — so that they fill the idle lanes of another wave of this
                        // stage instead of running here at a fraction of the wave (cloud segments have a heavy tail)
                        if (nb > 0 && nb <= (int)A.yield_max && work_elsewhere(Q)) {
                            if (busy) {
                                st.suspend(p, cold + slot);
                                p.meta |= 0x80000000u;
                                store_rec(rec, slot, p);
                            }
                            release_wg();
                            push<Q>(slot, busy);
                            stat(ST_YIELD, 1); stat(ST_YIELDED, (unsigned long long)nb);
                            busy = false;
                            break;
                        }
                    } else {
                        const uint32_t my = (uint32_t)__popcll(m_need & ((1ull << lane_id()) - 1ull));
                        const bool mine = !busy && my < got;
                        uint32_t s = 0;
                        if (mine) s = take(Q, first + my);
                        if (__ballot(mine && s == DE_V6_EMPTY) != 0ull) { raise(ERR_ENTRY_TIMEOUT); break; }
                        acquire_wg();
                        if (mine) {
                            slot = s; load_rec(rec, slot, p);
                            const bool resumed = (p.meta & 0x80000000u) != 0u;
                            p.meta &= 0x7fffffffu;
                            st.begin(p);
                            if (resumed) st.restore(cold + slot);
                            busy = true;
                        }
                        stat(ST_RECS + Q, got);
                    }
                }
                if (__ballot(busy) == 0ull) { if (dry || aborted) break; else continue; }
            }
            if (STATS) { stat(ST_TRIPS + Q, 1); stat(ST_LANETRIPS + Q, (unsigned long long)__popcll(__ballot(busy))); }
            if (busy) {
                if (st.step(p)) { busy = false; fin = true; }
            }
            trips++;
        }
    }
    // is there a wave-load of work outside ring q (another stage's ring, or free slots for new paths)?
    DE_DEV bool work_elsewhere(uint32_t q) {
        int ready = 0;
        const uint32_t l = (uint32_t)lane_id();
        if (l < (uint32_t)NR && l != q) ready = (int)(ld_wg(tail(l)) - ld_wg(head(l)));
        if (l == (uint32_t)Q_FREE && !primary_left) ready = 0;
        return __ballot(ready >= (int)A.elsewhere_min) != 0ull;
    }

    // ---- a straight-line stage: up to 64 slots per pass, one record per lane, written back in place
    template <int Q, typename Body, int... TARGETS>
    DE_DEV void run_flat(Body body, Targets<TARGETS...>) {
        for (;;) {
            uint32_t first = 0;
            const uint32_t n = claim(Q, 1u, 64u, &first);
            if (n == 0u) break;
            const bool active = (uint32_t)lane_id() < n;
            uint32_t slot = 0;
            if (active) slot = take(Q, first + (uint32_t)lane_id());
            if (__ballot(active && slot == DE_V6_EMPTY) != 0ull) { raise(ERR_ENTRY_TIMEOUT); break; }
            acquire_wg();
            Path p;
            int target = -2;
            if (active) {
                load_rec(rec, slot, p);
                p.pid = wf::asu(cold[slot].pad0);
                target = body(p, ColdAt{cold + slot});
                if (target >= 0) store_rec(rec, slot, p);
            }
            stat(ST_RECS + Q, n);
            if (STATS) { stat(ST_TRIPS + Q, 1); stat(ST_LANETRIPS + Q, n); }
            release_wg();
            (push<TARGETS>(slot, target == TARGETS), ...);
            push<Q_FREE>(slot, target == -1);
        }
    }

    // ---- new paths: 64 work items = the 64 pixels of one screen tile for one sample; the primary sphere trace runs here (coherent rays)
    DE_DEV void run_primary(const RenderArgs& a, const FrameConsts& fc) {
        if (w_lo >= w_hi) {          // 256 work items per device-wide atomic: four passes
            uint32_t w = 0;
            if (lane_id() == 0) w = __hip_atomic_fetch_add(A.ctl + G_WORK * DE_V6_CTL_STRIDE, 256u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            w = uni(w);
            if (w >= n_work) { primary_left = false; if (STATS) t_drain = __builtin_amdgcn_s_memrealtime(); return; }
            w_lo = w; w_hi = min(w + 256u, n_work);
        }
        const uint32_t n = min(64u, w_hi - w_lo);
        uint32_t first = 0;
        if (claim(Q_FREE, n, n, &first) == 0u) { stat(ST_CLAIMFAIL, 1); return; }      // another wave was quicker: back to the scheduler
        const uint32_t w = w_lo + (uint32_t)lane_id();
        w_lo += n;
        const bool active = (uint32_t)lane_id() < n;
        uint32_t slot = 0;
        if (active) slot = take(Q_FREE, first + (uint32_t)lane_id());
        if (__ballot(active && slot == DE_V6_EMPTY) != 0ull) { raise(ERR_ENTRY_TIMEOUT); return; }
        acquire_wg();
        const float MAX_RAY_DIST = (float)(6371e3 * 10.0);
        Path p;
        int target = -2;
        if (active) {
            wf::new_path(a, fc, bs::s_node_val, w, p);
            const float t = wf::sphere_trace<CLAMP>(a, p, fc.land_height_scale, DE_PLANET_R + de_max(fc.land_height_scale, 0.0f) + 8100.0f, MAX_RAY_DIST);
            target = wf::after_sphere_trace(p, t, MAX_RAY_DIST);
            store_rec(rec, slot, p);
            cold[slot].pad0 = wf::asf(p.pid);
        }
        stat(ST_RECS + NQ, n);
        release_wg();
        push<Q_GAS>(slot, target == Q_GAS);
        push<Q_CLOUD>(slot, target == Q_CLOUD);
        push<Q_EVENT>(slot, target == Q_EVENT);
    }
};

}  // namespace bs


// zero the control words and put the launch's RenderArgs where the persistent kernel reads them
__global__ void v6_init_kernel(uint32_t* ctl, RenderArgs a, RenderArgs* dst) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (uint32_t)bs::G_WORDS) { ctl[i * DE_V6_CTL_STRIDE] = 0u; ctl[i * DE_V6_CTL_STRIDE + 1u] = 0u; }
    if (i == 0u) *dst = a;
}

// ---- the stage functions: NOT inlined into the kernel — one register allocation per stage (see "Geometry").  A stage function builds
// its view of the wave from LDS, runs the stage until its ring is dry (or it hands its records back) and returns to the scheduler.
#define DE_V6_STAGE template <bool CLAMP, bool STATS> __device__ __attribute__((noinline)) void
DE_V6_STAGE v6_stage_st(bool primary_left) {
    using namespace bs;
    Wave<CLAMP, STATS> W = Wave<CLAMP, STATS>::enter(primary_left);
    const RenderArgs& a = *W.A.ra;
    const FrameConsts& fc = *W.A.fc;
    const float MAX_RAY_DIST = (float)(6371e3 * 10.0);
    StStage<CLAMP> st{a.topo, fc.land_height_scale, DE_PLANET_R + de_max(fc.land_height_scale, 0.0f) + 8100.0f, MAX_RAY_DIST, 0.0f, 0};
    W.template run_loop<Q_ST>(st, Targets<Q_GAS, Q_CLOUD, Q_EVENT, Q_SURFFIN>{});
}
DE_V6_STAGE v6_stage_gas(bool primary_left) {
    using namespace bs;
    Wave<CLAMP, STATS> W = Wave<CLAMP, STATS>::enter(primary_left);
    const RenderArgs& a = *W.A.ra;
    GasStage st{a.nodes, a.dens_table};
    W.template run_loop<Q_GAS>(st, Targets<Q_CLOUD, Q_EVENT, Q_SCFIN, Q_SURFFIN>{});
}
DE_V6_STAGE v6_stage_cloud(bool primary_left) {
    using namespace bs;
    Wave<CLAMP, STATS> W = Wave<CLAMP, STATS>::enter(primary_left);
    const RenderArgs& a = *W.A.ra;
    CloudStage<CLAMP> st{a.clouds};
    W.template run_loop<Q_CLOUD>(st, Targets<Q_EVENT, Q_SCFIN, Q_SURFFIN>{});
}
DE_V6_STAGE v6_stage_event(bool primary_left) {
    using namespace bs;
    Wave<CLAMP, STATS> W = Wave<CLAMP, STATS>::enter(primary_left);
    const RenderArgs& a = *W.A.ra;
    const FrameConsts& fc = *W.A.fc;
    W.template run_flat<Q_EVENT>([&](Path& p, const ColdAt& cold) { return wf::event_body<CLAMP>(a, fc, cold, p); }, Targets<Q_GAS, Q_CLOUD, Q_SCFIN, Q_SURF>{});
}
DE_V6_STAGE v6_stage_surf(bool primary_left) {
    using namespace bs;
    Wave<CLAMP, STATS> W = Wave<CLAMP, STATS>::enter(primary_left);
    const RenderArgs& a = *W.A.ra;
    const FrameConsts& fc = *W.A.fc;
    W.template run_flat<Q_SURF>([&](Path& p, const ColdAt& cold) { return wf::surf_body<CLAMP>(a, fc, cold, p); }, Targets<Q_ST>{});
}
DE_V6_STAGE v6_stage_scfin(bool primary_left) {
    using namespace bs;
    Wave<CLAMP, STATS> W = Wave<CLAMP, STATS>::enter(primary_left);
    const RenderArgs& a = *W.A.ra;
    const FrameConsts& fc = *W.A.fc;
    W.template run_flat<Q_SCFIN>([&](Path& p, const ColdAt& cold) { return wf::scfin_body(a, fc, cold, p); }, Targets<Q_ST>{});
}
DE_V6_STAGE v6_stage_surffin(bool primary_left) {
    using namespace bs;
    Wave<CLAMP, STATS> W = Wave<CLAMP, STATS>::enter(primary_left);
    const RenderArgs& a = *W.A.ra;
    W.template run_flat<Q_SURFFIN>([&](Path& p, const ColdAt& cold) { return wf::surffin_body(a, cold, p); }, Targets<Q_ST>{});
}

template <bool CLAMP, bool STATS>
__global__ void __attribute__((amdgpu_flat_work_group_size(64 * DE_V6_WAVES, 64 * DE_V6_WAVES), amdgpu_waves_per_eu(DE_V6_WPE, DE_V6_WPE)))
v6_kernel(bs::Args A, const FrameConsts* fcp) {
    using namespace bs;
    for (uint32_t i = threadIdx.x; i < NR * DE_V6_CAP; i += blockDim.x) {
        const uint32_t q = i / DE_V6_CAP, k = i % DE_V6_CAP;
        s_rings[i] = (uint16_t)((q == (uint32_t)Q_FREE && k < DE_V6_P) ? k : DE_V6_EMPTY);
    }
    if (threadIdx.x < 32u) s_cw[threadIdx.x] = (threadIdx.x == 8u + (uint32_t)Q_FREE) ? DE_V6_P : 0u;
    if (threadIdx.x < 64u) { s_node_val[threadIdx.x] = A.ra->node_val[threadIdx.x]; s_stats[threadIdx.x] = 0ull; }
    if (threadIdx.x == 0u) {
        Cfg c;
        c.ra = A.ra; c.fc = fcp; c.cold = A.cold + (size_t)blockIdx.x * DE_V6_P; c.ctl = A.ctl; c.status = A.status;
        c.n_items = A.n_items; c.idle_limit = A.idle_limit;
        c.svc[0] = A.svc[0]; c.svc[1] = A.svc[1]; c.svc[2] = A.svc[2];
        c.yield_max = A.yield_max; c.elsewhere_min = A.elsewhere_min; c.retry = A.retry; c.enter_min = A.enter_min; c.flat_min = A.flat_min;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&c);
        for (uint32_t k = 0; k < sizeof(Cfg) / 4u; ++k) s_cfg[k] = src[k];
    }
    __syncthreads();
    Wave<CLAMP, STATS> W = Wave<CLAMP, STATS>::enter(true);
    W.t_mark = STATS ? __builtin_amdgcn_s_memrealtime() : 0ull;

    for (;;) {
        const int s = W.schedule();
        W.template mark<-1>();
        if (s == Q_EXIT) { if (STATS && W.t_drain) W.stat(ST_DRAIN, __builtin_amdgcn_s_memrealtime() - W.t_drain); break; }
        switch (s) {
        case Q_ST: v6_stage_st<CLAMP, STATS>(W.primary_left); W.template mark<Q_ST>(); break;
        case Q_GAS: v6_stage_gas<CLAMP, STATS>(W.primary_left); W.template mark<Q_GAS>(); break;
        case Q_CLOUD: v6_stage_cloud<CLAMP, STATS>(W.primary_left); W.template mark<Q_CLOUD>(); break;
        case Q_EVENT: v6_stage_event<CLAMP, STATS>(W.primary_left); W.template mark<Q_EVENT>(); break;
        case Q_SURF: v6_stage_surf<CLAMP, STATS>(W.primary_left); W.template mark<Q_SURF>(); break;
        case Q_SCFIN: v6_stage_scfin<CLAMP, STATS>(W.primary_left); W.template mark<Q_SCFIN>(); break;
        case Q_SURFFIN: v6_stage_surffin<CLAMP, STATS>(W.primary_left); W.template mark<Q_SURFFIN>(); break;
        default: {
            const RenderArgs& a = *v5_opaque(W.A.ra);
            const FrameConsts& fc = *v5_opaque(W.A.fc);
            W.run_primary(a, fc);
            W.template mark<7>();
        } break;
        }
    }
    if (STATS) {
        __syncthreads();
        if (threadIdx.x < (uint32_t)ST_N && s_stats[threadIdx.x])
            __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(A.ctl + (G_STAT0 + threadIdx.x) * DE_V6_CTL_STRIDE), s_stats[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
