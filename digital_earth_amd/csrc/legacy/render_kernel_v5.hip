// render_kernel_v5.hip — path_tracer (pathtracer.py:316-469) as ONE PERSISTENT KERNEL: a stage scheduler over chunked HBM queues.
//
// Why.  render_kernel_v3 runs the integrator as a level-synchronous pipeline: ten launches per path vertex, each draining one stage
// queue with every wave in the same loop.  Its waves are coherent, but (i) a SIMD that holds waves of ONE class cannot cover that
// class's dependent loads (gas 22 %, cloud 37 % of the VALU issue ceiling), (ii) every launch ends with its own tail, (iii) the
// deep vertex rounds are all tail, which is why the default handed them to render_kernel_v2.  Here the same seven stage bodies
// (shared with render_kernel_v3.hip: one statement of the integrator per family) run inside one launch of persistent one-wave
// workgroups.  A wave asks the scheduler for the fullest queue, claims chunks of 64 records from it, runs that stage's body on them
// — the loop stages with persistent lanes that refill from the queue, exactly as in v3 — and appends the results to the queues of
// the next stages; when its queue runs dry it goes back to the scheduler.  Waves of different stages share every SIMD, there is no
// level barrier and no hand-over, and new paths are only started when no queued work is left for a wave: the population in flight
// stays near the minimum that keeps the machine busy (a few hundred MB of queue memory instead of 512 B per path of a batch).
//
// Results are IDENTICAL to render_kernel_v2 / _v3 / the CPU oracle: every path runs the same arithmetic in the same order on its
// own RNG stream; only where and when changes.
//
// Inter-workgroup visibility (the per-XCD L2s are not coherent with each other, a CU's L1 is never refreshed by other CUs' stores):
// records and cold records are written THROUGH (sc1) and read with sc1 loads; a chunk is published — one agent-scope store of its
// ring cell — only after the writing wave has drained its stores (s_waitcnt vmcnt(0)); control words and ring cells are agent-scope
// atomics.  No fences, no spinning on data: the only waits are on a ring cell between its producer's ticket and its store (a few
// hundred ns), and on the free-chunk ring when the pool is exhausted (sized so that it is not).  Every wait is bounded: a wave that
// waits too long raises the abort word, every wave leaves, and the host reports DE_ERR_HIP instead of a hung GPU.
#include "../de_stages.h"

namespace ps {

using wf::Path;
using wf::NQ;
using wf::Q_ST; using wf::Q_GAS; using wf::Q_CLOUD; using wf::Q_EVENT; using wf::Q_SURF; using wf::Q_SCFIN; using wf::Q_SURFFIN;

enum { Q_PRIMARY = NQ, Q_EXIT = NQ + 1 };
#define DE_V5_CH 64u                       // records per chunk = one wave-load
#define DE_V5_EMPTY 0xffffffffu
#define DE_V5_NONE 0xffffffffu
#define DE_V5_CTL_STRIDE 32u               // one control word per 128-byte line
#define DE_V5_MAX_SHARDS 256
// control words (index x DE_V5_CTL_STRIDE dwords)
enum { C_WORK = 0, C_DONE = 1, C_ABORT = 2, C_FREE_HEAD = 3, C_FREE_TAIL = 4, C_STAT0 = 8, C_QUEUE0 = 56, C_WORDS = C_QUEUE0 + 2 * NQ * DE_V5_MAX_SHARDS };
// statistics (C_STAT0 + k): 0-7 records taken per stage (7 = primary), 8-15 wave-cycles (s_memtime / 64) per stage, 16 idle loops,
// 17 partial chunks published, 18 failed claims, 19 stage entries, 20 chunk allocations from the global ring, 21 steals
enum { ST_RECS = 0, ST_CYC = 8, ST_IDLE = 16, ST_PARTIAL = 17, ST_CLAIMFAIL = 18, ST_ENTRIES = 19, ST_GALLOC = 20, ST_STEAL = 21, ST_SCHED_CYC = 22, ST_T_INPUT = 23, ST_T_PUSH = 24, ST_T_SERVICE = 25, ST_T_BODY = 26, ST_T_ENSURE = 27, ST_T_FLUSH = 28, ST_T_SNAP = 29, ST_T_CLAIM = 30, ST_TRIPS = 32, ST_LANETRIPS = 35, ST_DRY = 38, ST_N = 40 };      // ST_N <= 64: one LDS slot per statistic
enum { ERR_CELL_TIMEOUT = 1, ERR_POOL_EMPTY = 2, ERR_IDLE_WATCHDOG = 3, ERR_RING_FULL = 4 };

struct Args {
    const RenderArgs* ra;       // the launch's RenderArgs in device memory (written by v5_init_kernel): each stage reads what it needs
                                // through a pointer the compiler cannot see through, so that nothing of it stays live between stages
    uint4* pool;                // pool_chunks x DE_V5_CH x 64-byte records
    uint32_t* ctl;              // C_WORDS x DE_V5_CTL_STRIDE dwords
    uint32_t* rings;            // [n_shards x NQ][ring_mask + 1] cells: chunk id x 128 + records, DE_V5_EMPTY when free
    uint32_t* free_ring;        // [pool_chunks] chunk ids
    unsigned long long* cold;   // 8 x 8 bytes per path
    uint32_t* status;           // host-visible: [0] abort code (0 = clean)
    uint32_t n_items, pool_chunks, ring_mask, n_shards;
    // as the FINISHER of the wavefront pipeline (render_kernel_v3.hip): the work items are not fresh samples but the path rays the
    // pipeline queued for its next vertex round — *resume_count 64-byte records of its sphere-trace queue (holes included) plus the
    // paths' throughput and radiance in its cold records.  nullptr = start from (pixel, sample).
    const uint4* resume_rec;
    const uint32_t* resume_count;
    const float* resume_cold;        // 16 floats per path: thr, Ls, ...
    uint32_t resume_capacity;
    uint32_t reserve_chunks;    // new paths are started only while at least this many chunks are free
    uint32_t backlog_target;    // ... and while fewer than this many chunks are queued in the wave's shard: deep queues keep the waves
                                // in their stages (no scheduler round trips, full chunks), a bounded population keeps them in the caches
    uint32_t claim_k;           // chunks per claim (1..4)
    uint32_t age_max;           // a partial output chunk is published once it is this old (ticks of the 100 MHz constant clock)
    uint32_t closers;           // workgroups that stay until the last path has ended; the others leave when they run out of work
    uint32_t idle_limit;        // watchdog: scheduler rounds without work before a wave raises the abort word
};
template <int... T> struct Targets {};
#ifndef DE_V5_DUAL
#define DE_V5_DUAL 0           // experiment: bit 0 = the sphere-trace stage holds two records per lane (run_loop2), bit 1 = the cloud stage
#endif

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define DE_V5_SC1 16            // aux bits of the raw buffer builtins on gfx950: 16 = sc1 (write-through / L1 bypass)

DE_DEV int lane_id() { return (int)(threadIdx.x & 63u); }
DE_DEV uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
DE_DEV uint32_t ld_agent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DE_DEV void st_agent(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DE_DEV uint32_t add_agent(uint32_t* p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DE_DEV void add_stat(uint32_t* p, unsigned long long v) { __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// records: four 16-byte sc1 accesses through a buffer descriptor over the pool (<= 4 GB).  A chunk is laid out QUAD-MAJOR — the first
// 16 bytes of its 64 records, then the second 16 bytes of each, ... (4 x 1 KB) — so that each of the four instructions of a wave that
// loads or stores consecutive records touches one contiguous run of bytes: whole 128-byte lines, each moved once.  (Record-major,
// every instruction touched all 32 lines of the chunk: 4x the L2 requests and 16-byte partial write-throughs, measured 1.4 TB of
// L2-miss traffic per frame.)
DE_DEV int rec_offset(uint32_t rec) { return (int)((rec >> 6) * (DE_V5_CH * 64u) + (rec & 63u) * 16u); }
DE_DEV void load_rec(__amdgpu_buffer_rsrc_t rs, uint32_t rec, Path& p) {
    const int off = rec_offset(rec);
    const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, DE_V5_SC1);
    const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 1024, 0, DE_V5_SC1);
    const u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 2048, 0, DE_V5_SC1);
    const u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 3072, 0, DE_V5_SC1);
    p.pid = a.x; p.meta = a.y; p.rng.s0 = a.z; p.rng.s1 = a.w; p.rng.draws = 0;
    p.P = v3(wf::asf(b.x), wf::asf(b.y), wf::asf(b.z)); p.W = v3(wf::asf(b.w), wf::asf(c.x), wf::asf(c.y));
    p.x[0] = wf::asf(c.z); p.x[1] = wf::asf(c.w); p.x[2] = wf::asf(d.x); p.x[3] = wf::asf(d.y); p.x[4] = wf::asf(d.z); p.x[5] = wf::asf(d.w);
}
template <bool LOCAL>
DE_DEV void store_rec(__amdgpu_buffer_rsrc_t rs, uint32_t rec, const Path& p) {
    constexpr int AUX = LOCAL ? 0 : DE_V5_SC1;      // LOCAL: producer and consumer share an XCD's L2 (a write-back store reaches it; the consumer's sc1 load bypasses its L1)
    const int off = rec_offset(rec);
    u32x4 a, b, c, d;
    a.x = p.pid; a.y = p.meta; a.z = p.rng.s0; a.w = p.rng.s1;
    b.x = wf::asu(p.P.x); b.y = wf::asu(p.P.y); b.z = wf::asu(p.P.z); b.w = wf::asu(p.W.x);
    c.x = wf::asu(p.W.y); c.y = wf::asu(p.W.z); c.z = wf::asu(p.x[0]); c.w = wf::asu(p.x[1]);
    d.x = wf::asu(p.x[2]); d.y = wf::asu(p.x[3]); d.z = wf::asu(p.x[4]); d.w = wf::asu(p.x[5]);
    __builtin_amdgcn_raw_buffer_store_b128(a, rs, off, 0, AUX);
    __builtin_amdgcn_raw_buffer_store_b128(b, rs, off + 1024, 0, AUX);
    __builtin_amdgcn_raw_buffer_store_b128(c, rs, off + 2048, 0, AUX);
    __builtin_amdgcn_raw_buffer_store_b128(d, rs, off + 3072, 0, AUX);
}

// cold records: 8-byte agent-scope (sc1) accesses, any buffer size.  Layout (8-byte words): 0 (thr, Ls), 1 (dx, dy), 2 (dz, -),
// 3 (nx, ny), 4 (nz, alb), 5 (ocean, bathy).  A path's cold record is written by the wave that then queues the path's record
// (drained before the chunk is published) and read by the wave that claimed it: ordered by the chunk hand-over.
template <bool LOCAL>
struct ColdWT {
    unsigned long long* base;
    DE_DEV static unsigned long long pack(float a, float b) { return (unsigned long long)wf::asu(a) | ((unsigned long long)wf::asu(b) << 32); }
    DE_DEV unsigned long long ld(uint32_t pid, int k) const { return __hip_atomic_load(base + (size_t)pid * 8u + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    DE_DEV void st(uint32_t pid, int k, unsigned long long v) const {
        if (LOCAL) base[(size_t)pid * 8u + k] = v;
        else __hip_atomic_store(base + (size_t)pid * 8u + k, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    DE_DEV void get_thr_ls(uint32_t pid, float& thr, float& Ls) const { const unsigned long long v = ld(pid, 0); thr = wf::asf((uint32_t)v); Ls = wf::asf((uint32_t)(v >> 32)); }
    DE_DEV float get_ls(uint32_t pid) const { return wf::asf((uint32_t)(ld(pid, 0) >> 32)); }
    DE_DEV void set_thr_ls(uint32_t pid, float thr, float Ls) const { st(pid, 0, pack(thr, Ls)); }
    DE_DEV vec3 get_dir(uint32_t pid) const {
        const unsigned long long a = ld(pid, 1), b = ld(pid, 2);
        return v3(wf::asf((uint32_t)a), wf::asf((uint32_t)(a >> 32)), wf::asf((uint32_t)b));
    }
    DE_DEV void set_dir(uint32_t pid, vec3 D) const { st(pid, 1, pack(D.x, D.y)); st(pid, 2, pack(D.z, 0.0f)); }
    DE_DEV void set_surf(uint32_t pid, vec3 N, float alb, float ocean, float bathy) const {
        st(pid, 3, pack(N.x, N.y)); st(pid, 4, pack(N.z, alb)); st(pid, 5, pack(ocean, bathy));
    }
    DE_DEV void get_surf(uint32_t pid, vec3& N, float& alb, float& ocean, float& bathy) const {
        const unsigned long long a = ld(pid, 3), b = ld(pid, 4), c = ld(pid, 5);
        N = v3(wf::asf((uint32_t)a), wf::asf((uint32_t)(a >> 32)), wf::asf((uint32_t)b));
        alb = wf::asf((uint32_t)(b >> 32)); ocean = wf::asf((uint32_t)c); bathy = wf::asf((uint32_t)(c >> 32));
    }
};

// Everything a wave carries between stages.  All members are wave-uniform (SGPRs) except `stash`: lane i holds free chunk id i.
template <bool CLAMP, bool STATS, bool LOCAL>
struct Wave {
    const Args& A;
    const float* s_node_val;
    __amdgpu_buffer_rsrc_t rs;
    uint32_t shard;
    // output side: one partially filled chunk per target queue
    uint32_t oc[NQ], of[NQ], oa[NQ];
    // free chunks held by this wave
    uint32_t stash, n_stash;
    // input side: the queue being drained
    uint32_t in_ring;           // index of the ring ((shard x NQ + q), possibly another shard's when stealing)
    uint32_t in_chunk, in_lo, in_hi, in_h, in_n;
    uint32_t in_cells, in_k;    // the cells of the current claim as first read (lane j = cell j; DE_V5_EMPTY = not yet stored), entries read so far
    uint32_t done_local;
    uint32_t n_work;            // work items of the launch: A.n_items, or the records to resume
    uint32_t w_lo, w_hi;        // work items (new paths) this wave has drawn from the device-wide counter
    uint32_t victim;            // the other team this wave looks at next when its own has nothing
    bool primary_left, aborted;
    // statistics (STATS): accumulated in the wave's LDS block, index = ST_* (registers would cost ~60 SGPRs and change the kernel measured)
    unsigned long long* s_st;
    unsigned long long t_mark;
    DE_DEV void stat(int k, unsigned long long v) { if (STATS && lane_id() == 0) s_st[k] += v; }

    DE_DEV uint32_t* ctl(uint32_t k) const { return A.ctl + k * DE_V5_CTL_STRIDE; }
    DE_DEV uint32_t* q_head(uint32_t ring) const { return ctl(C_QUEUE0 + 2u * ring); }
    DE_DEV uint32_t* q_tail(uint32_t ring) const { return ctl(C_QUEUE0 + 2u * ring + 1u); }
    DE_DEV uint32_t* ring_cells(uint32_t ring) const { return A.rings + (size_t)ring * ((size_t)A.ring_mask + 1u); }

    DE_DEV void raise(uint32_t code) {
        if (lane_id() == 0) {
            st_agent(ctl(C_ABORT), code);
            __hip_atomic_store(A.status, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        aborted = true;
    }
    DE_DEV bool abort_seen() { if (!aborted && uni(ld_agent(ctl(C_ABORT))) != 0u) aborted = true; return aborted; }

    // ---- ring cells: the producer takes a ticket (tail), then stores the cell; the consumer that owns the ticket waits for the store
    DE_DEV uint32_t take_cell(uint32_t* cells, uint32_t mask, uint32_t idx) {
        uint32_t c = DE_V5_EMPTY;
        if (lane_id() == 0) {
            uint32_t* cell = cells + (idx & mask);
#pragma nounroll
            for (uint32_t spins = 0; spins < (1u << 22); ++spins) {
                c = ld_agent(cell);
                if (c != DE_V5_EMPTY) break;
                __builtin_amdgcn_s_sleep(2);
            }
            if (c != DE_V5_EMPTY) st_agent(cell, DE_V5_EMPTY);
        }
        c = uni(c);
        if (c == DE_V5_EMPTY) raise(ERR_CELL_TIMEOUT);
        return c;
    }
    // claim up to kmax entries of a ring whose head / tail words are given: a compare-and-swap on the head, never past the tail
    DE_DEV uint32_t claim(uint32_t* head, uint32_t* tail, uint32_t kmin, uint32_t kmax, uint32_t* first) {
        uint32_t got = 0, hh = 0;
        if (lane_id() == 0) {
            uint32_t h = ld_agent(head);
#pragma nounroll
            for (int tries = 0; tries < 8; ++tries) {
                const uint32_t t = ld_agent(tail);
                const int avail = (int)(t - h);
                if (avail < (int)kmin) break;
                const uint32_t k = (uint32_t)avail < kmax ? (uint32_t)avail : kmax;
                uint32_t expected = h;
                if (__hip_atomic_compare_exchange_strong(head, &expected, h + k, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { got = k; hh = h; break; }
                h = expected;
            }
        }
        *first = uni(hh);
        return uni(got);
    }

    // ---- free chunks.  The wave keeps a few in `stash` (lane i = entry i).  ensure_stash() — called once per service / chunk / pass,
    // i.e. between any two of its calls at most eight chunks are taken — is the only place that goes to the device-wide ring for more.
    DE_DEV void ensure_stash() {
        if (n_stash >= 8u) return;
        // eight tickets of the free ring with ONE fetch-add (thousands of waves on a compare-and-swap retry each other quadratically:
        // measured 27 % of all wave-time at 16 waves per CU); a ticket past the tail waits for a chunk to come back — the pool is sized
        // so that this does not happen, and the wait is bounded
        uint32_t first = 0;
        if (lane_id() == 0) first = add_agent(ctl(C_FREE_HEAD), 8u);
        first = uni(first);
        stat(ST_GALLOC, 1);
        const uint32_t l = (uint32_t)lane_id();
        uint32_t c = 0u;
        if (l >= n_stash && l < n_stash + 8u) {
            uint32_t* cell = A.free_ring + ((first + (l - n_stash)) & (A.pool_chunks - 1u));
            c = DE_V5_EMPTY;
#pragma nounroll
            for (uint32_t spins = 0; spins < (1u << 21); ++spins) {
                c = ld_agent(cell);
                if (c != DE_V5_EMPTY) break;
                __builtin_amdgcn_s_sleep(4);
            }
            if (c != DE_V5_EMPTY) st_agent(cell, DE_V5_EMPTY);
            stash = c;
        }
        if (__ballot(c == DE_V5_EMPTY) != 0ull) { raise(ERR_POOL_EMPTY); n_stash += 8u; return; }
        n_stash += 8u;
    }
    DE_DEV uint32_t alloc_chunk() {
        if (n_stash == 0u) { if (!aborted) raise(ERR_POOL_EMPTY); return 0u; }      // cannot happen (ensure_stash); chunk 0 is scribbled on and the frame reported as failed
        n_stash--;
        return (uint32_t)__builtin_amdgcn_readlane((int)stash, (int)n_stash);
    }
    DE_DEV void free_chunk(uint32_t chunk) {
        if ((uint32_t)lane_id() == n_stash) stash = chunk;
        n_stash++;
        if (n_stash >= 24u) {           // give eight back
            uint32_t i = 0;
            if (lane_id() == 0) i = add_agent(ctl(C_FREE_TAIL), 8u);
            i = uni(i);
            const uint32_t l = (uint32_t)lane_id();
            if (l >= n_stash - 8u && l < n_stash) st_agent(A.free_ring + ((i + (l - (n_stash - 8u))) & (A.pool_chunks - 1u)), stash);
            n_stash -= 8u;
        }
    }

    // ---- output side
    DE_DEV void publish(uint32_t q, uint32_t chunk, uint32_t count) {
        // the ticket first: its round trip overlaps the drain of the record stores (a consumer that claims the ticket early waits for
        // the cell, which is stored only after the drain)
        const uint32_t ring = shard * NQ + q;
        uint32_t i = 0;
        if (lane_id() == 0) i = add_agent(q_tail(ring), 1u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's record and cold stores have reached the coherence point
        if (lane_id() == 0) st_agent(ring_cells(ring) + (i & A.ring_mask), chunk * 128u + count);
    }
    template <int Q>
    DE_DEV void push(const Path& p, bool pred) {
        const unsigned long long mask = __ballot(pred);
        if (mask == 0ull) return;
        const uint32_t n = (uint32_t)__popcll(mask), rank = (uint32_t)__popcll(mask & ((1ull << lane_id()) - 1ull));
        if (oc[Q] == DE_V5_NONE) { oc[Q] = alloc_chunk(); of[Q] = 0u; oa[Q] = now(); }
        const uint32_t rem = DE_V5_CH - of[Q];
        if (pred && rank < rem) store_rec<LOCAL>(rs, oc[Q] * DE_V5_CH + of[Q] + rank, p);
        if (n >= rem) {
            publish(Q, oc[Q], DE_V5_CH);
            if (n > rem) {
                oc[Q] = alloc_chunk(); of[Q] = n - rem; oa[Q] = now();
                if (pred && rank >= rem) store_rec<LOCAL>(rs, oc[Q] * DE_V5_CH + (rank - rem), p);
            } else { oc[Q] = DE_V5_NONE; of[Q] = 0u; }
        } else of[Q] += n;
    }
    // publish partially filled chunks: all of them (an idle wave: whatever it holds may be the only work left), or those that have
    // waited for age_max scheduler rounds (a record must not sit in a rarely used target's chunk until the end of the launch)
    DE_DEV bool flush_outs(bool all) {
        bool any = false;
        const uint32_t t = now();
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (oc[q] != DE_V5_NONE && of[q] > 0u && (all || t - oa[q] >= A.age_max)) {
                publish((uint32_t)q, oc[q], of[q]);
                oc[q] = DE_V5_NONE; of[q] = 0u;
                stat(ST_PARTIAL, 1);
                any = true;
            }
        }
        return any;
    }
    DE_DEV static uint32_t now() { return (uint32_t)__builtin_amdgcn_s_memrealtime(); }

    // ---- input side
    DE_DEV bool next_input() {
        const unsigned long long t0 = STATS ? __builtin_amdgcn_s_memrealtime() : 0ull;
        const bool r = next_input_();
        if (STATS) stat(ST_T_INPUT, __builtin_amdgcn_s_memrealtime() - t0);
        return r;
    }
    DE_DEV bool next_input_() {
        if (in_chunk != DE_V5_NONE) { free_chunk(in_chunk); in_chunk = DE_V5_NONE; }
        in_lo = in_hi = 0u;
        if (in_n == 0u) {
            in_n = claim(q_head(in_ring), q_tail(in_ring), 1u, A.claim_k, &in_h);
            if (in_n == 0u) return false;
            fetch_cells();
        }
        uint32_t cell = (uint32_t)__builtin_amdgcn_readlane((int)in_cells, (int)in_k);
        if (cell == DE_V5_EMPTY) cell = take_cell(ring_cells(in_ring), A.ring_mask, in_h);      // its producer had not stored it yet: wait for it
        in_h++; in_n--; in_k++;
        if (cell == DE_V5_EMPTY) return false;
        in_chunk = cell >> 7; in_lo = 0u; in_hi = cell & 127u;
        return true;
    }
    // one round trip for all cells of a fresh claim (in_h .. in_h + in_n): lane j reads cell j and, if it is there, frees it
    DE_DEV void fetch_cells() {
        in_cells = DE_V5_EMPTY; in_k = 0u;
        if ((uint32_t)lane_id() < in_n) {
            uint32_t* cell = ring_cells(in_ring) + ((in_h + (uint32_t)lane_id()) & A.ring_mask);
            in_cells = ld_agent(cell);
            if (in_cells != DE_V5_EMPTY) st_agent(cell, DE_V5_EMPTY);
        }
    }
    DE_DEV void leave_input() {
        if (in_chunk != DE_V5_NONE && in_lo == in_hi) { free_chunk(in_chunk); in_chunk = DE_V5_NONE; }
    }

    // statistics: the time since the last mark goes to stage STAGE (-1: the scheduler).  A template parameter: a run-time index
    // would put the counters — and with them the whole wave state — into scratch memory
    template <int STAGE>
    DE_DEV void mark() {
        if (!STATS) return;
        const unsigned long long t = __builtin_amdgcn_s_memrealtime();
        stat(STAGE >= 0 ? ST_CYC + STAGE : ST_SCHED_CYC, t - t_mark);
        t_mark = t;
    }

    // ---- the scheduler: which stage this wave runs next.  Returns a queue id (its first chunks are claimed: in_ring / in_h / in_n),
    // Q_PRIMARY, or Q_EXIT.
    DE_DEV int schedule() {
        uint32_t idle = 0;
        for (;;) {
            if (abort_seen()) return Q_EXIT;
            unsigned long long tq = STATS ? __builtin_amdgcn_s_memrealtime() : 0ull;
            flush_outs(false);
            if (STATS) { const unsigned long long t = __builtin_amdgcn_s_memrealtime(); stat(ST_T_FLUSH, t - tq); tq = t; }
            ensure_stash();
            if (STATS) { const unsigned long long t = __builtin_amdgcn_s_memrealtime(); stat(ST_T_ENSURE, t - tq); tq = t; }
            // one look at this team's queues: lane q reads head and tail of ring (shard, q); lanes 62 / 63 the work counter and the
            // free-chunk count.  (A team = the waves of one shard: an XCD in LOCAL mode, else the blocks with index = shard mod n_shards.)
            int ready = 0;
            const uint32_t l = (uint32_t)lane_id();
            if (l < (uint32_t)NQ) { const uint32_t r = shard * NQ + l; const uint32_t h = ld_agent(q_head(r)), t = ld_agent(q_tail(r)); ready = (int)(t - h); }
            else if (l == 62u) ready = (int)ld_agent(ctl(C_WORK));
            else if (l == 63u) ready = (int)(ld_agent(ctl(C_FREE_TAIL)) - ld_agent(ctl(C_FREE_HEAD)));
            const uint32_t work_next = (uint32_t)__builtin_amdgcn_readlane(ready, 62);
            const int free_chunks = __builtin_amdgcn_readlane(ready, 63);
            if (STATS) { const unsigned long long t = __builtin_amdgcn_s_memrealtime(); stat(ST_T_SNAP, t - tq); tq = t; }
            if (work_next >= n_work && w_lo >= w_hi) primary_left = false;
            // the fullest queue, and how much is queued in all
            int best_q = -1, best = 0, backlog = 0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int r = __builtin_amdgcn_readlane(ready, q);
                if (r > 0) backlog += r;
                if (r > best) { best = r; best_q = q; }
            }
            const bool may_start = primary_left && free_chunks >= (int)A.reserve_chunks;
            // shallow queues: start new paths rather than fight over the few chunks there are
            if (may_start && backlog < (int)A.backlog_target) return Q_PRIMARY;
            uint32_t ring = shard * NQ + (uint32_t)(best_q < 0 ? 0 : best_q);
            if (best_q < 0 && !may_start && A.n_shards > (LOCAL ? 8u : 1u)) {
                // nothing here and nothing to start: look at one other team (LOCAL: one of the same XCD — a path never leaves its XCD)
                if (LOCAL) {
                    const uint32_t per_xcd = A.n_shards >> 3, base = shard - shard % per_xcd;
                    victim = victim + 1u >= base + per_xcd ? base : victim + 1u;
                    if (victim == shard) victim = victim + 1u >= base + per_xcd ? base : victim + 1u;
                } else {
                    victim = victim + 1u >= A.n_shards ? 0u : victim + 1u;
                    if (victim == shard) victim = victim + 1u >= A.n_shards ? 0u : victim + 1u;
                }
                int vr = 0;
                if (l < (uint32_t)NQ) { const uint32_t r = victim * NQ + l; const uint32_t h = ld_agent(q_head(r)), t = ld_agent(q_tail(r)); vr = (int)(t - h); }
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int r = __builtin_amdgcn_readlane(vr, q);
                    if (r > best) { best = r; best_q = q; }
                }
                if (best_q >= 0) { ring = victim * NQ + (uint32_t)best_q; stat(ST_STEAL, 1); }
            }
            if (best_q >= 0) {
                in_ring = ring;
                in_n = claim(q_head(ring), q_tail(ring), 1u, A.claim_k, &in_h);
                if (STATS) { const unsigned long long t = __builtin_amdgcn_s_memrealtime(); stat(ST_T_CLAIM, t - tq); tq = t; }
                if (in_n) { fetch_cells(); stat(ST_ENTRIES, 1); return best_q; }
                stat(ST_CLAIMFAIL, 1);
            }
            if (may_start) return Q_PRIMARY;
            // nothing to do: whatever this wave still holds may be what the others are waiting for
            if (flush_outs(true)) continue;
            if (done_local) { if (lane_id() == 0) add_agent(ctl(C_DONE), done_local); done_local = 0u; }
            if (uni(ld_agent(ctl(C_DONE))) >= n_work) return Q_EXIT;
            stat(ST_IDLE, 1);
            // nothing will be started any more and this wave is not one of the closers: make room for the next launch's waves
            if (!primary_left && blockIdx.x >= A.closers && idle >= 32u) return Q_EXIT;
            if (++idle > A.idle_limit) { raise(ERR_IDLE_WATCHDOG); return Q_EXIT; }
            // back off: 0.2 us at first, ~25 us after a few hundred rounds (thousands of waves polling the same lines starve the producers)
            { const uint32_t reps = idle < 8u ? 1u : (idle < 64u ? 4u : 32u);
#pragma nounroll
              for (uint32_t k = 0; k < reps; ++k) __builtin_amdgcn_s_sleep(32); }
        }
    }

    // ---- a loop stage: persistent lanes.  A lane takes a record, runs the stage's loop on it one step per trip of the wave loop,
    // keeps the finished record until the wave's next service; a service pushes the finished records and refills the idle lanes
    // from the input chunk (the next chunks of the queue as needed).  When the queue is dry the wave keeps stepping what it has,
    // looks again every few trips while it is still reasonably full, and returns to the scheduler when its last lane has finished.
    template <int Q, int SERVICE, typename Stage, int... TARGETS>
    DE_DEV void run_loop(Stage& st, Targets<TARGETS...>) {
        Path p;
        bool busy = false, fin = false;
        int target = -1;
        bool dry = false;
        uint32_t trips = 0, retry_at = 0, services = 0;
        for (;;) {
            const int nb = __popcll(__ballot(busy));
            const bool may_refill = !dry || (nb >= 16 && trips >= retry_at);
            if (nb == 0 || (64 - nb >= SERVICE && (may_refill || __ballot(fin) != 0ull))) {
                const unsigned long long ts0 = STATS ? __builtin_amdgcn_s_memrealtime() : 0ull;
                ensure_stash();
                (push<TARGETS>(p, fin && target == TARGETS), ...);
                fin = false;
                if (STATS) stat(ST_T_PUSH, __builtin_amdgcn_s_memrealtime() - ts0);
                if (may_refill || nb == 0) {
                    dry = false;
                    bool got = false;
                    uint32_t idx = 0;
                    for (;;) {
                        const unsigned long long m_need = __ballot(!busy && !got);
                        if (m_need == 0ull) break;
                        const uint32_t avail = in_hi - in_lo;
                        if (avail == 0u) {
                            if (!next_input()) { dry = true; retry_at = trips + 24u; stat(ST_DRY, 1); break; }
                            continue;
                        }
                        const uint32_t my = (uint32_t)__popcll(m_need & ((1ull << lane_id()) - 1ull));
                        if (!busy && !got && my < avail) { idx = in_chunk * DE_V5_CH + in_lo + my; got = true; }
                        const uint32_t take = min((uint32_t)__popcll(m_need), avail);
                        in_lo += take;
                        stat(ST_RECS + Q, take);
                    }
                    if (got) { load_rec(rs, idx, p); st.begin(p); busy = true; }
                }
                if (STATS) stat(ST_T_SERVICE, __builtin_amdgcn_s_memrealtime() - ts0);
                if (__ballot(busy) == 0ull) { if (dry || aborted) break; else continue; }
                if ((++services & 31u) == 0u) flush_outs(false);
            }
            if (STATS) { stat(ST_TRIPS + Q, 1); stat(ST_LANETRIPS + Q, (unsigned long long)__popcll(__ballot(busy))); }
            if (busy) {
                if (st.step(p)) { target = st.finish(p); busy = false; fin = true; }
            }
            trips++;
        }
        leave_input();
    }

#if DE_V5_DUAL
    // ---- (experiment, off: profiles/experiments/README.md) a loop stage with TWO records per lane.  A single-class wave has one dependent load per trip (the map tap) and nothing else to
    // issue while it is in flight; four such waves per SIMD do not cover it (58 % of the wave-cycles were waits).  Here a lane holds two
    // records, A and B; a trip issues A's load, then B's, and only then consumes either: two loads in flight per lane, the address
    // arithmetic of B under A's latency.  Each slot is serviced exactly like run_loop's one (own thresholds, shared input cursor).
    template <int Q, int SERVICE, typename Stage, int... TARGETS>
    DE_DEV void run_loop2(Stage& sa, Stage& sb, Targets<TARGETS...>) {
        Path pa, pb;
        bool busy_a = false, busy_b = false, fin_a = false, fin_b = false;
        int tgt_a = -1, tgt_b = -1;
        bool dry = false;
        uint32_t trips = 0, retry_at = 0, services = 0;
        auto service = [&](Path& p, Stage& st, bool& busy, bool& fin, int target, bool refill) {
            const unsigned long long ts0 = STATS ? __builtin_amdgcn_s_memrealtime() : 0ull;
            ensure_stash();
            (push<TARGETS>(p, fin && target == TARGETS), ...);
            fin = false;
            if (STATS) stat(ST_T_PUSH, __builtin_amdgcn_s_memrealtime() - ts0);
            if (refill) {
                dry = false;
                bool got = false;
                uint32_t idx = 0;
                for (;;) {
                    const unsigned long long m_need = __ballot(!busy && !got);
                    if (m_need == 0ull) break;
                    const uint32_t avail = in_hi - in_lo;
                    if (avail == 0u) {
                        if (!next_input()) { dry = true; retry_at = trips + 24u; stat(ST_DRY, 1); break; }
                        continue;
                    }
                    const uint32_t my = (uint32_t)__popcll(m_need & ((1ull << lane_id()) - 1ull));
                    if (!busy && !got && my < avail) { idx = in_chunk * DE_V5_CH + in_lo + my; got = true; }
                    const uint32_t take = min((uint32_t)__popcll(m_need), avail);
                    in_lo += take;
                    stat(ST_RECS + Q, take);
                }
                if (got) { load_rec(rs, idx, p); st.begin(p); busy = true; }
            }
            if (STATS) stat(ST_T_SERVICE, __builtin_amdgcn_s_memrealtime() - ts0);
            if ((++services & 31u) == 0u) flush_outs(false);
        };
        for (;;) {
            const int na = __popcll(__ballot(busy_a)), nb = __popcll(__ballot(busy_b));
            const bool none = na + nb == 0;
            const bool may_refill = !dry || none || (na + nb >= 32 && trips >= retry_at);
            if (none || (64 - na >= SERVICE && (may_refill || __ballot(fin_a) != 0ull))) service(pa, sa, busy_a, fin_a, tgt_a, may_refill);
            if (none || (64 - nb >= SERVICE && ((may_refill && !dry) || __ballot(fin_b) != 0ull))) service(pb, sb, busy_b, fin_b, tgt_b, may_refill && !dry);
            if (__ballot(busy_a || busy_b) == 0ull) { if (dry || aborted) break; else continue; }
            if (STATS) { stat(ST_TRIPS + Q, 1); stat(ST_LANETRIPS + Q, (unsigned long long)(__popcll(__ballot(busy_a)) + __popcll(__ballot(busy_b)))); }
            typename Stage::Tok ka, kb;
            bool da = false, db = false;
            if (busy_a) da = sa.issue(pa, ka);
            if (busy_b) db = sb.issue(pb, kb);
            if (busy_a && !da) da = sa.resolve(pa, ka);
            if (busy_b && !db) db = sb.resolve(pb, kb);
            if (da) { tgt_a = sa.finish(pa); busy_a = false; fin_a = true; }
            if (db) { tgt_b = sb.finish(pb); busy_b = false; fin_b = true; }
            trips++;
        }
        leave_input();
    }
#endif

    // ---- a straight-line stage: one record per lane, chunk after chunk
    template <int Q, typename Body, int... TARGETS>
    DE_DEV void run_flat(Body body, Targets<TARGETS...>) {
        for (;;) {
            if (in_lo == in_hi && !next_input()) break;
            ensure_stash();
            const uint32_t n = in_hi - in_lo;
            const bool active = (uint32_t)lane_id() < n;
            Path p;
            int target = -2;
            const unsigned long long tb0 = STATS ? __builtin_amdgcn_s_memrealtime() : 0ull;
            if (active) { load_rec(rs, in_chunk * DE_V5_CH + in_lo + (uint32_t)lane_id(), p); target = body(p); }
            if (STATS) stat(ST_T_BODY, __builtin_amdgcn_s_memrealtime() - tb0);
            in_lo += n;
            stat(ST_RECS + Q, n);
            done_local += (uint32_t)__popcll(__ballot(target == -1));
            const unsigned long long tp0 = STATS ? __builtin_amdgcn_s_memrealtime() : 0ull;
            (push<TARGETS>(p, target == TARGETS), ...);
            if (STATS) stat(ST_T_PUSH, __builtin_amdgcn_s_memrealtime() - tp0);
        }
        leave_input();
    }

    // ---- new paths: 64 work items = the 64 pixels of one screen tile for one sample; the primary sphere trace runs here (coherent rays)
    DE_DEV void run_primary(const RenderArgs& a, const FrameConsts& fc) {
        if (w_lo >= w_hi) {          // 256 work items per atomic: four passes
            uint32_t w = 0;
            if (lane_id() == 0) w = add_agent(ctl(C_WORK), 256u);
            w = uni(w);
            if (w >= n_work) { primary_left = false; return; }
            w_lo = w; w_hi = min(w + 256u, n_work);
        }
        const uint32_t w0 = w_lo;
        w_lo += 64u;
        ensure_stash();
        const uint32_t w = w0 + (uint32_t)lane_id();
        const bool active = w < w_hi;
        if (A.resume_rec) {
            // finisher: a queued path ray of the pipeline goes into this launch's sphere-trace queue, its throughput and radiance into
            // this launch's cold record (everything else of a cold record is rewritten before it is read: the ray starts a vertex)
            Path p;
            bool live = false;
            if (active) {
                const uint4* r = A.resume_rec + (size_t)w * 4u;
                const uint4 a0 = r[0], b0 = r[1], c0 = r[2], d0 = r[3];
                p.pid = a0.x; p.meta = a0.y; p.rng.s0 = a0.z; p.rng.s1 = a0.w; p.rng.draws = 0;
                p.P = v3(wf::asf(b0.x), wf::asf(b0.y), wf::asf(b0.z)); p.W = v3(wf::asf(b0.w), wf::asf(c0.x), wf::asf(c0.y));
                p.x[0] = wf::asf(c0.z); p.x[1] = wf::asf(c0.w); p.x[2] = wf::asf(d0.x); p.x[3] = wf::asf(d0.y); p.x[4] = wf::asf(d0.z); p.x[5] = wf::asf(d0.w);
                live = p.pid != DE_V3_HOLE;
                if (live) {
                    const float* cr = A.resume_cold + (size_t)p.pid * 16u;
                    const ColdWT<LOCAL> cold{A.cold};
                    cold.set_thr_ls(p.pid, cr[0], cr[1]);
                }
            }
            done_local += (uint32_t)__popcll(__ballot(active && !live));      // holes count as finished work items
            stat(ST_RECS + NQ, min(64u, w_hi - w0));
            push<Q_ST>(p, live);
            return;
        }
        const float MAX_RAY_DIST = (float)(6371e3 * 10.0);
        Path p;
        int target = -2;
        if (active) {
            wf::new_path(a, fc, s_node_val, w, p);
            const float t = wf::sphere_trace<CLAMP>(a, p, fc.land_height_scale, DE_PLANET_R + de_max(fc.land_height_scale, 0.0f) + 8100.0f, MAX_RAY_DIST);
            target = wf::after_sphere_trace(p, t, MAX_RAY_DIST);
        }
        stat(ST_RECS + NQ, min(64u, w_hi - w0));
        push<Q_GAS>(p, target == Q_GAS);
        push<Q_CLOUD>(p, target == Q_CLOUD);
        push<Q_EVENT>(p, target == Q_EVENT);
    }
};

}  // namespace ps

#ifndef DE_V5_MIN_WAVES
#define DE_V5_MIN_WAVES 4          // waves per SIMD the register allocator must leave room for (128 VGPRs)
#endif
#ifndef DE_V5_SVC_ST
#define DE_V5_SVC_ST 32
#endif
#ifndef DE_V5_SVC_GAS
#define DE_V5_SVC_GAS 32
#endif
#ifndef DE_V5_SVC_CLOUD
#define DE_V5_SVC_CLOUD 12
#endif

// zero the control words, hand every chunk to the free ring (the queue rings are all-EMPTY: set by the host before the launch), and
// put the launch's RenderArgs where the persistent kernel reads them
__global__ void v5_init_kernel(uint32_t* ctl, uint32_t* free_ring, uint32_t pool_chunks, RenderArgs a, RenderArgs* dst) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (uint32_t)ps::C_WORDS) { ctl[i * DE_V5_CTL_STRIDE] = (i == (uint32_t)ps::C_FREE_TAIL) ? pool_chunks : 0u; ctl[i * DE_V5_CTL_STRIDE + 1u] = 0u; }
    if (i < pool_chunks) free_ring[i] = i;
    if (i == 0u) *dst = a;
}

template <typename T> DE_DEV const T* v5_opaque(const T* p) { return wf::opaque_ptr(p); }

template <bool CLAMP, bool STATS, bool LOCAL>
__global__ void __launch_bounds__(64, DE_V5_MIN_WAVES) v5_kernel(ps::Args A, const FrameConsts* fcp) {
    using namespace ps;
    __shared__ float s_node_val[64];
    s_node_val[threadIdx.x & 63u] = A.ra->node_val[threadIdx.x & 63u];
    __syncthreads();
    Wave<CLAMP, STATS, LOCAL> W{A, s_node_val};
    W.rs = __builtin_amdgcn_make_buffer_rsrc(A.pool, 0, (int)(A.pool_chunks * DE_V5_CH * 64u), 0x00020000);
    {
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        // a team = the waves of one shard.  LOCAL: n_shards / 8 teams per XCD (block b runs on XCD b mod 8: the blocks of an XCD are dealt
        // round robin to its teams; correctness only needs that a team's waves share the XCD, which the register says)
        const uint32_t per_xcd = A.n_shards >> 3;
        W.shard = LOCAL ? ((xcc & 7u) * per_xcd + (blockIdx.x >> 3) % per_xcd) : (blockIdx.x % A.n_shards);
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) { W.oc[q] = DE_V5_NONE; W.of[q] = 0u; W.oa[q] = 0u; }
    W.stash = 0u; W.n_stash = 0u;
    W.in_ring = 0u; W.in_chunk = DE_V5_NONE; W.in_lo = W.in_hi = W.in_h = W.in_n = 0u; W.in_cells = DE_V5_EMPTY; W.in_k = 0u;
    W.done_local = 0u; W.primary_left = true; W.aborted = false;
    W.w_lo = W.w_hi = 0u; W.victim = W.shard;
    W.n_work = A.resume_rec ? min(*A.resume_count, A.resume_capacity) : A.n_items;
    __shared__ unsigned long long s_stats[STATS ? 64 : 1];
    if (STATS) s_stats[threadIdx.x & 63u] = 0ull;
    __syncthreads();
    W.s_st = s_stats;
    W.t_mark = STATS ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const ColdWT<LOCAL> cold{A.cold};
    const float MAX_RAY_DIST = (float)(6371e3 * 10.0);

    for (;;) {
        const int s = W.schedule();
        W.template mark<-1>();
        if (s == Q_EXIT) break;
        const RenderArgs& a = *v5_opaque(A.ra);
        const FrameConsts& fc = *v5_opaque(fcp);
        switch (s) {
        case Q_ST: {
            StStage<CLAMP> st{a.topo, fc.land_height_scale, DE_PLANET_R + de_max(fc.land_height_scale, 0.0f) + 8100.0f, MAX_RAY_DIST, 0.0f, 0};
#if DE_V5_DUAL & 1
            StStage<CLAMP> st2 = st;
            W.template run_loop2<Q_ST, DE_V5_SVC_ST>(st, st2, Targets<Q_GAS, Q_CLOUD, Q_EVENT, Q_SURFFIN>{});
#else
            W.template run_loop<Q_ST, DE_V5_SVC_ST>(st, Targets<Q_GAS, Q_CLOUD, Q_EVENT, Q_SURFFIN>{});
#endif
            W.template mark<Q_ST>();
        } break;
        case Q_GAS: {
            GasStage st{a.nodes, a.dens_table};
            W.template run_loop<Q_GAS, DE_V5_SVC_GAS>(st, Targets<Q_CLOUD, Q_EVENT, Q_SCFIN, Q_SURFFIN>{});
            W.template mark<Q_GAS>();
        } break;
        case Q_CLOUD: {
            CloudStage<CLAMP> st{a.clouds};
#if DE_V5_DUAL & 2
            CloudStage<CLAMP> st2{a.clouds};
            W.template run_loop2<Q_CLOUD, DE_V5_SVC_CLOUD>(st, st2, Targets<Q_EVENT, Q_SCFIN, Q_SURFFIN>{});
#else
            W.template run_loop<Q_CLOUD, DE_V5_SVC_CLOUD>(st, Targets<Q_EVENT, Q_SCFIN, Q_SURFFIN>{});
#endif
            W.template mark<Q_CLOUD>();
        } break;
        case Q_EVENT:
            W.template run_flat<Q_EVENT>([&](Path& p) { return wf::event_body<CLAMP>(a, fc, cold, p); }, Targets<Q_GAS, Q_CLOUD, Q_SCFIN, Q_SURF>{});
            W.template mark<Q_EVENT>();
            break;
        case Q_SURF:
            W.template run_flat<Q_SURF>([&](Path& p) { return wf::surf_body<CLAMP>(a, fc, cold, p); }, Targets<Q_ST>{});
            W.template mark<Q_SURF>();
            break;
        case Q_SCFIN:
            W.template run_flat<Q_SCFIN>([&](Path& p) { return wf::scfin_body(a, fc, cold, p); }, Targets<Q_ST>{});
            W.template mark<Q_SCFIN>();
            break;
        case Q_SURFFIN:
            W.template run_flat<Q_SURFFIN>([&](Path& p) { return wf::surffin_body(a, cold, p); }, Targets<Q_ST>{});
            W.template mark<Q_SURFFIN>();
            break;
        default:
            W.run_primary(a, fc);
            W.template mark<Q_PRIMARY>();
            break;
        }
    }
    if (STATS) {
        __syncthreads();
        const int k = lane_id();
        if (k < ST_N && s_stats[k]) add_stat(A.ctl + (C_STAT0 + k) * DE_V5_CTL_STRIDE, s_stats[k]);
    }
}
