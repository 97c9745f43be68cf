// legacy/de_launch_legacy.h — host side of the kernel families the product no longer runs (library built with -DDE_LEGACY_VARIANTS only:
// libdigitalearth_hip_legacy.so, loaded by the cross-check tests through DE_LIB_PATH): render_kernel (per-lane loops, variant 1), the wavefront
// pipeline through HBM queues with its launch policy of rounds 3-4 (variants 3 and, with DE_AUTO_V6=0, 4) and the HBM-queue stage scheduler
// (variant 5).  They are three further statements of path_tracer that must give the product's bits; nothing here is product surface.  This
// library reads its experiment knobs from the environment as rounds 1-4 did (the product library reads none: de_set_tuning).
#pragma once

namespace {

// render_kernel_v3: the level-synchronous launch sequence of one batch of paths — ten launches per path vertex, each draining
// one stage queue (render_kernel_v3.hip).  Every launch is sized to the machine, not to the queue: a block that finds its queue
// empty leaves after one atomic.
// Returns DE_OK, an error, or V3_NO_MEMORY: the device does not have room for queues of this size (the caller shrinks the batch).
static const int V3_NO_MEMORY = 1000;
int v3_reserve(de_ctx* c, int pipe, size_t n_items) {
    de_ctx::V3Pipe& P = c->v3p[pipe];
    if (n_items <= P.capacity) return DE_OK;
    // room for every path plus the holes the waves' last bulk reservations leave (render_kernel_v3.hip: push / close_out)
    const size_t slack = 4u * (size_t)c->n_cus * (size_t)c->v3_slack_wpc * DE_V3_BULK;   // up to 3 producing launches fill a queue before it is drained
    // 512 B per path: ask first — other contexts, the host framework and the next pipe want memory too (8 GB stay free) — and BEFORE
    // giving up what the pipe holds: a request that cannot be met must leave the pipe as it is (the caller then asks for less, which the
    // pipe may already hold; freeing first made every such call re-allocate its queues: 1 s per frame under a binding budget)
    size_t free_b = 0, total_b = 0;
    const size_t need_b = (size_t)wf::NQ * (n_items + slack) * sizeof(wf::Rec) + n_items * sizeof(wf::Cold);
    const size_t held_b = P.capacity ? (size_t)wf::NQ * (P.capacity + slack) * sizeof(wf::Rec) + P.capacity * sizeof(wf::Cold) : 0;
    if (c->mem_budget) {
        // the call's pipes run side by side and are sized alike: each gets its share of the budget
        const size_t share = c->mem_budget / (size_t)(c->launch_pipes > 0 ? c->launch_pipes : 1);
        if (need_b > share) return V3_NO_MEMORY;
    }
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b + held_b < need_b + ((size_t)8 << 30)) return V3_NO_MEMORY;
    if (c->v3_test_budget && n_items > c->v3_test_budget) return V3_NO_MEMORY;      // tests: pretend larger queues do not fit
    { int rc = sync_all(c); if (rc) return rc; }
    for (int i = 0; i < wf::NQ; ++i) { if (P.rec[i]) hipFree(P.rec[i]); P.rec[i] = nullptr; }
    if (P.cold) hipFree(P.cold);
    P.cold = nullptr; P.capacity = 0;
    bool ok = true;
    for (int i = 0; i < wf::NQ && ok; ++i) ok = hipMalloc(&P.rec[i], (n_items + slack) * sizeof(wf::Rec)) == hipSuccess;
    if (ok) ok = hipMalloc(&P.cold, n_items * sizeof(wf::Cold)) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        for (int i = 0; i < wf::NQ; ++i) { if (P.rec[i]) hipFree(P.rec[i]); P.rec[i] = nullptr; }
        if (P.cold) hipFree(P.cold);
        P.cold = nullptr;
        return V3_NO_MEMORY;
    }
    if (!P.ctl) {
        HIP_TRY(hipMalloc(&P.ctl, wf::NQ * 4 * sizeof(uint32_t)));
        // hipMemset runs on the null stream and may return before it has executed; the launch slots are non-blocking streams
        // that do not wait for the null stream: make sure the control words are zero before any kernel can read them
        HIP_TRY(hipMemset(P.ctl, 0, wf::NQ * 4 * sizeof(uint32_t)));
        HIP_TRY(hipDeviceSynchronize());
        for (int i = 0; i < wf::NQ; ++i) P.epoch[i] = 0;
        if (hipHostMalloc(reinterpret_cast<void**>(&P.h_round2), sizeof(uint32_t), hipHostMallocDefault) == hipSuccess) {
            *P.h_round2 = 0;
            bool ok = hipEventCreateWithFlags(&P.stat_ev, hipEventDisableTiming) == hipSuccess;
            for (int i = 0; i < 3 && ok; ++i) ok = hipEventCreate(&P.ev_t[i]) == hipSuccess;
            if (!ok) { hipHostFree(P.h_round2); P.h_round2 = nullptr; }
        } else P.h_round2 = nullptr;
        (void)hipGetLastError();
    }
    P.capacity = n_items;
    return DE_OK;
}
int v5_reserve(de_ctx* c, int idx, hipStream_t stream, size_t n_items);
hipError_t launch_v3(de_ctx* c, const RenderArgs& a, hipStream_t stream, int pipe) {
    de_ctx::V3Pipe& P = c->v3p[pipe];
    const bool clamp = (c->p.flags & DE_FLAG_CLAMP_SAMPLER) != 0;
    wf::Args A;
    A.a = a;
    A.cold = P.cold;
    A.n_items = (uint32_t)((size_t)a.n_tiles * 64u * (size_t)a.spp_count);
    A.capacity = (uint32_t)(P.capacity + 4u * (size_t)c->n_cus * (size_t)c->v3_slack_wpc * DE_V3_BULK);
    const unsigned by_work = (unsigned)((A.n_items + DE_V3_BLOCK - 1) / DE_V3_BLOCK);
    dim3 block(DE_V3_BLOCK), grid(1);
    auto size_grid = [&](int cls) {
        unsigned wpc = (unsigned)(c->v3_wpc_class[cls] > 0 ? c->v3_wpc_class[cls] : c->launch_wpc);
        // the cloud stage's larger budget is for two pipes side by side; with four (a rank's share of a tile partition) 4 x 16 waves per CU
        // only queue behind each other: 30.3 -> 28.8 ms per 1/8 frame with the call's own budget
        if (cls == 2 && c->launch_pipes > 2 && c->kernel_variant == 4 && wpc > (unsigned)c->launch_wpc) wpc = (unsigned)c->launch_wpc;
        if (wpc > (unsigned)c->v3_slack_wpc) wpc = (unsigned)c->v3_slack_wpc;      // the queues' room for holes is sized for that many waves per CU
        const unsigned machine = (unsigned)c->n_cus * wpc;
        grid = dim3(by_work < machine ? by_work : machine);
    };
    // the control words each launch sees: per queue, the set being filled (pushes, and the drain of that queue) and the other one
    auto bind = [&]() {
        for (int i = 0; i < wf::NQ; ++i) {
            A.q[i].rec = P.rec[i];
            A.q[i].ctl = P.ctl + 4 * i + 2 * (P.epoch[i] & 1);
            A.q[i].ctl_next = P.ctl + 4 * i + 2 * ((P.epoch[i] + 1) & 1);
        }
    };
#define V3_DRAIN_T(qid, k, ...) { size_grid(qid == wf::Q_ST ? 0 : (qid == wf::Q_CLOUD ? 2 : 3)); bind(); if (clamp) hipLaunchKernelGGL((k<true>), grid, block, 0, stream, __VA_ARGS__); else hipLaunchKernelGGL((k<false>), grid, block, 0, stream, __VA_ARGS__); P.epoch[qid]++; }
#define V3_DRAIN(qid, k, ...) { size_grid(qid == wf::Q_GAS ? 1 : 3); bind(); hipLaunchKernelGGL(k, grid, block, 0, stream, __VA_ARGS__); P.epoch[qid]++; }
    // Vertex rounds the pipeline runs itself; the path rays it has queued for round n_depths are finished by render_kernel_v2
    // (resume mode): the wavefront stages are at their best while the rays are many and coherent, the megakernel when they
    // are few, deep and divergent — every pipeline launch ends with its own tail, the megakernel has one.
    const int n_depths = c->launch_depths;
    // (launches of a call issued as ONE batch overlap other calls' launches: their duration says nothing about the view — the statistic
    // behind the deep setting is taken from calls with batches side by side only)
    const bool timed = P.h_round2 && !P.stat_pending && n_depths < 25 && !c->launch_one_batch;
    if (timed) hipEventRecord(P.ev_t[0], stream);
    auto note_round2 = [&]() {      // the ST queue's current set holds the count of round 2's path rays until that queue has been drained twice more
        if (!P.h_round2 || P.stat_pending) return;
        bind();
        if (hipMemcpyAsync(P.h_round2, A.q[wf::Q_ST].ctl, sizeof(uint32_t), hipMemcpyDeviceToHost, stream) != hipSuccess) return;
        if (hipEventRecord(P.stat_ev, stream) != hipSuccess) return;
        P.stat_items = A.n_items; P.stat_pending = true; P.stat_timed = false;
    };
    for (int depth = 0; depth < n_depths; ++depth) {
        if (depth == 2) note_round2();
        if (depth == 0) { size_grid(0); bind(); if (clamp) hipLaunchKernelGGL((v3_primary_kernel<true>), grid, block, 0, stream, A, a.fc); else hipLaunchKernelGGL((v3_primary_kernel<false>), grid, block, 0, stream, A, a.fc); }
        else V3_DRAIN_T(wf::Q_ST, v3_st_kernel, A, a.fc)               // path rays of this vertex
        if (c->v3_fuse_track) { V3_DRAIN_T(wf::Q_CLOUD, v3_track_kernel, A) P.epoch[wf::Q_GAS]++; }   // delta tracking: gases, then the cloud shell, in one lane
        else {
        V3_DRAIN(wf::Q_GAS, v3_gas_kernel, A)                          // delta tracking, gases
        V3_DRAIN_T(wf::Q_CLOUD, v3_cloud_kernel, A)                    // delta tracking, cloud shell
        }
        V3_DRAIN_T(wf::Q_EVENT, v3_event_kernel, A, a.fc)              // event, light sample, branch
        V3_DRAIN_T(wf::Q_SURF, v3_surf_kernel, A, a.fc)                // surface hits: normal, material, shadow ray
        V3_DRAIN_T(wf::Q_ST, v3_st_kernel, A, a.fc)                    // shadow rays
        if (c->v3_fuse_track) { V3_DRAIN_T(wf::Q_CLOUD, v3_track_kernel, A) P.epoch[wf::Q_GAS]++; }   // ratio tracking towards the sun
        else {
        V3_DRAIN(wf::Q_GAS, v3_gas_kernel, A)                          // ratio tracking towards the sun, gases
        V3_DRAIN_T(wf::Q_CLOUD, v3_cloud_kernel, A)                    // ... cloud shell
        }
        V3_DRAIN(wf::Q_SCFIN, v3_scfin_kernel, A, a.fc)                // volume vertices: shade, next direction
        V3_DRAIN(wf::Q_SURFFIN, v3_surffin_kernel, A, a.fc)            // surface vertices
    }
#undef V3_DRAIN_T
#undef V3_DRAIN
    if (n_depths == 2) note_round2();
    if (n_depths < 25) {
        bind();
        RenderArgs b = a;
        b.resume_rec = reinterpret_cast<const uint4*>(P.rec[wf::Q_ST]);
        b.resume_count = A.q[wf::Q_ST].ctl;
        b.resume_cold = reinterpret_cast<const float*>(P.cold);
        b.resume_capacity = A.capacity;
        hipError_t me = hipMemsetAsync(A.q[wf::Q_ST].ctl_next, 0, 2 * sizeof(uint32_t), stream);      // what begin_drain does in the pipeline's kernels
        if (me == hipSuccess) me = hipMemsetAsync(b.work_counter, 0, sizeof(uint32_t), stream);
        if (me != hipSuccess) return me;
        dim3 pgrid((unsigned)c->n_cus * (unsigned)c->tune_wpc), pblock(DE_V2_BLOCK);
        if (timed && P.stat_pending) hipEventRecord(P.ev_t[1], stream);
        if (c->v3_finisher == 5) {
            // the persistent stage scheduler finishes the queued path rays (round 4): the same records, resumed into its own queues
            me = launch_v5(c, a, stream, DE_MAX_SLOTS + pipe, b.resume_rec, b.resume_count, b.resume_cold, b.resume_capacity, c->v5_fin_wpc);
            if (me != hipSuccess) return me;
        } else
        if (clamp) hipLaunchKernelGGL((render_kernel_v2<true, 0>), pgrid, pblock, 0, stream, b, b.fc);
        else hipLaunchKernelGGL((render_kernel_v2<false, 0>), pgrid, pblock, 0, stream, b, b.fc);
        if (timed && P.stat_pending) { hipEventRecord(P.ev_t[2], stream); hipEventRecord(P.stat_ev, stream); P.stat_timed = true; P.stat_pipes = c->launch_pipes; }
        P.epoch[wf::Q_ST]++;
    }
    return hipGetLastError();
}

// render_kernel_v5: buffers of one launch slot.  Pool, rings and control words are sized once; the cold records grow with the
// largest launch the slot has seen.
int v5_reserve(de_ctx* c, int idx, hipStream_t stream, size_t n_items) {
    de_ctx::V5State& S = c->v5s[idx];
    const uint32_t pool_chunks = 1u << c->v5_pool_log2;
    if (!S.pool || !S.ctl || !S.rings || !S.free_ring || !S.d_args || !S.h_status || S.pool_chunks != pool_chunks || S.n_shards != (uint32_t)c->v5_shards) {      // all or none (a reservation that failed half way is redone)
        HIP_TRY(hipStreamSynchronize(stream));
        if (S.pool) hipFree(S.pool); if (S.ctl) hipFree(S.ctl); if (S.rings) hipFree(S.rings); if (S.free_ring) hipFree(S.free_ring);
        S.pool = nullptr; S.ctl = nullptr; S.rings = nullptr; S.free_ring = nullptr;
        S.pool_chunks = pool_chunks; S.n_shards = (uint32_t)c->v5_shards;
        S.ring_cells = pool_chunks;
        HIP_TRY(hipMalloc(&S.pool, (size_t)pool_chunks * DE_V5_CH * 64u));
        HIP_TRY(hipMalloc(&S.ctl, (size_t)ps::C_WORDS * DE_V5_CTL_STRIDE * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&S.rings, (size_t)S.n_shards * wf::NQ * S.ring_cells * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&S.free_ring, (size_t)pool_chunks * sizeof(uint32_t)));
        if (!S.d_args) HIP_TRY(hipMalloc(&S.d_args, sizeof(RenderArgs)));
        if (!S.h_status) {
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&S.h_status), 64, hipHostMallocMapped));
            S.h_status[0] = 0u;
        }
    }
    if (n_items > S.cold_items) {
        HIP_TRY(hipStreamSynchronize(stream));
        if (S.cold) hipFree(S.cold);
        S.cold = nullptr; S.cold_items = 0;
        if (hipMalloc(&S.cold, n_items * 64u) != hipSuccess) { (void)hipGetLastError(); return fail(DE_ERR_NOMEM, "no room for the cold records of a render_kernel_v5 launch (64 B per path): lower DE_V5_MAX_ITEMS"); }
        S.cold_items = n_items;
    }
    return DE_OK;
}
hipError_t launch_v5(de_ctx* c, const RenderArgs& a, hipStream_t stream, int idx, const uint4* resume_rec, const uint32_t* resume_count, const float* resume_cold,
                     uint32_t resume_capacity, int wpc) {
    de_ctx::V5State& S = c->v5s[idx];
    const bool clamp = (c->p.flags & DE_FLAG_CLAMP_SAMPLER) != 0;
    ps::Args A;
    A.ra = S.d_args;
    A.pool = S.pool; A.ctl = S.ctl; A.rings = S.rings; A.free_ring = S.free_ring; A.cold = S.cold;
    uint32_t* d_status = nullptr;
    hipError_t e = hipHostGetDevicePointer(reinterpret_cast<void**>(&d_status), S.h_status, 0);
    if (e != hipSuccess) return e;
    A.status = d_status;
    A.n_items = (uint32_t)((size_t)a.n_tiles * 64u * (size_t)a.spp_count);
    A.resume_rec = resume_rec; A.resume_count = resume_count; A.resume_cold = resume_cold; A.resume_capacity = resume_capacity;
    if (wpc <= 0) wpc = c->v5_wpc;
    A.pool_chunks = S.pool_chunks; A.ring_mask = S.ring_cells - 1u; A.n_shards = S.n_shards;
    A.reserve_chunks = (uint32_t)((unsigned long long)S.pool_chunks * (unsigned)c->v5_reserve_pct / 100ull);
    A.backlog_target = (uint32_t)((unsigned long long)c->n_cus * (unsigned)wpc * (unsigned)c->v5_backlog / S.n_shards);
    A.claim_k = (uint32_t)c->v5_claim_k; A.age_max = (uint32_t)c->v5_age;
    A.closers = (uint32_t)(c->n_cus * c->v5_closers_per_cu);
    A.idle_limit = 1u << 20;
    // every launch starts from all-EMPTY rings, zeroed control words and a full free ring (a launch that ended on its abort word
    // leaves them in any state)
    e = hipMemsetAsync(S.rings, 0xff, (size_t)S.n_shards * wf::NQ * S.ring_cells * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(v5_init_kernel, dim3((S.pool_chunks + 255u) / 256u), dim3(256), 0, stream, S.ctl, S.free_ring, S.pool_chunks, a, S.d_args);
    const unsigned by_work = (unsigned)((A.n_items + 63u) / 64u);
    const unsigned machine = (unsigned)c->n_cus * (unsigned)wpc;
    dim3 grid(by_work < machine ? (by_work ? by_work : 1u) : machine), block(64);
    // LOCAL: one queue set per XCD, a path lives and dies on the XCD that started it — records and cold records are handed over through
    // that XCD's L2 with write-back stores.  Fewer shards (tests, experiments): any wave may take any chunk, everything is written through.
    const bool local = c->v5_local && (S.n_shards % 8u) == 0u;
    if (c->v5_stats && !clamp) {
        if (local) hipLaunchKernelGGL((v5_kernel<false, true, true>), grid, block, 0, stream, A, a.fc); else hipLaunchKernelGGL((v5_kernel<false, true, false>), grid, block, 0, stream, A, a.fc);
    } else if (clamp) {
        if (local) hipLaunchKernelGGL((v5_kernel<true, false, true>), grid, block, 0, stream, A, a.fc); else hipLaunchKernelGGL((v5_kernel<true, false, false>), grid, block, 0, stream, A, a.fc);
    } else {
        if (local) hipLaunchKernelGGL((v5_kernel<false, false, true>), grid, block, 0, stream, A, a.fc); else hipLaunchKernelGGL((v5_kernel<false, false, false>), grid, block, 0, stream, A, a.fc);
    }
    return hipGetLastError();
}

void legacy_note_abort(de_ctx* c) {
    for (int i = 0; i < DE_MAX_SLOTS + 4; ++i) {
        de_ctx::V5State& S = c->v5s[i];
        if (S.h_status && S.h_status[0] != 0u) {
            const uint32_t code = S.h_status[0];
            S.h_status[0] = 0u;
            c->frame_invalid = true;
            c->invalid_msg = std::string("render_kernel_v5 left on its abort word (") + (code == ps::ERR_CELL_TIMEOUT ? "a ring cell was never stored" : code == ps::ERR_POOL_EMPTY ? "the chunk pool ran empty: raise DE_V5_POOL_LOG2" : code == ps::ERR_RING_FULL ? "a queue ring wrapped: fewer shards or a lower DE_V5_BACKLOG" : "idle watchdog: work was lost") + "): the frame is invalid until de_reset";
        }
    }
}

// Every environment override of the library, read ONCE per context here (experiments, tests, A/B sweeps; none is needed in production
// and none is read again later):
//   DE_KERNEL             kernel variant 1..5 (de_set_kernel_variant)             DE_AUTO_TRACE   print what the launch policy chose
//   DE_V2_THR / _A / _B / _G / _CHUNK / _WPC / _MAX_SPP     render_kernel_v2's scheduler thresholds, work chunk, waves per CU, test split
//   DE_V3_MAX_ITEMS / _WPC / _WPC_{ST,GAS,CLOUD,FLAT} / _PIPES / _BATCHES / _DEPTHS / _TEST_BUDGET      the pipeline alone (variant 3)
//   DE_AUTO_DEPTHS / _WPC / _PIPES / _MIN_ITEMS / _TWO_PIPE_ITEMS / _MID / _DEEP_NS / _DEEP_DEPTHS / _DEEP_PIPES     the default's policy
//   DE_V5_WPC / _SHARDS / _LOCAL / _POOL_LOG2 / _CLAIM_K / _AGE / _CLOSERS / _STATS / _BACKLOG / _RESERVE_PCT / _BATCHES / _MAX_ITEMS   variant 5
//   DE_SLOTS / DE_BIG_SLOTS   launch slots (read next to their creation in de_create)
void read_legacy_env(de_ctx* c) {
    c->trace = getenv("DE_AUTO_TRACE") != nullptr;
    if (const char* kv = getenv("DE_KERNEL")) { if (kv[0] >= '1' && kv[0] <= '6') c->kernel_variant = kv[0] - '0'; }
    c->launch_variant = (c->kernel_variant >= 4) ? 2 : c->kernel_variant;
    {   // scheduler knobs of render_kernel_v2 (experiments): read once here, not per launch
        auto env_int = [](const char* name, int* v, int lo) { if (const char* e = getenv(name)) { int x = atoi(e); if (x >= lo) *v = x; } };
        env_int("DE_V2_THR", &c->tune_pend, 0); env_int("DE_V2_A", &c->tune_heavy, 0); env_int("DE_V2_B", &c->tune_b, 0);
        env_int("DE_V2_G", &c->tune_gas, 0); env_int("DE_V2_CHUNK", &c->tune_chunk, 1); env_int("DE_V2_WPC", &c->tune_wpc, 1);
        env_int("DE_V2_MAX_SPP", &c->tune_max_spp, 1);      // test hook: split launches at this many samples per pixel
        if (const char* e3 = getenv("DE_V3_MAX_ITEMS")) { long long x = atoll(e3); if (x >= 4096) c->v3_max_items = (size_t)x; }
        env_int("DE_V3_WPC", &c->v3_wpc, 1); env_int("DE_V3_FUSE_TRACK", &c->v3_fuse_track, 0); env_int("DE_V3_FINISHER", &c->v3_finisher, 2); env_int("DE_V5_FIN_WPC", &c->v5_fin_wpc, 1);
        env_int("DE_V3_PIPES", &c->v3_pipes, 1); if (c->v3_pipes > 4) c->v3_pipes = 4;
        env_int("DE_V3_BATCHES", &c->v3_batches, 1);
        env_int("DE_V3_DEPTHS", &c->v3_depths, 1); if (c->v3_depths > 25) c->v3_depths = 25;
        env_int("DE_AUTO_DEPTHS", &c->auto_depths, 1); if (c->auto_depths > 25) c->auto_depths = 25;
        env_int("DE_AUTO_WPC", &c->auto_wpc, 1); env_int("DE_AUTO_PIPES", &c->auto_pipes, 0); if (c->auto_pipes > 4) c->auto_pipes = 4;
        if (const char* e6 = getenv("DE_V3_TEST_BUDGET")) { long long x = atoll(e6); if (x > 0) c->v3_test_budget = (size_t)x; }
        env_int("DE_AUTO_MID", &c->auto_mid_mode, 1); env_int("DE_AUTO_LONE_V5", &c->auto_lone_v5, 0); env_int("DE_AUTO_BUDGET_V5", &c->auto_budget_v5, 0); env_int("DE_AUTO_V6", &c->auto_v6, 0);
        if (const char* e6 = getenv("DE_AUTO_V6_MIN_ITEMS")) { long long x = atoll(e6); if (x >= 64) c->auto_v6_min_items = (unsigned long long)x; } env_int("DE_AUTO_MID_BATCHES1", &c->auto_mid_batches1, 0); env_int("DE_AUTO_FULL_BATCHES1", &c->auto_full_batches1, 0);
        env_int("DE_AUTO_DEEP_NS", &c->auto_deep_ns_x100, 0); env_int("DE_AUTO_DEEP_DEPTHS", &c->auto_deep_depths, 1); env_int("DE_AUTO_DEEP_PIPES", &c->auto_deep_pipes, 1);
        if (c->auto_deep_depths > 25) c->auto_deep_depths = 25; if (c->auto_deep_pipes > 4) c->auto_deep_pipes = 4;
        if (const char* e5 = getenv("DE_AUTO_TWO_PIPE_ITEMS")) { long long x = atoll(e5); if (x >= 0) c->auto_two_pipe_items = (unsigned long long)x; }
        if (const char* e8 = getenv("DE_AUTO_LONE_MAX_ITEMS")) { long long x = atoll(e8); if (x >= 0) c->auto_lone_max_items = (unsigned long long)x; }
        if (const char* e4 = getenv("DE_AUTO_MIN_ITEMS")) { long long x = atoll(e4); if (x >= 0) c->auto_min_items = (unsigned long long)x; }
        env_int("DE_V3_WPC_ST", &c->v3_wpc_class[0], 1); env_int("DE_V3_WPC_GAS", &c->v3_wpc_class[1], 1);
        env_int("DE_V3_WPC_CLOUD", &c->v3_wpc_class[2], 1); env_int("DE_V3_WPC_FLAT", &c->v3_wpc_class[3], 1);
        env_int("DE_V5_WPC", &c->v5_wpc, 1); env_int("DE_V5_SHARDS", &c->v5_shards, 1); env_int("DE_V5_POOL_LOG2", &c->v5_pool_log2, 10);
        env_int("DE_V5_CLAIM_K", &c->v5_claim_k, 1); env_int("DE_V5_AGE", &c->v5_age, 1); env_int("DE_V5_CLOSERS", &c->v5_closers_per_cu, 1);
        env_int("DE_V6_STATS", &c->v6_stats, 0); env_int("DE_V6_SVC_ST", &c->v6_svc[0], 1); env_int("DE_V6_SVC_GAS", &c->v6_svc[1], 1); env_int("DE_V6_SVC_CLOUD", &c->v6_svc[2], 1);
        env_int("DE_V6_AREA_ST", &c->v6_svc_area[0], 0); env_int("DE_V6_AREA_GAS", &c->v6_svc_area[1], 0); env_int("DE_V6_AREA_CLOUD", &c->v6_svc_area[2], 0);
        env_int("DE_V6_YIELD", &c->v6_yield, 0); env_int("DE_V6_ELSEWHERE", &c->v6_elsewhere, 1); env_int("DE_V6_RETRY", &c->v6_retry, 1); env_int("DE_V6_ENTER_MIN", &c->v6_enter_min, 0); env_int("DE_V6_FLAT_MIN", &c->v6_flat_min, 0); env_int("DE_V6_FLAT_AGAIN", &c->v6_flat_again, 1); env_int("DE_V5_STATS", &c->v5_stats, 0); env_int("DE_V5_LOCAL", &c->v5_local, 0); env_int("DE_V5_BACKLOG", &c->v5_backlog, 0); env_int("DE_V5_RESERVE_PCT", &c->v5_reserve_pct, 1); env_int("DE_V5_BATCHES", &c->v5_batches, 1);
        if (const char* e7 = getenv("DE_V5_MAX_ITEMS")) { long long x = atoll(e7); if (x >= 4096) c->v5_max_items = (size_t)x; }
        if (c->v5_shards > 32) c->v5_shards = 32;            // every ring has a cell per chunk of the pool (it cannot wrap): 1 MB x 7 per shard
        if (c->v5_pool_log2 > 19) c->v5_pool_log2 = 19;      // 2^19 chunks x 4 KB = 2 GB: the buffer descriptor addresses < 4 GB
        if (c->v5_claim_k > 64) c->v5_claim_k = 64;
        if (c->v5_wpc > 32) c->v5_wpc = 32;
        c->v3_slack_wpc = c->v3_wpc > c->auto_wpc ? c->v3_wpc : c->auto_wpc;
        for (int i = 0; i < 4; ++i) if (c->v3_wpc_class[i] > c->v3_slack_wpc) c->v3_slack_wpc = c->v3_wpc_class[i];
        if (c->v3_slack_wpc > 32) c->v3_slack_wpc = 32;
        if (c->v3_slack_wpc < 1) c->v3_slack_wpc = 1;
    }
    if (const char* sv = getenv("DE_SLOTS")) { int x = atoi(sv); if (x >= 1 && x <= DE_MAX_SLOTS) c->n_slots = x; }
    if (const char* sv = getenv("DE_BIG_SLOTS")) { int x = atoi(sv); if (x >= 1 && x <= DE_MAX_SLOTS) c->big_slots = x; }
}

// de_accumulate of rounds 1-4, from the point where the call's RenderArgs are filled: every kernel variant and the launch policy of the automatic one
int accumulate_legacy(de_ctx* c, RenderArgs& a, int spp, int first_index) {
    const bool v2 = c->kernel_variant != 1 && !(c->p.flags & DE_FLAG_RAY_MARCHER);     // per-sample records + accumulate_kernel
    // does this call go through the wavefront pipeline, and with which settings
    const unsigned long long call_items = (unsigned long long)a.n_tiles * 64ull * (unsigned long long)(spp > 0 ? spp : 0);
    bool pipeline = false, mid_whole = false;
    int pipes = 1, batches = c->v3_batches;
    // The automatic variant: every call of at least 4096 paths runs the per-CU stage scheduler (render_kernel_v6.hip) in one launch per
    // 2^29 paths — it ends a frame sooner than the pipeline + state machine on every camera and call size measured (1920x1080x64 spp:
    // 122 against 162 ms per frame; a 1/8 share of it 21 against 26; 1 spp + fetch_image 10 against 15) and needs no queue memory.
    // Smaller calls (and counting launches) run the state machine.  DE_AUTO_V6=0 brings back the policy of rounds 3-4 below.
    const bool auto_v6 = c->kernel_variant == 4 && c->auto_v6 && v2 && !c->count && call_items >= c->auto_v6_min_items;
    if (c->kernel_variant == 3 && v2 && !c->count) {
        pipeline = true; pipes = c->v3_pipes; c->launch_depths = c->v3_depths; c->launch_wpc = c->v3_wpc;
    } else if (c->kernel_variant == 4 && !auto_v6 && v2 && !c->count && call_items >= c->auto_min_items && spp >= 2) {
        // what earlier launches found (whichever of them have completed: nothing is waited for; results do not depend on it)
        for (int i = 0; i < 4; ++i) {
            de_ctx::V3Pipe& P = c->v3p[i];
            if (P.stat_pending && hipEventQuery(P.stat_ev) == hipSuccess) {
                if (P.stat_items) c->auto_ratio = (float)((double)*P.h_round2 / (double)P.stat_items);
                float t_ms = 0.f;
                if (P.stat_timed && P.stat_items && hipEventElapsedTime(&t_ms, P.ev_t[0], P.ev_t[2]) == hipSuccess && t_ms > 0.f)
                    c->auto_ns_per_path = t_ms * 1e6f / (float)P.stat_items / (float)(P.stat_pipes > 0 ? P.stat_pipes : 1);
                P.stat_pending = false;
            }
        }
        (void)hipGetLastError();      // hipErrorNotReady of a query is not an error of this call
        const bool two_pipe_size = call_items / 2ull > c->auto_two_pipe_items;
        const bool mid_pipeline = c->auto_mid_mode != 2;      // calls below the two-pipe size
        if (two_pipe_size || c->auto_pipes > 0 || mid_pipeline) {
            pipeline = true; c->launch_depths = c->auto_depths; c->launch_wpc = c->auto_wpc;
            pipes = c->auto_pipes > 0 ? c->auto_pipes : (two_pipe_size ? 2 : 4);
            // a mid-size call (a rank's share of a tile partition) while earlier calls are still in flight — frames issued back to back —
            // goes to ONE pipe as one batch: four frames on four pipes overlap better than two half-frames of each (1/8 of the frame, ms per
            // step: default camera 29.5 -> 24.5, florida 28.5 -> 25.1, sunset hurricane 53.5 -> 49.0, Apollo 19.7 -> 14.9; 1/4: 48.5 -> 42.7).
            // A lone call keeps two batches side by side: its latency is 10-35 % better that way (43.8 against 48.3 ms).
            // (Only for calls a pipe's queues hold in one piece at the size the two-pipe mode uses anyway, 48 M paths; and every pipe such
            // a call may land on is sized for the whole call from the start, so that going from a lone call to calls in flight does not
            // re-allocate queues in the middle of a run.)
            mid_whole = !two_pipe_size && c->auto_pipes <= 0 && c->auto_mid_batches1 && call_items <= c->auto_two_pipe_items;
            if (mid_whole && launches_in_flight(c)) batches = 1;
            if (c->auto_ns_per_path >= 0.f) {
                const float thr = 0.01f * (float)c->auto_deep_ns_x100;
                if (!c->auto_deep && c->auto_ns_per_path >= thr) c->auto_deep = true;
                else if (c->auto_deep && c->auto_ns_per_path < 0.8f * thr) c->auto_deep = false;
            }
            if (c->auto_deep && pipes == 2 && c->auto_deep_depths > c->auto_depths) {
                c->launch_depths = c->auto_deep_depths;
                if (c->auto_pipes <= 0 && c->auto_deep_pipes >= 2) { pipes = c->auto_deep_pipes; if (batches < pipes) batches = pipes; }
            } else if (two_pipe_size && c->auto_pipes <= 0 && c->auto_full_batches1 && launches_in_flight(c)) {
                // frames back to back on a view that is not in the deep setting: the whole call as ONE batch, consecutive calls on the two
                // pipes — half the stage launches (and their tails) per frame, the overlap comes from the caller.  1920x1080x64 spp, ms per
                // step: default camera 171.4 -> 160.8, florida 200.3 -> 195.7, Apollo 108.4 -> 100.3 (three / four pipes: 158.7 / 153.2, for
                // 68 GB of queues each); sunset hurricane 353.7 -> 369.5, which is why the deep setting keeps its batches.  A lone call
                // keeps two batches (173 ms against 208).  The queues grow to the whole call the first time this happens (68 GB per pipe for
                // this frame; when the device or the memory budget has no room the batch is halved as always).
                batches = 1;
            }
        }
        if (c->trace)
            fprintf(stderr, "[de] auto: %llu paths; pipeline launch %.2f ns per path (round-2 ratio %.3f): %s, depths %d pipes %d batches %d\n",
                    call_items, c->auto_ns_per_path, c->auto_ratio, pipeline ? "pipeline" : "state machine", c->launch_depths, pipes, batches);
    }
    size_t budget_fit = 0;           // paths per pipeline launch the memory budget allows (0: no budget)
    bool budget_v5 = false;          // the budget has no room for the pipeline: the persistent stage scheduler takes the call
    if (pipeline && c->mem_budget && a.n_tiles > 0) {
        // A budget that is binding — a pipe's share holds less than the batch the call would use — cuts the call into more, smaller
        // launches, each with its own tails.  Measured (profiles/memory_budget.md): batches under ~24 M paths lose to the state
        // machine in one launch on light views.  So: fewer pipes first (two), and below that size no pipeline for this call.
        const size_t slack_b = (size_t)wf::NQ * (4u * (size_t)c->n_cus * (size_t)c->v3_slack_wpc * DE_V3_BULK) * sizeof(wf::Rec);
        auto fit_of = [&](int np) { const size_t share = c->mem_budget / (size_t)np; return share > slack_b ? (share - slack_b) / 512u : (size_t)0; };
        const int nb = batches > pipes ? batches : pipes;
        size_t nominal = (size_t)(call_items / (unsigned long long)(nb > 0 ? nb : 1));
        if (nominal > c->v3_max_items) nominal = c->v3_max_items;
        const size_t floor_items = (size_t)24 << 20;
        if (fit_of(pipes) < nominal && c->launch_depths != c->auto_depths) {
            // the deep setting (more rounds, a third pipe) is chosen from the GPU time per path of earlier launches, which small batches inflate
            c->launch_depths = c->auto_depths;
            if (c->auto_pipes <= 0 && pipes > 2) { pipes = 2; batches = c->v3_batches; }
        }
        if (fit_of(pipes) < nominal && fit_of(pipes) < floor_items) {
            if (pipes > 2 && fit_of(2) >= floor_items) { pipes = 2; if (batches > 2) batches = 2; }
            else pipeline = false;
        }
        if (pipeline) budget_fit = fit_of(pipes);
        else if (c->kernel_variant == 4 && c->auto_budget_v5 && v2 && !c->count) {
            // No room for the pipeline's queues: the persistent stage scheduler needs a fixed pool and 64 B per path instead of 512, and
            // ends the frame sooner than the state machine on every camera measured (1920x1080x64 spp: 186 / 254 / 491 / 138 ms against
            // 211 / 305 / 685 / 170) — when at least a quarter of the call fits one of its three launch slots.
            const unsigned long long fixed = ((unsigned long long)DE_V5_CH * 64ull << c->v5_pool_log2) + ((unsigned long long)c->v5_shards * wf::NQ * 4ull << c->v5_pool_log2);
            const unsigned long long share = (unsigned long long)c->mem_budget / 3ull;
            budget_v5 = share > fixed && (share - fixed) / 64ull >= call_items / 4ull;
        }
    }
    // The automatic variant's LONE calls.  A call that finds NO launch of this context in flight — the host has fetched / synchronised
    // since the last one: every frame of the reference's window loop, accumulate() x 1 then fetch_image() (earth_viewer.py:241-243), or a
    // progressive render that displays after every call — is alone on the GPU and its latency is what the user sees.  The persistent stage
    // scheduler ends a small such call sooner than the state machine: 1920x1080 x 1 spp + fetch_image 14.7 against 20.4 ms (a launch of the
    // state machine cannot be shorter than its longest path in ONE lane; the scheduler's paths hop between waves that are all busy).  Calls
    // issued while others are in flight overlap through the launch slots on the state machine (64 x 1 spp: 295 against 523 ms).
    bool lone_small = false;
    if (c->kernel_variant == 4 && !auto_v6 && c->auto_lone_v5 && v2 && !c->count && call_items >= (1ull << 18) && call_items <= c->auto_lone_max_items) {
        lone_small = !launches_in_flight(c);
        if (c->mem_budget) {      // its pool, rings and cold records must fit the budget (one slot): otherwise the state machine, which needs none
            const unsigned long long fixed = ((unsigned long long)DE_V5_CH * 64ull << c->v5_pool_log2) + ((unsigned long long)c->v5_shards * wf::NQ * 4ull << c->v5_pool_log2);
            if (fixed + call_items * 64ull > (unsigned long long)c->mem_budget) lone_small = false;
        }
        if (lone_small) pipeline = false;
    }
    const bool v5k = (c->kernel_variant == 5 || lone_small || budget_v5) && v2 && !c->count;      // the persistent stage scheduler (counting launches use the state machine)
    const bool v6k = (c->kernel_variant == 6 || auto_v6) && v2 && !c->count;      // the per-CU stage scheduler (LDS queues)
    const int base_variant = c->kernel_variant == 4 ? (v6k ? 6 : (v5k ? 5 : 2)) : (c->kernel_variant == 5 ? (v5k ? 5 : 2) : (c->kernel_variant == 6 ? (v6k ? 6 : 2) : c->kernel_variant));
    const bool v3k = pipeline && v2 && !c->count;      // the call's launches run the wavefront pipeline
    c->launch_one_batch = v3k && batches == 1 && pipes >= 2;
    // A launch of render_kernel_v2 handles n_tiles*64*spp work items: the item index must fit 32 bits, the tile index is
    // recovered with a 2^40 magic multiplication (exact while n_tiles*spp^2 < 2^40), and each item owns 8 bytes of the
    // contribution buffer (kept <= 4 GB).  Larger requests are split here; the sample index continues from launch to launch
    // and the per-pixel sum keeps sample order, so split launches give the same bits as one launch would.
    int max_spp = spp > 0 ? spp : 1;
    if (v2 && a.n_tiles > 0) {
        const unsigned long long nt = (unsigned long long)a.n_tiles;
        unsigned long long cap = ((1ull << 29) / (nt * 64ull));                          // 2^29 items = 4 GB of contributions
        while (cap > 1 && nt * cap * cap >= (1ull << 40)) --cap;
        if (cap < 1) cap = 1;
        if (c->tune_max_spp > 0 && cap > (unsigned long long)c->tune_max_spp) cap = (unsigned long long)c->tune_max_spp;
        if (v3k) {   // the stage queues hold 512 B per path
            unsigned long long cap3 = (unsigned long long)(budget_fit && budget_fit < c->v3_max_items ? budget_fit : c->v3_max_items) / (nt * 64ull);
            if (cap3 < 1) cap3 = 1;
            if (cap > cap3) cap = cap3;
            // two pipes: a call is cut in (at least) two batches that run side by side
            const int nb = batches < spp ? batches : spp;
            if (pipes >= 2 && nb >= 2 && cap > (unsigned long long)((spp + nb - 1) / nb)) cap = (unsigned long long)((spp + nb - 1) / nb);
        }
        if (v5k) {   // 64 B of cold record per path of a launch
            unsigned long long max5 = (unsigned long long)c->v5_max_items;
            if (c->mem_budget && !lone_small) {      // three launch slots in flight, each: the chunk pool + rings, and 64 B of cold record per path
                const unsigned long long fixed = ((unsigned long long)DE_V5_CH * 64ull << c->v5_pool_log2) + ((unsigned long long)c->v5_shards * wf::NQ * 4ull << c->v5_pool_log2);
                const unsigned long long share = (unsigned long long)c->mem_budget / 3ull;
                const unsigned long long fit = share > fixed ? (share - fixed) / 64ull : 0ull;
                if (fit < max5) max5 = fit;
                if (max5 < nt * 64ull) return fail(DE_ERR_NOMEM, "the memory budget has no room for one sample per pixel on the persistent stage scheduler (de_set_memory_budget)");
            }
            unsigned long long cap5 = max5 / (nt * 64ull);
            if (cap5 < 1) cap5 = 1;
            if (cap > cap5) cap = cap5;
            const int nb = c->v5_batches < spp ? c->v5_batches : spp;
            if (nb >= 2 && call_items >= c->auto_min_items && cap > (unsigned long long)((spp + nb - 1) / nb)) cap = (unsigned long long)((spp + nb - 1) / nb);
        }
        if ((unsigned long long)max_spp > cap) max_spp = (int)cap;
        if (max_spp < spp) { const int nl = (spp + max_spp - 1) / max_spp; max_spp = (spp + nl - 1) / nl; }      // launches of equal size: 64 spp under a cap of 15 is 5 x 13, not 4 x 15 + 4
        if (nt * 64ull * (unsigned long long)max_spp >= (1ull << 32) || nt * (unsigned long long)max_spp * (unsigned long long)max_spp >= (1ull << 40))
            return fail(DE_ERR_INVALID, "image too large for render_kernel_v2's 32-bit work-item index");
    }
    // What the context stream has done since the last launch and a launch must be ordered after (set-up kernels, map packing,
    // the tile list: before the render kernel; reset / display / reduce / uploads of the HDR buffer: before accumulate_kernel).
    if (c->rec_render != c->gen_render || c->rec_hdr != c->gen_hdr) {
        HIP_TRY(hipEventRecord(c->ev_main, c->stream));
        c->rec_render = c->gen_render; c->rec_hdr = c->gen_hdr;
    }
    bool first = true;
    c->last_call[0] = 0; c->last_call[1] = 1; c->last_call[2] = 0; c->last_call[3] = 0;
    c->timing_empty = true;
    for (int done = 0; done < spp && a.n_tiles > 0;) {
        bool this_v3 = v3k;
        int n = (spp - done < max_spp) ? spp - done : max_spp;
        // small launches (the interactive 1-spp pattern: a 16 MB record buffer, 3.5 ms of work behind a 15 ms tail) use all the
        // slots; big ones only the first few — each slot keeps a record buffer of the largest launch it has seen, and more
        // than three big persistent kernels side by side only dilute each other (1/8 frame at 64 spp, 137 MB of records:
        // 30.9 ms per frame with 3 slots, 35.1 ms with 8)
        const size_t need_bytes = v2 ? (size_t)a.n_tiles * 64u * (size_t)n * sizeof(uint2) : 0;
        // pipeline launches cycle through the call's pipes (one launch slot each)
        const int n_cycle_want = v3k ? pipes : ((need_bytes > ((size_t)96 << 20)) ? (c->big_slots < c->n_slots ? c->big_slots : c->n_slots) : c->n_slots);
        const int n_cycle = n_cycle_want < c->n_slots ? n_cycle_want : c->n_slots;      // never a slot the context stream does not join
        const int si = lone_small ? 0 : c->next_slot % n_cycle;      // a lone small call keeps ONE slot's pool and rings warm
        const int pipe = (v3k && pipes >= 2) ? si : 0;
        c->launch_pipe = pipe; c->launch_pipes = n_cycle; c->launch_slot = si;
        if (v5k) { int rc5 = v5_reserve(c, si, c->slot[si].stream, (size_t)a.n_tiles * 64u * (size_t)n); if (rc5) return rc5; }
        if (v6k) { int rc6 = v6_reserve(c, si, c->slot[si].stream, (unsigned long long)a.n_tiles * 64ull * (unsigned long long)n); if (rc6) return rc6; }
        if (this_v3 && c->v3_finisher == 5 && c->launch_depths < 25) { int rc5 = v5_reserve(c, DE_MAX_SLOTS + pipe, c->slot[si].stream, (size_t)a.n_tiles * 64u * (size_t)n); if (rc5) return rc5; }
        if (this_v3) {
            // queues for this batch; when the device has no room for them the batch shrinks (split launches give the same bits),
            // and a batch of one sample per pixel that still does not fit runs on the state machine
            int rc3 = v3_reserve(c, pipe, mid_whole ? (size_t)call_items : (size_t)a.n_tiles * 64u * (size_t)n);
            if (rc3 == V3_NO_MEMORY && mid_whole) rc3 = v3_reserve(c, pipe, (size_t)a.n_tiles * 64u * (size_t)n);      // no room for the whole call: this batch's size then
            while (rc3 == V3_NO_MEMORY && n > 1) {
                n = (n + 1) / 2; max_spp = n;
                rc3 = v3_reserve(c, pipe, (size_t)a.n_tiles * 64u * (size_t)n);
            }
            if (rc3 == V3_NO_MEMORY) { this_v3 = false; rc3 = DE_OK; }
            if (rc3) return rc3;
        }
        // a batch whose queues do not fit runs on the state machine — also under kernel variant 3: its pipe has just been freed
        c->launch_variant = this_v3 ? 3 : (base_variant == 3 ? 2 : base_variant);
        LaunchSlot& s = c->slot[si];
        if (v2) {
            const size_t need = (size_t)a.n_tiles * 64u * (size_t)n;
            if (need > s.contrib_items) {
                // grow the record buffers of EVERY slot of this launch's cycle now: the first launch of a new size pays for all
                // the allocations, none falls between later launches (hipFree / hipMalloc stall the streams)
                for (int j = 0; j < n_cycle; ++j) {
                    LaunchSlot& sj = c->slot[j];
                    if (need <= sj.contrib_items) continue;
                    HIP_TRY(hipStreamSynchronize(sj.stream));      // the slot's previous launch still reads the old buffer
                    if (sj.contrib) hipFree(sj.contrib);
                    sj.contrib = nullptr; sj.contrib_items = 0;
                    HIP_TRY(hipMalloc(&sj.contrib, need * sizeof(uint2)));
                    sj.contrib_items = need;
                }
            }
        }
        a.contrib = s.contrib;
        a.work_counter = c->d_work_counter + 16 * si;
        a.spp_begin = first_index + done * a.spp_stride; a.spp_count = n;
        a.spp_magic = ((1ull << 40) / (unsigned long long)n) + 1ull;
        if (s.seen_render != c->gen_render) {
            HIP_TRY(hipStreamWaitEvent(s.stream, c->ev_main, 0));
            s.seen_render = c->gen_render; s.seen_hdr = c->gen_hdr;
        }
        if (this_v3) {
            if (c->last_v3_slot[pipe] >= 0 && c->last_v3_slot[pipe] != si) HIP_TRY(hipStreamWaitEvent(s.stream, c->slot[c->last_v3_slot[pipe]].done, 0));
            c->last_v3_slot[pipe] = si;
        }
        if (first) { HIP_TRY(hipEventRecord(s.t0, s.stream)); c->t0_slot = si; first = false; }
        const int prev = c->last_slot;
        auto pre_acc = [&]() -> hipError_t {
            if (s.seen_hdr != c->gen_hdr) {
                hipError_t e = hipStreamWaitEvent(s.stream, c->ev_main, 0);
                if (e != hipSuccess) return e;
                s.seen_hdr = c->gen_hdr;
            }
            // HDR sums keep launch order: this launch's accumulate_kernel after the previous launch's
            if (prev >= 0 && prev != si) return hipStreamWaitEvent(s.stream, c->slot[prev].done, 0);
            return hipSuccess;
        };
        HIP_TRY(c->count ? launch_render<1>(c, a, s.stream, pre_acc) : launch_render<0>(c, a, s.stream, pre_acc));
        c->last_call[0] = (c->p.flags & DE_FLAG_RAY_MARCHER) ? 0 : c->launch_variant;
        c->last_call[1] = this_v3 ? n_cycle : 1; c->last_call[2] = this_v3 ? c->launch_depths : 0; c->last_call[3]++;
        HIP_TRY(hipEventRecord(s.done, s.stream));
        s.pending = true; s.launched = true;
        c->last_slot = si;
        c->next_slot = (si + 1) % n_cycle;
        done += n;
        if (done >= spp) {
            HIP_TRY(hipEventRecord(s.t1, s.stream)); c->t1_slot = si; c->timing_valid = true; c->timing_empty = false;
        }
    }
    return DE_OK;
}

void free_queue_memory(de_ctx* c) {
    for (auto& P : c->v3p) {
        for (int i = 0; i < wf::NQ; ++i) { if (P.rec[i]) hipFree(P.rec[i]); P.rec[i] = nullptr; }
        if (P.cold) hipFree(P.cold);
        P.cold = nullptr; P.capacity = 0;
    }
    for (auto& S : c->v5s) {
        if (S.pool) hipFree(S.pool); if (S.rings) hipFree(S.rings); if (S.free_ring) hipFree(S.free_ring); if (S.cold) hipFree(S.cold);
        S.pool = nullptr; S.rings = nullptr; S.free_ring = nullptr; S.cold = nullptr; S.cold_items = 0; S.pool_chunks = 0;
    }
}
void legacy_destroy(de_ctx* c) {
    for (auto& P : c->v3p) {
        for (int i = 0; i < wf::NQ; ++i) if (P.rec[i]) hipFree(P.rec[i]);
        if (P.ctl) hipFree(P.ctl);
        if (P.cold) hipFree(P.cold);
        if (P.h_round2) hipHostFree(P.h_round2);
        if (P.stat_ev) hipEventDestroy(P.stat_ev);
        for (int i = 0; i < 3; ++i) if (P.ev_t[i]) hipEventDestroy(P.ev_t[i]);
    }
    for (auto& S : c->v5s) {
        if (S.pool) hipFree(S.pool); if (S.ctl) hipFree(S.ctl); if (S.rings) hipFree(S.rings); if (S.free_ring) hipFree(S.free_ring);
        if (S.cold) hipFree(S.cold);
        if (S.d_args) hipFree(S.d_args);
        if (S.h_status) hipHostFree(S.h_status);
    }
}
uint64_t legacy_memory_use(de_ctx* c) {
    uint64_t n = 0;
    for (auto& P : c->v3p) {
        if (!P.capacity) continue;
        const size_t slack = 4u * (size_t)c->n_cus * (size_t)c->v3_slack_wpc * DE_V3_BULK;
        n += (uint64_t)wf::NQ * (P.capacity + slack) * sizeof(wf::Rec) + (uint64_t)P.capacity * sizeof(wf::Cold);
    }
    for (auto& S : c->v5s) {
        if (S.pool) n += (uint64_t)S.pool_chunks * DE_V5_CH * 64u + (uint64_t)S.n_shards * wf::NQ * S.ring_cells * 4u + (uint64_t)S.pool_chunks * 4u;
        n += (uint64_t)S.cold_items * 64u;
    }
    return n;
}

}  // namespace

extern "C" {
/* statistics block of the persistent stage scheduler (DE_V5_STATS=1), summed over the launch slots' LAST launches: out[k] = word k of
 * render_kernel_v5.hip's ST_* list */
int de_debug_v5_stats(de_ctx* c, uint64_t* out, int n) {
    if (!c || !out || n < 0) return fail(DE_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    { int rc = sync_all(c); if (rc) return rc; }
    for (int k = 0; k < n; ++k) out[k] = 0;
    std::vector<uint32_t> h((size_t)ps::C_WORDS * DE_V5_CTL_STRIDE);
    for (auto& S : c->v5s) {
        if (!S.ctl) continue;
        HIP_TRY(hipMemcpy(h.data(), S.ctl, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (int k = 0; k < n && k < ps::ST_N; ++k) { uint64_t v; memcpy(&v, &h[(size_t)(ps::C_STAT0 + k) * DE_V5_CTL_STRIDE], 8); out[k] += v; }
    }
    return DE_OK;
}

}  // extern "C"
