// de_launch.h — what the entry points of de_api.hip share: ordering between the context stream and the launch slots, map packing, the
// set-up kernel, a launch's RenderArgs, and the launches themselves (render_kernel_v6 for calls of 4096 paths and more, render_kernel_v2 for
// smaller calls / counting / single-sample traces, ray_march_kernel).  The legacy kernel families (per-lane loops, the wavefront pipeline and
// its launch policy, the HBM-queue scheduler) hang in through legacy/de_launch_legacy.h when the library is built with -DDE_LEGACY_VARIANTS.
#pragma once
#include "de_host_consts.h"

namespace {


// The context stream waits (on the device; the host does not block) for every launch issued so far.  Called before anything on
// the context stream that touches the HDR buffer or rewrites data a render launch reads.
int join_slots(de_ctx* c) {
    for (int i = 0; i < c->n_slots; ++i) {
        LaunchSlot& s = c->slot[i];
        if (!s.pending) continue;
        HIP_TRY(hipStreamWaitEvent(c->stream, s.done, 0));
        s.pending = false;
    }
    return DE_OK;
}
// Is a launch of this context still running (or queued) on the GPU?  Asked, never waited for: it tells a caller that keeps calls in flight
// (frames back to back: throughput is what counts) from one that fetches or synchronises after every call (its latency counts).
bool launches_in_flight(de_ctx* c) {
    bool busy = false;
    for (int i = 0; i < c->n_slots && !busy; ++i)
        if (c->slot[i].launched && hipEventQuery(c->slot[i].done) == hipErrorNotReady) busy = true;
    (void)hipGetLastError();      // hipErrorNotReady is an answer, not an error
    return busy;
}
// context-stream work that the next launches must be ordered after: what a render kernel reads / the HDR buffer
void touched_render_inputs(de_ctx* c) { c->gen_render++; }
void touched_hdr(de_ctx* c) { c->gen_hdr++; }
void note_abort(de_ctx* c);
int frame_status(de_ctx* c);
// Wait for everything the context has issued.  An abort word found afterwards is RECORDED (the frame is invalid until de_reset) but does not
// fail the caller: uploads, partition changes and the like synchronise for their own reasons; the entry points that hand a frame to the host
// (fetch, reduce, flush, synchronize) ask frame_status().
int sync_all(de_ctx* c) {
    for (int i = 0; i < c->n_slots; ++i)
        if (c->slot[i].stream) HIP_TRY(hipStreamSynchronize(c->slot[i].stream));
    for (int i = 0; i < c->n_slots; ++i) c->slot[i].pending = false;
    HIP_TRY(hipStreamSynchronize(c->stream));
    note_abort(c);
    return DE_OK;
}
void release_loan(de_ctx* c) {
    if (c->lender) { c->lender->loans--; c->lender = nullptr; }
}

int ensure_packed(de_ctx* c, int slot, bool clamp) {
    DevTexture& t = c->tex[slot];
    if (t.packed_clamp == (clamp ? 1 : 0)) return DE_OK;
    if (t.borrowed) return fail(DE_ERR_STATE, "a shared texture keeps its owner's address mode: set DE_FLAG_CLAMP_SAMPLER on the owner and render there once, then share again");
    if (!t.linear) return fail(DE_ERR_STATE, "the sampler address mode cannot change after de_trim_textures released the as-uploaded maps: upload or generate them again");
    if (c->loans > 0 && t.packed_clamp != -1) return fail(DE_ERR_STATE, "the maps are lent to another context (de_share_textures): they cannot be repacked for another address mode until every borrower is destroyed or holds maps of its own");
    { int rc = join_slots(c); if (rc) return rc; }
    touched_render_inputs(c);
    if (t.ch == 1) {
        dim3 grid((unsigned)((t.w + 255) / 256), (unsigned)t.h);
        hipLaunchKernelGGL(pack_r8_kernel, grid, dim3(256), 0, c->stream, t.linear, t.packed, t.w, t.h, t.tiles_x, clamp ? 1 : 0);
    } else {
        size_t n = (size_t)t.w * t.h;
        hipLaunchKernelGGL(pack_rgb_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, t.linear, t.packed, n);
    }
    HIP_TRY(hipGetLastError());
    t.packed_clamp = clamp ? 1 : 0;
    return DE_OK;
}

int alloc_texture(de_ctx* c, int slot, int w, int h, int ch) {
    DevTexture& t = c->tex[slot];
    if ((unsigned long long)(ch == 1 ? ((w + 7) / 8) * 8 : w) * (unsigned long long)(ch == 1 ? ((h + 3) / 4) * 4 : h) >= (1ull << 30))
        return fail(DE_ERR_INVALID, "map too large: the packed copy (one dword per texel) must stay below 4 GB — the kernels address it with a 32-bit byte offset");
    if (c->loans > 0) return fail(DE_ERR_STATE, "the maps are lent to another context (de_share_textures): destroy the borrowers, or give them maps of their own, before replacing a map");
    { int rc = sync_all(c); if (rc) return rc; }      // no launch may still read the map that is being replaced
    if (t.linear && !t.borrowed) hipFree(t.linear);
    if (t.packed && !t.borrowed) hipFree(t.packed);
    t.linear = nullptr; t.packed = nullptr; t.borrowed = false;
    {   // a borrower that now owns every map again gives the loan back
        bool any = false;
        for (int i = 0; i < DE_TEX_COUNT; ++i) any = any || (i != slot && c->tex[i].borrowed);
        if (!any && !c->luts_borrowed) release_loan(c);
    }
    touched_render_inputs(c);
    t.w = w; t.h = h; t.ch = ch; t.set = false; t.packed_clamp = -1;
    HIP_TRY(hipMalloc(&t.linear, (size_t)w * h * ch));
    if (ch == 1) {
        t.tiles_x = (w + 7) / 8; t.tiles_y = (h + 3) / 4;
        HIP_TRY(hipMalloc(&t.packed, (size_t)t.tiles_x * t.tiles_y * 32 * sizeof(uint32_t)));
    } else {
        t.tiles_x = t.tiles_y = 0;
        HIP_TRY(hipMalloc(&t.packed, (size_t)w * h * sizeof(uint32_t)));
    }
    return DE_OK;
}

int run_setup(de_ctx* c) {
    if (!c->params_dirty && !c->nodes_dirty) return DE_OK;
    { int rc = join_slots(c); if (rc) return rc; }    // launches in flight still read the tables this rewrites
    touched_render_inputs(c);
    SetupArgs s;
    s.p = c->p;
    opendrt_consts(&s.drt[0], &s.drt[1], &s.drt[2], &s.drt[3], &s.drt[4], &s.drt[5]);
    agx_consts(s.agx);
    s.W = c->W; s.H = c->H;
    s.topo_w = c->tex[DE_TEX_TOPOGRAPHY].w > 0 ? c->tex[DE_TEX_TOPOGRAPHY].w : 1;
    s.n_crf = c->n_crf;
    s.cie.data = c->d_cie; s.cie.w = 441; s.cie.h = 2;
    s.srgb2spec = c->d_srgb2spec; s.o3 = c->d_o3;
    s.fc = c->d_fc; s.nodes = c->d_nodes; s.node_val = c->d_node_val;
    s.do_nodes = c->nodes_dirty ? 1 : 0;
    // camera / sun / display changes rebuild the frame constants only (one wave); the 512-node wavelength table depends on
    // the LUTs, the address mode and the fixed wavelength alone
    hipLaunchKernelGGL(setup_kernel, dim3(c->nodes_dirty ? DE_N_NODES / 64 : 1), dim3(64), 0, c->stream, s);
    HIP_TRY(hipGetLastError());
    c->params_dirty = false; c->nodes_dirty = false;
    return DE_OK;
}

int build_tiles(de_ctx* c, int rank, int world) {
    if (c->tiles_rank == rank && c->tiles_world == world && c->d_tiles) return DE_OK;
    { int rc = sync_all(c); if (rc) return rc; }      // launches in flight read the tile list
    touched_render_inputs(c);
    const int tx = c->W / 8, ty = c->H / 8;
    std::vector<uint32_t> tiles;
    tiles.reserve((size_t)tx * ty / world + tx);
    for (int y = 0; y < ty; ++y)
        for (int x = 0; x < tx; ++x)
            if (world <= 1 || ((x + y) % world) == rank) tiles.push_back((uint32_t)(y * tx + x));
    if (!c->d_tiles) HIP_TRY(hipMalloc(&c->d_tiles, (size_t)tx * ty * sizeof(uint32_t)));
    if (!tiles.empty()) HIP_TRY(hipMemcpyAsync(c->d_tiles, tiles.data(), tiles.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));   // `tiles` goes out of scope
    c->n_tiles = (int)tiles.size(); c->tiles_rank = rank; c->tiles_world = world;
    return DE_OK;
}

int fill_render_args(de_ctx* c, RenderArgs* a) {
    for (int i = 0; i < DE_TEX_COUNT; ++i)
        if (!c->tex[i].set) return fail(DE_ERR_STATE, "all 7 textures must be uploaded or generated before rendering");
    if (!c->luts_set) return fail(DE_ERR_STATE, "LUTs must be uploaded before rendering");
    const bool clamp = (c->p.flags & DE_FLAG_CLAMP_SAMPLER) != 0;
    for (int i = 0; i < DE_TEX_COUNT; ++i) { int rc = ensure_packed(c, i, clamp); if (rc) return rc; }
    int rc = run_setup(c);
    if (rc) return rc;
    auto r8 = [&](int slot) {
        const DevTexture& t = c->tex[slot];
        TexR8 o; o.data = t.packed; o.w = t.w; o.h = t.h; o.tiles_x = t.tiles_x; o.wf = (float)t.w; o.hf = (float)t.h;
        return o;
    };
    auto rgb = [&](int slot) {
        const DevTexture& t = c->tex[slot];
        TexRGB o; o.data = t.packed; o.w = t.w; o.h = t.h; o.wf = (float)t.w; o.hf = (float)t.h;
        return o;
    };
    a->fc = c->d_fc; a->nodes = c->d_nodes; a->node_val = c->d_node_val; a->dens_table = c->d_dens_table;
    a->topo = r8(DE_TEX_TOPOGRAPHY); a->ocean = r8(DE_TEX_OCEAN); a->clouds = r8(DE_TEX_CLOUDS);
    a->bathy = r8(DE_TEX_BATHYMETRY); a->emissive = r8(DE_TEX_EMISSIVE);
    a->albedo = rgb(DE_TEX_ALBEDO); a->stars = rgb(DE_TEX_STARS);
    a->hdr = c->d_hdr; a->W = c->W; a->H = c->H;
    a->tiles = c->d_tiles; a->n_tiles = c->n_tiles; a->tiles_x = c->W / 8; a->tiles_identity = (c->tiles_world <= 1) ? 1 : 0;
    a->fixed_wavelength = (c->p.flags & DE_FLAG_FIXED_WAVELENGTH) ? 1 : 0;
    a->counters = c->d_counters; a->debug_out = c->d_scratch; a->work_counter = c->d_work_counter;   // work_counter / contrib: per launch slot, set by the caller
    a->resume_rec = nullptr; a->resume_count = nullptr; a->resume_cold = nullptr; a->resume_capacity = 0;
    a->pend_thr = c->tune_pend; a->heavy_thr = c->tune_heavy; a->tune_b = c->tune_b; a->gas_thr = c->tune_gas; a->chunk = c->tune_chunk;
    return DE_OK;
}

#ifdef DE_LEGACY_VARIANTS
// legacy/de_launch_legacy.h (included at the end of this file)
hipError_t launch_v3(de_ctx* c, const RenderArgs& a, hipStream_t stream, int pipe);
hipError_t launch_v5(de_ctx* c, const RenderArgs& a, hipStream_t stream, int idx, const uint4* resume_rec = nullptr, const uint32_t* resume_count = nullptr,
                     const float* resume_cold = nullptr, uint32_t resume_capacity = 0, int wpc = 0);
void legacy_note_abort(de_ctx* c);
#endif

// render_kernel_v6: buffers of one launch slot (independent of the launch's size)
int v6_reserve(de_ctx* c, int idx, hipStream_t stream, unsigned long long n_items) {
    de_ctx::V6State& S = c->v6s[idx];
    const uint32_t n_wg = (uint32_t)(c->cu_withhold > 0 ? c->n_cus - 8 * c->cu_withhold : c->n_cus) * DE_V6_WG_PER_CU;
    // the tail's pools: level k exports at most (workgroups of level k) x v6_tail_export[k] records (render_kernel_v6.hip: "The tail").  Only a launch that can have
    // a tail needs them: the interactive 1-spp pattern (all 8 launch slots, calls far below v6_tail_min_paths) holds none
    uint32_t want[2] = {0u, 0u};
    const bool tail = n_items >= (unsigned long long)c->v6_tail_min_paths;
    if (tail && c->v6_tail_levels >= 1) want[0] = n_wg * (uint32_t)c->v6_tail_export[0];
    if (tail && c->v6_tail_levels >= 2) want[1] = std::min((uint32_t)c->v6_tail_grid[0], n_wg) * (uint32_t)c->v6_tail_export[1];
    for (int k = 0; k < 2; ++k) {
        if (S.pool_cap[k] >= want[k]) continue;
        HIP_TRY(hipStreamSynchronize(stream));
        if (S.pool[k]) hipFree(S.pool[k]);
        S.pool[k] = nullptr; S.pool_cap[k] = 0;
        if (hipMalloc(&S.pool[k], (size_t)want[k] * DE_V6_POOL_ENTRY_BYTES + 64u) != hipSuccess) { (void)hipGetLastError(); S.pool[k] = nullptr; return fail(DE_ERR_NOMEM, "no room for the tail pool of a render_kernel_v6 launch slot"); }
        S.pool_cap[k] = want[k];
    }
    if (S.cold && S.ctl && S.d_args && S.h_status && S.n_wg == n_wg) return DE_OK;      // all four or none: a slot whose allocation failed half way is completed (or fails again) here
    HIP_TRY(hipStreamSynchronize(stream));
    if (S.cold) hipFree(S.cold);
    S.cold = nullptr; S.n_wg = 0;
    if (hipMalloc(&S.cold, (size_t)n_wg * DE_V6_P * sizeof(wf::Cold)) != hipSuccess) { (void)hipGetLastError(); S.cold = nullptr; return fail(DE_ERR_NOMEM, "no room for the cold records of a render_kernel_v6 launch slot (37 MB)"); }
    S.n_wg = n_wg;
    if (!S.ctl) HIP_TRY(hipMalloc(&S.ctl, (size_t)bs::G_WORDS * DE_V6_CTL_STRIDE * sizeof(uint32_t)));
    if (!S.d_args) HIP_TRY(hipMalloc(&S.d_args, sizeof(RenderArgs)));
    if (!S.h_status) {
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&S.h_status), 64, hipHostMallocMapped));
        S.h_status[0] = 0u;
    }
    return DE_OK;
}
}  // namespace
hipError_t de_fast_launch_v6(const bs::Args& A, const FrameConsts* fc, dim3 grid, dim3 block, hipStream_t stream, bool clamp, bool tail);      // de_fast.hip
namespace {
hipError_t launch_v6(de_ctx* c, const RenderArgs& a, hipStream_t stream, int idx) {
    de_ctx::V6State& S = c->v6s[idx];
    const bool clamp = (c->p.flags & DE_FLAG_CLAMP_SAMPLER) != 0;
    bs::Args A;
    A.ra = S.d_args; A.cold = S.cold; A.ctl = S.ctl;
    uint32_t* d_status = nullptr;
    hipError_t e = hipHostGetDevicePointer(reinterpret_cast<void**>(&d_status), S.h_status, 0);
    if (e != hipSuccess) return e;
    A.status = d_status;
    A.n_items = (uint32_t)((size_t)a.n_tiles * 64u * (size_t)a.spp_count);
    A.idle_limit = 1u << 22;      // ~10 s of idling (an idle round ends in a sleep of up to 2.7 us: render_kernel_v6.hip schedule())
    for (int k = 0; k < 3; ++k) { A.svc[k] = (uint32_t)c->v6_svc[k]; A.svc_area[k] = (uint32_t)c->v6_svc_area[k]; }
    A.yield_max = (uint32_t)c->v6_yield; A.elsewhere_min = (uint32_t)c->v6_elsewhere; A.retry = (uint32_t)c->v6_retry; A.enter_min = (uint32_t)c->v6_enter_min; A.flat_min = (uint32_t)c->v6_flat_min; A.flat_again = (uint32_t)(c->v6_flat_again > 0 ? c->v6_flat_again : 1);
    A.bands = c->v6_bands == 8 ? 8u : 1u;
    hipLaunchKernelGGL(v6_init_kernel, dim3(1), dim3(256), 0, stream, S.ctl, a, S.d_args);
    // DE_V6_WG_PER_CU workgroups per CU; a small launch: one per wave-load of new paths per wave (a workgroup's waves start 64 at a time)
    const unsigned by_work = (unsigned)((A.n_items + 64u * DE_V6_WAVES - 1u) / (64u * DE_V6_WAVES));
    dim3 grid(by_work < S.n_wg ? (by_work ? by_work : 1u) : S.n_wg), block(64 * DE_V6_WAVES);
    // The frame's chain (render_kernel_v6.hip: "The tail"): a launch that fills the GPU exports its workgroups' last paths instead of waiting for
    // them, and up to two small launches of the same kernel — on this stream: the pool is complete when they start — finish them.
    const int levels = A.n_items >= c->v6_tail_min_paths ? c->v6_tail_levels : 0;
    if (!c->h_issued) {
        e = hipHostMalloc(reinterpret_cast<void**>(&c->h_issued), 64, hipHostMallocMapped);
        if (e != hipSuccess) return e;
        c->h_issued[0] = 0u;
    }
    uint32_t* d_issued = nullptr;
    e = hipHostGetDevicePointer(reinterpret_cast<void**>(&d_issued), c->h_issued, 0);
    if (e != hipSuccess) return e;
    // This launch's number.  The word the kernels read holds the number of the newest launch that HAS A SUCCESSOR ON ANOTHER STREAM: written here, when that
    // successor is issued (a successor on the same launch slot waits for the whole chain anyway: exporting would only add the tail launch's latency)
    // (the slot of the newest render_kernel_v6 launch, not of the newest launch of any kernel: a state-machine call in between is no successor — ADVICE r5)
    const uint32_t seq = ++c->v6_launch_seq;
    if (c->last_v6_slot >= 0 && c->last_v6_slot != idx) __atomic_store_n(c->h_issued, seq - 1u, __ATOMIC_RELEASE);
    c->last_v6_slot = idx;
    A.issued = d_issued; A.seq = c->v6_tail_when_alone ? 0xffffffffu : seq;      // (4 Mi paths by default: launches that fill the GPU several times over)
    const bool fast = (c->p.flags & DE_FLAG_FAST_MATH) != 0u;
    for (int l = 0; l <= levels; ++l) {
        A.level = (uint32_t)l;
        A.export_at = l < levels ? (uint32_t)c->v6_tail_export[l] : 0u;
        A.pool_out = l < levels ? S.pool[l & 1] : nullptr; A.cap_out = l < levels ? S.pool_cap[l & 1] : 0u;
        A.pool_in = l > 0 ? S.pool[(l - 1) & 1] : nullptr; A.cap_in = l > 0 ? S.pool_cap[(l - 1) & 1] : 0u;
        const dim3 g = l == 0 ? grid : dim3(std::min((unsigned)c->v6_tail_grid[l - 1], (unsigned)S.n_wg));      // the cold records are per workgroup slot: at most n_wg
        if (fast) { hipError_t e2 = de_fast_launch_v6(A, a.fc, g, block, stream, clamp, l > 0); if (e2 != hipSuccess) return e2; continue; }      // de_fast.hip: the same kernel on the hardware's transcendentals (opt-in, not bit-reproducible)
        if (l == 0) {
            if (c->v6_stats && !clamp) hipLaunchKernelGGL((v6_kernel<false, true, false>), g, block, 0, stream, A, a.fc);
            else if (clamp) hipLaunchKernelGGL((v6_kernel<true, false, false>), g, block, 0, stream, A, a.fc);
            else hipLaunchKernelGGL((v6_kernel<false, false, false>), g, block, 0, stream, A, a.fc);
        } else {
            if (c->v6_stats && !clamp) hipLaunchKernelGGL((v6_kernel<false, true, true>), g, block, 0, stream, A, a.fc);
            else if (clamp) hipLaunchKernelGGL((v6_kernel<true, false, true>), g, block, 0, stream, A, a.fc);
            else hipLaunchKernelGGL((v6_kernel<false, false, true>), g, block, 0, stream, A, a.fc);
        }
    }
    return hipGetLastError();
}
// Did a persistent launch leave on its abort word?  The word is host-visible pinned memory written by the kernel when it gives up; what has
// arrived is folded into the context's sticky flag (and the word re-armed for the next launch).  Sticky: the HDR buffer of this frame is
// missing paths until de_reset, however often it is fetched.
void note_abort(de_ctx* c) {
    for (int i = 0; i < DE_MAX_SLOTS; ++i) {
        de_ctx::V6State& S = c->v6s[i];
        if (S.h_status && S.h_status[0] != 0u) {
            const uint32_t code = S.h_status[0];
            S.h_status[0] = 0u;
            c->frame_invalid = true;
            c->invalid_msg = std::string("render_kernel_v6 left on its abort word (") + (code == bs::ERR_ENTRY_TIMEOUT ? "a ring entry was never stored" : code == bs::ERR_CELL_BUSY_TIMEOUT ? "a ring cell was never emptied" : code == bs::ERR_POOL_FULL ? "the tail pool overflowed" : "idle watchdog: work was lost") + "): the frame is invalid until de_reset";
        }
    }
#ifdef DE_LEGACY_VARIANTS
    legacy_note_abort(c);
#endif
}
int frame_status(de_ctx* c) {
    note_abort(c);
    return c->frame_invalid ? fail(DE_ERR_HIP, c->invalid_msg) : DE_OK;
}

// One launch on `stream`.  v2 (the default path tracer): render_kernel_v2 writes the per-sample records, then — after `pre_acc`
// has made the stream wait for the previous launch's accumulate_kernel and for context-stream work on the HDR buffer — accumulate_kernel
// adds them to the HDR buffer.  The other kernels read-modify-write the HDR buffer themselves: `pre_acc` runs before them.
template <int MODE, typename PreAcc>
hipError_t launch_render(de_ctx* c, const RenderArgs& a, hipStream_t stream, PreAcc pre_acc) {
    const bool clamp = (c->p.flags & DE_FLAG_CLAMP_SAMPLER) != 0;
    const bool march = (c->p.flags & DE_FLAG_RAY_MARCHER) != 0;
    dim3 grid((unsigned)((a.n_tiles + 3) / 4)), block(256);
#ifdef DE_LEGACY_VARIANTS
    if (!march && (c->launch_variant == 5 || c->launch_variant == 3) && MODE == 0) {
        hipError_t el = c->launch_variant == 5 ? launch_v5(c, a, stream, c->launch_slot) : launch_v3(c, a, stream, c->launch_pipe);
        if (el != hipSuccess) return el;
        hipError_t pe = pre_acc();
        if (pe != hipSuccess) return pe;
        hipLaunchKernelGGL(accumulate_kernel, grid, block, 0, stream, a);
        return hipGetLastError();
    }
#endif
    if (!march && c->launch_variant == 6 && MODE == 0) {
        hipError_t e6 = launch_v6(c, a, stream, c->launch_slot);
        if (e6 != hipSuccess) return e6;
        hipError_t pe = pre_acc();
        if (pe != hipSuccess) return pe;
        hipLaunchKernelGGL(accumulate_kernel, grid, block, 0, stream, a);
        return hipGetLastError();
    }
    if (!march && c->launch_variant != 1) {
        // persistent waves pulling pixels from a device-wide counter: size the grid to the machine, not to the image
        hipError_t me = hipMemsetAsync(a.work_counter, 0, sizeof(uint32_t), stream);
        if (me != hipSuccess) return me;
        // waves: as many as the CUs hold at 5 waves/SIMD (c->tune_wpc = 20 per CU), fewer only when the launch has less than one
        // chunk of work items per wave — a rank of an 8-way tile partition owns 4050 tiles but 16.6 M work items at 64 spp,
        // and still fills all 5120 wave slots.  (More waves would only queue behind the resident ones and find the work
        // counter exhausted; they would also keep the small kernels of the other launches in flight waiting for a slot.)
        const unsigned waves_per_block = DE_V2_BLOCK / 64;
        const unsigned waves_per_cu = (unsigned)c->tune_wpc;
        const unsigned max_waves = (unsigned)c->n_cus * waves_per_cu;
        const unsigned long long items = (unsigned long long)a.n_tiles * 64ull * (unsigned long long)a.spp_count;
        const unsigned long long by_work = (items + (unsigned long long)a.chunk - 1ull) / (unsigned long long)a.chunk;
        const unsigned n_waves = by_work >= (unsigned long long)max_waves ? max_waves : (unsigned)(by_work > 0 ? by_work : 1ull);
        dim3 pgrid((n_waves + waves_per_block - 1) / waves_per_block), pblock(DE_V2_BLOCK);
        if (clamp) hipLaunchKernelGGL((render_kernel_v2<true, MODE>), pgrid, pblock, 0, stream, a, a.fc);
        else hipLaunchKernelGGL((render_kernel_v2<false, MODE>), pgrid, pblock, 0, stream, a, a.fc);
        if (MODE != 2) {
            hipError_t pe = pre_acc();
            if (pe != hipSuccess) return pe;
            hipLaunchKernelGGL(accumulate_kernel, grid, block, 0, stream, a);
        }
        return hipGetLastError();
    }
    hipError_t pe = pre_acc();
    if (pe != hipSuccess) return pe;
    if (march) {
        if (clamp) hipLaunchKernelGGL((ray_march_kernel<true, MODE>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((ray_march_kernel<false, MODE>), grid, block, 0, stream, a);
    }
#ifdef DE_LEGACY_VARIANTS
    else if (clamp) hipLaunchKernelGGL((render_kernel<true, false, MODE>), grid, block, 0, stream, a);      // kernel variant 1: per-lane loops
    else hipLaunchKernelGGL((render_kernel<false, false, MODE>), grid, block, 0, stream, a);
#endif
    return hipGetLastError();
}

// device (W*H*3 floats) -> caller's buffer through the pinned staging buffer
int copy_out(de_ctx* c, float* out, const float* d_src) {
    const size_t bytes = (size_t)c->W * c->H * 3 * sizeof(float);
    if (!c->h_stage && hipHostMalloc((void**)&c->h_stage, bytes, hipHostMallocDefault) != hipSuccess) c->h_stage = nullptr;
    if (c->h_stage) {
        HIP_TRY(hipMemcpyAsync(c->h_stage, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        memcpy(out, c->h_stage, bytes);
    } else {
        HIP_TRY(hipMemcpyAsync(out, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return frame_status(c);
}

// The launches of one de_accumulate call on the product's kernels.  `spp` = the samples per pixel THIS context renders of the call (all of
// them, or its share of a sample partition), `first_index` = the frame's sample index of the first one (the others follow at a.spp_stride).
//   * render_kernel_v6 (one persistent launch per call, stage queues in LDS) for calls of at least auto_v6_min_items paths, or when forced;
//   * render_kernel_v2 (wave-level state machine) for smaller calls and for counting launches;
//   * ray_march_kernel under DE_FLAG_RAY_MARCHER (read-modify-writes the HDR buffer itself).
// A launch handles n_tiles * 64 * spp work items: the item index must fit 32 bits, the tile index is recovered with a 2^40 magic
// multiplication (exact while n_tiles * spp^2 < 2^40), and each item owns 8 bytes of the contribution buffer (kept <= 4 GB).  Larger calls are
// split here; the sample index continues from launch to launch and the per-pixel sum keeps sample order, so split launches give the same bits.
// Launch k goes to launch slot k mod n (its own stream, work counter, contribution buffer): consecutive launches overlap, the
// accumulate_kernels are chained with events in launch order.
int accumulate_default(de_ctx* c, RenderArgs& a, int spp, int first_index) {
    const bool march = (c->p.flags & DE_FLAG_RAY_MARCHER) != 0;
    const bool records = !march;
    const unsigned long long call_items = (unsigned long long)a.n_tiles * 64ull * (unsigned long long)(spp > 0 ? spp : 0);
    const bool fast = (c->p.flags & DE_FLAG_FAST_MATH) != 0u;      // one kernel carries the fast arithmetic: every call runs it, whatever its size
    const bool v6k = records && !c->count && (fast || c->kernel_variant == 6 || (c->kernel_variant == 4 && call_items >= c->auto_v6_min_items));
    int max_spp = spp > 0 ? spp : 1;
    if (records && a.n_tiles > 0) {
        const unsigned long long nt = (unsigned long long)a.n_tiles;
        unsigned long long cap = ((1ull << 29) / (nt * 64ull));                          // 2^29 items = 4 GB of contributions
        while (cap > 1 && nt * cap * cap >= (1ull << 40)) --cap;
        if (cap < 1) cap = 1;
        if (c->tune_max_spp > 0 && cap > (unsigned long long)c->tune_max_spp) cap = (unsigned long long)c->tune_max_spp;
        if ((unsigned long long)max_spp > cap) max_spp = (int)cap;
        if (max_spp < spp) { const int nl = (spp + max_spp - 1) / max_spp; max_spp = (spp + nl - 1) / nl; }      // launches of equal size: 64 spp under a cap of 15 is 5 x 13, not 4 x 15 + 4
        if (nt * 64ull * (unsigned long long)max_spp >= (1ull << 32) || nt * (unsigned long long)max_spp * (unsigned long long)max_spp >= (1ull << 40))
            return fail(DE_ERR_INVALID, "image too large for the 32-bit work-item index of a launch");
    }
    // What the context stream has done since the last launch and a launch must be ordered after (set-up kernels, map packing,
    // the tile list: before the render kernel; reset / display / reduce / uploads of the HDR buffer: before accumulate_kernel).
    if (c->rec_render != c->gen_render || c->rec_hdr != c->gen_hdr) {
        HIP_TRY(hipEventRecord(c->ev_main, c->stream));
        c->rec_render = c->gen_render; c->rec_hdr = c->gen_hdr;
    }
    bool first = true;
    c->last_call[0] = 0; c->last_call[1] = 1; c->last_call[2] = 0; c->last_call[3] = 0;
    c->timing_empty = true;       // a call that launches nothing (an empty share of a sample partition) has no duration: de_last_accumulate_ms says so instead of repeating the previous call's
    for (int done = 0; done < spp && a.n_tiles > 0;) {
        const int n = (spp - done < max_spp) ? spp - done : max_spp;
        // small launches (the interactive 1-spp pattern) use all the slots; big ones only the first few — each slot keeps a record buffer of
        // the largest launch it has seen, and more than three big persistent kernels side by side only dilute each other
        const size_t need = records ? (size_t)a.n_tiles * 64u * (size_t)n : 0;
        const int n_want = (need * sizeof(uint2) > ((size_t)96 << 20)) ? (c->big_slots < c->n_slots ? c->big_slots : c->n_slots) : c->n_slots;
        const int n_cycle = n_want < c->n_slots ? n_want : c->n_slots;      // never a slot the context stream does not join
        const int si = c->next_slot % n_cycle;
        c->launch_slot = si;
        LaunchSlot& s = c->slot[si];
        if (v6k) { int rc6 = v6_reserve(c, si, s.stream, (unsigned long long)a.n_tiles * 64ull * (unsigned long long)n); if (rc6) return rc6; }
        c->launch_variant = v6k ? 6 : 2;
        if (need > s.contrib_items) {
            // grow the record buffers of EVERY slot of this launch's cycle now: the first launch of a new size pays for all the
            // allocations, none falls between later launches (hipFree / hipMalloc stall the streams)
            for (int j = 0; j < n_cycle; ++j) {
                LaunchSlot& sj = c->slot[j];
                if (need <= sj.contrib_items) continue;
                HIP_TRY(hipStreamSynchronize(sj.stream));      // the slot's previous launch still reads the old buffer
                if (sj.contrib) hipFree(sj.contrib);
                sj.contrib = nullptr; sj.contrib_items = 0;
                if (hipMalloc(&sj.contrib, need * sizeof(uint2)) != hipSuccess) { (void)hipGetLastError(); return fail(DE_ERR_NOMEM, "no room for the per-sample records of a launch (8 bytes per path)"); }
                sj.contrib_items = need;
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
        c->last_call[0] = march ? 0 : c->launch_variant;
        c->last_call[3]++;
        HIP_TRY(hipEventRecord(s.done, s.stream));
        s.pending = true; s.launched = true;
        c->last_slot = si;
        c->next_slot = (si + 1) % n_cycle;
        done += n;
        if (done >= spp) { HIP_TRY(hipEventRecord(s.t1, s.stream)); c->t1_slot = si; c->timing_valid = true; c->timing_empty = false; }
    }
    return DE_OK;
}

}  // namespace

#ifdef DE_LEGACY_VARIANTS
#include "legacy/de_launch_legacy.h"
#endif
