// de_api.hip — the C ABI of libdigitalearth_hip.so (include/digital_earth.h): context life cycle, uploads, parameters, the frame loop, fetches.
// No CPU fallback exists: every entry point needs a HIP device.  The launches are in de_launch.h, the collectives in de_rccl.h, the context in
// de_context.h; the kernel families the product no longer runs hang in under -DDE_LEGACY_VARIANTS (legacy/).
#include "de_rccl.h"

namespace {
// A launch slot's stream.  withhold > 0: the stream may use every CU but the LAST `withhold` of each XCD (hipExtStreamCreateWithCUMask; bit i of the mask is
// CU i / 8 of XCD i % 8 — the driver deals the mask's bits round robin over the XCDs — so the top 8 x withhold bits are `withhold` CUs of every XCD):
// persistent workgroups then leave those CUs to the small kernels of the context stream (collective, accumulate, display).
hipError_t create_slot_stream(de_ctx* c, hipStream_t* out) {
    if (c->cu_withhold <= 0) return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
    const int n = c->n_cus, keep = n - 8 * c->cu_withhold;
    if (keep < 8) return hipErrorInvalidValue;
    uint32_t mask[16];
    memset(mask, 0, sizeof(mask));
    for (int i = 0; i < keep && i < 512; ++i) mask[i >> 5] |= 1u << (i & 31);
    return hipExtStreamCreateWithCUMask(out, (uint32_t)((n + 31) / 32), mask);
}
}  // namespace

extern "C" {

const char* de_last_error(void) { return g_err.c_str(); }
int de_abi_version(void) { return DE_ABI_VERSION; }
int de_arithmetic_contract(void) { return DE_ARITHMETIC_CONTRACT; }

int de_create(int device, int width, int height, de_ctx** out) {
    if (!out) return fail(DE_ERR_INVALID, "out is null");
    if (width <= 0 || height <= 0 || width % 16 || height % 8)
        return fail(DE_ERR_INVALID, "image size must be a positive multiple of (16, 8) (renderer.py:46)");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(DE_ERR_NO_DEVICE, "no HIP device: libdigitalearth_hip has no CPU path");
    if (device < 0 || device >= n_dev) return fail(DE_ERR_INVALID, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(DE_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code only");
    de_ctx* c = new de_ctx();
    c->device = device; c->W = width; c->H = height;
    default_params(&c->p);
#ifdef DE_LEGACY_VARIANTS
    read_legacy_env(c);       // the legacy library keeps the experiment knobs of rounds 1-4 in the environment; the product reads none (de_set_tuning)
#endif
    memset(&c->counters, 0, sizeof(c->counters));
    c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    size_t npx = (size_t)width * height;
    // the context stream carries the small operations between frames and the collective: highest priority, so that wave slots
    // freed by the draining render kernels of the launch slots go to them first
    int prio_lo = 0, prio_hi = 0;
    hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    hipError_t e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_hi);
    c->own_stream = (e == hipSuccess);
    {   // launch slots: launches in flight per context (de_set_launch_slots; 1 = every launch waits for the previous one)
        for (int i = 0; i < c->n_slots && e == hipSuccess; ++i) {
            LaunchSlot& s = c->slot[i];
            e = create_slot_stream(c, &s.stream);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreate(&s.t0);
            if (e == hipSuccess) e = hipEventCreate(&s.t1);
        }
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_main, hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipMalloc(&c->d_hdr_own, npx * 3 * sizeof(float));
    c->d_hdr = c->d_hdr_own;
    if (e == hipSuccess) e = hipMalloc(&c->d_image, npx * 3 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&c->d_scratch, npx * 4 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&c->d_fc, sizeof(FrameConsts));
    if (e == hipSuccess) e = hipMalloc(&c->d_nodes, DE_N_NODES * sizeof(LambdaNode));
    if (e == hipSuccess) e = hipMalloc(&c->d_node_val, 256 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&c->d_counters, DE_N_COUNTERS * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc(&c->d_work_counter, 16 * (DE_MAX_SLOTS + 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&c->d_dens_table, (size_t)DE_DENS_TABLE_N * DE_DENS_STRIDE * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&c->d_cie, 441 * 2 * 3 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&c->d_srgb2spec, 900 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&c->d_o3, 441 * sizeof(float));
    if (e == hipSuccess) e = hipMemsetAsync(c->d_hdr, 0, npx * 3 * sizeof(float), c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->d_counters, 0, DE_N_COUNTERS * sizeof(unsigned long long), c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(dens_table_kernel, dim3((DE_DENS_TABLE_N + 255) / 256), dim3(256), 0, c->stream, c->d_dens_table);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        std::string msg = std::string("de_create: ") + hipGetErrorString(e);
        de_destroy(c);
        return fail(e == hipErrorOutOfMemory ? DE_ERR_NOMEM : DE_ERR_HIP, msg);
    }
    *out = c;
    return DE_OK;
}

int de_destroy(de_ctx* c) {
    if (!c) return DE_OK;
    if (c->loans > 0) return fail(DE_ERR_STATE, "this context lends its maps to another one (de_share_textures): destroy the borrowers first");
    hipSetDevice(c->device);
    for (int i = 0; i < DE_MAX_SLOTS; ++i) if (c->slot[i].stream) hipStreamSynchronize(c->slot[i].stream);
    hipStreamSynchronize(c->stream);
    if (c->comm && g_rccl.CommDestroy) { g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
    release_loan(c);
    for (auto& t : c->tex) { if (t.borrowed) continue; if (t.linear) hipFree(t.linear); if (t.packed) hipFree(t.packed); }
    if (c->luts_borrowed) { c->d_cie = nullptr; c->d_srgb2spec = nullptr; c->d_o3 = nullptr; c->d_crf = nullptr; }
    void* ptrs[] = {c->d_cie, c->d_srgb2spec, c->d_o3, c->d_crf, c->d_fc, c->d_nodes, c->d_node_val, c->d_hdr_own, c->d_image, c->d_scratch, c->d_tiles, c->d_counters, c->d_work_counter, c->d_dens_table, c->d_assembled, c->d_gather, c->d_standin};
    for (void* p : ptrs) if (p) hipFree(p);
    if (c->h_stage) hipHostFree(c->h_stage);
    for (int k = 0; k < DE_FETCH_RING; ++k) { if (c->h_ring[k]) hipHostFree(c->h_ring[k]); if (c->ev_ring[k]) hipEventDestroy(c->ev_ring[k]); }
    if (c->h_issued) hipHostFree(c->h_issued);
#ifdef DE_LEGACY_VARIANTS
    legacy_destroy(c);
#endif
    for (auto& S : c->v6s) {
        if (S.cold) hipFree(S.cold); if (S.ctl) hipFree(S.ctl); if (S.d_args) hipFree(S.d_args); if (S.pool[0]) hipFree(S.pool[0]); if (S.pool[1]) hipFree(S.pool[1]);
        if (S.h_status) hipHostFree(S.h_status);
    }
    for (int i = 0; i < DE_MAX_SLOTS; ++i) {
        LaunchSlot& s = c->slot[i];
        if (s.contrib) hipFree(s.contrib);
        if (s.done) hipEventDestroy(s.done);
        if (s.t0) hipEventDestroy(s.t0);
        if (s.t1) hipEventDestroy(s.t1);
        if (s.stream) hipStreamDestroy(s.stream);
    }
    if (c->ev_main) hipEventDestroy(c->ev_main);
    if (c->ev_r0) hipEventDestroy(c->ev_r0);
    if (c->ev_r1) hipEventDestroy(c->ev_r1);
    for (auto& pair : c->ev_standin) for (hipEvent_t ev : pair) if (ev) hipEventDestroy(ev);
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    delete c;
    return DE_OK;
}

int de_upload_texture(de_ctx* c, int slot, const uint8_t* texels, int w, int h, int channels) {
    if (!c || slot < 0 || slot >= DE_TEX_COUNT || !texels || w <= 0 || h <= 0) return fail(DE_ERR_INVALID, "bad texture arguments");
    const bool colour = (slot == DE_TEX_ALBEDO || slot == DE_TEX_STARS);
    if (channels != (colour ? 3 : 1)) return fail(DE_ERR_INVALID, "albedo/stars take 3 channels, the grey maps 1");
    HIP_TRY(hipSetDevice(c->device));
    int rc = alloc_texture(c, slot, w, h, channels);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c->tex[slot].linear, texels, (size_t)w * h * channels, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->tex[slot].set = true;
    if (slot == DE_TEX_TOPOGRAPHY) c->params_dirty = true;
    return DE_OK;
}

int de_generate_texture(de_ctx* c, int slot, int w, int h, uint32_t seed, int variant) {
    if (!c || slot < 0 || slot >= DE_TEX_COUNT || w <= 0 || h <= 0) return fail(DE_ERR_INVALID, "bad texture arguments");
    HIP_TRY(hipSetDevice(c->device));
    const bool colour = (slot == DE_TEX_ALBEDO || slot == DE_TEX_STARS);
    int rc = alloc_texture(c, slot, w, h, colour ? 3 : 1);
    if (rc) return rc;
    dim3 grid((unsigned)((w + 255) / 256), (unsigned)h);
    hipLaunchKernelGGL(synth_kernel, grid, dim3(256), 0, c->stream, c->tex[slot].linear, slot, w, h, seed, variant);
    HIP_TRY(hipGetLastError());
    c->tex[slot].set = true;
    if (slot == DE_TEX_TOPOGRAPHY) c->params_dirty = true;
    return DE_OK;
}

int de_share_textures(de_ctx* dst, de_ctx* src) {
    if (!dst || !src || dst == src) return fail(DE_ERR_INVALID, "two different contexts are needed");
    if (dst->device != src->device) return fail(DE_ERR_INVALID, "contexts on different devices cannot share maps");
    if (src->lender) return fail(DE_ERR_STATE, "the lending context borrows its maps itself: share from their owner");
    if (dst->loans > 0) return fail(DE_ERR_STATE, "the borrowing context lends its own maps to another one");
    for (int i = 0; i < DE_TEX_COUNT; ++i)
        if (!src->tex[i].set) return fail(DE_ERR_STATE, "the lending context must hold all 7 maps");
    if (!src->luts_set) return fail(DE_ERR_STATE, "the lending context must hold the LUTs");
    HIP_TRY(hipSetDevice(src->device));
    // bring the lender's packed copies up to date for ITS address mode, then wait: the borrower reads them from other streams
    const bool clamp = (src->p.flags & DE_FLAG_CLAMP_SAMPLER) != 0;
    for (int i = 0; i < DE_TEX_COUNT; ++i) { int rc = ensure_packed(src, i, clamp); if (rc) return rc; }
    HIP_TRY(hipStreamSynchronize(src->stream));
    { int rc = sync_all(dst); if (rc) return rc; }
    release_loan(dst);
    for (int i = 0; i < DE_TEX_COUNT; ++i) {
        DevTexture& t = dst->tex[i];
        if (!t.borrowed) { if (t.linear) hipFree(t.linear); if (t.packed) hipFree(t.packed); }
        t = src->tex[i];
        t.linear = nullptr;       // the as-uploaded copy stays the lender's alone: de_download_texture on the borrower fails cleanly
        t.borrowed = true;
    }
    if (!dst->luts_borrowed) {
        void* own[] = {dst->d_cie, dst->d_srgb2spec, dst->d_o3, dst->d_crf};
        for (void* q : own) if (q) hipFree(q);
    }
    dst->d_cie = src->d_cie; dst->d_srgb2spec = src->d_srgb2spec; dst->d_o3 = src->d_o3; dst->d_crf = src->d_crf;
    dst->n_crf = src->n_crf; dst->luts_set = true; dst->luts_borrowed = true;
    dst->params_dirty = true; dst->nodes_dirty = true;
    // the loan is on record: while it lasts the lender refuses to free, replace or repack its maps and LUTs, and to be destroyed
    dst->lender = src; src->loans++;
    touched_render_inputs(dst);
    return DE_OK;
}

int de_trim_textures(de_ctx* c) {
    if (!c) return fail(DE_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    const bool clamp = (c->p.flags & DE_FLAG_CLAMP_SAMPLER) != 0;
    for (int i = 0; i < DE_TEX_COUNT; ++i) {
        DevTexture& t = c->tex[i];
        if (!t.set || t.borrowed || !t.linear) continue;
        int rc = ensure_packed(c, i, clamp);          // the packed copy must exist before its source goes (fails if that means repacking lent maps)
        if (rc) return rc;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int i = 0; i < DE_TEX_COUNT; ++i) {
        DevTexture& t = c->tex[i];
        if (!t.set || t.borrowed || !t.linear) continue;
        hipFree(t.linear);            // borrowers never hold this pointer (de_share_textures)
        t.linear = nullptr;
    }
    return DE_OK;
}

int de_download_texture(de_ctx* c, int slot, uint8_t* out, uint64_t out_bytes) {
    if (!c || slot < 0 || slot >= DE_TEX_COUNT || !out || !c->tex[slot].set) return fail(DE_ERR_INVALID, "texture not set");
    const DevTexture& t = c->tex[slot];
    if (t.borrowed) return fail(DE_ERR_STATE, "this map is borrowed (de_share_textures): download it from the context that owns it");
    if (!t.linear) return fail(DE_ERR_STATE, "the as-uploaded copy of this map was released (de_trim_textures)");
    size_t n = (size_t)t.w * t.h * t.ch;
    if (out_bytes < n) return fail(DE_ERR_INVALID, "output buffer too small");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(out, t.linear, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return DE_OK;
}

int de_texture_info(de_ctx* c, int slot, int* w, int* h, int* ch) {
    if (!c || slot < 0 || slot >= DE_TEX_COUNT) return fail(DE_ERR_INVALID, "bad slot");
    if (w) *w = c->tex[slot].w;
    if (h) *h = c->tex[slot].h;
    if (ch) *ch = c->tex[slot].ch;
    return DE_OK;
}

int de_upload_luts(de_ctx* c, const float* cie, const uint16_t* srgb2spec_f16, const float* o3, const float* crf, int n_crf) {
    if (!c || !cie || !srgb2spec_f16 || !o3 || !crf || n_crf <= 0) return fail(DE_ERR_INVALID, "bad LUT arguments");
    if (c->loans > 0) return fail(DE_ERR_STATE, "the LUTs are lent to another context (de_share_textures): destroy the borrowers first");
    HIP_TRY(hipSetDevice(c->device));
    { int rc = sync_all(c); if (rc) return rc; }
    touched_render_inputs(c);
    std::vector<float> q(441 * 2 * 3), s(900), r((size_t)1024 * n_crf * 3);
    for (int i = 0; i < 441 * 2 * 3; ++i) q[i] = quantize_f16(cie[i]);
    for (int i = 0; i < 900; ++i) s[i] = half_to_float(srgb2spec_f16[i]);
    for (int x = 0; x < 1024; ++x)
        for (int y = 0; y < n_crf; ++y)
            for (int ch = 0; ch < 3; ++ch) r[((size_t)y * 1024 + x) * 3 + ch] = crf[((size_t)x * n_crf + y) * 3 + ch];
    if (c->luts_borrowed) {                  // stop borrowing: own copies again
        c->d_cie = nullptr; c->d_srgb2spec = nullptr; c->d_o3 = nullptr; c->d_crf = nullptr; c->luts_borrowed = false;
        {   bool any = false;
            for (int i = 0; i < DE_TEX_COUNT; ++i) any = any || c->tex[i].borrowed;
            if (!any) release_loan(c); }
        HIP_TRY(hipMalloc(&c->d_cie, 441 * 2 * 3 * sizeof(float)));
        HIP_TRY(hipMalloc(&c->d_srgb2spec, 900 * sizeof(float)));
        HIP_TRY(hipMalloc(&c->d_o3, 441 * sizeof(float)));
    }
    if (c->d_crf) { hipFree(c->d_crf); c->d_crf = nullptr; }
    HIP_TRY(hipMalloc(&c->d_crf, r.size() * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(c->d_cie, q.data(), q.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_srgb2spec, s.data(), s.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_o3, o3, 441 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_crf, r.data(), r.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->n_crf = n_crf; c->luts_set = true; c->params_dirty = true; c->nodes_dirty = true;
    return DE_OK;
}

int de_set_params(de_ctx* c, const de_params* p) {
    if (!c || !p) return fail(DE_ERR_INVALID, "null argument");
    if (p->flags != c->p.flags || memcmp(&p->fixed_wavelength, &c->p.fixed_wavelength, sizeof(float)) != 0) c->nodes_dirty = true;
    c->p = *p;
    c->params_dirty = true;
    return DE_OK;
}
int de_get_params(de_ctx* c, de_params* p) {
    if (!c || !p) return fail(DE_ERR_INVALID, "null argument");
    *p = c->p;
    return DE_OK;
}

int de_reset(de_ctx* c) {
    if (!c) return fail(DE_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    { int rc = join_slots(c); if (rc) return rc; }
    touched_hdr(c);
    HIP_TRY(hipMemsetAsync(c->d_hdr, 0, (size_t)c->W * c->H * 3 * sizeof(float), c->stream));
    HIP_TRY(hipMemsetAsync(c->d_counters, 0, DE_N_COUNTERS * sizeof(unsigned long long), c->stream));
    memset(&c->counters, 0, sizeof(c->counters));
    c->current_spp = 0;
    note_abort(c);                         // re-arm the abort words ...
    c->frame_invalid = false;              // ... a new frame starts
    if (c->display_src == c->d_assembled) c->display_src = nullptr;    // the assembled frame of a progressive reduce is history now
    return DE_OK;
}

int de_accumulate(de_ctx* c, int spp, uint64_t seed, int tile_rank, int tile_world) {
    if (!c || spp < 0 || tile_world < 1 || tile_rank < 0 || tile_rank >= tile_world) return fail(DE_ERR_INVALID, "bad accumulate arguments");
    HIP_TRY(hipSetDevice(c->device));
    int rc = build_tiles(c, tile_rank, tile_world);
    if (rc) return rc;
    RenderArgs a;
    rc = fill_render_args(c, &a);
    if (rc) return rc;
    a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32);
    // Sample partition (renderer.py:371-380 dealt round robin): the call covers the frame's sample indices
    // [current_spp, current_spp + spp); this context renders those = sample_rank (mod sample_world) — first_index, first_index + world, ... —
    // and the frame's sample counter advances by the whole spp on every rank.
    const int frame_spp = spp;
    int first_index = c->current_spp;
    if (c->sample_world > 1) {
        const int rem = first_index % c->sample_world;
        first_index += (c->sample_rank - rem + c->sample_world) % c->sample_world;
        spp = first_index < c->current_spp + frame_spp ? (c->current_spp + frame_spp - first_index + c->sample_world - 1) / c->sample_world : 0;
    }
    a.spp_stride = c->sample_world;
#ifdef DE_LEGACY_VARIANTS
    rc = accumulate_legacy(c, a, spp, first_index);
#else
    rc = accumulate_default(c, a, spp, first_index);
#endif
    if (rc) return rc;
    c->current_spp += frame_spp;
    return DE_OK;
}

int de_set_sample_partition(de_ctx* c, int rank, int world) {
    if (!c || world < 1 || rank < 0 || rank >= world) return fail(DE_ERR_INVALID, "sample partition: 0 <= rank < world");
    c->sample_rank = rank; c->sample_world = world;      // read by the next de_accumulate; launches in flight keep what they were issued with
    return DE_OK;
}

int de_set_memory_budget(de_ctx* c, uint64_t bytes) {
    if (!c) return fail(DE_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    { int rc = sync_all(c); if (rc) return rc; }
#ifdef DE_LEGACY_VARIANTS
    free_queue_memory(c);                 // what is held may exceed the new budget: the next large call allocates within it
#endif
    c->mem_budget = (size_t)bytes;        // the product's kernels need 37 MB per launch slot whatever the call: the budget binds the legacy pipeline's queues only
    return DE_OK;
}
int de_get_memory_use(de_ctx* c, uint64_t* queue_bytes) {
    if (!c || !queue_bytes) return fail(DE_ERR_INVALID, "null argument");
    uint64_t n = 0;
#ifdef DE_LEGACY_VARIANTS
    n += legacy_memory_use(c);
#endif
    for (auto& S : c->v6s) if (S.cold) n += (uint64_t)S.n_wg * DE_V6_P * sizeof(wf::Cold) + ((uint64_t)S.pool_cap[0] + S.pool_cap[1]) * DE_V6_POOL_ENTRY_BYTES;      // the per-CU scheduler: 37 MB per launch slot, whatever the call
    *queue_bytes = n;
    return DE_OK;
}

/* Phases of the LAST render_kernel_v6 launch of the last de_accumulate call, from the kernel's own clock (100 MHz): ms[0] = first workgroup start
 * to last wave exit, ms[1] = the DRAIN — from the first wave that found no work item left to the last wave's exit (the launch's last long
 * paths, which no scheduling shortens: what separates a rank's 1/N share of a frame from 1/N of the frame's time).  Waits for that launch.
 * DE_ERR_STATE when the last call did not run render_kernel_v6. */
int de_last_launch_phases(de_ctx* c, float* ms2) {
    if (!c || !ms2) return fail(DE_ERR_INVALID, "null argument");
    if (c->last_call[0] != 6 || c->last_slot < 0) return fail(DE_ERR_STATE, "the last de_accumulate did not run render_kernel_v6");
    HIP_TRY(hipSetDevice(c->device));
    de_ctx::V6State& S = c->v6s[c->last_slot];
    if (!S.ctl) return fail(DE_ERR_STATE, "no launch yet");
    HIP_TRY(hipStreamSynchronize(c->slot[c->last_slot].stream));
    unsigned long long t[3] = {0ull, 0ull, 0ull};
    for (int k = 0; k < 3; ++k) HIP_TRY(hipMemcpy(&t[k], S.ctl + (size_t)(bs::G_T_START + k) * DE_V6_CTL_STRIDE, 8, hipMemcpyDeviceToHost));
    ms2[0] = t[2] > t[0] ? (float)((double)(t[2] - t[0]) * 1e-5) : 0.f;
    ms2[1] = (t[2] > t[1] && t[1] != ~0ull) ? (float)((double)(t[2] - t[1]) * 1e-5) : 0.f;
    return DE_OK;
}

int de_last_call_info(de_ctx* c, int* variant, int* pipes, int* depths, int* launches) {
    if (!c) return fail(DE_ERR_INVALID, "null context");
    if (variant) *variant = c->last_call[0];
    if (pipes) *pipes = c->last_call[1];
    if (depths) *depths = c->last_call[2];
    if (launches) *launches = c->last_call[3];
    return DE_OK;
}

/* Make the context stream wait (on the device) for every launch issued so far.  Needed only by a host framework that enqueues
 * its OWN work on the stream it handed to de_set_stream — e.g. a torch.distributed reduce of the bound HDR tensor; the
 * library's own entry points (fetch, reduce, reset, display ...) do it themselves. */
int de_flush(de_ctx* c) {
    if (!c) return fail(DE_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    int rc = join_slots(c);
    if (rc) return rc;
    touched_hdr(c);          // whatever the host enqueues next on the stream may touch the HDR buffer
    return frame_status(c);  // an abort already known (the words are host-visible): a host framework's own collective must not ship that frame
}

int de_render_to_image(de_ctx* c, const float** device_image) {
    if (!c) return fail(DE_ERR_INVALID, "null context");
    if (!c->luts_set) return fail(DE_ERR_STATE, "LUTs must be uploaded before fetch_image");
    HIP_TRY(hipSetDevice(c->device));
    int rc = run_setup(c);
    if (rc) return rc;
    rc = join_slots(c);
    if (rc) return rc;
    touched_hdr(c);              // the next accumulate_kernel must not overwrite what this reads
    DisplayArgs d;
    d.fc = c->d_fc; d.hdr = c->display_src ? c->display_src : c->d_hdr; d.image = c->d_image;
    d.crf.data = c->d_crf; d.crf.w = 1024; d.crf.h = c->n_crf;
    d.W = c->W; d.H = c->H; d.samples = c->current_spp; d.clamp = (c->p.flags & DE_FLAG_CLAMP_SAMPLER) ? 1 : 0;
    size_t npx = (size_t)c->W * c->H;
    hipLaunchKernelGGL(display_kernel, dim3((unsigned)((c->W + 31) / 32), (unsigned)((c->H + 31) / 32)), dim3(256), 0, c->stream, d);
    HIP_TRY(hipGetLastError());
    // what the next accumulate_kernel must wait for ends HERE (the display has read the HDR buffer): recorded now, not lazily at the next de_accumulate, so that
    // a device-to-host copy of the image enqueued behind the display (de_fetch_image_begin) does not hold the next frame's sums back
    HIP_TRY(hipEventRecord(c->ev_main, c->stream));
    c->rec_render = c->gen_render; c->rec_hdr = c->gen_hdr;
    if (device_image) *device_image = c->d_image;
    return DE_OK;
}

int de_fetch_image(de_ctx* c, float* out) {
    if (!out) return fail(DE_ERR_INVALID, "out is null");
    int rc = de_render_to_image(c, nullptr);
    if (rc) return rc;
    return copy_out(c, out, c->d_image);
}

int de_fetch_image_view(de_ctx* c, const float** host_image) {
    if (!host_image) return fail(DE_ERR_INVALID, "host_image is null");
    int rc = de_render_to_image(c, nullptr);
    if (rc) return rc;
    const size_t bytes = (size_t)c->W * c->H * 3 * sizeof(float);
    if (!c->h_stage && hipHostMalloc((void**)&c->h_stage, bytes, hipHostMallocDefault) != hipSuccess) { c->h_stage = nullptr; (void)hipGetLastError(); return fail(DE_ERR_NOMEM, "no pinned host memory for the staging buffer"); }
    HIP_TRY(hipMemcpyAsync(c->h_stage, c->d_image, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *host_image = c->h_stage;
    return frame_status(c);
}

/* The window loop pipelined (include/digital_earth.h).  begin: the display transform of the frame as it stands and its copy into one of four pinned
 * buffers are ENQUEUED on the context stream (which waits, on the device, for the launches issued so far); the host returns at once and may issue the
 * next frame's de_accumulate — its render launch runs beside the display and the copy.  end: wait for the oldest fetch begun and hand out its image. */
int de_fetch_image_begin(de_ctx* c) {
    if (!c) return fail(DE_ERR_INVALID, "null context");
    if (c->fetch_begun - c->fetch_ended >= (unsigned)DE_FETCH_RING) return fail(DE_ERR_STATE, "four fetches are in flight already: de_fetch_image_end first");
    const int k = (int)(c->fetch_begun % (unsigned)DE_FETCH_RING);
    const size_t bytes = (size_t)c->W * c->H * 3 * sizeof(float);
    HIP_TRY(hipSetDevice(c->device));
    if (!c->h_ring[k]) {
        if (hipHostMalloc((void**)&c->h_ring[k], bytes, hipHostMallocDefault) != hipSuccess) { c->h_ring[k] = nullptr; (void)hipGetLastError(); return fail(DE_ERR_NOMEM, "no pinned host memory for the staging buffers"); }
        HIP_TRY(hipEventCreateWithFlags(&c->ev_ring[k], hipEventDisableTiming));
    }
    int rc = de_render_to_image(c, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c->h_ring[k], c->d_image, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipEventRecord(c->ev_ring[k], c->stream));
    c->fetch_begun++;
    return DE_OK;
}
int de_fetch_image_end(de_ctx* c, const float** host_image) {
    if (!c || !host_image) return fail(DE_ERR_INVALID, "null argument");
    if (c->fetch_begun == c->fetch_ended) return fail(DE_ERR_STATE, "no fetch in flight: de_fetch_image_begin first");
    const int k = (int)(c->fetch_ended % (unsigned)DE_FETCH_RING);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(c->ev_ring[k]));
    c->fetch_ended++;
    *host_image = c->h_ring[k];
    return frame_status(c);
}

int de_fetch_hdr(de_ctx* c, float* out) {
    if (!c || !out) return fail(DE_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    { int rc = join_slots(c); if (rc) return rc; }
    touched_hdr(c);
    hipLaunchKernelGGL(hdr_transpose_kernel, dim3((unsigned)((c->W + 31) / 32), (unsigned)((c->H + 31) / 32)), dim3(256), 0, c->stream, c->display_src ? c->display_src : c->d_hdr, c->d_scratch, c->W, c->H);
    HIP_TRY(hipGetLastError());
    return copy_out(c, out, c->d_scratch);
}

int de_upload_hdr(de_ctx* c, const float* hdr, int spp) {
    if (!c || !hdr || spp < 0) return fail(DE_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    size_t npx = (size_t)c->W * c->H;
    std::vector<float> t(npx * 3);
    for (int i = 0; i < c->W; ++i)
        for (int j = 0; j < c->H; ++j)
            for (int ch = 0; ch < 3; ++ch) t[((size_t)j * c->W + i) * 3 + ch] = hdr[((size_t)i * c->H + j) * 3 + ch];
    { int rc = join_slots(c); if (rc) return rc; }
    touched_hdr(c);
    HIP_TRY(hipMemcpyAsync(c->d_hdr, t.data(), npx * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->current_spp = spp;
    return DE_OK;
}
int de_current_spp(de_ctx* c, int* spp) { if (!c || !spp) return fail(DE_ERR_INVALID, "null argument"); *spp = c->current_spp; return DE_OK; }
int de_set_current_spp(de_ctx* c, int spp) { if (!c || spp < 0) return fail(DE_ERR_INVALID, "bad spp"); c->current_spp = spp; return DE_OK; }

int de_hdr_device_ptr(de_ctx* c, void** ptr, uint64_t* n_floats) {
    if (!c || !ptr) return fail(DE_ERR_INVALID, "null argument");
    *ptr = c->d_hdr;
    if (n_floats) *n_floats = (uint64_t)c->W * c->H * 3;
    return DE_OK;
}
int de_bind_hdr(de_ctx* c, void* device_ptr, uint64_t n_floats) {
    if (!c) return fail(DE_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    { int rc = sync_all(c); if (rc) return rc; }
    touched_hdr(c);
    if (!device_ptr) { c->d_hdr = c->d_hdr_own; return DE_OK; }
    if (n_floats < (uint64_t)c->W * c->H * 3) return fail(DE_ERR_INVALID, "bound HDR buffer is smaller than W*H*3 floats");
    c->d_hdr = (float*)device_ptr;
    return DE_OK;
}
int de_set_stream(de_ctx* c, void* hip_stream) {
    if (!c) return fail(DE_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    { int rc = sync_all(c); if (rc) return rc; }
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    // NULL is HIP's null stream, as everywhere in HIP (a framework whose current stream IS the null stream — torch's default
    // stream — passes 0 here and must get exactly that stream, or its own work would not be ordered with the context's)
    c->stream = (hipStream_t)hip_stream; c->own_stream = false;
    touched_render_inputs(c); touched_hdr(c);      // whatever the new stream holds, the next launch is ordered after it
    return DE_OK;
}
int de_use_own_stream(de_ctx* c) {
    if (!c) return fail(DE_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    { int rc = sync_all(c); if (rc) return rc; }
    if (c->own_stream) return DE_OK;
    hipStream_t s = nullptr;
    int prio_lo = 0, prio_hi = 0;
    hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    HIP_TRY(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio_hi));
    c->stream = s; c->own_stream = true;
    touched_render_inputs(c); touched_hdr(c);
    return DE_OK;
}
int de_synchronize(de_ctx* c) {
    if (!c) return fail(DE_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    { int rc = sync_all(c); if (rc) return rc; }
    return frame_status(c);
}
int de_last_accumulate_ms(de_ctx* c, float* ms) {
    if (!c || !ms) return fail(DE_ERR_INVALID, "null argument");
    if (c->timing_empty) { *ms = 0.0f; return DE_OK; }      // the last call's share was empty: nothing was launched
    if (!c->timing_valid) return fail(DE_ERR_STATE, "no de_accumulate has been issued");
    HIP_TRY(hipSetDevice(c->device));
    // from the start of the call's first launch (after its waits) to the end of its last accumulate_kernel
    HIP_TRY(hipEventSynchronize(c->slot[c->t1_slot].t1));
    HIP_TRY(hipEventElapsedTime(ms, c->slot[c->t0_slot].t0, c->slot[c->t1_slot].t1));
    return DE_OK;
}
int de_set_kernel_variant(de_ctx* c, int variant) {
    if (!c || variant < 1 || variant > 6) return fail(DE_ERR_INVALID, "kernel variant must be 1 ... 6");
#ifndef DE_LEGACY_VARIANTS
    if (variant != 2 && variant != 4 && variant != 6)
        return fail(DE_ERR_INVALID, "kernel variants 1, 3 and 5 (per-lane loops, wavefront pipeline, HBM-queue scheduler) live in the legacy library: build with -DDE_LEGACY_VARIANTS (digital_earth_amd/build.py: build_legacy) and load it through DE_LIB_PATH");
#endif
    c->kernel_variant = variant; c->launch_variant = variant >= 4 ? 2 : variant;
    return DE_OK;
}

/* Every tuning knob of the product's kernels as ONE plain struct (include/digital_earth.h: de_tuning).  The library reads no environment
 * variable for them: a host that wants experiment overrides reads its own environment and calls this (the Python layer does: _native.py). */
int de_get_tuning(de_ctx* c, de_tuning* t) {
    if (!c || !t) return fail(DE_ERR_INVALID, "null argument");
    memset(t, 0, sizeof(*t));
    t->struct_bytes = (uint32_t)sizeof(de_tuning);
    t->kernel_variant = c->kernel_variant;
    t->launch_slots = c->n_slots; t->big_launch_slots = c->big_slots;
    t->v6_min_paths = (uint64_t)c->auto_v6_min_items;
    for (int k = 0; k < 3; ++k) { t->v6_service_area[k] = c->v6_svc_area[k]; t->v6_service_lanes[k] = c->v6_svc[k]; }
    t->v6_yield_max = c->v6_yield; t->v6_elsewhere_min = c->v6_elsewhere; t->v6_retry = c->v6_retry; t->v6_enter_min = c->v6_enter_min;
    t->v6_flat_min = c->v6_flat_min; t->v6_flat_again = c->v6_flat_again; t->v6_bands = c->v6_bands; t->v6_stats = c->v6_stats;
    t->v6_tail_levels = c->v6_tail_levels; t->v6_tail_min_paths = c->v6_tail_min_paths; t->v6_tail_when_alone = c->v6_tail_when_alone;
    for (int k = 0; k < 2; ++k) { t->v6_tail_export[k] = c->v6_tail_export[k]; t->v6_tail_grid[k] = c->v6_tail_grid[k]; }
    t->v2_pend = c->tune_pend; t->v2_heavy = c->tune_heavy; t->v2_b = c->tune_b; t->v2_gas = c->tune_gas; t->v2_chunk = c->tune_chunk;
    t->v2_waves_per_cu = c->tune_wpc; t->v2_max_spp = c->tune_max_spp;
    t->trace = c->trace ? 1 : 0;
    t->v6_cu_withhold = c->cu_withhold;
    return DE_OK;
}
int de_set_tuning(de_ctx* c, const de_tuning* t) {
    if (!c || !t) return fail(DE_ERR_INVALID, "null argument");
    if (t->struct_bytes != (uint32_t)sizeof(de_tuning)) return fail(DE_ERR_INVALID, "de_tuning.struct_bytes does not match this library's struct: fill it with de_get_tuning first");
    if (t->launch_slots < 1 || t->launch_slots > DE_MAX_SLOTS || t->big_launch_slots < 1 || t->big_launch_slots > DE_MAX_SLOTS) return fail(DE_ERR_INVALID, "launch slots must be 1..8");
    if (t->v2_chunk < 1 || t->v2_waves_per_cu < 1 || t->v2_waves_per_cu > 20 || t->v6_flat_again < 1 || t->v6_min_paths < 64) return fail(DE_ERR_INVALID, "tuning value out of range");
    for (int k = 0; k < 3; ++k) if (t->v6_service_lanes[k] < 1 || t->v6_service_lanes[k] > 64 || t->v6_service_area[k] < 0) return fail(DE_ERR_INVALID, "service thresholds out of range");
    if (t->v6_tail_levels < 0 || t->v6_tail_levels > 2) return fail(DE_ERR_INVALID, "v6_tail_levels must be 0..2");
    for (int k = 0; k < 2; ++k) if (t->v6_tail_export[k] < 1 || t->v6_tail_export[k] > 1024 || t->v6_tail_grid[k] < 1 || t->v6_tail_grid[k] > 4096) return fail(DE_ERR_INVALID, "tail settings out of range");
    if (t->v6_cu_withhold < 0 || 8 * t->v6_cu_withhold > c->n_cus - 8) return fail(DE_ERR_INVALID, "v6_cu_withhold: CUs per XCD withheld from the render streams, 0 .. (CUs / 8 - 1)");
    int rc = de_set_kernel_variant(c, t->kernel_variant);
    if (rc) return rc;
    if (t->v6_cu_withhold != c->cu_withhold) {
        // the launch slots' streams carry the mask: make them again (nothing may be in flight on them)
        HIP_TRY(hipSetDevice(c->device));
        rc = sync_all(c);
        if (rc) return rc;
        c->cu_withhold = t->v6_cu_withhold;
        for (int i = 0; i < DE_MAX_SLOTS; ++i) {
            LaunchSlot& sl = c->slot[i];
            if (!sl.stream) continue;
            HIP_TRY(hipStreamDestroy(sl.stream));
            sl.stream = nullptr;
            HIP_TRY(create_slot_stream(c, &sl.stream));
        }
    }
    if (t->launch_slots != c->n_slots || t->big_launch_slots != c->big_slots) { rc = de_set_launch_slots(c, t->launch_slots, t->big_launch_slots); if (rc) return rc; }
    c->auto_v6_min_items = (unsigned long long)t->v6_min_paths;
    for (int k = 0; k < 3; ++k) { c->v6_svc_area[k] = t->v6_service_area[k]; c->v6_svc[k] = t->v6_service_lanes[k]; }
    c->v6_yield = t->v6_yield_max; c->v6_elsewhere = t->v6_elsewhere_min; c->v6_retry = t->v6_retry; c->v6_enter_min = t->v6_enter_min;
    c->v6_flat_min = t->v6_flat_min; c->v6_flat_again = t->v6_flat_again; c->v6_bands = t->v6_bands == 8 ? 8 : 1; c->v6_stats = t->v6_stats;
    c->v6_tail_levels = t->v6_tail_levels; c->v6_tail_min_paths = t->v6_tail_min_paths; c->v6_tail_when_alone = t->v6_tail_when_alone != 0;
    for (int k = 0; k < 2; ++k) { c->v6_tail_export[k] = t->v6_tail_export[k]; c->v6_tail_grid[k] = t->v6_tail_grid[k]; }
    c->tune_pend = t->v2_pend; c->tune_heavy = t->v2_heavy; c->tune_b = t->v2_b; c->tune_gas = t->v2_gas; c->tune_chunk = t->v2_chunk;
    c->tune_wpc = t->v2_waves_per_cu; c->tune_max_spp = t->v2_max_spp;
    c->trace = t->trace != 0;
    return DE_OK;
}
int de_enable_counters(de_ctx* c, int enable) { if (!c) return fail(DE_ERR_INVALID, "null context"); c->count = enable != 0; return DE_OK; }
int de_get_counters(de_ctx* c, de_counters* out) {
    if (!c || !out) return fail(DE_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    unsigned long long h[16];
    { int rc = join_slots(c); if (rc) return rc; }
    HIP_TRY(hipMemcpyAsync(h, c->d_counters, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    memset(out, 0, sizeof(*out));
    out->samples = h[0]; out->taps_r8 = h[1]; out->taps_rgb8 = h[2]; out->sphere_steps = h[3];
    out->tracking_steps = h[4]; out->vertices = h[5]; out->rng_draws = h[6];
    for (int i = 0; i < 9; ++i) out->reserved[i] = h[7 + i];   // scheduler statistics of render_kernel_v2 (see its MODE 1 epilogue)
    return DE_OK;
}


int de_set_display_source(de_ctx* c, const void* device_ptr) {
    if (!c) return fail(DE_ERR_INVALID, "null context");
    c->display_src = (const float*)device_ptr;
    return DE_OK;
}
int de_set_launch_slots(de_ctx* c, int n_slots, int n_big) {
    if (!c || n_slots < 1 || n_slots > DE_MAX_SLOTS || n_big < 1 || n_big > DE_MAX_SLOTS) return fail(DE_ERR_INVALID, "slot counts must be 1..8");
    HIP_TRY(hipSetDevice(c->device));
    { int rc = sync_all(c); if (rc) return rc; }
    for (int i = c->n_slots; i < n_slots; ++i) {          // slots that did not exist yet
        LaunchSlot& s = c->slot[i];
        if (s.stream) continue;
        HIP_TRY(create_slot_stream(c, &s.stream));
        HIP_TRY(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
        HIP_TRY(hipEventCreate(&s.t0));
        HIP_TRY(hipEventCreate(&s.t1));
    }
    c->n_slots = n_slots; c->big_slots = n_big;
    c->next_slot = 0; c->last_slot = -1; c->last_v6_slot = -1;
    return DE_OK;
}
int de_set_wave_budget(de_ctx* c, int waves_per_cu) {
    if (!c || waves_per_cu < 1 || waves_per_cu > 20) return fail(DE_ERR_INVALID, "waves per CU must be 1..20");
    c->tune_wpc = waves_per_cu;
    return DE_OK;
}

/* the same for render_kernel_v6 (DE_V6_STATS=1): out[k] = word k of its ST_* list */
int de_debug_v6_stats(de_ctx* c, uint64_t* out, int n) {
    if (!c || !out || n < 0) return fail(DE_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    { int rc = sync_all(c); if (rc) return rc; }
    for (int k = 0; k < n; ++k) out[k] = 0;
    std::vector<uint32_t> h((size_t)bs::G_WORDS * DE_V6_CTL_STRIDE);
    for (auto& S : c->v6s) {
        if (!S.ctl) continue;
        HIP_TRY(hipMemcpy(h.data(), S.ctl, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (int k = 0; k < n && k < bs::ST_N; ++k) { uint64_t v; memcpy(&v, &h[(size_t)(bs::G_STAT0 + k) * DE_V6_CTL_STRIDE], 8); out[k] += v; }
        for (int k = 64; k < n && k < 128; ++k) { uint64_t v; memcpy(&v, &h[(size_t)(bs::G_STAT2 + k - 64) * DE_V6_CTL_STRIDE], 8); out[k] += v; }      // region statistics
        for (int k = 128; k < n && k < 192; ++k) { uint64_t v; memcpy(&v, &h[(size_t)(bs::G_DRAIN + k - 128) * DE_V6_CTL_STRIDE], 8); out[k] += v; }      // the drain's population histogram
    }
    return DE_OK;
}

int de_debug_sched_stats(de_ctx* c, uint64_t* out, int n) {
    if (!c || !out || n < 0 || n > DE_N_COUNTERS - 16) return fail(DE_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    unsigned long long h[DE_N_COUNTERS];
    { int rc = join_slots(c); if (rc) return rc; }
    HIP_TRY(hipMemcpyAsync(h, c->d_counters, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int i = 0; i < n; ++i) out[i] = h[16 + i];
    return DE_OK;
}

int de_debug_samples(de_ctx* c, uint64_t seed, int sample_index, float* out) {
    if (!c || !out || sample_index < 0) return fail(DE_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    int rc = build_tiles(c, 0, 1);
    if (rc) return rc;
    RenderArgs a;
    rc = fill_render_args(c, &a);
    if (rc) return rc;
    a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32);
    a.spp_begin = sample_index; a.spp_count = 1; a.spp_stride = 1;
    a.spp_magic = (1ull << 40) + 1ull;
    a.contrib = nullptr;
    a.work_counter = c->d_work_counter + 16 * DE_MAX_SLOTS;     // a counter of its own: launches in the slots keep theirs
    c->launch_variant = c->kernel_variant == 1 ? 1 : 2;         // (the pipeline has no single-sample trace mode: variants 3 and 4 trace with the state machine)
    rc = join_slots(c);
    if (rc) return rc;
    HIP_TRY(launch_render<2>(c, a, c->stream, []() -> hipError_t { return hipSuccess; }));
    HIP_TRY(hipMemcpyAsync(out, c->d_scratch, (size_t)c->W * c->H * 4 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return DE_OK;
}

int de_debug_math(de_ctx* c, int fn, const float* a, const float* b, float* out, uint64_t n) {
    if (!c || !a || !out || fn < 0 || fn > 31) return fail(DE_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    struct Bufs {   // freed on every exit path
        float *da = nullptr, *db = nullptr, *dout = nullptr;
        ~Bufs() { if (da) hipFree(da); if (db) hipFree(db); if (dout) hipFree(dout); }
    } m;
    HIP_TRY(hipMalloc(&m.da, n * sizeof(float)));
    HIP_TRY(hipMalloc(&m.dout, n * sizeof(float)));
    if (b) HIP_TRY(hipMalloc(&m.db, n * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(m.da, a, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    if (b) HIP_TRY(hipMemcpyAsync(m.db, b, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, fn, m.da, m.db, m.dout, (size_t)n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, m.dout, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return DE_OK;
}

}  // extern "C"

