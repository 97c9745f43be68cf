/* digital_earth.h — C ABI of libdigitalearth_hip.so (MI355X / gfx950).
 *
 * The reference (AntonioFerreras/Digital-Earth) has no FFI layer: its "operator API" for the hot path is
 * the Python class `Renderer` (renderer.py:15-401), whose @ti.kernel methods are the device boundary.
 * Each entry point below replaces one group of those methods; the Python `Renderer` in
 * digital_earth_amd/renderer.py binds them through ctypes.
 *
 * Conventions: every function returns 0 on success or a negative DE_ERR_* code and never throws;
 * de_last_error() returns a thread-local message for the last failure.  The caller owns every host
 * pointer; the library owns all device memory.  One context = one GPU; calls on one context are not thread-safe
 * (the reference is single-threaded, earth_viewer.py:183-318).  A context has ONE stream visible to the caller (the
 * "context stream": reset, display, fetch, reduce, uploads) and a few internal launch slots with streams of their own on
 * which de_accumulate runs, so that consecutive de_accumulate calls overlap (see de_accumulate); every entry point orders
 * itself against them, a caller only needs de_flush when it enqueues its OWN work on a stream given to de_set_stream.
 * There is no CPU fallback: de_create fails with DE_ERR_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef DIGITAL_EARTH_H
#define DIGITAL_EARTH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 1: round 1.  2: round 2 (contract-2 arithmetic, de_share_textures / de_trim_textures / de_comm_* added, DE_FLAG_NO_FAST_DIVISION
 * retired) — shipped still reporting 1.  3: launch slots (de_flush), de_use_own_stream, de_set_stream(NULL) = the null stream,
 * loans recorded by de_share_textures.  4: additions only — de_set_memory_budget / de_get_memory_use, de_last_call_info, de_fetch_image_view, de_debug_v5_stats, de_debug_v6_stats,
 * kernel variants 5 and 6.  5 (round 5): de_set_tuning / de_get_tuning replace every environment variable the library used to read for its
 * tuning; de_set_sample_partition, de_reduce_ordered, de_last_launch_phases added; kernel variants 1, 3 and 5 and de_debug_v5_stats moved to the
 * legacy library (digital_earth_legacy.h); an abort of a persistent launch is sticky until de_reset.  6 (round 6): the measurement, experiment and test hooks
 * (timers, counters, statistics, de_debug_*, map read-back, de_set_kernel_variant / _launch_slots / _wave_budget / _memory_budget) are declared in
 * digital_earth_debug.h — still exported, no longer in the binder's header; de_fetch_image_begin / _end (the window loop pipelined), de_tuning.v6_cu_withhold,
 * de_debug_ordered_sum, de_debug_standin_reduce added. */
#define DE_ABI_VERSION 6
/* version of the arithmetic contract the kernels and the oracle implement: golden vectors are tied to it */
#define DE_ARITHMETIC_CONTRACT 2

enum {
    DE_OK = 0,
    DE_ERR_INVALID = -1,   /* bad argument */
    DE_ERR_NO_DEVICE = -2, /* no HIP device / wrong architecture */
    DE_ERR_HIP = -3,       /* a HIP runtime call failed (message in de_last_error) */
    DE_ERR_STATE = -4,     /* call order violated (e.g. accumulate before textures/LUTs are uploaded) */
    DE_ERR_NOMEM = -5
};

/* texture slots — renderer.py:61-94 (albedo, topography, ocean, clouds, bathymetry, emissive, stars) */
enum {
    DE_TEX_ALBEDO = 0,     /* rgb8  (reference: rgba8, alpha unused)  renderer.py:61,171 */
    DE_TEX_TOPOGRAPHY = 1, /* r8    renderer.py:66,177 */
    DE_TEX_OCEAN = 2,      /* r8    renderer.py:71,183 */
    DE_TEX_CLOUDS = 3,     /* r8    renderer.py:76,189 */
    DE_TEX_BATHYMETRY = 4, /* r8    renderer.py:81,195 */
    DE_TEX_EMISSIVE = 5,   /* r8    renderer.py:86,201 */
    DE_TEX_STARS = 6,      /* rgb8  renderer.py:91,207 */
    DE_TEX_COUNT = 7
};

/* de_params.flags */
enum {
    DE_FLAG_FIXED_WAVELENGTH = 1u << 0, /* BASELINE cfg1: skip spectrum_sample's draw, use fixed_wavelength */
    DE_FLAG_CLAMP_SAMPLER = 1u << 1,    /* CLAMP_TO_EDGE instead of REPEAT texture addressing */
    DE_FLAG_RAY_MARCHER = 1u << 2,      /* integrate with ray_marcher (pathtracer.py:544-685) instead of path_tracer */
    DE_FLAG_AGX = 1u << 3,              /* display with srgb_transfer(AgX(linear)) — the alternative at renderer.py:356 (lib/AgX.py) — instead of OpenDRT + CRF + gamma */
    DE_FLAG_RESERVED_4 = 1u << 4,       /* was a round-1 test hook (IEEE-divide fallback of the tracking loops); ignored since contract 2 */
    DE_FLAG_FAST_MATH = 1u << 5         /* OPT-IN, not bit-reproducible: path_tracer on the hardware's exp / log / sin / cos / sqrt / rcp units and shorter atan2 / asin
                                           polynomials instead of the arithmetic contract's fixed sequences (render_kernel_v6 only; counting and trace launches and the ray
                                           marcher keep the contract).  A sample whose collision test lands on the other side follows a different path: the image agrees with
                                           the contract's statistically, not sample by sample (profiles/r5_fast_math.md: time, flip rate and L2 per camera). */
};

/* Scalar state of the reference Renderer (renderer.py:18-58), set through its set_* kernels (:224-266). */
typedef struct de_params {
    float camera_pos[3];      /* set_camera_pos  renderer.py:225 */
    float look_at[3];         /* set_look_at     renderer.py:233 */
    float up[3];              /* set_up (caller passes it normalised, as the kernel stores it) renderer.py:229 */
    float fov;                /* half-angle used as tangent-plane half extent  renderer.py:237,271-275 */
    float aspect_scale;       /* renderer.py:241 */
    float sun_angle;          /* renderer.py:261 */
    float sun_path_rot;       /* renderer.py:265 */
    float land_height_scale;  /* renderer.py:58 */
    float exposure;           /* renderer.py:245 */
    float gamma;              /* renderer.py:249 */
    int32_t selected_crf;     /* renderer.py:253 */
    float vignette_strength;  /* renderer.py:20 */
    float vignette_radius;    /* renderer.py:21 */
    float vignette_center[2]; /* renderer.py:22 */
    uint32_t flags;           /* DE_FLAG_* */
    float fixed_wavelength;   /* nm, with DE_FLAG_FIXED_WAVELENGTH */
    int32_t topo_res_override; /* land_normal's TOPOGRAPHY_TEX_RES[0] (pathtracer.py:20); 0 = width of the uploaded map */
    int32_t reserved[7];
} de_params;


typedef struct de_ctx de_ctx;

/* Renderer.__init__ (renderer.py:17-58): allocate color_buffer / _rendered_image for a W x H image on HIP
 * device `device`.  W must be a multiple of 16 and H of 8 (renderer.py:46). */
int de_create(int device, int width, int height, de_ctx** out);
int de_destroy(de_ctx* ctx);

/* imread + from_numpy + copy_*_texture (renderer.py:61-94,171-210).  texels: row-major, row j = 0 is v = 0
 * (south pole; Taichi imread is [x][y] with y up), `channels` bytes per texel (1 or 3). */
int de_upload_texture(de_ctx* ctx, int slot, const uint8_t* texels, int width, int height, int channels);
/* Synthetic stand-in for the NASA maps that the reference does not ship: generated on the GPU,
 * deterministic in (slot, width, height, seed, variant).  variant 1 = cloud-heavy (BASELINE cfg4). */
int de_generate_texture(de_ctx* ctx, int slot, int width, int height, uint32_t seed, int variant);
/* Let `dst` use the maps and LUTs that `src` holds on the same device instead of copies of its own (a second view of the
 * same Earth: 9 GB at quality 2).  Read-only borrowing; the borrowed maps keep `src`'s address mode.  The loan is recorded on
 * `src`: until every borrower has been destroyed or holds maps and LUTs of its own again, de_destroy, de_upload_texture,
 * de_generate_texture, de_upload_luts and a change of the sampler address mode on `src` fail with DE_ERR_STATE (nothing a
 * borrower reads can be freed or repacked under it).  de_download_texture on a borrower fails with DE_ERR_STATE. */
int de_share_textures(de_ctx* dst, de_ctx* src);
/* Release the as-uploaded (linear u8) copies of the maps, 2.1 GB at quality 2, keeping the packed copies the kernels read.
 * Afterwards de_download_texture and a change of the sampler address mode fail with DE_ERR_STATE until the maps are
 * uploaded or generated again. */
int de_trim_textures(de_ctx* ctx);

/* LUT loading + copy_CIE_LUT_texture / copy_CRF_LUT_texture (renderer.py:96-134,147-167,212-222).
 * cie: CIE.dat as read, f32[(x + y*441)*3 + c]; quantised to f16 inside, like the rgba16f texture.
 * srgb2spec_f16: srgb2spec.dat as read, 300x3 binary16.  o3: ozone_cross_section.dat, f32[441].
 * crf: (1024, n_crf, 3) f32 as returned by Renderer.load_crfs(). */
int de_upload_luts(de_ctx* ctx, const float* cie, const uint16_t* srgb2spec_f16, const float* o3,
                   const float* crf, int n_crf);

int de_set_params(de_ctx* ctx, const de_params* p);
int de_get_params(de_ctx* ctx, de_params* p);

/* reset_framebuffer (renderer.py:367-369) */
int de_reset(de_ctx* ctx);

/* accumulate (renderer.py:371-380) `spp` times: adds samples current_spp .. current_spp+spp-1 of every pixel
 * owned by (tile_rank, tile_world) to color_buffer, then current_spp += spp.  The RNG stream of a sample is a
 * function of (seed, absolute pixel index, sample index) only, so images do not depend on tile_world.
 * Ownership: 8x8-pixel tile (tx, ty) belongs to rank (tx + ty) % tile_world.
 * Asynchronous.  Consecutive calls go to internal launch slots in turn (8; 3 for launches with more than 96 MB of per-sample records; de_tuning.launch_slots / big_launch_slots) and overlap on
 * the GPU: k calls of 1 spp — the reference's interactive pattern, earth_viewer.py:241-243 — cost about what one call of
 * k spp costs.  The additions to color_buffer stay in call order (same bits as serial execution). */
int de_accumulate(de_ctx* ctx, int spp, uint64_t seed, int tile_rank, int tile_world);

/* fetch_image (renderer.py:382-384, 346-365): runs _render_to_image with samples = current_spp and copies
 * the (W, H, 3) f32 image to `out` (index (u*H + v)*3 + c, like the reference field). */
int de_fetch_image(de_ctx* ctx, float* out);
/* The same without the last copy: *host_image points at the context's pinned staging buffer, which holds the image until the next
 * de_fetch_* call on this context (the window loop hands it to the canvas before it renders again: earth_viewer.py:241-243, 316-317).
 * Saves the 24.9 MB host-to-host copy and the page faults of a fresh destination: 1920x1080, accumulate(1) + fetch per frame
 * 10.4 -> 9.7 ms (tools/one_spp.py; 9.1 ms of it is the lone launch itself: its longest paths, tools/loop_breakdown.py). */
int de_fetch_image_view(de_ctx* ctx, const float** host_image);
/* The window loop PIPELINED (earth_viewer.py:241-251 is accumulate() + fetch_image() per frame; with the synchronous fetch every frame pays its launch's whole
 * latency — ~10 ms at 1920x1080 for 1.7 ms of work: the launch's longest path is a serial chain — before the host may issue the next).  de_fetch_image_begin
 * enqueues the display transform of the frame as it stands and its copy into one of FOUR pinned buffers on the context stream and returns at once; the host
 * issues the next frame's de_accumulate, whose render launch runs beside them; de_fetch_image_end waits for the OLDEST fetch begun and not yet ended and hands
 * out its image (its buffer is filled again by the fourth de_fetch_image_begin after the one that filled it).  At most four fetches in flight.  The loop `accumulate; begin; (once
 * L fetches are in flight) end` shows frame k - L while frame k + 1 renders, L = 1, 2 or 3; every image equals the synchronous loop's image of the same frame
 * bit for bit. */
int de_fetch_image_begin(de_ctx* ctx);
int de_fetch_image_end(de_ctx* ctx, const float** host_image);
/* Same transform, result left on the device (pointer valid until the context is destroyed). */
int de_render_to_image(de_ctx* ctx, const float** device_image);
/* color_buffer as (W, H, 3) f32 — the raw HDR accumulation (renderer.py:25,330) */
int de_fetch_hdr(de_ctx* ctx, float* out);
/* Load an accumulation state: (W, H, 3) f32 HDR sums and the sample count they hold (resume of a progressive
 * render; the reference keeps this state only in memory, renderer.py:23,25). */
int de_upload_hdr(de_ctx* ctx, const float* hdr, int spp);
int de_current_spp(de_ctx* ctx, int* spp);
int de_set_current_spp(de_ctx* ctx, int spp);

/* Multi-GPU plumbing (no reference counterpart): the HDR buffer's device address and element count
 * (row-major [H][W][3] f32), so that the host can wrap it as a tensor and sum it across ranks with RCCL, and the
 * stream the context launches on.  de_set_stream makes the context use an existing hipStream_t (e.g. torch's). */
int de_hdr_device_ptr(de_ctx* ctx, void** ptr, uint64_t* n_floats);
/* Use caller-owned device memory (n_floats >= W*H*3, e.g. a torch tensor's data_ptr) as the HDR buffer; the current
 * contents are kept as the accumulation state.  ptr = NULL returns to the library's own buffer. */
int de_bind_hdr(de_ctx* ctx, void* device_ptr, uint64_t n_floats);
/* Use the caller's hipStream_t as the context stream.  NULL is HIP's null stream (what a framework whose current stream is the
 * default stream passes), NOT "the context's own stream": that is de_use_own_stream. */
int de_set_stream(de_ctx* ctx, void* hip_stream);
int de_use_own_stream(de_ctx* ctx);
/* Make the context stream wait, on the device, for every de_accumulate issued so far (returns at once).  For hosts that enqueue
 * their own work on a stream given to de_set_stream — e.g. a torch.distributed reduce of the tensor bound with de_bind_hdr. */
int de_flush(de_ctx* ctx);

/* The collective of the tile partition inside the library (SURVEY §8b `de_reduce`, §8e): one in-place
 * ncclReduce(sum, f32, count = W*H*3) of the HDR buffer to `root` over RCCL / xGMI, enqueued on the context's stream
 * after the accumulate launches.  librccl.so is bound at run time (dlopen; an instance already in the process, e.g.
 * PyTorch's, is reused), so single-GPU users never load it.
 *   de_comm_unique_id   rank 0: ncclGetUniqueId into a 128-byte buffer; the HOST carries those bytes to the other ranks
 *   de_comm_init        every rank: ncclCommInitRank (collective call); the communicator belongs to the context
 *   de_reduce           comm = NULL uses the context's communicator; otherwise an ncclComm_t the host framework owns */
int de_comm_unique_id(void* out128);
int de_comm_init(de_ctx* ctx, const void* id128, int rank, int world);
int de_comm_destroy(de_ctx* ctx);
int de_reduce(de_ctx* ctx, void* comm, int root);
/* Progressive display ("reduce every K spp"): like de_reduce, but OUT OF PLACE — every rank's accumulation buffer
 * stays its own (it keeps accumulating), the root receives the assembled frame in a buffer of the library, and that buffer
 * becomes what de_render_to_image / de_fetch_image / de_fetch_hdr read until the next de_reset.  (An in-place reduce would add
 * the other ranks' running sums into the root's again at the next reduce.) */
int de_reduce_progressive(de_ctx* ctx, void* comm, int root);
/* SAMPLE partition (SURVEY §8e "alternative for cfg5", reference renderer.py:371-380: one accumulate() per sample index, here dealt round
 * robin): after de_set_sample_partition(rank, world) a context renders, of the frame's sample indices covered by a de_accumulate call,
 * those = rank (mod world) of EVERY pixel it owns (combine with tile_rank = 0, tile_world = 1); the frame's sample counter advances by the
 * call's whole spp on every rank, and the RNG stream of a sample depends on (seed, pixel, sample index) only.  Every pixel then has a
 * partial sum on every rank, so the order of the cross-rank additions is part of the result: de_reduce_ordered gathers the ranks' buffers
 * on the root (ncclSend / ncclRecv, one group) and adds them in rank order — ((p0 + p1) + p2) + ... — in place, or into the assembled
 * buffer the display reads when out_of_place != 0 (progressive display: every rank keeps its running sums).  The assembled frame equals
 * the same partial sums added in the same order by any other means bit for bit; against the one-rank frame (one running sum in sample
 * order) it differs by f32 re-association: relative L2 <= 1e-6 (tests).  Perfect balance, no per-tile tail; 12 W H (world - 1) bytes into the root. */
int de_set_sample_partition(de_ctx* ctx, int rank, int world);
int de_reduce_ordered(de_ctx* ctx, void* comm, int root, int out_of_place);
/* Display transform source: a device buffer [H][W][3] f32 to read instead of the accumulation buffer (NULL = back to it) — for
 * hosts that run the out-of-place collective themselves (torch.distributed on a second tensor). */
int de_set_display_source(de_ctx* ctx, const void* device_ptr);
int de_synchronize(de_ctx* ctx);


/* Tuning of the product's kernels as one plain struct — the library reads NO environment variable for any of this (the only ones it
 * reads are DE_RCCL_PATH / DE_RCCL_STRICT, where to find librccl.so); a host that wants experiment overrides reads its own environment
 * (digital_earth_amd/_native.py does) and calls de_set_tuning.  Usage: de_get_tuning (fills struct_bytes and the current values), change
 * fields, de_set_tuning.  None of the values moves a bit of the image: they move time. */
typedef struct de_tuning {
    uint32_t struct_bytes;            /* sizeof(de_tuning) of the library that filled it */
    int32_t kernel_variant;           /* which kernels de_accumulate launches — same results, bit for bit, whatever the choice.  4 (default) automatic: a call of at least v6_min_paths
                                         paths runs the per-CU stage scheduler (render_kernel_v6.hip: ONE persistent launch, + a small tail launch of the same kernel for big
                                         calls), smaller calls and counting launches the wave-level state machine (render_kernel_v2.hip); 2 / 6: that kernel for every call.
                                         (1, 3, 5: the legacy library only, digital_earth_legacy.h.)  The ray marcher always runs its own kernel */
    int32_t launch_slots, big_launch_slots;   /* launches in flight per context (see de_accumulate): for small launches / for launches with more than 96 MB of per-sample records; 1 = serial */
    uint64_t v6_min_paths;            /* automatic variant: calls of at least this many paths run render_kernel_v6 (default 4096) */
    int32_t v6_service_area[3];       /* sphere trace / gas / cloud: idle lane-trips since its last service at which a loop stage services; 0 = use v6_service_lanes */
    int32_t v6_service_lanes[3];      /* ... idle lanes at which it does (the rule of round 4) */
    int32_t v6_yield_max, v6_elsewhere_min, v6_retry, v6_enter_min, v6_flat_min, v6_flat_again;   /* render_kernel_v6.hip: bs::Args */
    int32_t v6_bands;                 /* 8: one work counter per XCD, each over its own band of the launch's tiles (default); 1: one counter */
    int32_t v6_tail_levels;           /* 0: one launch per call.  1, 2: a launch does not wait for its last paths — a workgroup with nothing left to start and few paths alive
                                         exports them to a pool in HBM and exits, and that many further (small) launches of the same kernel finish them (render_kernel_v6.hip:
                                         "The tail").  Same bits; applies to launches that fill the GPU */
    int32_t v6_tail_export[2];        /* level k exports when at most this many paths are alive in the workgroup (1 .. 1024) */
    uint32_t v6_tail_min_paths;       /* ... and start at least this many paths (default 4 Mi: below that the tail launch's latency — the chain of a frame rendered alone ends ~1 ms
                                         later — outweighs the workgroup slots it frees; the interactive one-sample frame stays one launch) */
    int32_t v6_tail_when_alone;       /* 0 (default): a launch's workgroups export only when another launch of the context is already queued behind it — then the slots they free are
                                         taken at once; a call with nothing behind it (a frame rendered alone) finishes in place: it would end ~1 ms LATER with a tail launch.
                                         1: export regardless (tests) */
    int32_t v6_tail_grid[2];          /* workgroups of tail launch k + 1 (they must hold what level k can export: grid x 1120 >= workgroups of level k x v6_tail_export[k]) */
    int32_t v6_stats;                 /* 1: the instrumented instantiation of the kernel (de_debug_v6_stats) */
    int32_t v2_pend, v2_heavy, v2_b, v2_gas, v2_chunk, v2_waves_per_cu, v2_max_spp;   /* render_kernel_v2's scheduler thresholds, work chunk, waves per CU, test split */
    int32_t trace;                    /* 1: print what each call ran to stderr */
    int32_t v6_cu_withhold;           /* CUs per XCD that the render launches may NOT use (their streams carry a CU mask; the persistent grid shrinks with it): room for the small
                                         kernels of the context stream — collective, accumulate, display — which otherwise wait for a persistent workgroup to leave.
                                         0 (default): none.  Changing it waits for the launches in flight */
} de_tuning;
int de_get_tuning(de_ctx* ctx, de_tuning* out);
int de_set_tuning(de_ctx* ctx, const de_tuning* in);

const char* de_last_error(void);
int de_abi_version(void);
/* DE_ARITHMETIC_CONTRACT of the build: which fixed f32 sequences the kernels use.  Golden vectors and the oracle are
 * tied to it: tests refuse to compare a library with vectors of another contract. */
int de_arithmetic_contract(void);

#ifdef __cplusplus
}
#endif
#endif /* DIGITAL_EARTH_H */
