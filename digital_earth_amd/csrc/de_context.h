// de_context.h — the context behind the C ABI (include/digital_earth.h): device memory, streams, launch slots, what the launches need to know.
#pragma once
#include "de_kernels.h"

#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <string>
#include <vector>

// single translation unit: the kernels are compiled together with the host API
#include "render_kernel.hip"
#include "render_kernel_v2.hip"
#include "de_stages.h"
#ifdef DE_LEGACY_VARIANTS
#include "legacy/render_kernel_v3.hip"
#include "legacy/render_kernel_v5.hip"
#endif
#include "render_kernel_v6.hip"
#include "aux_kernels.hip"

namespace {

#define DE_N_COUNTERS 64   // 0-6 work counters, 7-8 trips / passes, 16-47 scheduler statistics, 48-59 section timers (render_kernel_v2 MODE 1)
thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(DE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

#define DE_MAX_SLOTS 8
#define DE_FETCH_RING 4      // de_fetch_image_begin / _end: fetches in flight (a lone one-sample launch takes ~10 ms, its longest path: 2 / 3 / 4 frames in flight run at 5.1 / 4.1 / ~3.5 ms per frame)
struct LaunchSlot {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;           // recorded after the slot's latest accumulate_kernel
    hipEvent_t t0 = nullptr, t1 = nullptr;
    uint2* contrib = nullptr;            // per-sample (radiance, wavelength node) records of the launch in this slot
    size_t contrib_items = 0;
    bool pending = false;                // `done` has not been waited for by the context stream yet
    bool launched = false;               // `done` has been recorded at least once (an event never recorded reads as complete)
    uint64_t seen_render = 0, seen_hdr = 0;
};

struct DevTexture {
    int w = 0, h = 0, ch = 0;
    uint8_t* linear = nullptr;    // as uploaded: [h][w][ch]
    uint32_t* packed = nullptr;   // footprint tiles (ch == 1) or rgbx dwords (ch == 3)
    int tiles_x = 0, tiles_y = 0;
    int packed_clamp = -1;        // address mode the packed copy was built with
    bool set = false;
    bool borrowed = false;        // the allocations belong to another context (de_share_textures)
};

}  // namespace

struct de_ctx {
    int device = 0;
    int W = 0, H = 0;
    de_params p;
    bool params_dirty = true;     // FrameConsts must be rebuilt (any scalar parameter, the topography width)
    bool nodes_dirty = true;      // the wavelength table must be rebuilt (LUTs, address mode, fixed wavelength)
    int tune_pend = 18, tune_heavy = 13, tune_b = 24, tune_gas = 6, tune_chunk = 128, tune_wpc = 20, tune_max_spp = 0;   // DE_V2_* overrides, read once in de_create
    hipStream_t stream = nullptr;   // the context stream: everything except the render launches (reset, display, fetch, reduce, uploads)
    bool own_stream = false;
    // Launch slots (round 3): de_accumulate does not launch on the context stream.  Launch k goes to slot k % n_slots, which has
    // its own stream, work counter and contribution buffer, so that consecutive launches overlap — the next render kernel
    // fills the machine while the last long paths of the previous one drain (the reference's interactive loop is
    // accumulate() x 1 per frame, earth_viewer.py:241-243).  The accumulate_kernels, which read-modify-write the HDR buffer,
    // are chained with events in launch order, so the per-pixel sums keep their association.  The context stream waits for
    // the slots lazily (join_slots) before anything that reads or writes the HDR buffer or rewrites what a launch reads.
    LaunchSlot slot[DE_MAX_SLOTS];
    int n_slots = 8, big_slots = 3, next_slot = 0, last_slot = -1;   // launches with > 96 MB of records cycle through the first big_slots only
    int last_v6_slot = -1;          // the slot of the newest render_kernel_v6 launch (the tail's "has a successor on another slot" word is about v6 launches only)
    int cu_withhold = 0;            // de_tuning.v6_cu_withhold: CUs per XCD the launch slots' streams may NOT use (hipExtStreamCreateWithCUMask), so that small kernels of
                                    // the context stream (collective, accumulate, display) find a free CU while persistent workgroups own the others; 0 = plain streams
    float* d_standin = nullptr;     // de_debug_standin_reduce: its second operand (zeros), its output and its copy target, [3][H][W][3]
    hipEvent_t ev_standin[16][2] = {};   // the last 16 stand-in collectives: de_last_reduce_ms answers their MEAN queue-to-finish time (frames in flight: no host wait per frame)
    unsigned standin_count = 0;
    bool last_reduce_standin = false;
    hipEvent_t ev_main = nullptr;   // last state of the context stream the slots may have to wait for
    uint64_t gen_render = 1, gen_hdr = 1, rec_render = 0, rec_hdr = 0;   // generations of context-stream work that launches depend on
    int t0_slot = -1, t1_slot = -1;
    bool timing_valid = false;
    bool timing_empty = false;      // the last de_accumulate launched nothing (an empty share): its duration is 0, not the previous call's
    DevTexture tex[DE_TEX_COUNT];
    float* d_cie = nullptr;      // 441 x 2 x 3 (f16-quantised)
    float* d_srgb2spec = nullptr;
    float* d_o3 = nullptr;
    float* d_crf = nullptr;      // [n][1024][3]
    int n_crf = 0;
    bool luts_set = false;
    bool luts_borrowed = false;
    FrameConsts* d_fc = nullptr;
    LambdaNode* d_nodes = nullptr;
    float* d_node_val = nullptr;
    float* d_hdr = nullptr;      // [H][W][3] (own allocation or bound by de_bind_hdr)
    float* d_hdr_own = nullptr;
    float* d_image = nullptr;    // (W, H, 3)
    float* d_scratch = nullptr;  // (W, H, 3) / debug [H][W][4]
    float* h_stage = nullptr;     // pinned host staging for fetch_image / fetch_hdr (a pageable destination copies at a fraction of the link rate)
    // de_fetch_image_begin / _end: the window loop pipelined — display + device-to-host copy of frame k run on the context stream while launch k + 1 renders.
    // DE_FETCH_RING pinned buffers and events; fetch_begun / fetch_ended count the calls (begun - ended = fetches in flight, at most DE_FETCH_RING).
    float* h_ring[DE_FETCH_RING] = {};
    hipEvent_t ev_ring[DE_FETCH_RING] = {};
    unsigned fetch_begun = 0, fetch_ended = 0;
    uint32_t* d_tiles = nullptr;
    int n_tiles = 0, tiles_rank = -1, tiles_world = -1;
    unsigned long long* d_counters = nullptr;
    uint32_t* d_work_counter = nullptr;   // 16 dwords per launch slot
    float* d_dens_table = nullptr;   // get_density by altitude index (DE_DENS_TABLE_N x 3)
    int n_cus = 256;
    bool count = false;
    de_counters counters;
    int current_spp = 0;
    int sample_rank = 0, sample_world = 1;   // de_set_sample_partition: of the frame's sample indices this context renders those = rank (mod world)
    float* d_assembled = nullptr;   // root's receive buffer of de_reduce_progressive ([H][W][3])
    float* d_gather = nullptr;      // root of de_reduce_ordered: the other ranks' buffers, [world][H][W][3]
    int gather_world = 0;
    const float* display_src = nullptr;   // what the display transform / de_fetch_hdr read instead of d_hdr (de_set_display_source)
    hipEvent_t ev_r0 = nullptr, ev_r1 = nullptr;   // around the last collective
    bool reduce_timing_valid = false;
    void* comm = nullptr;        // the context's own RCCL communicator (de_comm_init)
    int comm_rank = 0, comm_world = 1;
    bool trace = false;          // env DE_AUTO_TRACE, read once in de_create: print what the launch policy measured and chose
    int kernel_variant = 4;      // 4 = automatic (default): large launches run the first vertex rounds in the wavefront pipeline and finish in the state machine, small ones run the state machine alone; 1 = per-lane loops (render_kernel.hip), 2 = wave-level state machine (render_kernel_v2.hip), 3 = wavefront pipeline through HBM queues (render_kernel_v3.hip)
    // render_kernel_v6 (kernel variant 6): one persistent launch per call, one workgroup per CU, stage queues in LDS.  Per launch slot:
    // the control words, one cold record per record slot of every workgroup, the launch's RenderArgs.
    struct V6State {
        wf::Cold* cold = nullptr;
        uint32_t* ctl = nullptr;
        RenderArgs* d_args = nullptr;
        uint32_t* h_status = nullptr;     // pinned, device-visible: the kernel's abort code
        uint32_t n_wg = 0;
        uint4* pool[2] = {nullptr, nullptr};      // the tail's pools (render_kernel_v6.hip: "The tail"): level k exports to pool[k & 1]
        uint32_t pool_cap[2] = {0, 0};            // entries
    } v6s[DE_MAX_SLOTS];
    int v6_tail_levels = 1, v6_tail_export[2] = {128, 96}, v6_tail_grid[2] = {64, 8};
    uint32_t v6_tail_min_paths = 4u << 20;
    int v6_tail_when_alone = 0;      // 1: export even when no launch is queued behind (tests; default: only then, see render_kernel_v6.hip schedule())
    uint32_t* h_issued = nullptr;    // pinned, device-visible: the number of the newest render_kernel_v6 launch with a successor on another launch slot (read once per workgroup)
    uint32_t v6_launch_seq = 0;
    int v6_bands = 8;                // work counters of a launch: 8 = one band of the image per XCD, 1 = one for the whole launch (render_kernel_v6.hip: run_primary)
    int v6_stats = 0;                // 1: the instrumented kernel (env DE_V6_STATS; de_debug_v6_stats)
    int v6_svc[3] = {24, 24, 20};    // idle lanes at which a loop stage services (env DE_V6_SVC_ST / _GAS / _CLOUD)
    int v6_svc_area[3] = {100, 72, 90};  // idle lane-trips since its last service at which a loop stage services (sphere trace, gas, cloud; 0 = the idle-lane threshold above): render_kernel_v6.hip run_loop
    int v6_yield = 56, v6_elsewhere = 48, v6_retry = 6, v6_enter_min = 0, v6_flat_min = 0, v6_flat_again = 32;      // render_kernel_v6.hip: bs::Args (env DE_V6_YIELD / _ELSEWHERE / _RETRY / _ENTER_MIN)
    int launch_variant = 2;          // variant of the sub-launch being issued
    int launch_pipe = 0;             // v3 pipe of the sub-launch being issued
    int launch_slot = 0;             // launch slot of the sub-launch being issued
    bool launch_one_batch = false;   // the call being issued is one batch on one pipe (calls in flight)
    int last_call[4] = {0, 0, 0, 0}; // what the last de_accumulate ran: variant, pipes, pipeline rounds, launches (de_last_call_info)
    int launch_pipes = 1;            // pipes the call being issued runs side by side
    int launch_depths = 25, launch_wpc = 14;   // pipeline settings of the call being issued
    int auto_v6 = 1;                 // 1: the automatic variant runs calls of at least auto_v6_min_items paths on the per-CU stage scheduler (env DE_AUTO_V6)
    unsigned long long auto_v6_min_items = 1ull << 12;      // (env DE_AUTO_V6_MIN_ITEMS; 32 768 paths per call: 4.9 against 8.4 ms per frame of the window loop, 0.74 against 1.21 ms per call back to back — tools/small_calls.py)
    size_t mem_budget = 0;           // de_set_memory_budget (binds the legacy pipeline's queues; the product's kernels hold 37 MB per launch slot)
#ifdef DE_LEGACY_VARIANTS
#include "legacy/de_ctx_legacy_members.inc"
#endif
    bool frame_invalid = false;  // a persistent launch left on its abort word since the last de_reset: every fetch / reduce / synchronize reports it until then
    std::string invalid_msg;
    de_ctx* lender = nullptr;    // the context whose maps and LUTs this one reads (de_share_textures)
    int loans = 0;               // contexts currently reading THIS context's maps: while > 0 they may not be freed or repacked
};
