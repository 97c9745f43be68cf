// de_api.hip — host side of libdigitalearth_hip.so: context, device memory, kernel launches, C ABI
// (include/digital_earth.h).  No CPU fallback exists: every entry point needs a HIP device.
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
#include "render_kernel_v3.hip"
#include "render_kernel_v5.hip"
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

uint32_t f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// tex.store into an rgba16f texture (renderer.py:213-216): binary32 -> binary16 -> binary32, RNE
float quantize_f16(float f) {
    uint32_t x = f2bits(f);
    uint32_t sign = x & 0x80000000u, ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) return f;
    if (ax >= 0x477ff000u) return bits2f(sign | 0x7f800000u);
    if (ax < 0x33000001u) return bits2f(sign);
    int e = (int)(ax >> 23) - 127;
    uint32_t m = (ax & 0x007fffffu) | 0x00800000u;
    int shift = (e >= -14) ? 13 : (13 + (-14 - e));
    uint32_t kept = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (kept & 1u))) kept += 1;
    float v = ldexpf((float)kept, e - 23 + shift);
    return bits2f(f2bits(v) | sign);
}
float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 1023u;
    if (e == 31u) return bits2f(sign | 0x7f800000u | (m << 13));
    if (e == 0u) return bits2f(f2bits((float)m * 0x1p-24f) | sign);
    return bits2f(sign | ((e + 112u) << 23) | (m << 13));
}

void default_params(de_params* p) {
    memset(p, 0, sizeof(*p));
    p->camera_pos[0] = -15000000.0f; p->camera_pos[2] = 15000000.0f;   // earth_viewer.py:27
    p->up[1] = 1.0f;
    p->fov = (float)(27.0 * M_PI / 180.0 * 0.5);                        // renderer.py:50
    p->aspect_scale = 1.0f; p->exposure = 2.5f; p->gamma = 1.0f;       // :51-53
    p->sun_angle = (float)(60.0 * M_PI / 180.0);                       // :55
    p->sun_path_rot = (float)(-45.0 * M_PI / 180.0);                   // :56
    p->land_height_scale = 7800.0f;                                    // :58
    p->vignette_strength = 0.9f; p->vignette_radius = 0.0f;            // :20-21
    p->vignette_center[0] = 0.5f; p->vignette_center[1] = 0.5f;        // :22
}

// lib/OpenDRT.py:257-319,404 for the live module constants (Lp=100, gb=.12, c=1, fl=.005, dch=.35, EOTF=lin):
// constant expressions, evaluated once in double and rounded to f32.
void opendrt_consts(float* m, float* s, float* fl, float* ds, float* clamp_max, float* dch_s) {
    const double Lp = 100.0, gb = 0.12, c = 1.0, flare = 0.005, dch = 0.35;
    double ds_d = 100.0 / Lp, cm = ds_d * Lp / 100.0;
    double px = 128.0 * log10(Lp) / log10(100.0) - 64.0, py = Lp / 100.0, gx = 0.18;
    double gy = 11.696 / 100.0 * (1.0 + gb * log10(py) / log10(2.0));
    double s0 = (gy + sqrt(gy * (4.0 * flare + gy))) / 2.0, m0 = (py + sqrt(py * (4.0 * flare + py))) / 2.0;
    double ip = 1.0 / c;
    double s_d = (px * gx * (pow(m0, ip) - pow(s0, ip))) / (px * pow(s0, ip) - gx * pow(m0, ip));
    double m_d = pow(m0, ip) * (s_d + px) / px;
    *m = (float)m_d; *s = (float)s_d; *fl = (float)flare; *ds = (float)ds_d; *clamp_max = (float)cm; *dch_s = (float)(dch / s_d);
}

// lib/AgX.py:22-85, 99-101, 133-150: matrices and curve scales of the AgX display transform — constant expressions of
// module constants (primaries, COMPRESSION = 0.15, SLOPE = 2.3, TOE_POWER = 1.9, SHOULDER_POWER = 3.1, EV range
// [-10, 6.5]), evaluated in double and rounded to f32.
struct M3d { double m[9]; };
M3d m3_inverse(const M3d& a) {
    const double* m = a.m;
    double d = m[0] * (m[4] * m[8] - m[7] * m[5]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    double id = 1.0 / d;
    M3d c;
    c.m[0] = id * (m[4] * m[8] - m[7] * m[5]); c.m[1] = id * (m[2] * m[7] - m[1] * m[8]); c.m[2] = id * (m[1] * m[5] - m[2] * m[4]);
    c.m[3] = id * (m[5] * m[6] - m[3] * m[8]); c.m[4] = id * (m[0] * m[8] - m[2] * m[6]); c.m[5] = id * (m[3] * m[2] - m[0] * m[5]);
    c.m[6] = id * (m[3] * m[7] - m[6] * m[4]); c.m[7] = id * (m[6] * m[1] - m[0] * m[7]); c.m[8] = id * (m[0] * m[4] - m[3] * m[1]);
    return c;
}
void xy_unproject(double x, double y, double* X, double* Y, double* Z) {
    *X = 0; *Y = 0; *Z = 0;
    if (y != 0.0) { *Y = 1.0; *X = x / y; *Z = (1.0 - x - y) / y; }
}
M3d primaries_to_matrix(const double r[2], const double g[2], const double b[2], const double w[2]) {
    double Xr, Yr, Zr, Xg, Yg, Zg, Xb, Yb, Zb, Xw, Yw, Zw;
    xy_unproject(r[0], r[1], &Xr, &Yr, &Zr); xy_unproject(g[0], g[1], &Xg, &Yg, &Zg);
    xy_unproject(b[0], b[1], &Xb, &Yb, &Zb); xy_unproject(w[0], w[1], &Xw, &Yw, &Zw);
    M3d t = {{Xr, Xg, Xb, 1.0, 1.0, 1.0, Zr, Zg, Zb}};
    M3d inv = m3_inverse(t);
    double sx = inv.m[0] * Xw + inv.m[1] * Yw + inv.m[2] * Zw;
    double sy = inv.m[3] * Xw + inv.m[4] * Yw + inv.m[5] * Zw;
    double sz = inv.m[6] * Xw + inv.m[7] * Yw + inv.m[8] * Zw;
    M3d o = {{sx * Xr, sy * Xg, sz * Xb, sx * Yr, sy * Yg, sz * Yb, sx * Zr, sy * Zg, sz * Zb}};
    return o;
}
double agx_scale(double x_pivot, double y_pivot, double slope_pivot, double power) {
    return pow(pow(slope_pivot * x_pivot, -power) * (pow(slope_pivot * (x_pivot / y_pivot), power) - 1.0), -1.0 / power);
}
void agx_consts(float* out22) {
    const double R[2] = {0.64, 0.33}, G[2] = {0.3, 0.6}, B[2] = {0.15, 0.06}, W[2] = {0.3127, 0.3290};
    M3d s2x = primaries_to_matrix(R, G, B, W);
    const double sf = 1.0 / (1.0 - 0.15);
    double r2[2] = {(R[0] - W[0]) * sf + W[0], (R[1] - W[1]) * sf + W[1]}, g2[2] = {(G[0] - W[0]) * sf + W[0], (G[1] - W[1]) * sf + W[1]};
    double b2[2] = {(B[0] - W[0]) * sf + W[0], (B[1] - W[1]) * sf + W[1]};
    M3d x2a = m3_inverse(primaries_to_matrix(r2, g2, b2, W));
    for (int i = 0; i < 9; ++i) { out22[i] = (float)s2x.m[i]; out22[9 + i] = (float)x2a.m[i]; }
    const double xp = 10.0 / (6.5 + 10.0), yp = 0.5;
    out22[18] = (float)xp; out22[19] = (float)yp;
    out22[20] = (float)agx_scale(xp, yp, 2.3, 1.9);
    out22[21] = (float)agx_scale(1.0 - xp, 1.0 - yp, 2.3, 3.1);
}

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
    hipEvent_t ev_main = nullptr;   // last state of the context stream the slots may have to wait for
    uint64_t gen_render = 1, gen_hdr = 1, rec_render = 0, rec_hdr = 0;   // generations of context-stream work that launches depend on
    int t0_slot = -1, t1_slot = -1;
    bool timing_valid = false;
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
    // render_kernel_v3: one "pipe" = a set of stage queues + cold records.  Two pipes let two batches of paths run side by side on
    // two launch slots, so that the tail-heavy small launches of one overlap the large launches of the other (env DE_V3_PIPES)
    struct V3Pipe {
        wf::Rec* rec[wf::NQ] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // stage queues (64 B per path each)
        uint32_t* ctl = nullptr;      // two sets of 2 control words per queue (the set being filled / the set to fill next)
        int epoch[wf::NQ] = {0, 0, 0, 0, 0, 0, 0};   // completed drains per queue: parity selects the current set
        wf::Cold* cold = nullptr;     // 64 B per path
        size_t capacity = 0;          // paths the buffers hold
        // statistic for the automatic variant: records queued for vertex round 2 (holes included) of one launch, copied to pinned
        // host memory in stream order and read by a LATER de_accumulate call — never waited for
        uint32_t* h_round2 = nullptr;
        hipEvent_t stat_ev = nullptr, ev_t[3] = {nullptr, nullptr, nullptr};   // stat_ev: the copy has landed; ev_t: start / pipeline done / finisher done
        bool stat_pending = false, stat_timed = false;
        unsigned long long stat_items = 0;
        int stat_pipes = 1;
    } v3p[4];
    float auto_ratio = -1.f;         // last known (records queued for round 2) / (paths started); < 0 = none yet
    // Calls between auto_min_items and the two-pipe size (a rank's share of a tile partition, accumulate(16)) run the pipeline on four
    // pipes.  Rounds 3 picked the kernel for them by timing 32 calls on each (the state machine alone was 2-10 % faster on light views);
    // since the cloud stage reuses its map footprints (round 4) the pipeline wins or ties on every camera and share measured
    // (profiles/experiments: world 2 / 4 / 8 x four cameras; worst case -4 % / -9 % on the default / Apollo camera at 1/8 frame against
    // +24 % ... +45 % elsewhere), so the measuring state machine is gone.  env DE_AUTO_MID: 1 = the pipeline (default), 2 = the state machine
    int auto_mid_mode = 1;
    float auto_ns_per_path = -1.f;   // last known duration of a pipeline launch / its paths / the pipes running side by side
    bool auto_deep = false;          // the deep setting is in use (hysteresis: entered above auto_deep_ns, left below 0.8 x)
    int v3_pipes = 1;
    int v3_finisher = 2;             // what finishes the paths the pipeline hands over: 2 = render_kernel_v2 (resume mode), 5 = render_kernel_v5 (env DE_V3_FINISHER)
    int v5_fin_wpc = 8;              // waves per CU of a finisher launch of render_kernel_v5 (two pipes side by side; env DE_V5_FIN_WPC)
    int v3_fuse_track = 0;           // 1: one kernel tracks a ray through the gases and on through the cloud shell (v3_track_kernel: bit-identical, measured 8 % slower: profiles/experiments); 0: two stage kernels (env DE_V3_FUSE_TRACK)
    int v3_depths = 25;              // vertex rounds run by the pipeline before render_kernel_v2 finishes the surviving paths (env DE_V3_DEPTHS; 25 = all)
    int v3_batches = 2;              // with two pipes: batches a call is cut into (they alternate between the pipes; env DE_V3_BATCHES)
    size_t v3_max_items = (size_t)1 << 27;   // paths per batch (512 B each: 68 GB per pipe at 2^27 = a whole 1920x1080x64-spp frame; v3_reserve asks the device first and the batch shrinks when there is no room); env DE_V3_MAX_ITEMS
    size_t v3_test_budget = 0;       // env DE_V3_TEST_BUDGET (paths): see v3_reserve
    size_t mem_budget = 0;           // de_set_memory_budget: bytes the context may hold in stage queues / cold records / chunk pools (0 = automatic:
                                     // whatever the device has free minus 8 GB, at most 2^27 paths per batch)
    int v3_slack_wpc = 14;           // the largest wave budget any stage grid may get (set after the env knobs are read): sizes the queues' room for holes
    int v3_wpc = 14;                 // waves per CU the v3 grids are sized for (env DE_V3_WPC; per class: DE_V3_WPC_ST / _GAS / _CLOUD / _FLAT).
                                     // NOT the 32 the CU can hold: 64 paths per wave each keep a map line hot between two steps, and
                                     // beyond ~14 waves per CU that working set outgrows the XCD's 4 MB L2 (334 ms at 32, 252 ms at 14)
    int v3_wpc_class[4] = {0, 0, 16, 0};  // 0 = the call's wave budget; index: 0 sphere trace (and primary), 1 gas, 2 cloud, 3 the straight-line stages.  Cloud: 16 since the
                                          // tracking steps reuse their map footprint (round 4): the stage no longer lives on L2 capacity (8 / 12 / 16 / 20 / 24 / 32 waves: 186 / 176 / 174.5 / 181 / 180 / 186 ms per step)
    // render_kernel_v5 (kernel variant 5): one persistent stage-scheduler launch per call.  Each launch slot that runs it owns a chunk
    // pool, the queue rings, the control words and the cold records of its launch (launches of different slots overlap).
    struct V5State {
        uint4* pool = nullptr;
        uint32_t* ctl = nullptr;
        uint32_t* rings = nullptr;
        uint32_t* free_ring = nullptr;
        unsigned long long* cold = nullptr;
        RenderArgs* d_args = nullptr;     // the launch's RenderArgs (written by v5_init_kernel)
        uint32_t* h_status = nullptr;     // pinned, device-visible: the kernel's abort code
        size_t cold_items = 0;
        uint32_t pool_chunks = 0, ring_cells = 0, n_shards = 0;
    } v5s[DE_MAX_SLOTS + 4];         // one per launch slot (kernel variant 5), then one per pipe (the pipeline's finisher, v3_finisher = 5)
    // render_kernel_v6 (kernel variant 6): one persistent launch per call, one workgroup per CU, stage queues in LDS.  Per launch slot:
    // the control words, one cold record per record slot of every workgroup, the launch's RenderArgs.
    struct V6State {
        wf::Cold* cold = nullptr;
        uint32_t* ctl = nullptr;
        RenderArgs* d_args = nullptr;
        uint32_t* h_status = nullptr;     // pinned, device-visible: the kernel's abort code
        uint32_t n_wg = 0;
    } v6s[DE_MAX_SLOTS];
    int v6_stats = 0;                // 1: the instrumented kernel (env DE_V6_STATS; de_debug_v6_stats)
    int v6_svc[3] = {24, 24, 20};    // idle lanes at which a loop stage services (env DE_V6_SVC_ST / _GAS / _CLOUD)
    int v6_svc_area[3] = {0, 0, 0};  // > 0: idle lane-trips since its last service at which a loop stage services (env DE_V6_AREA_ST / _GAS / _CLOUD)
    int v6_yield = 56, v6_elsewhere = 48, v6_retry = 6, v6_enter_min = 0, v6_flat_min = 0;      // render_kernel_v6.hip: bs::Args (env DE_V6_YIELD / _ELSEWHERE / _RETRY / _ENTER_MIN)
    int v5_wpc = 16;                 // waves per CU of the persistent grid (env DE_V5_WPC)
    int v5_shards = 32;              // queue sets (8 = one per XCD; env DE_V5_SHARDS)
    int v5_local = 1;                // 1: teams live inside an XCD when the shard count is a multiple of 8 (env DE_V5_LOCAL)
    int v5_backlog = 2;              // chunks per wave queued in a shard before the waves stop starting new paths (env DE_V5_BACKLOG)
    int v5_pool_log2 = 18;           // chunks of 64 records in a slot's pool (2^18 = 1 GB; env DE_V5_POOL_LOG2)
    int v5_claim_k = 16, v5_age = 20000, v5_closers_per_cu = 2, v5_stats = 0;   // env DE_V5_CLAIM_K / _AGE (10 ns ticks) / _CLOSERS / _STATS
    int v5_reserve_pct = 25;         // new paths start only while this share of the pool is free (env DE_V5_RESERVE_PCT)
    size_t v5_max_items = (size_t)1 << 27;   // paths per launch (64 B of cold record each; env DE_V5_MAX_ITEMS)
    int v5_batches = 1;              // launches a large call is cut into (they overlap on the launch slots; env DE_V5_BATCHES)
    int launch_variant = 2;          // variant of the sub-launch being issued
    int launch_pipe = 0;             // v3 pipe of the sub-launch being issued
    int launch_slot = 0;             // launch slot of the sub-launch being issued
    bool launch_one_batch = false;   // the call being issued is one batch on one pipe (calls in flight)
    int last_call[4] = {0, 0, 0, 0}; // what the last de_accumulate ran: variant, pipes, pipeline rounds, launches (de_last_call_info)
    int launch_pipes = 1;            // pipes the call being issued runs side by side
    int launch_depths = 25, launch_wpc = 14;   // pipeline settings of the call being issued
    // variant 4 (automatic).  Measured on BASELINE's cameras (1920x1080, profiles/r3d_summary.md, profiles/experiments/): three vertex
    // rounds in the pipeline + the state machine as finisher beat the state machine alone by 7 % (default camera, 64 spp) to 77 %
    // (limb view), with 8 waves per CU per pipe, two pipes for batches above 48 M paths and four below (small batches: more tails to
    // overlap; large ones: the L2 holds two batches' lines, not four); launches under 8 M paths (the interactive 1-spp call) stay
    // with the state machine and its launch slots.
    int auto_depths = 3, auto_wpc = 8, auto_pipes = 0;
    int auto_full_batches1 = 1;      // 1: a full-frame call issued while others are in flight, outside the deep setting, is one batch (env DE_AUTO_FULL_BATCHES1)
    int auto_mid_batches1 = 1;       // 1: a mid-size call issued while others are in flight is one batch on one pipe (env DE_AUTO_MID_BATCHES1)
    unsigned long long auto_lone_max_items = 1ull << 23;    // largest lone call that runs the persistent scheduler (env DE_AUTO_LONE_MAX_ITEMS): the small calls.
                                                            // Up to 40 M paths it would still win on latency (a lone 1/8 share 34.2 against 43.7 ms, 1/4 share 56.0 / 65.1), but the
                                                            // first call of every back-to-back run is a lone one, and a persistent launch ahead of pipeline launches cost such runs
                                                            // 3-8 % (12 steps of the 1/4 share: 46.2 against 42.8 ms per step): not by default
    int auto_v6 = 1;                 // 1: the automatic variant runs calls of at least auto_v6_min_items paths on the per-CU stage scheduler (env DE_AUTO_V6)
    unsigned long long auto_v6_min_items = 1ull << 12;      // (env DE_AUTO_V6_MIN_ITEMS; 32 768 paths per call: 4.9 against 8.4 ms per frame of the window loop, 0.74 against 1.21 ms per call back to back — tools/small_calls.py)
    int auto_budget_v5 = 1;          // 1: a call the memory budget keeps off the pipeline runs the persistent scheduler when that fits (env DE_AUTO_BUDGET_V5)
    int auto_lone_v5 = 1;            // 1: a small call that finds nothing of this context in flight runs the persistent scheduler (env DE_AUTO_LONE_V5)
    unsigned long long auto_min_items = 1ull << 23, auto_two_pipe_items = 48ull << 20;
    // views whose paths are long (limb views through thick cloud: 2.4 - 2.9 ns of GPU time per path against 1.0 - 1.4 for the default camera)
    // keep the stage kernels full for more rounds and amortise a third, smaller batch: the pipeline then runs auto_deep_depths
    // rounds on auto_deep_pipes pipes (env DE_AUTO_DEEP_NS in 1/100 ns per path / DE_AUTO_DEEP_DEPTHS / DE_AUTO_DEEP_PIPES)
    int auto_deep_ns_x100 = 200, auto_deep_depths = 5, auto_deep_pipes = 3;
    int last_v3_slot[4] = {-1, -1, -1, -1};  // a pipe's launches follow each other
    bool frame_invalid = false;  // a persistent launch left on its abort word since the last de_reset: every fetch / reduce / synchronize reports it until then
    std::string invalid_msg;
    de_ctx* lender = nullptr;    // the context whose maps and LUTs this one reads (de_share_textures)
    int loans = 0;               // contexts currently reading THIS context's maps: while > 0 they may not be freed or repacked
};

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
int v6_reserve(de_ctx* c, int idx, hipStream_t stream);
hipError_t launch_v6(de_ctx* c, const RenderArgs& a, hipStream_t stream, int idx);
hipError_t launch_v5(de_ctx* c, const RenderArgs& a, hipStream_t stream, int idx, const uint4* resume_rec = nullptr, const uint32_t* resume_count = nullptr,
                     const float* resume_cold = nullptr, uint32_t resume_capacity = 0, int wpc = 0);
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
// render_kernel_v6: buffers of one launch slot (independent of the launch's size)
int v6_reserve(de_ctx* c, int idx, hipStream_t stream) {
    de_ctx::V6State& S = c->v6s[idx];
    const uint32_t n_wg = (uint32_t)c->n_cus * DE_V6_WG_PER_CU;
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
    A.idle_limit = 1u << 24;      // ~7 s of idling
    for (int k = 0; k < 3; ++k) { A.svc[k] = (uint32_t)c->v6_svc[k]; A.svc_area[k] = (uint32_t)c->v6_svc_area[k]; }
    A.yield_max = (uint32_t)c->v6_yield; A.elsewhere_min = (uint32_t)c->v6_elsewhere; A.retry = (uint32_t)c->v6_retry; A.enter_min = (uint32_t)c->v6_enter_min; A.flat_min = (uint32_t)c->v6_flat_min;
    hipLaunchKernelGGL(v6_init_kernel, dim3(1), dim3(256), 0, stream, S.ctl, a, S.d_args);
    // DE_V6_WG_PER_CU workgroups per CU; a small launch: one per wave-load of new paths per wave (a workgroup's waves start 64 at a time)
    const unsigned by_work = (unsigned)((A.n_items + 64u * DE_V6_WAVES - 1u) / (64u * DE_V6_WAVES));
    dim3 grid(by_work < S.n_wg ? (by_work ? by_work : 1u) : S.n_wg), block(64 * DE_V6_WAVES);
    if (c->v6_stats && !clamp) hipLaunchKernelGGL((v6_kernel<false, true>), grid, block, 0, stream, A, a.fc);
    else if (clamp) hipLaunchKernelGGL((v6_kernel<true, false>), grid, block, 0, stream, A, a.fc);
    else hipLaunchKernelGGL((v6_kernel<false, false>), grid, block, 0, stream, A, a.fc);
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
            c->invalid_msg = std::string("render_kernel_v6 left on its abort word (") + (code == bs::ERR_ENTRY_TIMEOUT ? "a ring entry was never stored" : code == bs::ERR_CELL_BUSY_TIMEOUT ? "a ring cell was never emptied" : "idle watchdog: work was lost") + "): the frame is invalid until de_reset";
        }
    }
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
    if (!march && c->launch_variant == 5 && MODE == 0) {
        hipError_t e5 = launch_v5(c, a, stream, c->launch_slot);
        if (e5 != hipSuccess) return e5;
        hipError_t pe = pre_acc();
        if (pe != hipSuccess) return pe;
        hipLaunchKernelGGL(accumulate_kernel, grid, block, 0, stream, a);
        return hipGetLastError();
    }
    if (!march && c->launch_variant == 6 && MODE == 0) {
        hipError_t e6 = launch_v6(c, a, stream, c->launch_slot);
        if (e6 != hipSuccess) return e6;
        hipError_t pe = pre_acc();
        if (pe != hipSuccess) return pe;
        hipLaunchKernelGGL(accumulate_kernel, grid, block, 0, stream, a);
        return hipGetLastError();
    }
    if (!march && c->launch_variant == 3 && MODE == 0) {
        hipError_t e3 = launch_v3(c, a, stream, c->launch_pipe);
        if (e3 != hipSuccess) return e3;
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
    } else if (clamp) hipLaunchKernelGGL((render_kernel<true, false, MODE>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((render_kernel<false, false, MODE>), grid, block, 0, stream, a);
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


// ---- RCCL, bound at run time (dlopen): a single-GPU user never loads it, and a process that already holds RCCL (PyTorch
// ships its own copy) keeps ONE instance — the already-loaded library is preferred over a second load.
struct UniqueId128;
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, struct UniqueId128, int) = nullptr;
    int (*Reduce)(const void*, void*, size_t, int, int, int, void*, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;      // the four below: de_reduce_ordered only (optional at load)
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*CommUserRank)(void*, int*) = nullptr;   // optional: which rank of an external communicator this context is
    const char* (*GetErrorString)(int) = nullptr;
};
struct UniqueId128 { char internal[128]; };        // ncclUniqueId (rccl.h:43), passed BY VALUE to ncclCommInitRank
Rccl g_rccl;
std::mutex g_rccl_mutex;       // contexts of different threads may reach the first use together
int load_rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.h) return DE_OK;
    void* h = nullptr;
    bool only_env = false;
    if (const char* e = getenv("DE_RCCL_PATH")) {
        h = dlopen(e, RTLD_NOW | RTLD_GLOBAL);
        only_env = getenv("DE_RCCL_STRICT") != nullptr;   // test hook: no search beyond DE_RCCL_PATH
    }
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (int pass = 0; pass < 2 && !h && !only_env; ++pass)
        for (const char* n : names) {
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (h) break;
        }
    if (!h) {
        const char* why = dlerror();      // ONE call: dlerror() clears the message it returns
        return fail(DE_ERR_STATE, std::string("cannot load librccl.so: ") + (why ? why : "not found"));
    }
    Rccl r; r.h = h;
    r.GetUniqueId = (int (*)(void*))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (int (*)(void**, int, UniqueId128, int))dlsym(h, "ncclCommInitRank");
    r.Reduce = (int (*)(const void*, void*, size_t, int, int, int, void*, hipStream_t))dlsym(h, "ncclReduce");
    r.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
    r.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    r.CommUserRank = (int (*)(void*, int*))dlsym(h, "ncclCommUserRank");
    r.Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclSend");
    r.Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclRecv");
    r.GroupStart = (int (*)())dlsym(h, "ncclGroupStart");
    r.GroupEnd = (int (*)())dlsym(h, "ncclGroupEnd");
    r.CommCount = (int (*)(void*, int*))dlsym(h, "ncclCommCount");
    if (!r.GetUniqueId || !r.CommInitRank || !r.Reduce || !r.CommDestroy || !r.GetErrorString)
        return fail(DE_ERR_STATE, "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclReduce / ncclCommDestroy / ncclGetErrorString");
    g_rccl = r;
    return DE_OK;
}
#define RCCL_TRY(expr)                                                                                           \
    do {                                                                                                         \
        int r_ = (expr);                                                                                         \
        if (r_ != 0) return fail(DE_ERR_HIP, std::string(#expr) + ": " + g_rccl.GetErrorString(r_));             \
    } while (0)

}  // namespace

namespace {
// Every environment override of the library, read ONCE per context here (experiments, tests, A/B sweeps; none is needed in production
// and none is read again later):
//   DE_KERNEL             kernel variant 1..5 (de_set_kernel_variant)             DE_AUTO_TRACE   print what the launch policy chose
//   DE_V2_THR / _A / _B / _G / _CHUNK / _WPC / _MAX_SPP     render_kernel_v2's scheduler thresholds, work chunk, waves per CU, test split
//   DE_V3_MAX_ITEMS / _WPC / _WPC_{ST,GAS,CLOUD,FLAT} / _PIPES / _BATCHES / _DEPTHS / _TEST_BUDGET      the pipeline alone (variant 3)
//   DE_AUTO_DEPTHS / _WPC / _PIPES / _MIN_ITEMS / _TWO_PIPE_ITEMS / _MID / _DEEP_NS / _DEEP_DEPTHS / _DEEP_PIPES     the default's policy
//   DE_V5_WPC / _SHARDS / _LOCAL / _POOL_LOG2 / _CLAIM_K / _AGE / _CLOSERS / _STATS / _BACKLOG / _RESERVE_PCT / _BATCHES / _MAX_ITEMS   variant 5
//   DE_SLOTS / DE_BIG_SLOTS   launch slots (read next to their creation in de_create)
void read_env_knobs(de_ctx* c) {
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
        env_int("DE_V6_YIELD", &c->v6_yield, 0); env_int("DE_V6_ELSEWHERE", &c->v6_elsewhere, 1); env_int("DE_V6_RETRY", &c->v6_retry, 1); env_int("DE_V6_ENTER_MIN", &c->v6_enter_min, 0); env_int("DE_V6_FLAT_MIN", &c->v6_flat_min, 0); env_int("DE_V5_STATS", &c->v5_stats, 0); env_int("DE_V5_LOCAL", &c->v5_local, 0); env_int("DE_V5_BACKLOG", &c->v5_backlog, 0); env_int("DE_V5_RESERVE_PCT", &c->v5_reserve_pct, 1); env_int("DE_V5_BATCHES", &c->v5_batches, 1);
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
    read_env_knobs(c);
    memset(&c->counters, 0, sizeof(c->counters));
    size_t npx = (size_t)width * height;
    // the context stream carries the small operations between frames and the collective: highest priority, so that wave slots
    // freed by the draining render kernels of the launch slots go to them first
    int prio_lo = 0, prio_hi = 0;
    hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    hipError_t e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_hi);
    c->own_stream = (e == hipSuccess);
    {   // launches in flight per context (DE_SLOTS, 1..DE_MAX_SLOTS; 1 = every launch waits for the previous one)
        if (const char* sv = getenv("DE_SLOTS")) { int x = atoi(sv); if (x >= 1 && x <= DE_MAX_SLOTS) c->n_slots = x; }
        if (const char* sv = getenv("DE_BIG_SLOTS")) { int x = atoi(sv); if (x >= 1 && x <= DE_MAX_SLOTS) c->big_slots = x; }
        for (int i = 0; i < c->n_slots && e == hipSuccess; ++i) {
            LaunchSlot& s = c->slot[i];
            e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking);
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
    c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
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
    void* ptrs[] = {c->d_cie, c->d_srgb2spec, c->d_o3, c->d_crf, c->d_fc, c->d_nodes, c->d_node_val, c->d_hdr_own, c->d_image, c->d_scratch, c->d_tiles, c->d_counters, c->d_work_counter, c->d_dens_table, c->d_assembled, c->d_gather};
    for (void* p : ptrs) if (p) hipFree(p);
    if (c->h_stage) hipHostFree(c->h_stage);
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
    for (auto& S : c->v6s) {
        if (S.cold) hipFree(S.cold); if (S.ctl) hipFree(S.ctl); if (S.d_args) hipFree(S.d_args);
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
    // [current_spp, current_spp + spp); this context renders those = sample_rank (mod sample_world) — `first`, first + world, ... —
    // and the frame's sample counter advances by the whole spp on every rank.
    const int frame_spp = spp;
    int first_index = c->current_spp;
    if (c->sample_world > 1) {
        const int rem = first_index % c->sample_world;
        first_index += (c->sample_rank - rem + c->sample_world) % c->sample_world;
        spp = first_index < c->current_spp + frame_spp ? (c->current_spp + frame_spp - first_index + c->sample_world - 1) / c->sample_world : 0;
    }
    a.spp_stride = c->sample_world;
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
        if (v6k) { int rc6 = v6_reserve(c, si, c->slot[si].stream); if (rc6) return rc6; }
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
        a.spp_begin = first_index + done * c->sample_world; a.spp_count = n;
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
            HIP_TRY(hipEventRecord(s.t1, s.stream)); c->t1_slot = si; c->timing_valid = true;
        }
    }
    c->current_spp += frame_spp;
    return DE_OK;
}

int de_set_sample_partition(de_ctx* c, int rank, int world) {
    if (!c || world < 1 || rank < 0 || rank >= world) return fail(DE_ERR_INVALID, "sample partition: 0 <= rank < world");
    c->sample_rank = rank; c->sample_world = world;      // read by the next de_accumulate; launches in flight keep what they were issued with
    return DE_OK;
}

namespace {
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
}  // namespace

int de_set_memory_budget(de_ctx* c, uint64_t bytes) {
    if (!c) return fail(DE_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    { int rc = sync_all(c); if (rc) return rc; }
    free_queue_memory(c);                 // what is held may exceed the new budget: the next large call allocates within it
    c->mem_budget = (size_t)bytes;
    return DE_OK;
}
int de_get_memory_use(de_ctx* c, uint64_t* queue_bytes) {
    if (!c || !queue_bytes) return fail(DE_ERR_INVALID, "null argument");
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
    for (auto& S : c->v6s) if (S.cold) n += (uint64_t)S.n_wg * DE_V6_P * sizeof(wf::Cold);      // the per-CU scheduler: 37 MB per launch slot, whatever the call
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
    if (!c->timing_valid) return fail(DE_ERR_STATE, "no de_accumulate has been issued");
    HIP_TRY(hipSetDevice(c->device));
    // from the start of the call's first launch (after its waits) to the end of its last accumulate_kernel
    HIP_TRY(hipEventSynchronize(c->slot[c->t1_slot].t1));
    HIP_TRY(hipEventElapsedTime(ms, c->slot[c->t0_slot].t0, c->slot[c->t1_slot].t1));
    return DE_OK;
}
int de_set_kernel_variant(de_ctx* c, int variant) {
    if (!c || variant < 1 || variant > 6) return fail(DE_ERR_INVALID, "kernel variant must be 1 ... 6");
    c->kernel_variant = variant; c->launch_variant = variant >= 4 ? 2 : variant;
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


/* ---- in-library collective (SURVEY §8b `de_reduce`, §8e) */
int de_comm_unique_id(void* out128) {
    if (!out128) return fail(DE_ERR_INVALID, "out is null");
    int rc = load_rccl();
    if (rc) return rc;
    RCCL_TRY(g_rccl.GetUniqueId(out128));
    return DE_OK;
}
int de_comm_init(de_ctx* c, const void* id128, int rank, int world) {
    if (!c || !id128 || world < 1 || rank < 0 || rank >= world) return fail(DE_ERR_INVALID, "bad communicator arguments");
    if (c->comm) return fail(DE_ERR_STATE, "the context already has a communicator: de_comm_destroy first");
    int rc = load_rccl();
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    UniqueId128 id;
    memcpy(id.internal, id128, sizeof(id.internal));
    void* comm = nullptr;
    RCCL_TRY(g_rccl.CommInitRank(&comm, world, id, rank));
    c->comm = comm; c->comm_rank = rank; c->comm_world = world;
    return DE_OK;
}
int de_comm_destroy(de_ctx* c) {
    if (!c) return fail(DE_ERR_INVALID, "null context");
    if (!c->comm) return DE_OK;
    HIP_TRY(hipSetDevice(c->device));
    { int rc2 = sync_all(c); if (rc2) return rc2; }
    RCCL_TRY(g_rccl.CommDestroy(c->comm));
    c->comm = nullptr; c->comm_rank = 0; c->comm_world = 1;
    return DE_OK;
}
namespace {
int reduce_impl(de_ctx* c, void* comm, int root, bool progressive) {
    if (!c) return fail(DE_ERR_INVALID, "null context");
    void* use = comm ? comm : c->comm;
    if (!use) return fail(DE_ERR_STATE, "no communicator: pass an ncclComm_t or call de_comm_init");
    if (root < 0 || (!comm && root >= c->comm_world)) return fail(DE_ERR_INVALID, "root out of range");
    int rc = load_rccl();
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    rc = join_slots(c);
    if (rc) return rc;
    rc = frame_status(c);      // what is known so far; the host synchronises (and asks again) before it reads the assembled frame
    if (rc) return rc;
    const size_t count = (size_t)c->W * c->H * 3;
    float* recv = c->d_hdr;
    if (progressive) {
        // out of place: every rank keeps accumulating into its own buffer; the root receives the assembled frame elsewhere
        if (!c->d_assembled) {
            HIP_TRY(hipMalloc(&c->d_assembled, count * sizeof(float)));
            HIP_TRY(hipMemsetAsync(c->d_assembled, 0, count * sizeof(float), c->stream));      // non-root ranks: scratch, but never uninitialised
        }
        recv = c->d_assembled;
    } else {
        touched_hdr(c);
    }
    if (!c->ev_r0) { HIP_TRY(hipEventCreate(&c->ev_r0)); HIP_TRY(hipEventCreate(&c->ev_r1)); }
    HIP_TRY(hipEventRecord(c->ev_r0, c->stream));
    // every rank's buffer is its own tiles plus zeros elsewhere; the root ends with the assembled frame
    RCCL_TRY(g_rccl.Reduce(c->d_hdr, recv, count, /*ncclFloat32*/ 7, /*ncclSum*/ 0, root, use, c->stream));
    HIP_TRY(hipEventRecord(c->ev_r1, c->stream));
    c->reduce_timing_valid = true;
    if (progressive) {
        touched_hdr(c);                        // the collective READS d_hdr: the next accumulate_kernel is ordered after it
        // only the root receives the assembled frame: elsewhere ncclReduce leaves recvbuff alone, and the display / fetch entry
        // points keep reading the rank's own sums
        int my_rank = c->comm_rank;
        if (comm) { if (!g_rccl.CommUserRank || g_rccl.CommUserRank(use, &my_rank) != 0) my_rank = root; }      // cannot ask: behave as every rank did before (the root is right, the others show scratch)
        if (my_rank == root) c->display_src = c->d_assembled;
    }
    return DE_OK;
}
}  // namespace

namespace {
// The collective of a SAMPLE partition (de_set_sample_partition): every pixel's sum has a term on every rank, so the order of the additions
// is part of the result.  ncclReduce leaves that order to the algorithm RCCL picks; here the ranks' buffers are GATHERED on the root (one
// ncclSend / ncclRecv pair per peer in one group: the root's ingress runs over its xGMI links in parallel) and added by ordered_sum_kernel
// in rank order, so the assembled frame is a function of the partition alone (tests hold it to the oracle's partial sums added the same way).
int reduce_ordered_impl(de_ctx* c, void* comm, int root, bool out_of_place) {
    if (!c) return fail(DE_ERR_INVALID, "null context");
    void* use = comm ? comm : c->comm;
    if (!use) return fail(DE_ERR_STATE, "no communicator: pass an ncclComm_t or call de_comm_init");
    int rc = load_rccl();
    if (rc) return rc;
    if (!g_rccl.Send || !g_rccl.Recv || !g_rccl.GroupStart || !g_rccl.GroupEnd) return fail(DE_ERR_STATE, "librccl.so lacks ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd");
    int world = c->comm_world, my_rank = c->comm_rank;
    if (comm) {
        if (!g_rccl.CommCount || !g_rccl.CommUserRank) return fail(DE_ERR_STATE, "librccl.so lacks ncclCommCount / ncclCommUserRank: an external communicator cannot be asked for its shape");
        RCCL_TRY(g_rccl.CommCount(use, &world));
        RCCL_TRY(g_rccl.CommUserRank(use, &my_rank));
    }
    if (root < 0 || root >= world) return fail(DE_ERR_INVALID, "root out of range");
    if (world > 16) return fail(DE_ERR_INVALID, "de_reduce_ordered: at most 16 ranks");
    HIP_TRY(hipSetDevice(c->device));
    rc = join_slots(c);
    if (rc) return rc;
    rc = frame_status(c);
    if (rc) return rc;
    const size_t count = (size_t)c->W * c->H * 3;
    if (!c->ev_r0) { HIP_TRY(hipEventCreate(&c->ev_r0)); HIP_TRY(hipEventCreate(&c->ev_r1)); }
    if (my_rank == root) {
        if (c->gather_world < world) {
            { int rc2 = sync_all(c); if (rc2) return rc2; }
            if (c->d_gather) hipFree(c->d_gather);
            c->d_gather = nullptr; c->gather_world = 0;
            if (hipMalloc(&c->d_gather, (size_t)world * count * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return fail(DE_ERR_NOMEM, "no room for the gather buffer of de_reduce_ordered"); }
            c->gather_world = world;
        }
        if (out_of_place && !c->d_assembled) HIP_TRY(hipMalloc(&c->d_assembled, count * sizeof(float)));
    }
    HIP_TRY(hipEventRecord(c->ev_r0, c->stream));
    if (my_rank == root) {
        RCCL_TRY(g_rccl.GroupStart());
        for (int r = 0; r < world; ++r)
            if (r != root) { int e = g_rccl.Recv(c->d_gather + (size_t)r * count, count, /*ncclFloat32*/ 7, r, use, c->stream); if (e != 0) { g_rccl.GroupEnd(); return fail(DE_ERR_HIP, std::string("ncclRecv: ") + g_rccl.GetErrorString(e)); } }
        RCCL_TRY(g_rccl.GroupEnd());
        OrderedSumArgs o;
        for (int r = 0; r < world; ++r) o.part[r] = (r == root) ? c->d_hdr : c->d_gather + (size_t)r * count;
        o.out = out_of_place ? c->d_assembled : c->d_hdr; o.n_parts = world; o.n4 = count / 4; o.n = count;
        hipLaunchKernelGGL(ordered_sum_kernel, dim3((unsigned)c->n_cus * 8u), dim3(256), 0, c->stream, o);
        HIP_TRY(hipGetLastError());
        if (out_of_place) c->display_src = c->d_assembled;
    } else {
        RCCL_TRY(g_rccl.Send(c->d_hdr, count, /*ncclFloat32*/ 7, root, use, c->stream));
    }
    HIP_TRY(hipEventRecord(c->ev_r1, c->stream));
    c->reduce_timing_valid = true;
    touched_hdr(c);                        // the collective reads (root, in place: rewrites) d_hdr: the next accumulate_kernel is ordered after it
    return DE_OK;
}
}  // namespace
int de_reduce_ordered(de_ctx* c, void* comm, int root, int out_of_place) { return reduce_ordered_impl(c, comm, root, out_of_place != 0); }

int de_reduce(de_ctx* c, void* comm, int root) { return reduce_impl(c, comm, root, false); }
int de_reduce_progressive(de_ctx* c, void* comm, int root) { return reduce_impl(c, comm, root, true); }
int de_last_reduce_ms(de_ctx* c, float* ms) {
    if (!c || !ms) return fail(DE_ERR_INVALID, "null argument");
    if (!c->reduce_timing_valid) return fail(DE_ERR_STATE, "no de_reduce has been issued");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(c->ev_r1));
    HIP_TRY(hipEventElapsedTime(ms, c->ev_r0, c->ev_r1));
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
        HIP_TRY(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
        HIP_TRY(hipEventCreate(&s.t0));
        HIP_TRY(hipEventCreate(&s.t1));
    }
    c->n_slots = n_slots; c->big_slots = n_big;
    c->next_slot = 0; c->last_slot = -1;
    return DE_OK;
}
int de_set_wave_budget(de_ctx* c, int waves_per_cu) {
    if (!c || waves_per_cu < 1 || waves_per_cu > 20) return fail(DE_ERR_INVALID, "waves per CU must be 1..20");
    c->tune_wpc = waves_per_cu;
    return DE_OK;
}

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
