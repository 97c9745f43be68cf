/* digital_earth_debug.h — measurement, experiment and test hooks of libdigitalearth_hip.so: exported by the same library, NOT part of what a binder of the
 * hot path needs (that is include/digital_earth.h: context, maps and LUTs, parameters, the frame loop, fetches, the multi-GPU plumbing, de_set_tuning).
 * What is here: timers and work counters behind the bench line, the kernel's own statistics, single-sample traces and the elementary functions for the
 * parity tests, map read-back for the oracle, the experiment switches that also exist as de_tuning fields, and two stand-ins that let ONE GPU exercise the
 * code of the N-rank collectives (de_debug_ordered_sum, de_debug_standin_reduce).  tests/test_abi.py holds both headers to the library and to the ctypes binding. */
#ifndef DIGITAL_EARTH_DEBUG_H
#define DIGITAL_EARTH_DEBUG_H
#include "digital_earth.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- timers */
/* Duration of the last de_accumulate call measured with HIP events on its launch slot (ms): from the start of its first
 * render launch to the end of its last accumulate_kernel. */
int de_last_accumulate_ms(de_ctx* ctx, float* ms);
/* Duration of the last de_reduce / de_reduce_progressive on the context stream (HIP events, ms). */
int de_last_reduce_ms(de_ctx* ctx, float* ms);
/* Phases of the LAST render_kernel_v6 launch of the last de_accumulate call, from the kernel's own clock (100 MHz): ms2[0] = first workgroup start to
 * last wave exit, ms2[1] = the DRAIN, from the first wave that found no work item left to the last wave's exit (what separates a rank's 1/N share of a
 * frame from 1/N of the frame's time).  Waits for that launch.  DE_ERR_STATE when the last call did not run render_kernel_v6. */
int de_last_launch_phases(de_ctx* ctx, float* ms2);
/* What the last de_accumulate call ran (any pointer may be NULL): *variant = the kernel of its launches (6 = the per-CU stage scheduler, 2 = the
 * state machine, 0 = the ray marcher; in the legacy library also 1, 3 = the wavefront pipeline — with the state machine as finisher when
 * *depths < 25 —, 5 = the HBM-queue scheduler), *pipes = pipelines side by side (legacy; 1 otherwise), *depths = vertex rounds run in the
 * pipeline (legacy; 0 otherwise), *launches = launches the call was cut into.  For a host that reports what it measured. */
int de_last_call_info(de_ctx* ctx, int* variant, int* pipes, int* depths, int* launches);


/* ---- work counters (algorithmic bytes per sample) and kernel statistics */
/* work counters of the last de_accumulate calls since de_reset (algorithmic bytes per sample) */
typedef struct de_counters {
    uint64_t samples;          /* paths traced */
    uint64_t taps_r8;          /* bilinear taps of single-channel maps (4 texels = 4 B each) */
    uint64_t taps_rgb8;        /* bilinear taps of colour maps (4 texels, 16 B each as rgba8 in the reference) */
    uint64_t sphere_steps;     /* intersect_land iterations (pathtracer.py:37-44) */
    uint64_t tracking_steps;   /* delta + ratio tracking iterations (pathtracer.py:91-112,130-141) */
    uint64_t vertices;         /* path vertices (pathtracer.py:349) */
    uint64_t rng_draws;
    uint64_t reserved[9];
} de_counters;
int de_get_counters(de_ctx* ctx, de_counters* out);
int de_enable_counters(de_ctx* ctx, int enable);
/* Scheduler statistics of render_kernel_v2 gathered while counters are enabled (sums over waves since de_reset):
 * out[0..31] = lane / run counts per section, out[32..43] = shader cycles per section; n <= 48.  Index meaning:
 * tools/sched_stats.py.  Profiling aid only. */
int de_debug_sched_stats(de_ctx* ctx, uint64_t* out, int n);
/* The same for the per-CU stage scheduler (render_kernel_v6 with de_tuning.v6_stats = 1): [0..7] records taken per stage (sphere trace, gas, cloud, event,
 * surface, volume finish, surface finish, new paths), [8..15] wave-time per stage in 10 ns ticks, [16] idle scheduler rounds, [17] stage entries, [18] failed
 * claims of free slots, [19] wave-time in the scheduler, [20] yields of a loop stage, [21] records they handed back, [22] wave-time between the end of the work
 * items and the wave's exit, [32..38] wave-trips of the loop stages / passes of the straight-line stages, [40..46] the busy lanes summed over them;
 * [64..127] the region statistics, [128..191] the drain's population histogram (render_kernel_v6.hip).  Profiling aid. */
int de_debug_v6_stats(de_ctx* ctx, uint64_t* out, int n);

/* ---- parity-test hooks
 * de_debug_samples: trace sample `sample_index` of every pixel WITHOUT accumulating; per pixel (row-major
 * [H][W]) writes 4 floats: radiance, wavelength, number of RNG draws, number of path vertices.
 * de_debug_math: evaluate a deterministic f32 elementary function on the GPU (fn: 0 exp, 1 log, 2 sin, 3 cos,
 * 4 atan2(a,b), 5 asin, 6 pow(a,b), 7 a/b, 8 sqrt, 9 the short sqrt de_sqrt_nr, 10 the 3-operation a/b for literal divisors, 11 a * RN(1/b) (contract-2 quotient),
 * 12 log of a random draw de_log_unit, 13 the 3-operation reciprocal de_rcp_nr,
 * 14 exp of a non-positive argument de_exp_nonpos), for bit-exact comparison against the oracle. */
int de_debug_samples(de_ctx* ctx, uint64_t seed, int sample_index, float* out);
int de_debug_math(de_ctx* ctx, int fn, const float* a, const float* b, float* out, uint64_t n);
/* Read a map back as uploaded (row-major, `channels` bytes per texel) — lets tests hand the same texels to the oracle. */
int de_download_texture(de_ctx* ctx, int slot, uint8_t* out, uint64_t out_bytes);
int de_texture_info(de_ctx* ctx, int slot, int* width, int* height, int* channels);

/* ---- the N-rank collectives' device code on ONE GPU
 * de_debug_ordered_sum: the root's half of de_reduce_ordered without a communicator.  `parts` = n_parts host buffers of W*H*3 floats each (device
 * layout [H][W][3]), rank by rank: part `root` is loaded into the accumulation buffer itself, every other part into the gather buffer at its rank's
 * offset — where ncclRecv would have put it — and the SAME launch as de_reduce_ordered's (ordered_sum_kernel, the root's part read in place) adds them
 * in rank order, in place or into the assembled buffer (out_of_place != 0, which then is what the display reads).  `out` receives the result
 * ([H][W][3]).  2 <= n_parts <= 16. */
int de_debug_ordered_sum(de_ctx* ctx, const float* parts, int n_parts, int root, int out_of_place, float* out);
/* de_debug_standin_reduce: a one-GPU stand-in for the per-frame collective of an N-rank run, enqueued on the context stream exactly where de_reduce goes
 * (after the device-side wait for the frame's launches): one streaming kernel that reads the accumulation buffer and a second buffer of the same size and
 * writes a third (3 x 12 W H bytes of HBM traffic: what a sum-reduce does on the root), then `extra_copies` further device-to-device copies of
 * 12 W H bytes (the ingress of an 8-rank gather: 7).  The frame's bits are untouched.  de_last_reduce_ms then reports the time from the moment the stream
 * reached the collective (the frame's last accumulate_kernel done) to its end: queueing behind the next frame's persistent workgroups included. */
int de_debug_standin_reduce(de_ctx* ctx, int extra_copies);

/* ---- experiment switches (every one also a de_tuning field, or without effect on the product's default kernels) */
/* Which kernels de_accumulate launches.  Same results, bit for bit, whatever the choice.
 *   4 (default)  automatic: a call of at least 4096 paths runs the per-CU stage scheduler (variant 6) in one launch per 2^29 paths; smaller
 *                calls (and counting launches) run the state machine (variant 2), overlapping through the launch slots.  The choice
 *                depends on the call's size only and moves time, never bits.
 *   2            wave-level state machine only (render_kernel_v2.hip)
 *   6            per-CU stage scheduler: ONE launch (+ a small tail launch of the same kernel for big calls, de_tuning.v6_tail_*), two workgroups per CU, the stage queues in LDS, a path stays on its CU (render_kernel_v6.hip)
 * Variants 1 (per-lane loops), 3 (wavefront pipeline through HBM stage queues; with DE_AUTO_V6=0 the launch policy of rounds 3-4) and 5 (stage
 * scheduler over HBM queues) are further statements of path_tracer kept for the cross-check tests: the LEGACY library only
 * (libdigitalearth_hip_legacy.so, include/digital_earth_legacy.h); this library answers DE_ERR_INVALID to them.
 * The ray marcher always runs its own kernel. */
int de_set_kernel_variant(de_ctx* ctx, int variant);
/* Launches in flight per context (see de_accumulate): n_slots for small launches, n_big for launches with more than 96 MB of
 * per-sample records; 1 = every launch waits for the previous one.  Waits for the launches in flight. */
int de_set_launch_slots(de_ctx* ctx, int n_slots, int n_big);
/* render_kernel_v2 only (calls below 4096 paths, counting launches): resident waves per CU the kernel takes (1..20, default 20 = 5 per SIMD, all of the CU's vector registers).  A host that
 * overlaps a collective with the next frame's render leaves room for the collective's kernel: 16 = 4 per SIMD. */
int de_set_wave_budget(de_ctx* ctx, int waves_per_cu);
/* Memory the context may hold for the wavefront pipeline's stage queues and cold records (512 B per path of a batch, per pipe) and the
 * HBM-queue scheduler's pools (kernel variants 3 and 5, and variant 4 under DE_AUTO_V6=0; the default's per-CU scheduler holds 37 MB of cold records + 8 MB of tail pool per
 * launch slot whatever the call and is not affected): a large call is cut into batches that fit (split launches give the same bits; a batch of one sample per
 * pixel that does not fit runs on the state machine).  0 (default) = automatic: what the device has free minus 8 GB, at most 2^27 paths
 * per batch.  Setting it releases what is held.  profiles/memory_budget.md: throughput against the budget.
 * de_get_memory_use: bytes currently held for these buffers. */
int de_set_memory_budget(de_ctx* ctx, uint64_t bytes);
int de_get_memory_use(de_ctx* ctx, uint64_t* queue_bytes);

#ifdef __cplusplus
}
#endif
#endif /* DIGITAL_EARTH_DEBUG_H */
