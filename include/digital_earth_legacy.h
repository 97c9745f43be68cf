/* digital_earth_legacy.h — entry points that exist only in libdigitalearth_hip_legacy.so (the product library built with
 * -DDE_LEGACY_VARIANTS: kernel variants 1, 3 and 5 and the launch policy of rounds 3-4, kept as independent statements of path_tracer for the
 * cross-check tests; digital_earth_amd/build.py: build_legacy).  Everything of digital_earth.h is there too. */
#ifndef DIGITAL_EARTH_LEGACY_H
#define DIGITAL_EARTH_LEGACY_H
#include "digital_earth.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Statistics block of the persistent stage scheduler (kernel variant 5, environment DE_V5_STATS=1), summed over the launch slots'
 * last launches; n <= 40: [0..7] records taken per stage (sphere trace, gas, cloud, event, surface, volume finish, surface finish,
 * new paths), [8..15] wave-time per stage in 10 ns ticks, [16] idle scheduler rounds, [17] partial chunks published, [18] failed
 * claims, [19] stage entries, [20] refills from the device-wide free ring, [21] steals, [22] wave-time in the scheduler, [23] in next_input, [24] in pushes, [25] in loop-stage services, [26] in straight-line bodies, [27..30] of the scheduler's: free-chunk refill, age flush, queue snapshot, claim, [32..34] wave-trips of the three loop
 * stages, [35..37] their lane-trips, [38] times a loop stage found its queue dry.  Profiling aid. */
int de_debug_v5_stats(de_ctx* ctx, uint64_t* out, int n);
#ifdef __cplusplus
}
#endif
#endif
