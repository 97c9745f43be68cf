#!/usr/bin/env python3
"""Where the lanes of render_kernel_v6 go: the kernel's own statistics (DE_V6_STATS=1 build of the kernel: template flag, same arithmetic) for one
workload, as a markdown table — per stage: records, wave-time share, busy lanes per trip / pass, and per REGION of the stage (step body, epilogue,
ring appends, refill, yield, trip control; body / ring work for the straight-line stages) its share of the wave-time and the lanes its vector code runs for.

  DE_V6_STATS=1 python tools/v6_lanes.py [--preset NAME] [--width W --height H --spp S] [--cloud-heavy] [--out FILE.md] [--world N]

The last line is the time-weighted active-lane fraction the regions add up to, to be read next to the hardware's
SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU) of the same workload (profiles/*_summary.md)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("DE_V6_STATS", "1")
from digital_earth_amd.renderer import Renderer
from digital_earth_amd.earth_viewer import load_config

STAGES = ["sphere trace", "gas", "cloud", "event", "surface", "vol finish", "surf finish", "new paths"]
REGIONS = ["step body", "epilogue + write-back", "ring appends", "refill (claim, take, load, begin)", "yield (suspend, hand back)", "trip control"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset"); ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64); ap.add_argument("--cloud-heavy", action="store_true"); ap.add_argument("--out")
    ap.add_argument("--world", type=int, default=1); ap.add_argument("--tex-size", type=int, nargs=2)
    a = ap.parse_args()
    kw = dict(texture_source="synthetic", cloud_heavy=a.cloud_heavy)
    if a.tex_size: kw["texture_size"] = tuple(a.tex_size)
    r = Renderer((a.width, a.height), (0, 1, 0), **kw); r.copy_textures(); r.trim_textures()
    if a.preset: load_config(a.preset).apply(r)
    if a.world > 1: r.set_tile_partition(0, a.world)
    r.set_launch_slots(1, 1)                       # one launch slot: the statistics block read below is that of the timed launch alone
    r.accumulate(a.spp); r.synchronize()
    r.reset_framebuffer(); r.synchronize()
    t0 = time.perf_counter(); r.accumulate(a.spp); r.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    st = r.v6_stats(192)
    phases = r.last_launch_phases()
    if not any(st):
        raise SystemExit("no statistics: run with DE_V6_STATS=1")
    name = "%dx%d, %d spp, %s%s%s" % (a.width, a.height, a.spp, a.preset or "default camera", ", cloud-heavy maps" if a.cloud_heavy else "", ", 1/%d share" % a.world if a.world > 1 else "")
    L = ["# `render_kernel_v6` statistics — %s" % name, "",
         "One launch, %.1f ms (statistics build of the kernel: ~8 %% slower than the shipped instantiation; shares, not times, are what to read)." % ms, "",
         "| stage | records | wave-time | busy lanes per trip / pass | trips / passes | us per trip |", "|---|---|---|---|---|---|"]
    tot = sum(st[8:16]) + st[19]
    for k, nm in enumerate(STAGES):
        trips = st[32 + k] if k < 7 else st[64 + 61]
        lu = "%.3f" % (st[40 + k] / 64.0 / max(st[32 + k], 1)) if k < 7 else "%.3f" % (st[7] / 64.0 / max(st[64 + 61], 1))
        L.append("| %s | %d | %.1f %% | %s | %d | %.2f |" % (nm, st[k], 100.0 * st[8 + k] / max(tot, 1), lu, trips, st[8 + k] * 0.01 / max(trips, 1)))
    L.append("| scheduler | | %.1f %% | idle rounds %d, stage entries %d, failed free-slot claims %d | | |" % (100.0 * st[19] / max(tot, 1), st[16], st[17], st[18]))
    if any(st[24:31]):
        L += ["", "Stage entries (calls of the stage function; each re-reads its configuration from LDS; until round 6 it also saved and restored up to 24 callee-saved VGPRs through scratch): "
              + ", ".join("%s %d (%.0f records per entry)" % (STAGES[k], st[24 + k], st[k] / max(st[24 + k], 1)) for k in range(7)) + "."]
    L += ["", "Yields: %d, handing back %d records (%.1f %% of the loop stages' record hops).  Drain: %.2f ms per wave between the end of the work items and the wave's exit."
          % (st[20], st[21], 100.0 * st[21] / max(st[0] + st[1] + st[2], 1), st[22] * 1e-5 / max(1, 12 * 512)), ""]
    # regions
    r2 = st[64:]
    ticks = sum(r2[14 * q + 2 * k] for q in range(3) for k in range(6)) + sum(r2[42 + 4 * j] + r2[42 + 4 * j + 2] for j in range(4)) + r2[58] + r2[60]
    lane_ticks = sum(r2[14 * q + 2 * k + 1] for q in range(3) for k in range(6)) + sum(r2[42 + 4 * j + 1] for j in range(4)) + r2[59]
    L += ["## Regions (shader-clock ticks between region boundaries, summed over all waves; lanes = the lanes the region's vector code is executed for)", "",
          "| stage | region | share of all stage time | lanes (of 64) | per service / trip |", "|---|---|---|---|---|"]
    for q in range(3):
        b = 14 * q
        sv, tr = r2[b + 12], r2[b + 13]
        for k, nm in enumerate(REGIONS):
            t, l = r2[b + 2 * k], r2[b + 2 * k + 1]
            per = ""
            if k == 0: per = "%.0f ticks per trip" % (t / max(tr, 1))
            elif k in (1, 2, 3): per = "%.0f ticks per service" % (t / max(sv, 1))
            elif k == 5: per = "%.0f ticks per trip" % (t / max(tr, 1))
            L.append("| %s | %s | %.2f %% | %s | %s |" % (STAGES[q], nm, 100.0 * t / max(ticks, 1), ("%.1f" % (l / max(t, 1))) if k != 5 else "-", per))
        L.append("| %s | *services: %d, one per %.1f trips; stage total %.2f %%* | | | |" % (STAGES[q], sv, tr / max(sv, 1), 100.0 * sum(r2[b + 2 * k] for k in range(6)) / max(ticks, 1)))
    for j in range(4):
        b = 42 + 4 * j
        L.append("| %s | body | %.2f %% | %.1f | %.0f ticks per pass |" % (STAGES[3 + j], 100.0 * r2[b] / max(ticks, 1), r2[b + 1] / max(r2[b], 1), r2[b] / max(r2[b + 3], 1)))
        L.append("| %s | ring work (claim, take, load / store, appends) | %.2f %% | - | %.0f ticks per pass |" % (STAGES[3 + j], 100.0 * r2[b + 2] / max(ticks, 1), r2[b + 2] / max(r2[b + 3], 1)))
    L.append("| new paths | body (camera ray, wavelength, primary sphere trace) | %.2f %% | %.1f | %.0f ticks per pass |" % (100.0 * r2[58] / max(ticks, 1), r2[59] / max(r2[58], 1), r2[58] / max(r2[61], 1)))
    L.append("| new paths | ring work | %.2f %% | - | %.0f ticks per pass |" % (100.0 * r2[60] / max(ticks, 1), r2[60] / max(r2[61], 1)))
    L += ["", "Time-weighted active lanes over all regions: **%.3f** of 64 (regions without vector work count as 0 lanes; divergence INSIDE a step body or a stage body is not seen here — "
          "the hardware counter of the same workload is the product of the two)." % (lane_ticks / 64.0 / max(ticks, 1)), ""]
    # the drain: population of the resident workgroups in 0.25 ms buckets after the work counter ran out
    dr = st[128:192]
    if any(dr):
        n_wg = 512      # workgroup slots of an MI355X (2 per CU); with the tail chain the first bucket also counts the tail launch's 64 workgroups
        L += ["## Drain (the launch's own clock: %.2f ms in all, %.2f ms after the last work item was handed out)" % phases, "",
              "Sampled by the first wave of each workgroup that passes the scheduler in a 0.25 ms bucket.  `slot-time held` = resident workgroups x bucket; `slot-time needed` = paths alive / %d slots x bucket: what "
              "the same paths would hold if they sat in full workgroups." % 1120, "",
              "| ms after exhaustion | resident workgroups (frame launch + tail launch) | paths alive in them | per resident workgroup | fill |", "|---|---|---|---|---|"]
        held = need = 0.0
        for k in range(32):
            if dr[32 + k] == 0: continue
            L.append("| %.2f%s | %d | %d | %.0f | %.3f |" % (0.25 * k, "+" if k == 31 else "", dr[32 + k], dr[k], dr[k] / dr[32 + k], dr[k] / dr[32 + k] / 1120.0))
            held += 0.25 * dr[32 + k]; need += 0.25 * dr[k] / 1120.0
        L += ["", "Workgroup-slot time held during the drain: **%.2f ms x %d slots**; needed at full workgroups: %.2f ms x %d slots." % (held / n_wg, n_wg, need / n_wg, n_wg), ""]
    text = "\n".join(L)
    print(text)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(text)


if __name__ == "__main__":
    main()
