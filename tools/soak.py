#!/usr/bin/env python3
"""Soak run of the shipped kernel: many frames back to back on one context — changing cameras, sample counts, seeds, partitions, calls in flight on every
launch slot, the tail chain exporting whenever a call is queued behind another, (round 6) pipelined fetches between the calls and CU-masked render streams switched on and off — and
after every batch a fetch (where an aborted launch would surface as DE_ERR_HIP) and,
for one frame of the batch, a second render of the same seed on a FRESH context with one launch per call: the two must agree bit for bit.

  python tools/soak.py [--minutes 3] [--width 1920 --height 1080]        -> a one-line summary per batch, a total at the end (exit code 1 on any mismatch)"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from digital_earth_amd.renderer import Renderer
from digital_earth_amd.earth_viewer import load_config

PRESETS = ["config - florida.txt", "config - sunset hurricane.txt", "config - Apollo 11.txt"]      # each sets every scalar parameter: the long-lived context and the fresh one agree


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=3.0); ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
    a = ap.parse_args()
    rng = np.random.default_rng(20240127)
    kw = dict(texture_source="synthetic", texture_size=(8192, 4096))
    r = Renderer((a.width, a.height), (0, 1, 0), **kw); r.copy_textures()
    t_end = time.time() + 60.0 * a.minutes
    frames = paths = checks = bad = 0
    batch = 0
    while time.time() < t_end:
        batch += 1
        preset = PRESETS[int(rng.integers(len(PRESETS)))]
        seed = int(rng.integers(1 << 30))
        calls = [int(x) for x in rng.integers(1, 24, size=int(rng.integers(2, 9)))]
        part = (int(rng.integers(0, 3)), 3) if rng.random() < 0.3 else (0, 1)
        def setup(x):
            cfg = load_config(preset)
            cfg.apply(x)
            x.set_fov(float(cfg.fov) * float(rng2.uniform(0.8, 1.25)))
            x.set_sun_angle(float(cfg.sun_angle) + float(rng2.uniform(-0.5, 0.5)))
            x.seed = seed
            x.set_tile_partition(*part)
        rng2 = np.random.default_rng(seed)
        setup(r)
        r.reset_framebuffer()
        if batch % 7 == 3:                          # (round 6) now and then the render streams leave a CU per XCD alone: another grid size, recreated streams
            t = r.tuning(); t.v6_cu_withhold = 1 - t.v6_cu_withhold; r.set_tuning(t)
        lag = int(rng.integers(0, 4)) if batch % 3 == 0 else 0
        for n in calls:
            r.accumulate(n)                         # nothing waits in between: the calls queue up on all launch slots
            if lag:
                r.fetch_image(copy=False, lag=lag)  # (round 6) the pipelined window loop's display + host copy between the calls, up to three in flight
        if lag:
            shown = r.fetch_pending(copy=True)
            assert shown is not None and np.isfinite(shown).all()
        got = r.fetch_hdr()                         # an aborted launch surfaces here
        frames += 1; paths += a.width * a.height * sum(calls) // part[1]
        line = "batch %3d  %-32s spp %-28s partition %d/%d  max %.3g" % (batch, preset, "+".join(map(str, calls)), part[0], part[1], float(got.max()))
        if batch % 4 == 1:
            ref = Renderer((a.width, a.height), (0, 1, 0), **kw); ref.copy_textures()
            t = ref.tuning(); t.v6_tail_levels = 0; t.launch_slots = 1; t.big_launch_slots = 1; ref.set_tuning(t)
            rng2 = np.random.default_rng(seed)
            setup(ref)
            ref.accumulate(sum(calls))              # one call, one launch per call, one at a time
            want = ref.fetch_hdr(); ref.close()
            # the sum of a pixel's samples is taken in sample order by accumulate_kernel whatever the split into calls: the same bits
            same = bool((got.view(np.uint32) == want.view(np.uint32)).all())
            checks += 1; bad += 0 if same else 1
            line += "  | fresh context, one launch: %s" % ("identical" if same else "DIFFERENT")
        print(line, flush=True)
    print("soak: %d frames, %.2f G paths, %d cross-checks, %d mismatches, %.1f min" % (frames, paths / 1e9, checks, bad, a.minutes), flush=True)
    r.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
