#!/usr/bin/env python3
"""render_kernel_v6 (per-CU stage scheduler, queues in LDS) against render_kernel_v2: bit-identity on small frames, then timing of
a frame with the scheduler's statistics block.

  python tools/v6_try.py [--small] [--time [SPP]] [--preset NAME] [--ref VARIANT] [--frames N] [--pipelined N]
Environment: DE_V6_STATS=1 for the statistics block.
"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from digital_earth_amd.renderer import Renderer
from digital_earth_amd.earth_viewer import load_config

STAGES = ["sphere trace", "gas", "cloud", "event", "surface", "vol finish", "surf finish", "new paths"]


def arg(name, default=None, cast=str):
    if name in sys.argv:
        i = sys.argv.index(name)
        if i + 1 < len(sys.argv) and not sys.argv[i + 1].startswith("--"):
            return cast(sys.argv[i + 1])
        return default if default is not None else True
    return None


def frame(variant, W, H, spp, preset=None, fov=None, **kw):
    r = Renderer((W, H), (0, 1, 0), seed=11, **kw)
    if preset: load_config(preset).apply(r)
    if fov: r.set_fov(fov)
    r.set_kernel_variant(variant)
    r.accumulate(spp)
    return r.fetch_hdr()


def print_stats(r, ms):
    st = r.v6_stats()
    if not any(st):
        return
    tot = sum(st[8:16]) + st[19]
    print("  stage            records      wave-time share   busy lanes per trip (loop stages) / pass (straight-line stages)")
    for k, name in enumerate(STAGES):
        lu = ("%.3f   %d trips, %.2f us per trip" % (st[40 + k] / 64.0 / max(st[32 + k], 1), st[32 + k], st[8 + k] * 0.01 / max(st[32 + k], 1))) if k < 7 else ""
        print("  %-14s %10d   %6.1f %%   %s" % (name, st[k], 100.0 * st[8 + k] / max(tot, 1), lu))
    print("  %-14s %10s   %6.1f %%   (idle rounds %d, stage entries %d, failed free-slot claims %d, yields %d handing back %d records)"
          % ("scheduler", "", 100.0 * st[19] / max(tot, 1), st[16], st[17], st[18], st[20], st[21]))
    print("  drain: %.2f ms per wave and launch between the end of the work items and the wave's exit (two launches in these statistics)" % (st[22] * 1e-5 / 6144 / 2))
    print("  wave-time total %.1f ms x waves (launch %.1f ms x 6144 waves = %.1f)" % (tot * 10e-6, ms, ms * 6144))


ok = True
if "--small" in sys.argv or "--time" not in sys.argv:
    for name, kw in (("default", dict()), ("wide", dict(fov=0.45)), ("florida", dict(preset="config - florida.txt")),
                     ("sunset", dict(preset="config - sunset hurricane.txt")), ("apollo", dict(preset="config - Apollo 11.txt"))):
        a = frame(2, 128, 64, 6, texture_source="synthetic", texture_size=(1024, 512), **kw)
        t0 = time.perf_counter()
        b = frame(6, 128, 64, 6, texture_source="synthetic", texture_size=(1024, 512), **kw)
        same = float((a.view(np.uint32) == b.view(np.uint32)).mean())
        print("%-8s v6 == v2 bit-identical fraction %.6f (max %g) %.2f s" % (name, same, a.max(), time.perf_counter() - t0), flush=True)
        ok = ok and same == 1.0
if "--time" in sys.argv:
    W, H = 1920, 1080
    SPP = arg("--time", 64, int)
    if SPP is True: SPP = 64
    preset = arg("--preset")
    ref_variant = arg("--ref", 2, int) or 2
    frames = arg("--frames", 3, int) or 3
    piped = arg("--pipelined", 0, int) or 0
    r = Renderer((W, H), (0, 1, 0), texture_source="synthetic"); r.copy_textures(); r.trim_textures()
    if preset: load_config(preset).apply(r)
    world = arg("--world", 1, int) or 1
    if world > 1: r.set_tile_partition(0, world)
    for variant in (ref_variant, 6):
        r.set_kernel_variant(variant)
        r.reset_framebuffer(); r.accumulate(SPP); r.synchronize()
        ts = []
        for _ in range(frames):
            r.reset_framebuffer(); r.synchronize()
            t0 = time.perf_counter(); r.accumulate(SPP); r.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        hdr = r.fetch_hdr()
        print("variant %d: %s ms per %d-spp frame -> %.0f Msamples/s" % (variant, ["%.1f" % t for t in ts], SPP, W * H * SPP / min(ts) / 1e3), flush=True)
        if piped:
            t0 = time.perf_counter()
            for _ in range(piped):
                r.reset_framebuffer(); r.accumulate(SPP)
            r.synchronize()
            print("variant %d: %.1f ms per frame over %d frames back to back" % (variant, (time.perf_counter() - t0) * 1e3 / piped, piped), flush=True)
        if variant == ref_variant: ref = hdr
        else:
            same = float((hdr.view(np.uint32) == ref.view(np.uint32)).mean())
            print("full frame v6 == v%d: %.6f" % (ref_variant, same))
            ok = ok and same == 1.0
            print_stats(r, min(ts))
sys.exit(0 if ok else 1)
