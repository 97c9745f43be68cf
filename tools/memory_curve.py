#!/usr/bin/env python3
"""Throughput against the memory budget of the pipeline's queues (de_set_memory_budget): BASELINE cfg2 and cfg4, pipelined frames.

  DE_AUTO_V6=0 python tools/memory_curve.py [--cfg4]        -> one line per budget (the pipeline policy; the default holds 37 MB whatever the budget) (GB; 0 = automatic): ms per frame, bytes held, launches per call
"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from digital_earth_amd.renderer import Renderer
from digital_earth_amd.earth_viewer import load_config

cfg4 = "--cfg4" in sys.argv
W, H, SPP, N = (3840, 2160, 128, 2) if cfg4 else (1920, 1080, 64, 6)
r = Renderer((W, H), (0, 1, 0), texture_source="synthetic", cloud_heavy=cfg4)
r.copy_textures(); r.trim_textures()
if cfg4: load_config("config - sunset hurricane.txt").apply(r)
print("%s: %dx%d x %d spp, %d frames back to back" % ("BASELINE cfg4 (cloud-heavy, sunset hurricane)" if cfg4 else "BASELINE cfg2", W, H, SPP, N))
for gb in (0, 68, 48, 32, 24, 16, 12, 8, 6):
    r.set_memory_budget(gb << 30)
    for _ in range(3):
        r.reset_framebuffer(); r.accumulate(SPP)
    r.synchronize()
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        for _ in range(N):
            r.reset_framebuffer(); r.accumulate(SPP)
        r.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3 / N)
    info = r.last_call_info()
    print("budget %4s GB: %8.1f ms per frame = %6.1f Msamples/s | held %5.1f GB | %s, %d launches per call"
          % (gb or "auto", best, W * H * SPP / best / 1e3, r.memory_use() / 2**30, info["kernel"], info["launches"]), flush=True)
