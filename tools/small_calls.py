#!/usr/bin/env python3
"""Where the per-CU stage scheduler starts to pay: calls of one sample per pixel at several image sizes, kernel variant 2 (state machine) against 6,
(a) the reference's window loop — accumulate(1) then fetch_image() — and (b) 32 calls back to back.   python tools/small_calls.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from digital_earth_amd.renderer import Renderer

for (W, H) in ((64, 64), (128, 64), (256, 128), (512, 256), (640, 360), (1024, 512), (1280, 720)):
    row = []
    for variant in (2, 6):
        r = Renderer((W, H), (0, 1, 0), texture_source="synthetic", texture_size=(8192, 4096))
        r.set_kernel_variant(variant)
        for _ in range(3):
            r.accumulate(1); r.fetch_image()
        t0 = time.perf_counter()
        for _ in range(30):
            r.accumulate(1); r.fetch_image()
        loop = (time.perf_counter() - t0) * 1e3 / 30
        r.reset_framebuffer(); r.synchronize()
        t0 = time.perf_counter()
        for _ in range(32):
            r.accumulate(1)
        r.synchronize()
        b2b = (time.perf_counter() - t0) * 1e3 / 32
        row.append((loop, b2b))
        del r
    print("%4dx%-4d = %7d paths per call | window loop ms per frame: state machine %.2f, per-CU scheduler %.2f | back to back ms per call: %.2f, %.2f"
          % (W, H, W * H, row[0][0], row[1][0], row[0][1], row[1][1]), flush=True)
