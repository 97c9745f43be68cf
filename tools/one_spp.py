#!/usr/bin/env python3
"""Small launches: k x accumulate(1) against accumulate(k) (the reference's interactive loop is accumulate() x 1
per frame, earth_viewer.py:241-243).  Run with DE_SLOTS=1 for the serial behaviour of round 2.

  python tools/one_spp.py [--check]      --check: the HDR sums of the k x 1 render equal those of the 1 x k render bit for bit
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from digital_earth_amd.renderer import Renderer

W, H = 1920, 1080
r = Renderer((W, H), (0, 1, 0), texture_source="synthetic")
r.copy_textures()
r.reset_framebuffer(); r.accumulate(16); r.synchronize()           # allocations, first-launch costs


def timed(spp, n):
    r.reset_framebuffer(); r.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r.accumulate(spp)
    r.synchronize()
    return (time.perf_counter() - t0) * 1e3


print("launch slots: DE_SLOTS=%s" % os.environ.get("DE_SLOTS", "default"))
base = {}
for total in (16, 64):
    base[total] = min(timed(total, 1) for _ in range(3))
    print("accumulate(%d) x 1: %.1f ms = %.0f Msamples/s" % (total, base[total], W * H * total / base[total] / 1e3))
for spp, total in ((1, 16), (4, 16), (1, 64), (4, 64), (16, 64)):
    n = total // spp
    ms = min(timed(spp, n) for _ in range(3))
    print("accumulate(%d) x %d: %.1f ms = %.0f Msamples/s, %.2fx the time of accumulate(%d)" % (spp, n, ms, W * H * total / ms / 1e3, ms / base[total], total))
t0 = time.perf_counter(); r.reset_framebuffer(); r.accumulate(1); img = r.fetch_image(); t1 = time.perf_counter()
print("reset + accumulate(1) + fetch_image (host copy): %.2f ms" % ((t1 - t0) * 1e3))
# the reference's window loop (earth_viewer.py:241-243, 316-317): accumulate() x 1, then fetch_image(), frame after frame
for _ in range(3):
    r.accumulate(1); r.fetch_image()
t0 = time.perf_counter()
for _ in range(30):
    r.accumulate(1); img = r.fetch_image()
t1 = time.perf_counter()
print("interactive loop, accumulate(1) + fetch_image per frame: %.2f ms (%s)" % ((t1 - t0) * 1e3 / 30, r.last_call_info()["kernel"]))
for _ in range(3):
    r.accumulate(1); r.fetch_image(copy=False)
t0 = time.perf_counter()
for _ in range(30):
    r.accumulate(1); img = r.fetch_image(copy=False)
t1 = time.perf_counter()
print("the same with fetch_image(copy=False) (a view of the pinned staging buffer): %.2f ms" % ((t1 - t0) * 1e3 / 30))
# the same loop PIPELINED (round 6: de_fetch_image_begin / _end): the display + host copy of frame k are enqueued, frame k + 1 is issued at once and the
# image handed to the canvas is frame k - 1's — with and without CUs withheld from the render streams for the display kernel and the copy
def pipelined(n, lag):
    r.fetch_pending()
    for _ in range(6):
        r.accumulate(1); r.fetch_image(copy=False, lag=lag)
    t0 = time.perf_counter()
    for _ in range(n):
        r.accumulate(1); img = r.fetch_image(copy=False, lag=lag)
    dt = (time.perf_counter() - t0) * 1e3 / n
    r.fetch_pending(copy=False)
    return dt


for lag in (1, 2, 3):
    print("pipelined loop, accumulate(1) + fetch_image(copy=False, lag=%d) per frame (the image shown is %d frame%s old): %.2f ms" % (lag, lag, "" if lag == 1 else "s", min(pipelined(90, lag) for _ in range(3))))
if "--check" in sys.argv:
    r.reset_framebuffer(); r.accumulate(16); a = r.fetch_hdr()
    r.reset_framebuffer()
    for _ in range(16):
        r.accumulate(1)
    b = r.fetch_hdr()
    print("16 x accumulate(1) vs accumulate(16): bit-identical fraction %.6f" % float((a.view(np.uint32) == b.view(np.uint32)).mean()))
