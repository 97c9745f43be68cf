"""One frame of the reference's window loop at 1920x1080, split: the accumulate(1) call, waiting for it, fetch_image(copy=False).   python tools/loop_breakdown.py"""
import sys, time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from digital_earth_amd.renderer import Renderer
r = Renderer((1920, 1080), (0, 1, 0), texture_source="synthetic"); r.copy_textures()
for _ in range(5): r.accumulate(1); r.fetch_image(copy=False)
N = 30
t = [0, 0, 0]
for _ in range(N):
    t0 = time.perf_counter(); r.accumulate(1); t1 = time.perf_counter(); r.synchronize(); t2 = time.perf_counter(); r.fetch_image(copy=False); t3 = time.perf_counter()
    t[0] += t1 - t0; t[1] += t2 - t1; t[2] += t3 - t2
print("accumulate(1) call %.2f ms, wait for it %.2f ms, fetch_image(copy=False) %.2f ms" % tuple(x * 1e3 / N for x in t))
print("last accumulate GPU ms:", r.last_accumulate_ms())
