import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from digital_earth_amd.renderer import Renderer
W, H, SPP = 1920, 1080, 64
r = Renderer((W, H), (0, 1, 0), texture_source="synthetic"); r.copy_textures(); r.trim_textures()
if len(sys.argv) > 1: r.set_wave_budget(int(sys.argv[1]))
for _ in range(2):
    r.reset_framebuffer(); r.accumulate(SPP); r.synchronize()
ts = []
for _ in range(3):
    r.reset_framebuffer(); r.synchronize()
    t0 = time.perf_counter(); r.accumulate(SPP); r.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
knobs = " ".join("%s=%s" % kv for kv in sorted(os.environ.items()) if kv[0].startswith("DE_"))
print("single frames, 1920x1080 x 64 spp [%s]: %s ms -> %.0f Msamples/s" % (knobs, ["%.1f" % t for t in ts], W * H * SPP / min(ts) / 1e3))
