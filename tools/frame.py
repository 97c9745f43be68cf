#!/usr/bin/env python3
"""Two 1920x1080 frames under the kernel variant of DE_KERNEL (default 5, the persistent stage scheduler): the workload of tools/v5_pmc.sh, v3_pmc.sh, v3_run.sh, cloud_l2.sh.  usage: frame.py [spp] [preset]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from digital_earth_amd.renderer import Renderer
from digital_earth_amd.earth_viewer import load_config
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
r = Renderer((1920, 1080), (0, 1, 0), texture_source="synthetic"); r.copy_textures(); r.trim_textures()
if len(sys.argv) > 2: load_config(sys.argv[2]).apply(r)
r.set_kernel_variant(int(os.environ.get("DE_KERNEL", "5")))
for _ in range(2):
    r.reset_framebuffer(); r.synchronize()
    t0 = time.perf_counter(); r.accumulate(spp); r.synchronize()
    print("frame %.1f ms" % ((time.perf_counter() - t0) * 1e3))
