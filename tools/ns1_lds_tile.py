#!/usr/bin/env python3
"""NS-1 experiment: the primary sphere traces of render_kernel_v3's v3_primary_kernel with the topography taps staged through an
LDS tile (build/ab/v3_lds*.so, -DDE_V3_LDS_TILE=<tiles>) against the footprint-packed dword tap (product build).
Run under rocprofv3 --kernel-trace / --pmc by tools/ns1_lds_tile.sh; prints bit-identity and the staging statistics."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from digital_earth_amd.renderer import Renderer
r = Renderer((1920, 1080), (0, 1, 0), texture_source="synthetic"); r.copy_textures(); r.trim_textures()
r.set_kernel_variant(3)
r.reset_framebuffer(); r.accumulate(4); r.synchronize()
st = r.sched_stats(48)[40:44]
hdr = r.fetch_hdr()
np.save("/tmp/ns1_%s.npy" % os.path.basename(os.environ.get("DE_LIB_PATH", "product")), hdr)
if st[0]:
    print("staging: %d wave-steps, %.1f %% staged, %.1f map tiles (128 B) staged per staged step for %.1f tapping lanes" % (st[0], 100.0 * st[1] / st[0], st[2] / max(st[1], 1), st[3] / max(st[1], 1)))
