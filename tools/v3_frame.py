import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from digital_earth_amd.renderer import Renderer
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 3
r = Renderer((1920, 1080), (0, 1, 0), texture_source="synthetic"); r.copy_textures(); r.trim_textures()
r.set_kernel_variant(variant)
for _ in range(2):
    r.reset_framebuffer(); r.accumulate(spp); r.synchronize()
print("ms", r.last_accumulate_ms())
