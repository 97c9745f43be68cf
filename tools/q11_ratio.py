#!/usr/bin/env python3
"""how much does the (Taichi-internal, unverifiable) sampler address mode change the image?  Renders BASELINE
configs[1] with REPEAT (default) and CLAMP and prints the ratio of mean radiance and the relative L2 difference."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from digital_earth_amd.renderer import Renderer
from digital_earth_amd import _native
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
out = {}
for name, clamp in (("repeat", False), ("clamp", True)):
    r = Renderer((1920, 1080), (0, 1, 0), texture_source="synthetic")
    r.set_flag(_native.DE_FLAG_CLAMP_SAMPLER, clamp)
    r.copy_textures(); r.accumulate(spp)
    out[name] = (r.fetch_hdr().astype(np.float64), r.fetch_image().astype(np.float64))
    r.close()
a, b = out["repeat"][0], out["clamp"][0]
print("mean HDR radiance: repeat %.6g clamp %.6g ratio clamp/repeat %.4f" % (a.mean(), b.mean(), b.mean() / a.mean()))
print("relative L2 of (clamp/ratio - repeat): %.4g" % (np.sqrt(((b * (a.mean() / b.mean()) - a) ** 2).sum()) / np.sqrt((a ** 2).sum())))
print("displayed image mean: repeat %.4f clamp %.4f" % (out["repeat"][1].mean(), out["clamp"][1].mean()))
