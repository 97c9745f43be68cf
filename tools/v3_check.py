#!/usr/bin/env python3
"""render_kernel_v3 (wavefront pipeline) against render_kernel_v2: bit-identity on small frames, then timing of the bench frame."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from digital_earth_amd.renderer import Renderer
from digital_earth_amd.earth_viewer import load_config

def frame(variant, W, H, spp, preset=None, fov=None, **kw):
    r = Renderer((W, H), (0, 1, 0), seed=11, **kw)
    if preset: load_config(preset).apply(r)
    if fov: r.set_fov(fov)
    r.set_kernel_variant(variant)
    r.accumulate(spp)
    return r.fetch_hdr()

ok = True
for name, kw in (("default", dict()), ("wide", dict(fov=0.45)), ("florida", dict(preset="config - florida.txt")),
                 ("sunset", dict(preset="config - sunset hurricane.txt")), ("apollo", dict(preset="config - Apollo 11.txt"))):
    a = frame(2, 128, 64, 6, texture_source="synthetic", texture_size=(1024, 512), **kw)
    b = frame(3, 128, 64, 6, texture_source="synthetic", texture_size=(1024, 512), **kw)
    same = float((a.view(np.uint32) == b.view(np.uint32)).mean())
    print("%-8s v3 == v2 bit-identical fraction %.6f (max %g)" % (name, same, a.max()))
    ok = ok and same == 1.0
if "--time" in sys.argv:
    W, H, SPP = 1920, 1080, 64
    r = Renderer((W, H), (0, 1, 0), texture_source="synthetic"); r.copy_textures(); r.trim_textures()
    for variant in (2, 3):
        r.set_kernel_variant(variant)
        r.reset_framebuffer(); r.accumulate(SPP); r.synchronize()
        ts = []
        for _ in range(3):
            r.reset_framebuffer(); r.synchronize()
            t0 = time.perf_counter(); r.accumulate(SPP); r.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        hdr = r.fetch_hdr()
        print("variant %d: %s ms per 64-spp frame -> %.0f Msamples/s" % (variant, ["%.1f" % t for t in ts], W * H * SPP / min(ts) / 1e3))
        if variant == 2: ref = hdr
        else: print("full frame v3 == v2: %.6f" % float((hdr.view(np.uint32) == ref.view(np.uint32)).mean()))
sys.exit(0 if ok else 1)
