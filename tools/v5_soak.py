#!/usr/bin/env python3
"""Soak test of the persistent stage scheduler: N frames of each of several shapes, every frame compared bit for bit with the state machine's
(the hand-over between workgroups is concurrency: a rare stale record or lost chunk shows up as a wrong pixel or as the abort word).
  python tools/v5_soak.py [frames]        DE_SOAK_VARIANT=6: the per-CU stage scheduler (render_kernel_v6.hip) instead"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from digital_earth_amd.renderer import Renderer
from digital_earth_amd.earth_viewer import load_config

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
bad = 0
t0 = time.time()
for (W, H, spp, preset, heavy) in ((1920, 1080, 1, None, False), (640, 360, 7, "config - sunset hurricane.txt", True), (1920, 1080, 8, "config - florida.txt", False),
                                   (256, 128, 3, None, False), (3840, 2160, 2, "config - sunset hurricane.txt", True)):
    ref = Renderer((W, H), (0, 1, 0), texture_source="synthetic", texture_size=(8192, 4096), cloud_heavy=heavy, seed=3)
    if preset: load_config(preset).apply(ref)
    ref.set_kernel_variant(2)
    r = Renderer((W, H), (0, 1, 0), texture_source="synthetic", texture_size=(8192, 4096), cloud_heavy=heavy, seed=3)
    r.share_textures_from(ref)
    if preset: load_config(preset).apply(r)
    r.set_kernel_variant(int(os.environ.get("DE_SOAK_VARIANT", "5")))
    ref.accumulate(spp); want = ref.fetch_hdr().view(np.uint32)
    n_bad = 0
    for k in range(N):
        r.reset_framebuffer(); r.accumulate(spp)
        got = r.fetch_hdr().view(np.uint32)
        if not (got == want).all():
            n_bad += 1
            print("  frame %d of %dx%dx%d differs in %d words" % (k, W, H, spp, int((got != want).sum())), flush=True)
    print("%4dx%-4d x %d spp, %s: %d frames, %d wrong (%.0f s)" % (W, H, spp, preset or "default camera", N, n_bad, time.time() - t0), flush=True)
    bad += n_bad
    del r, ref
sys.exit(1 if bad else 0)
