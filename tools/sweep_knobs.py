#!/usr/bin/env python3
"""Joint sweep of render_kernel_v2's scheduler knobs (DE_V2_* env overrides, read at de_create) on the bench workload.
One process: every configuration is a fresh context that borrows the maps of the first (de_share_textures).
  python tools/sweep_knobs.py THR=14,18 A=10,13 G=6,10 B=24 [--spp 64] [--preset name]"""
import itertools, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from digital_earth_amd.renderer import Renderer

spp, preset, axes = 64, None, {}
args = sys.argv[1:]
while args:
    a = args.pop(0)
    if a == "--spp": spp = int(args.pop(0))
    elif a == "--preset": preset = args.pop(0)
    else:
        k, v = a.split("="); axes[k] = [int(x) for x in v.split(",")]
base = Renderer((1920, 1080), (0, 1, 0), texture_source="synthetic")
base.copy_textures()
names = {"THR": "DE_V2_THR", "A": "DE_V2_A", "G": "DE_V2_G", "B": "DE_V2_B", "CHUNK": "DE_V2_CHUNK", "WPC": "DE_V2_WPC"}
keys = list(axes)
for combo in itertools.product(*[axes[k] for k in keys]):
    for k, v in zip(keys, combo):
        os.environ[names[k]] = str(v)
    r = Renderer((1920, 1080), (0, 1, 0), texture_source="synthetic")
    if preset:
        from digital_earth_amd.earth_viewer import load_config
        load_config(preset).apply(r)
    r.share_textures_from(base)
    ms = []
    for _ in range(2):
        r.reset_framebuffer(); r.accumulate(spp); r.synchronize(); ms.append(r.last_accumulate_ms())
    print(" ".join("%s=%d" % kv for kv in zip(keys, combo)), "kernel ms %.1f" % min(ms), flush=True)
    r.close()
