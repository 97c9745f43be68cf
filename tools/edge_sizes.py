"""Odd image shapes, call patterns and tile partitions under the default kernel variant against the state machine, bit for bit (sizes are multiples of (16, 8): renderer.py:46).  python tools/edge_sizes.py"""
import sys, itertools
sys.path.insert(0, "/root/repo")
import numpy as np
from digital_earth_amd.renderer import Renderer
bad = 0
for (W, H, spp_calls) in ((16, 8, [1]), (80, 40, [3, 1]), (208, 104, [7]), (1008, 8, [2, 2, 2]), (16, 1000, [5]), (1936, 1088, [2]), (640, 360, [1] * 9), (320, 200, [130])):
    out = {}
    for variant in (2, 4):
        r = Renderer((W, H), (0, 1, 0), seed=31, texture_source="synthetic", texture_size=(2048, 1024))
        r.set_kernel_variant(variant)
        kinds = []
        for s in spp_calls:
            r.accumulate(s); kinds.append(r.last_call_info()["variant"])
        out[variant] = (r.fetch_hdr(), kinds)
    same = (out[2][0].view(np.uint32) == out[4][0].view(np.uint32)).all()
    print("%4dx%-4d spp %s: default ran %s, bit-identical %s, max %g" % (W, H, spp_calls, sorted(set(out[4][1])), same, out[4][0].max()), flush=True)
    bad += 0 if same else 1
# tile partitions of the default at odd world sizes
for world in (3, 5, 7):
    full = Renderer((256, 128), (0, 1, 0), seed=5, texture_source="synthetic", texture_size=(2048, 1024)); full.set_kernel_variant(2); full.accumulate(6)
    acc = np.zeros_like(full.fetch_hdr())
    for rank in range(world):
        r = Renderer((256, 128), (0, 1, 0), seed=5, texture_source="synthetic", texture_size=(2048, 1024))
        r.set_tile_partition(rank, world); r.accumulate(6); acc += r.fetch_hdr()
    same = (acc.view(np.uint32) == full.fetch_hdr().view(np.uint32)).all()
    print("world %d partition sum == full frame: %s" % (world, same)); bad += 0 if same else 1
sys.exit(bad)
