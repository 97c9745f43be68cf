#!/usr/bin/env python3
"""Exhaustive device checks of candidate short sequences against the IEEE result (run on the GPU box).
  python tools/exhaustive_math.py sqrt_rsq"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from digital_earth_amd.renderer import Renderer

r = Renderer((16, 8), (0, 1, 0), texture_source="constant")
what = sys.argv[1] if len(sys.argv) > 1 else "sqrt_rsq"
m = np.arange(1 << 23, dtype=np.uint32)
if what == "sqrt_rsq":
    bad_total = 0
    for expo in list(range(64, 192)):
        x = ((np.uint32(expo) << np.uint32(23)) | m).view(np.float32)
        got = r.debug_math(9, x)
        want = np.sqrt(x.astype(np.float64)).astype(np.float32)
        bad = int((got.view(np.uint32) != want.view(np.uint32)).sum())
        bad_total += bad
        if bad and expo < 70 or expo in (126, 127, 128):
            i = np.nonzero(got.view(np.uint32) != want.view(np.uint32))[0][:3]
            print("expo", expo, "bad", bad, [(float(x[k]).hex(), float(got[k]).hex(), float(want[k]).hex()) for k in i])
    print("sqrt_rsq: total mismatches over exponents 64..191:", bad_total)
