#!/bin/bash
# round 6, GPU job 22: launch slots for frame-sized launches (2 / 3 / 4 / 5) on the round-6 kernel: how much of the drain do consecutive frames hide?
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6_job22; mkdir -p $OUT
for rep in 1 2 3; do for bs in 2 3 4 5; do
  echo -n "big launch slots $bs: "; python3 - $bs <<'PY'
import sys, time, os
sys.path.insert(0, os.getcwd())
from digital_earth_amd.renderer import Renderer
r = Renderer((1920, 1080), (0, 1, 0), texture_source="synthetic"); r.copy_textures(); r.trim_textures()
r.set_launch_slots(8, int(sys.argv[1]))
for _ in range(6):
    r.reset_framebuffer(); r.accumulate(64); r.render_to_image_device()
r.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    r.reset_framebuffer(); r.accumulate(64); r.render_to_image_device()
r.synchronize()
print("%.2f ms per step" % ((time.perf_counter() - t0) * 1e3 / 30))
PY
done; done | tee $OUT/slots.txt
