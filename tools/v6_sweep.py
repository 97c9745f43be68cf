#!/usr/bin/env python3
"""Sweep render_kernel_v6's DE_V6_* knobs: one subprocess per setting (the knobs are read when a context is created).

  python tools/v6_sweep.py "DE_V6_YIELD=16,20,24,32" "DE_V6_SVC_CLOUD=8,12,16" ...   (one knob at a time around the defaults; --grid: all combinations)
  options: --preset NAME, --frames N (back to back, default 6), --spp N
"""
import itertools, os, subprocess, sys
REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
from digital_earth_amd.renderer import Renderer
from digital_earth_amd.earth_viewer import load_config
spp, frames, preset = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
r = Renderer((1920, 1080), (0, 1, 0), texture_source="synthetic"); r.copy_textures(); r.trim_textures()
if preset != "-": load_config(preset).apply(r)
r.set_kernel_variant(int(os.environ.get("DE_KERNEL", "6")))
r.reset_framebuffer(); r.accumulate(spp); r.synchronize()
best1 = 1e9
for _ in range(2):
    r.reset_framebuffer(); r.synchronize()
    t0 = time.perf_counter(); r.accumulate(spp); r.synchronize(); best1 = min(best1, (time.perf_counter() - t0) * 1e3)
t0 = time.perf_counter()
for _ in range(frames):
    r.reset_framebuffer(); r.accumulate(spp)
r.synchronize()
print("%%.1f %%.1f" %% (best1, (time.perf_counter() - t0) * 1e3 / frames))
''' % REPO
args = [a for a in sys.argv[1:] if "=" in a]
def opt(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default
spp, frames, preset = opt("--spp", "64"), opt("--frames", "6"), opt("--preset", "-")
def run(env_over):
    env = dict(os.environ); env.update(env_over)
    out = subprocess.run([sys.executable, "-c", CHILD, spp, frames, preset], env=env, capture_output=True, text=True, timeout=600)
    line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "FAILED " + out.stderr.strip()[-200:]
    print("%-60s single / back-to-back ms: %s" % (" ".join("%s=%s" % kv for kv in env_over.items()) or "(defaults)", line), flush=True)
run({})
knobs = [(a.split("=")[0], a.split("=")[1].split(",")) for a in args]
if "--grid" in sys.argv:
    for combo in itertools.product(*[v for _, v in knobs]):
        run({k: x for (k, _), x in zip(knobs, combo)})
else:
    for k, vals in knobs:
        for x in vals:
            run({k: x})
