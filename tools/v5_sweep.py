#!/usr/bin/env python3
"""Time the 1920x1080 frame under several settings of a kernel variant (default: the persistent stage scheduler, 5) in one process.

  python tools/v5_sweep.py [--spp N] [--preset NAME] [--stats] "DE_V5_WPC=16,DE_V5_SHARDS=8" "DE_V5_WPC=20" ...
Each argument is one configuration (environment knobs read at de_create); "" = defaults.  The first context's maps are shared.
"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from digital_earth_amd.renderer import Renderer
from digital_earth_amd.earth_viewer import load_config

args = sys.argv[1:]
spp, preset, stats, ref, variant, pipelined = 64, None, False, None, 5, 0
cfgs = []
while args:
    a = args.pop(0)
    if a == "--spp": spp = int(args.pop(0))
    elif a == "--preset": preset = args.pop(0)
    elif a == "--stats": stats = True
    elif a == "--ref": ref = int(args.pop(0))
    elif a == "--variant": variant = int(args.pop(0))
    elif a == "--pipelined": pipelined = int(args.pop(0))      # frames issued back to back (launch slots overlap them), one wait at the end
    else: cfgs.append(a)
if not cfgs: cfgs = [""]
W, H = 1920, 1080
STAGES = ["sphere trace", "gas", "cloud", "event", "surface", "vol finish", "surf finish", "new paths"]
if stats: os.environ["DE_V5_STATS"] = "1"; os.environ["DE_BIG_SLOTS"] = "1"
owner = Renderer((W, H), (0, 1, 0), texture_source="synthetic"); owner.copy_textures(); owner.trim_textures()
if preset: load_config(preset).apply(owner)
ref_hdr = None
if ref:
    owner.set_kernel_variant(ref)
    owner.reset_framebuffer(); owner.accumulate(spp); owner.synchronize()
    ts = []
    for _ in range(3):
        owner.reset_framebuffer(); owner.synchronize()
        t0 = time.perf_counter(); owner.accumulate(spp); owner.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ref_hdr = owner.fetch_hdr()
    print("variant %d: %s ms per %d-spp frame" % (ref, " ".join("%.1f" % t for t in ts), spp), flush=True)
for cfg in cfgs:
    saved = {}
    for kv in [x for x in cfg.split(",") if x]:
        k, v = kv.split("=")
        saved[k] = os.environ.get(k); os.environ[k] = v
    r = Renderer((W, H), (0, 1, 0), texture_source="synthetic"); r.share_textures_from(owner)
    if preset: load_config(preset).apply(r)
    r.set_kernel_variant(variant)
    try:
        r.reset_framebuffer(); r.accumulate(spp); r.synchronize()
        ts = []
        for _ in range(3):
            r.reset_framebuffer(); r.synchronize()
            t0 = time.perf_counter(); r.accumulate(spp); r.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        line = "%-60s %s ms -> %.0f Msamples/s" % (cfg or "(defaults)", " ".join("%.1f" % t for t in ts), W * H * spp / min(ts) / 1e3)
        if pipelined:
            for _ in range(2):
                r.synchronize(); t0 = time.perf_counter()
                for _ in range(pipelined):
                    r.reset_framebuffer(); r.accumulate(spp)
                r.synchronize(); tp = (time.perf_counter() - t0) * 1e3 / pipelined
            line += "  | pipelined %.1f ms/frame" % tp
        if ref_hdr is not None:
            line += "  bit-identical %.6f" % float((r.fetch_hdr().view(np.uint32) == ref_hdr.view(np.uint32)).mean())
        print(line, flush=True)
        if stats:
            st = r.v5_stats(40)
            tot = sum(st[8:16]) + st[22]
            print("    " + " | ".join("%s %.1fM %.0f%%" % (STAGES[k], st[k] / 1e6, 100.0 * st[8 + k] / max(tot, 1)) for k in range(8)))
            print("    scheduler %.0f%%; idle rounds %d, partial chunks %d, failed claims %d, stage entries %d, ring refills %d, steals %d; wave-time %.0f ms x waves"
                  % (100.0 * st[22] / max(tot, 1), st[16], st[17], st[18], st[19], st[20], st[21], tot * 10e-6), flush=True)
            print("    of the wave-time: next_input %.1f%%, loop-stage push %.1f%% + flat push (in the same counter), loop-stage service in all %.1f%%, flat load+body %.1f%%"
                  % (100.0 * st[23] / max(tot, 1), 100.0 * st[24] / max(tot, 1), 100.0 * st[25] / max(tot, 1), 100.0 * st[26] / max(tot, 1)), flush=True)
            print("    loop stages (sphere trace, gas, cloud): trips %s M, lanes per trip %s, steps per record %s, step time per trip %s us; dry events %d"
                  % (" ".join("%.1f" % (st[32 + k] / 1e6) for k in range(3)), " ".join("%.1f" % (st[35 + k] / max(st[32 + k], 1)) for k in range(3)),
                     " ".join("%.1f" % (st[35 + k] / max(st[k], 1)) for k in range(3)),
                     " ".join("%.2f" % ((st[8 + k]) * 0.01 / max(st[32 + k], 1)) for k in range(3)), st[38]), flush=True)
            print("    scheduler: ensure_stash %.1f%%, flush %.1f%%, snapshot %.1f%%, claim %.1f%%" % tuple(100.0 * st[k] / max(tot, 1) for k in (27, 28, 29, 30)), flush=True)
    except Exception as e:
        print("%-60s FAILED: %s" % (cfg or "(defaults)", e), flush=True)
    del r
    import gc; gc.collect()
    for k, v in saved.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
