#!/usr/bin/env python3
"""The window loop pipelined (accumulate(1) + fetch_image(lag=1) per frame) for a kernel trace, and the trace's timeline.

  rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/pipelined_trace.py run [frames] [lag]
  python3 tools/pipelined_trace.py show DIR            # per frame: when its kernels started and ended, relative to the frame's launch
  python3 tools/pipelined_trace.py tail DIR [N]        # the last N kernels of any kernel trace, with their queues

`run` also prints the loop's own ms per frame (DE_V6_* / DE_SLOTS overrides apply as everywhere).
"""
import csv
import glob
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def run(frames, lag=1):
    from digital_earth_amd.renderer import Renderer
    r = Renderer((1920, 1080), (0, 1, 0), texture_source="synthetic")
    r.copy_textures()
    r.reset_framebuffer(); r.accumulate(16); r.synchronize()
    for _ in range(8):
        r.accumulate(1); r.fetch_image(copy=False, lag=lag)
    r.fetch_pending(copy=False); r.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        r.accumulate(1); r.fetch_image(copy=False, lag=lag)
    r.fetch_pending(copy=False)
    print("pipelined loop, lag %d: %.2f ms per frame over %d frames" % (lag, (time.perf_counter() - t0) * 1e3 / frames, frames))
    r.close()


def show(d):
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for row in csv.DictReader(open(f)):
            rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")))
    rows.sort()
    # the last 12 frames: a frame = one v6_init_kernel and what follows it until the next
    inits = [i for i, r_ in enumerate(rows) if r_[2] == "v6_init_kernel"]
    if len(inits) < 14:
        print("too few frames in the trace"); return
    base = rows[inits[-13]][0]
    prev_init = None
    for i in inits[-13:-1]:
        t = rows[i][0]
        print("frame launched at %+8.3f ms%s" % ((t - base) / 1e6, "" if prev_init is None else "  (%.3f ms after the previous)" % ((t - prev_init) / 1e6)))
        prev_init = t
    print()
    t_lo, t_hi = rows[inits[-13]][0], rows[inits[-1]][0]
    for s, e, n in rows:
        if t_lo <= s <= t_hi:
            print("%+9.3f .. %+9.3f  (%7.3f ms)  %s" % ((s - base) / 1e6, (e - base) / 1e6, (e - s) / 1e6, n))


def tail(d, n):
    """the last n kernels of a trace (any workload), times relative to the first of them; queue = the stream's queue id"""
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for row in csv.DictReader(open(f)):
            rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row["Kernel_Name"].split("(")[0].replace("void ", "")[:60], row.get("Queue_Id", "?"), row.get("Workgroup_Size", "?"), row.get("Grid_Size", "?")))
    rows.sort()
    rows = rows[-n:]
    base = rows[0][0]
    for s, e, name, q, wg, grid in rows:
        print("%+9.3f .. %+9.3f  (%7.3f ms)  queue %-3s grid %-9s %s" % ((s - base) / 1e6, (e - base) / 1e6, (e - s) / 1e6, q, grid, name))


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "tail":
        tail(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 60)
    elif len(sys.argv) >= 2 and sys.argv[1] == "show":
        show(sys.argv[2])
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 60, int(sys.argv[3]) if len(sys.argv) > 3 else 1)
