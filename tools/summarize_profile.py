#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (tools/profile_gpu.sh) into profiles/<tag>_summary.md + profiles/hbm_traffic.json.

A frame of the bench workload is one de_accumulate(64) call.  With the default variant 4 that is one v6_kernel launch + its tail launch (round 5; kernel
variants 2 / 5: one render_kernel_v2 / v5_kernel launch); with the wavefront pipeline (variant 3, and variant 4 until round 4) it is a
sequence: batches side by side, each = the pipeline's stage kernels for the first vertex rounds (v3_*) + render_kernel_v2 as finisher.
Counters are therefore summed over ALL render kernels of a run and divided by its number of frames; the frame duration is the busy span
of the render kernels in the kernel trace between two display_kernel launches."""
import csv, glob, hashlib, json, os, re, sys
from collections import defaultdict


def source_sha16():
    """sha256 (first 16 hex digits) over the device and host sources of the library, in name order: what the profiled counters belong to.  bench.py
    recomputes it and withholds replayed counters when the sources have changed since the profile (box-to-box speed differences of 5 % were measured
    for one build in round 5: a duration gate alone either passes changed kernels or refuses unchanged ones)."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "digital_earth_amd", "csrc")
    h = hashlib.sha256()
    for d, _, files in sorted(os.walk(root)):
        for f in sorted(files):
            if f.endswith((".h", ".hip", ".inc")):
                h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]

RENDER = re.compile(r"render_kernel|ray_march_kernel|v3_|v5_kernel|v6_kernel")


def read_csvs(pattern):
    rows = []
    for f in glob.glob(pattern, recursive=True):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    return rows


def main(tag, workload="1920x1080_64spp_q2", write_json="json"):
    root = os.path.join("gpurun_out", "prof_" + tag)
    out = ["# rocprofv3 summary `%s` (workload %s)" % (tag, workload), ""]
    stats = read_csvs(os.path.join(root, "stats", "**", "*kernel_stats.csv"))
    out += ["## kernel stats (rocprofv3 --kernel-trace --stats)", "", "| kernel | calls | total ms | avg ms | % |", "|---|---|---|---|---|"]
    for r in sorted(stats, key=lambda r: -float(r.get("TotalDurationNs", 0)))[:14]:
        out.append("| %s | %s | %.3f | %.3f | %s |" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                      float(r["AverageNs"]) / 1e6, r.get("Percentage", "")))
    out.append("")
    trace = read_csvs(os.path.join(root, "stats", "**", "*kernel_trace.csv"))
    n_primary = sum(1 for r in trace if "v3_primary" in r["Kernel_Name"])
    n_display = sum(1 for r in trace if "display_kernel" in r["Kernel_Name"])
    n_acc = sum(1 for r in trace if "accumulate_kernel" in r["Kernel_Name"])
    batches = max(1, round(n_primary / n_display)) if (n_primary and n_display) else 1
    kernel_ms = None
    if n_display:
        # frames run back to back: a frame = the kernels between two display_kernel launches (the bench displays every frame)
        ev = sorted(trace, key=lambda r: int(r["Start_Timestamp"]))
        spans, start, end = [], None, None
        for r in ev:
            if RENDER.search(r["Kernel_Name"]) and "<false, 1>" not in r["Kernel_Name"] and "<true, 1>" not in r["Kernel_Name"]:
                if start is None: start = int(r["Start_Timestamp"])
                end = max(end or 0, int(r["End_Timestamp"]))
            elif "display_kernel" in r["Kernel_Name"] and start is not None:
                spans.append((end - start) / 1e6); start = end = None
        med = sorted(spans)[len(spans) // 2]
        spans = [x for x in spans if 0.5 * med <= x <= 1.5 * med]      # not the priming block (several calls in flight between two displays), not the small parity launches
        kernel_ms = sorted(spans)[len(spans) // 2]        # the first frames of a run also allocate the pipeline's queues
        out.append("* frames in the trace: %d (one de_accumulate(64) call each, displayed after it); first render-kernel start to last render-kernel end, median: **%.2f ms** (min %.2f, max %.2f)"
                   % (len(spans), kernel_ms, min(spans), max(spans)))
        if n_primary: out.append("* a frame = %d batches side by side, each: the wavefront pipeline's stage kernels for the first vertex rounds + render_kernel_v2 as finisher of the surviving paths" % batches)
        else: out.append("* a frame = one launch of the persistent kernel (the per-CU stage scheduler, render_kernel_v6.hip, unless a kernel variant was forced) + its tail launch (`v6_kernel<.., true>`: 64 workgroups that finish the paths the frame's workgroups exported).  NOTE: these passes run one frame at a time (`--sync-frames`, one launch slot), so no launch has a successor queued behind it, every workgroup decides to FINISH IN PLACE and the tail launch imports nothing: the counters are those of a frame rendered alone; the export / import path the timed region of bench.py uses (frames back to back) is measured by the `_tail` passes when present (`DE_V6_TAIL_ALONE=1`: export regardless)")
        rsum = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in trace if RENDER.search(r["Kernel_Name"])) / 1e6 / max(n_display, 1)
        out.append("* sum of the render kernels' own durations per displayed frame: %.1f ms%s" % (rsum, " (the batches overlap)" if n_primary else " (includes the bench's parity and counting launches)"))
        out.append("")
    else:
        for r in stats:
            if r["Name"].startswith("void v6_kernel<false, false, false>") or r["Name"].startswith("void render_kernel_v2<false, 0>") or (kernel_ms is None and "ray_march_kernel<false, 0>" in r["Name"]):
                kernel_ms = float(r["AverageNs"]) / 1e6
    # counters: sum over every render kernel of a pass, per frame of that pass
    pmc, frames_in_pass = defaultdict(float), {}
    passes_with = defaultdict(int)                             # a counter collected in several passes (SQ_WAVE_CYCLES) is averaged, not summed
    per_kernel = defaultdict(lambda: defaultdict(float))      # short kernel name -> counter -> per frame; "ms:<pass>" = duration sum
    for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
        if not os.path.isdir(d): continue
        rows = [r for r in read_csvs(os.path.join(d, "**", "*counter_collection.csv")) if RENDER.search(r.get("Kernel_Name", ""))]
        disp = defaultdict(dict)
        for r in rows:
            disp[r["Dispatch_Id"]]["k"] = r["Kernel_Name"]
            disp[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
            disp[r["Dispatch_Id"]]["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        prim = sum(1 for v in disp.values() if "v3_primary" in v["k"])
        if prim:
            nf = prim / batches
            use = [v for v in disp.values() if not re.search(r"render_kernel_v2<\w+, 1>", v["k"])]      # not the 2-spp counter launch
        else:
            vals = [v for v in disp.values()]
            key = next((k for k in vals[0] if k != "k"), None) if vals else None
            mx = max((v.get(key, 0.0) for v in vals), default=0.0)
            TAIL = re.compile(r"v6_kernel\w*<\w+, \w+, true>")                                                # a tail launch of the frame's chain (render_kernel_v6.hip: "The tail")
            use, nf, parent_in = [], 0, False
            for did in sorted(disp, key=lambda x: int(x)):                                                     # --sync-frames: a tail launch follows its frame's launch
                v = disp[did]
                if TAIL.search(v["k"]):
                    if parent_in: use.append(v)
                    continue
                parent_in = v.get(key, 0.0) >= 0.5 * mx                                                         # the 64-spp launches
                if parent_in: use.append(v); nf += 1
        for k in {k for v in use for k in v if k not in ("k", "ms")}: passes_with[k] += 1
        for v in use:
            short = re.sub(r"^void ", "", v["k"]).split("(")[0]
            for k, x in v.items():
                if k == "ms": per_kernel[short]["ms:" + os.path.basename(d)] += x / max(nf, 1)
                elif k != "k":
                    pmc[k] += x / max(nf, 1)
                    per_kernel[short][k] += x / max(nf, 1)
    out += ["## PMC counters, all render kernels, per frame", "", "| counter | per frame |", "|---|---|"]
    means = {k: x / max(passes_with[k], 1) for k, x in pmc.items()}
    for c in per_kernel.values():
        for k in list(c):
            if not k.startswith("ms:") and passes_with[k] > 1: c[k] /= passes_with[k]
    for k in sorted(means):
        out.append("| %s | %.6g |" % (k, means[k]))
    out.append("")
    if means.get("SQ_ACTIVE_INST_VALU"):
        out.append("* VALU lane utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU) = %.3f"
                   % (means["SQ_THREAD_CYCLES_VALU"] / (64.0 * means["SQ_ACTIVE_INST_VALU"])))
    if means.get("SQ_WAVE_CYCLES") and means.get("SQ_ACTIVE_INST_VALU"):
        out.append("* VALU issue share of wave-cycles = %.3f" % (means["SQ_ACTIVE_INST_VALU"] / means["SQ_WAVE_CYCLES"]))
    if len(per_kernel) > 1 and means.get("SQ_INSTS_VALU"):
        out += ["", "## The same counters by kernel, per frame", "",
                "| kernel | launches' duration, ms (sum; batches overlap) | VALU wave-instructions | share | lane utilisation | L2 requests | L2 hit | read GB (2 x FETCH_SIZE) | write GB | fabric TB/s while it runs |",
                "|---|---|---|---|---|---|---|---|---|---|"]
        for k, c in sorted(per_kernel.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0.0)):
            hit, miss = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0)
            rd, wr = 2 * c.get("FETCH_SIZE", 0.0) * 1024, c.get("WRITE_SIZE", 0.0) * 1024
            ms_f = c.get("ms:pmc_fetch", 0.0)
            out.append("| `%s` | %.1f | %.3g | %.1f %% | %.3f | %.3g | %.1f %% | %.1f | %.1f | %.2f |" % (
                k, c.get("ms:pmc_sq1", 0.0), c.get("SQ_INSTS_VALU", 0.0), 100 * c.get("SQ_INSTS_VALU", 0.0) / means["SQ_INSTS_VALU"],
                c.get("SQ_THREAD_CYCLES_VALU", 0.0) / max(64.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0), 1.0), c.get("TCP_TCC_READ_REQ_sum", 0.0),
                100 * hit / max(hit + miss, 1.0), rd / 1e9, wr / 1e9, (rd + wr) / 1e12 / max(ms_f / 1e3, 1e-9)))
        out.append("")
    if "FETCH_SIZE" in means:
        fetch_kb, wr_kb = means["FETCH_SIZE"], means.get("WRITE_SIZE", 0.0)
        # FETCH_SIZE tallies 64 B per L2->fabric read request; every request moves a 128-byte line — for wide streams (the
        # guide's gfx950 note) AND for scattered dword taps (profiles/r2_fetch_calibration/README.md): true read bytes = 2 x FETCH_SIZE.
        out.append("* L2-miss read bytes (2 x FETCH_SIZE, calibrated): %.4g B per frame (raw counter %.4g B)" % (2 * fetch_kb * 1024, fetch_kb * 1024))
        out.append("* write bytes (WRITE_SIZE): %.4g B per frame" % (wr_kb * 1024))
        if write_json == "json":
            miss = means.get("TCC_MISS_sum")
            lane = (means["SQ_THREAD_CYCLES_VALU"] / (64.0 * means["SQ_ACTIVE_INST_VALU"])) if means.get("SQ_ACTIVE_INST_VALU") and means.get("SQ_THREAD_CYCLES_VALU") else None
            with open(os.path.join("profiles", "hbm_traffic.json"), "w") as f:
                json.dump({"workload": workload, "hbm_bytes_per_launch": 2 * fetch_kb * 1024 + wr_kb * 1024,
                           "fetch_raw_bytes": fetch_kb * 1024, "fetch_x2_bytes": 2 * fetch_kb * 1024, "write_bytes": wr_kb * 1024,
                           "tcc_miss_x_128B": (miss * 128 if miss else None), "profile": tag, "kernel_ms": kernel_ms,
                           "valu_insts_per_launch": means.get("SQ_INSTS_VALU"), "valu_lane_utilisation": lane,
                           "salu_insts_per_launch": means.get("SQ_INSTS_SALU"), "source_sha16": source_sha16(),
                           "note": "rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE (separate passes), summed over every render kernel of a frame (one "
                                   "de_accumulate(64) call: one v6_kernel launch and its tail launch under the default variant — profiled one frame at a time, so the launch finishes in place and the tail launch imports nothing); read bytes = 2 x FETCH_SIZE (a request "
                                   "moves a 128-B line and is tallied as 64 B: profiles/r2_fetch_calibration). L2-miss traffic at the fabric: "
                                   "Infinity-Cache hits are included"},
                          f, indent=1)
    # the same frame with the tail chain exporting (profile_gpu.sh: DE_V6_TAIL_ALONE=1 passes) — the path frames back to back take
    tstats = read_csvs(os.path.join(root, "tail_stats", "**", "*kernel_stats.csv"))
    if tstats:
        out += ["", "## The tail chain exporting (`DE_V6_TAIL_ALONE=1`: what frames back to back run; one frame at a time here)", ""]
        for r in sorted(tstats, key=lambda r: -float(r.get("TotalDurationNs", 0))):
            if "v6_kernel" in r["Name"]:
                out.append("* `%s`: %s launches, average %.3f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6))
        tp = defaultdict(float)
        for d in sorted(glob.glob(os.path.join(root, "tail_pmc_*"))):
            rows = [r for r in read_csvs(os.path.join(d, "**", "*counter_collection.csv")) if "v6_kernel" in r.get("Kernel_Name", "")]
            disp = defaultdict(dict)
            for r in rows:
                disp[r["Dispatch_Id"]]["k"] = r["Kernel_Name"]
                disp[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
            names = {k for v in disp.values() for k in v if k != "k"}
            for name in names:
                frame_vals = [v.get(name, 0.0) for v in disp.values() if ", false>" in v["k"]]
                big = [x for x in frame_vals if x >= 0.5 * max(frame_vals)]                      # the 64-spp launches
                tail_vals = [v.get(name, 0.0) for v in disp.values() if ", true>" in v["k"]]
                nt = max(len(big), 1)
                tp[name] = sum(big) / nt + sum(sorted(tail_vals)[-len(big):]) / nt       # a frame = its launch + its tail launch
        if tp:
            out.append("* per frame (frame launch + tail launch): WRITE_SIZE %.1f GB, 2 x FETCH_SIZE %.1f GB, VALU %.4g, SALU %.4g wave-instructions, lane utilisation %.3f"
                       % (tp.get("WRITE_SIZE", 0) * 1024 / 1e9, 2 * tp.get("FETCH_SIZE", 0) * 1024 / 1e9, tp.get("SQ_INSTS_VALU", 0), tp.get("SQ_INSTS_SALU", 0),
                          tp.get("SQ_THREAD_CYCLES_VALU", 0) / max(64.0 * tp.get("SQ_ACTIVE_INST_VALU", 1), 1)))
    os.makedirs("profiles", exist_ok=True)
    with open(os.path.join("profiles", tag + "_summary.md"), "w") as f:
        f.write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main(*sys.argv[1:])
