#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (tools/profile_gpu.sh) into profiles/<tag>_summary.md + hbm_traffic.json."""
import csv, glob, json, os, sys
from collections import defaultdict

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
    for r in sorted(stats, key=lambda r: -float(r.get("TotalDurationNs", 0)))[:8]:
        out.append("| %s | %s | %.3f | %.3f | %s |" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                      float(r["AverageNs"]) / 1e6, r.get("Percentage", "")))
    out.append("")
    # average duration of the dominant kernel (the 64-spp launches of the production mode), for bench.py's staleness check
    kernel_ms = None
    for r in stats:
        if r["Name"].startswith("void render_kernel_v2<false, 0>") or (kernel_ms is None and "ray_march_kernel<false, 0>" in r["Name"]):
            kernel_ms = float(r["AverageNs"]) / 1e6
    pmc = defaultdict(list)
    for d in glob.glob(os.path.join(root, "pmc_*")):
        for r in read_csvs(os.path.join(d, "**", "*counter_collection.csv")):
            if "render_kernel" in r.get("Kernel_Name", "") or "ray_march_kernel" in r.get("Kernel_Name", ""):
                pmc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out += ["## PMC counters, render_kernel, mean per launch (64-spp launches only)", "", "| counter | mean per launch | launches |", "|---|---|---|"]
    means = {}
    for k, v in sorted(pmc.items()):
        big = [x for x in v if x >= 0.5 * max(v)] if v else v   # drop the 2-spp counter launch
        means[k] = sum(big) / max(len(big), 1)
        out.append("| %s | %.6g | %d |" % (k, means[k], len(big)))
    out.append("")
    if "SQ_THREAD_CYCLES_VALU" in means and "SQ_ACTIVE_INST_VALU" in means and means["SQ_ACTIVE_INST_VALU"]:
        out.append("* VALU lane utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU) = %.3f"
                   % (means["SQ_THREAD_CYCLES_VALU"] / (64.0 * means["SQ_ACTIVE_INST_VALU"])))
    if "SQ_ACTIVE_INST_VALU" in means and "SQ_WAVE_CYCLES" in means and means["SQ_WAVE_CYCLES"]:
        out.append("* VALU issue share of wave-cycles = %.3f" % (means["SQ_ACTIVE_INST_VALU"] / means["SQ_WAVE_CYCLES"]))
    if "FETCH_SIZE" in means:
        fetch_kb = means["FETCH_SIZE"]
        wr_kb = means.get("WRITE_SIZE", 0.0)
        # FETCH_SIZE tallies 64 B per L2->fabric read request; every request moves a 128-byte line — for wide streams (the
        # guide's gfx950 note) AND for this kernel's scattered dword taps (profiles/r2_fetch_calibration/README.md, measured
        # with tools/micro/fetch_calib.hip): true read bytes = 2 x FETCH_SIZE.  Infinity-Cache hits are included.
        out.append("* L2-miss read bytes (2 x FETCH_SIZE, calibrated): %.4g B per launch (raw counter %.4g B)" % (2 * fetch_kb * 1024, fetch_kb * 1024))
        out.append("* write bytes (WRITE_SIZE): %.4g B per launch" % (wr_kb * 1024))
        if write_json == "json":
            with open(os.path.join("profiles", "hbm_traffic.json"), "w") as f:
                miss = means.get("TCC_MISS_sum")
                json.dump({"workload": workload, "hbm_bytes_per_launch": 2 * fetch_kb * 1024 + wr_kb * 1024,
                           "fetch_raw_bytes": fetch_kb * 1024, "fetch_x2_bytes": 2 * fetch_kb * 1024, "write_bytes": wr_kb * 1024,
                           "tcc_miss_x_128B": (miss * 128 if miss else None), "profile": tag, "kernel_ms": kernel_ms,
                           "valu_insts_per_launch": means.get("SQ_INSTS_VALU"), "valu_lane_utilisation": (means["SQ_THREAD_CYCLES_VALU"] / (64.0 * means["SQ_ACTIVE_INST_VALU"]) if means.get("SQ_ACTIVE_INST_VALU") and means.get("SQ_THREAD_CYCLES_VALU") else None),
                           "note": "rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE (separate passes), per 64-spp launch; read bytes = 2 x FETCH_SIZE "
                                   "(a request moves a 128-B line and is tallied as 64 B — calibrated for scattered dword loads in "
                                   "profiles/r2_fetch_calibration). L2-miss traffic at the fabric: Infinity-Cache hits (the 3 MB altitude "
                                   "table) are included, so this over-states HBM traffic"},
                          f, indent=1)
    os.makedirs("profiles", exist_ok=True)
    with open(os.path.join("profiles", tag + "_summary.md"), "w") as f:
        f.write("\n".join(out) + "\n")
    print("\n".join(out))

if __name__ == "__main__":
    main(*sys.argv[1:])
