"""The counters the bench line quotes (profiles/hbm_traffic.json) against the raw rocprofv3 CSVs committed beside them: recomputed
here without tools/summarize_profile.py, so that a stale or hand-edited JSON cannot pass."""
import csv, json, os, re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RENDER = re.compile(r"render_kernel|ray_march_kernel|v3_|v5_kernel|v6_kernel")


def _per_frame(path, counters):
    rows = list(csv.DictReader(open(path)))
    disp = {}
    for r in rows:
        if RENDER.search(r["Kernel_Name"]):
            d = disp.setdefault(r["Dispatch_Id"], {"k": r["Kernel_Name"]})
            d[r["Counter_Name"]] = float(r["Counter_Value"])
    prim = sum(1 for d in disp.values() if "v3_primary" in d["k"])
    if prim:                                                                       # the wavefront pipeline: two batches per frame
        frames = prim / 2.0
        use = [d for d in disp.values() if not re.search(r"render_kernel_v2<\w+, 1>", d["k"])]   # not the 2-spp counting launch
    else:                                                                          # one persistent launch per frame: the 64-spp launches
        key = counters[0]                                                          # (+ the tail launch that follows each of them, round 5: v6_kernel<.., true>)
        mx = max(d.get(key, 0.0) for d in disp.values())
        use, frames, take = [], 0.0, False
        for did in sorted(disp, key=int):
            d = disp[did]
            if re.search(r"v6_kernel\w*<\w+, \w+, true>", d["k"]):
                if take: use.append(d)
                continue
            take = d.get(key, 0.0) >= 0.5 * mx
            if take: use.append(d); frames += 1.0
    return {c: sum(d.get(c, 0.0) for d in use) / frames for c in counters}


def test_bench_counters_match_the_committed_raw_profile():
    j = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    raw = os.path.join(ROOT, "profiles", j["profile"] + "_raw")
    assert os.path.isdir(raw), "the raw CSVs of profile %s are not committed" % j["profile"]
    sq = _per_frame(os.path.join(raw, "pmc_sq1_counter_collection.csv"), ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU"])
    assert abs(sq["SQ_INSTS_VALU"] / j["valu_insts_per_launch"] - 1.0) < 1e-6
    lane = sq["SQ_THREAD_CYCLES_VALU"] / (64.0 * sq["SQ_ACTIVE_INST_VALU"])
    assert abs(lane - j["valu_lane_utilisation"]) < 1e-6
    fetch = _per_frame(os.path.join(raw, "pmc_fetch_counter_collection.csv"), ["FETCH_SIZE"])["FETCH_SIZE"] * 1024.0
    write = _per_frame(os.path.join(raw, "pmc_write_counter_collection.csv"), ["WRITE_SIZE"])["WRITE_SIZE"] * 1024.0
    assert abs(fetch / j["fetch_raw_bytes"] - 1.0) < 1e-6 and abs(write / j["write_bytes"] - 1.0) < 1e-6
    assert abs((2 * fetch + write) / j["hbm_bytes_per_launch"] - 1.0) < 1e-6      # 2 x FETCH_SIZE: profiles/r2_fetch_calibration


def test_frame_time_of_the_profile_comes_from_its_kernel_trace():
    j = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", j["profile"] + "_raw", "stats_kernel_trace.csv"))))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    spans, start, end = [], None, None
    for r in rows:
        k = r["Kernel_Name"]
        if RENDER.search(k) and not re.search(r"render_kernel_v2<\w+, 1>", k):
            start = int(r["Start_Timestamp"]) if start is None else start
            end = max(end or 0, int(r["End_Timestamp"]))
        elif "display_kernel" in k and start is not None:
            spans.append((end - start) / 1e6); start = end = None
    med = sorted(spans)[len(spans) // 2]
    spans = sorted(x for x in spans if 0.5 * med <= x <= 1.5 * med)      # the synchronised frames: not the priming block, not the small parity launches
    assert abs(spans[len(spans) // 2] - j["kernel_ms"]) < 0.01
