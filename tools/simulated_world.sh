# profiles/simulated_world.json: rank 0's share of an N-rank tile partition rendered on ONE GPU (bench.py --simulate-world N, no collective),
# N = 1 / 2 / 4 / 8 x four cameras, default kernel variant: the prediction the first hardware SCALE run can be read against
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import json, subprocess, sys
cams = [None, "config - florida.txt", "config - sunset hurricane.txt", "config - Apollo 11.txt"]
out = {"note": "ms per step of rank 0's share of the 1920x1080x64-spp frame on ONE MI355X (bench.py --simulate-world N --steps 12 --warmup 3; frames overlap through the launch slots; no collective: the 24.9 MB reduce per frame comes on top); speedup = N=1 time / N time", "cameras": {}}
for cam in cams:
    row = {}
    for n in (1, 2, 4, 8):
        cmd = [sys.executable, "bench.py", "--no-cpu-baseline", "--warmup", "3", "--steps", "12"] + (["--simulate-world", str(n)] if n > 1 else []) + (["--preset", cam] if cam else [])
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        j = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
        row[str(n)] = {"ms_per_step": round(j["ms_per_step"], 2), "single_frame_ms": round(j["single_frame_ms"], 2), "kernel": j["config"]["kernel"]}
    for n in ("2", "4", "8"):
        row[n]["speedup_vs_1"] = round(row["1"]["ms_per_step"] / row[n]["ms_per_step"], 2)
    out["cameras"][cam or "default camera"] = row
    print(cam or "default camera", {k: (v["ms_per_step"], v.get("speedup_vs_1")) for k, v in row.items()}, flush=True)
json.dump(out, open("gpurun_out/simulated_world.json", "w"), indent=1)
PY
