# profiles/simulated_world.json: rank 0's share of an N-rank partition rendered on ONE GPU (bench.py --simulate-world N [--split samples], no collective),
# N = 1 / 2 / 4 / 8, at the BASELINE configs' OWN sizes and sample counts (VERDICT r4 #3a), both partitions: the prediction the first hardware SCALE run
# can be read against.  cfg2: 1920x1080 x 64, default camera; cfg3: x 256, florida; cfg4: 3840x2160 x 128, sunset hurricane, cloud-heavy maps;
# cfg5: 1920x1080 x 1024 progressive every 128, Apollo 11.
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import json, subprocess, sys
CFG = {"cfg2 default camera 64 spp": [],
       "cfg3 florida 256 spp": ["--preset", "config - florida.txt", "--spp", "256"],
       "cfg4 sunset hurricane 3840x2160 128 spp cloud-heavy": ["--preset", "config - sunset hurricane.txt", "--width", "3840", "--height", "2160", "--spp", "128", "--cloud-heavy"],
       "cfg5 Apollo 11 1024 spp progressive 128": ["--preset", "config - Apollo 11.txt", "--spp", "1024", "--progressive", "128"]}
out = {"note": "ms per step of rank 0's share of the config's frame on ONE MI355X (bench.py --simulate-world N; frames overlap through the launch slots; no collective: "
               "the reduce / gather of 12 W H bytes per rank comes on top); speedup = N=1 time / N time; drain_ms = the launch's own clock from 'no work item left' to the last wave's exit",
       "configs": {}}
for name, extra in CFG.items():
    heavy = "cfg4" in name or "cfg5" in name
    steps, warm = ("3", "1") if heavy else ("8", "2")
    out["configs"][name] = {}
    for split in ("tiles", "samples"):
        row = {}
        for n in (1, 2, 4, 8):
            if n == 1 and split == "samples":
                row["1"] = out["configs"][name]["tiles"]["1"]
                continue
            cmd = [sys.executable, "bench.py", "--no-cpu-baseline", "--warmup", warm, "--steps", steps, "--split", split] + extra + (["--simulate-world", str(n)] if n > 1 else [])
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
            j = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
            row[str(n)] = {"ms_per_step": round(j["ms_per_step"], 2), "single_frame_ms": round(j["single_frame_ms"], 2), "drain_ms": j.get("drain_ms"), "kernel": j["config"]["kernel"]}
        for n in ("2", "4", "8"):
            row[n]["speedup_vs_1"] = round(row["1"]["ms_per_step"] / row[n]["ms_per_step"], 2)
        out["configs"][name][split] = row
        print(name, split, {k: (v["ms_per_step"], v.get("speedup_vs_1")) for k, v in row.items()}, flush=True)
json.dump(out, open("gpurun_out/simulated_world.json", "w"), indent=1)
PY
