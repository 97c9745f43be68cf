# profiles/simulated_world.json: rank 0's share of an N-rank partition rendered on ONE GPU (bench.py --simulate-world N [--split samples], no collective),
# N = 1 / 2 / 4 / 8, at the BASELINE configs' OWN sizes and sample counts (VERDICT r4 #3a), both partitions: the prediction the first hardware SCALE run
# can be read against.  cfg2: 1920x1080 x 64, default camera; cfg3: x 256, florida; cfg4: 3840x2160 x 128, sunset hurricane, cloud-heavy maps;
# cfg5: 1920x1080 x 1024 progressive every 128, Apollo 11.
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import json, subprocess, sys
CFG = {"cfg2 default camera 64 spp": ["--config", "cfg2", "--no-other-configs"],
       "cfg3 florida 256 spp": ["--config", "cfg3"],
       "cfg4 sunset hurricane 3840x2160 128 spp cloud-heavy": ["--config", "cfg4"],
       "cfg5 Apollo 11 1024 spp progressive 128": ["--config", "cfg5"]}
out = {"note": "ms per step of rank 0's share of the config's frame on ONE MI355X (bench.py --simulate-world N; frames overlap through the launch slots); speedup = N=1 time / N time; "
               "drain_ms = the launch's own clock from 'no work item left' to the last wave's exit.  `with_standin` (round 6, tile partition): the same steps with a ONE-GPU STAND-IN for the "
               "frame's collective enqueued where the reduce goes — a streaming kernel over the HDR buffer (3 x 12 W H bytes) + 7 device-to-device copies of 12 W H bytes (the ingress of an "
               "8-rank gather), behind the next frame's persistent workgroups (include/digital_earth_debug.h: de_debug_standin_reduce); standin_ms = its mean queue-to-finish time.  What a "
               "stand-in cannot show: xGMI transfer time and the ranks' collective kernels waiting for each other",
       "configs": {}}
for name, extra in CFG.items():
    heavy = "cfg4" in name or "cfg5" in name
    out["configs"][name] = {}
    for split in ("tiles", "samples"):
        row = {}
        for n in (1, 2, 4, 8):
            if n == 1 and split == "samples":
                row["1"] = out["configs"][name]["tiles"]["1"]
                continue
            # enough steps for a stable figure: ~2 s of frames, at least 4
            steps = str(max(4, min(120, int((24 if "cfg2" in name else 6) * n * (0.4 if heavy else 1.0)))))
            warm = "2" if not heavy else "1"
            def run(extra2):
                cmd = [sys.executable, "bench.py", "--no-cpu-baseline", "--warmup", warm, "--steps", steps, "--split", split] + extra + (["--simulate-world", str(n)] if n > 1 else []) + extra2
                p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
                return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
            j = run([])
            row[str(n)] = {"ms_per_step": round(j["ms_per_step"], 2), "single_frame_ms": round(j["single_frame_ms"], 2), "drain_ms": j.get("drain_ms"), "kernel": j["config"]["kernel"], "steps": int(steps)}
            if n > 1 and split == "tiles":
                js = run(["--standin-collective", "7"])
                row[str(n)]["with_standin"] = {"ms_per_step": round(js["ms_per_step"], 2), "standin_ms": round(js["standin_collective"]["queue_to_finish_ms_mean_of_last_16"], 2)}
        for n in ("2", "4", "8"):
            row[n]["speedup_vs_1"] = round(row["1"]["ms_per_step"] / row[n]["ms_per_step"], 2)
            if "with_standin" in row[n]:
                row[n]["with_standin"]["speedup_vs_1"] = round(row["1"]["ms_per_step"] / row[n]["with_standin"]["ms_per_step"], 2)
        out["configs"][name][split] = row
        print(name, split, {k: (v["ms_per_step"], v.get("speedup_vs_1"), v.get("with_standin", {}).get("speedup_vs_1")) for k, v in row.items()}, flush=True)
json.dump(out, open("gpurun_out/simulated_world.json", "w"), indent=1)
PY
