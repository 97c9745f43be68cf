#!/bin/bash
# round 6, GPU job 21 (run twice: after the experiment switches and after the last dead code were taken out of the source — same kernel code) — the whole GPU suite, the profile of record again (the replayed counters are tied to a
# hash of the sources), the bench line
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6_job21; mkdir -p $OUT
timeout 2400 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_all.txt 2>&1; tail -4 $OUT/pytest_all.txt | head -2
bash tools/profile_gpu.sh r6f > $OUT/prof_r6f.log 2>&1
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_r6f.json 2> $OUT/bench_r6f.err
python3 -c "
import json
j=json.loads([l for l in open('$OUT/bench_r6f.json') if l.startswith('{')][-1])
print(j['value'], j['ms_per_step'], j['single_frame_ms'], j['roofline']['frac'], j['roofline']['traffic_note'][:60])"
timeout 400 python3 tools/soak.py --minutes 3 > $OUT/soak.txt 2>&1; tail -1 $OUT/soak.txt
