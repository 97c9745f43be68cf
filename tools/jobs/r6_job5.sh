#!/bin/bash
# round 6, GPU job 5: checkpoint of the IPRA build — the whole GPU suite, the default bench line, smoke
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6_job5; mkdir -p $OUT
timeout 2400 python3 -m pytest tests -q -m gpu -x --durations=10 > $OUT/pytest_all.txt 2>&1; tail -16 $OUT/pytest_all.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 900 python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; python3 -c "
import json
j=json.loads([l for l in open('$OUT/bench.json') if l.startswith('{')][-1])
print(j['metric'], j['value'], j['ms_per_step'], j['single_frame_ms'], j['roofline']['kernel_ms'], j['roofline']['frac'], j['roofline']['traffic_note'][:80])
for k,v in j['other_configs'].items(): print(k, {a:b for a,b in v.items() if a in ('ms_per_step','value','kernel_ms','roofline_frac','bit_identical_vs_oracle_on_sparse_tiles','sparse_tiles','error')})
print(j['parity'])"
