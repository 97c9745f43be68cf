#!/bin/bash
# round 6, GPU job 20: what the driver runs at round end, on the committed tree: pytest -m gpu, smoke, bench.py with default flags
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6_job20; mkdir -p $OUT
timeout 2400 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_all.txt 2>&1; tail -4 $OUT/pytest_all.txt | head -2
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time python3 bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -3 $OUT/bench_default.err
python3 -c "
import json
j=json.loads([l for l in open('$OUT/bench_default.json') if l.startswith('{')][-1])
print(j['metric'], '|', j['value'], j['ms_per_step'], j['single_frame_ms'], j['roofline']['bound'], j['roofline']['frac'], j['roofline']['traffic'], j['valu']['issue']['frac_of_measured'])
print({k:(v.get('value'), v.get('bit_identical_vs_oracle_on_sparse_tiles'), v.get('error')) for k,v in j['other_configs'].items()})
print(j['cpu_baseline']['value'], j['parity']['bit_identical_fraction'])"
