#!/bin/bash
# round 5, GPU job 38: the final set for the committed sources: rocprofv3 passes (r5f) condensed ON THE BOX, the bench lines with those counters replayed, the
# kernel-stats pass with every frame synchronised, smoke(), the whole GPU suite
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_job38; mkdir -p $OUT
bash tools/profile_gpu.sh r5f > $OUT/prof_r5f.log 2>&1
bash tools/commit_profile.sh r5f > $OUT/commit_profile.log 2>&1
mkdir -p $OUT/profiles; cp -r profiles/hbm_traffic.json profiles/r5f_summary.md profiles/r5f_raw $OUT/profiles/
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_r5f.json 2> $OUT/bench_r5f.err
export TMPDIR=/tmp
( cd /tmp; DE_BIG_SLOTS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/stats -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --frames-in-flight 1 --sync-frames > $GRAFT_REPO_ROOT/$OUT/stats.log 2>&1 )
find $OUT -name "*.csv" -size +24M -delete
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1
timeout 1500 python3 -m pytest tests -q -m gpu --durations=6 > $OUT/pytest_all.txt 2>&1
tail -3 $OUT/smoke.txt; grep -E "passed|failed" $OUT/pytest_all.txt; tail -c 400 $OUT/bench_r5f.json
