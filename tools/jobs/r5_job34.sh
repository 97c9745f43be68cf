#!/bin/bash
# round 5, GPU job 34: the bench line of record with the r5f counters replayed (same sources as the profile: the sha gate passes), and the profile's kernel-trace pass
# again with every frame synchronised from the first launch on (bench.py --sync-frames now covers the priming frames)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_job34; mkdir -p $OUT
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_r5f.json 2> $OUT/bench_r5f.err
export TMPDIR=/tmp DE_BIG_SLOTS=1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/stats -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --frames-in-flight 1 --sync-frames > $GRAFT_REPO_ROOT/$OUT/stats.log 2>&1
cd $GRAFT_REPO_ROOT
find $OUT -name "*.csv" -size +24M -delete
tail -c 600 $OUT/bench_r5f.json
