#!/bin/bash
# round 6, GPU job 15: the stateless scheduler loop through the whole GPU suite and a soak run
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6_job15; mkdir -p $OUT
timeout 2400 python3 -m pytest tests -q -m gpu -x --durations=6 > $OUT/pytest_all.txt 2>&1; tail -12 $OUT/pytest_all.txt
timeout 600 python3 tools/soak.py --minutes 5 > $OUT/soak.txt 2>&1; tail -6 $OUT/soak.txt
