#!/bin/bash
# round 5, GPU job 37: soak run of the shipped kernel (tools/soak.py): four minutes of frames back to back with calls in flight on every launch slot
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_job37; mkdir -p $OUT
timeout 600 python3 tools/soak.py --minutes 4 > $OUT/soak.txt 2>&1
echo "rc $?" >> $OUT/soak.txt
tail -5 $OUT/soak.txt
