#!/bin/bash
# round 5, GPU job 5: where do the 4 % between the round-4 tree and this one come from? — counters of one frame under both
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_job5; mkdir -p $OUT
(cd build/r4_tree && DE_KERNEL=6 GRAFT_REPO_ROOT=$PWD bash tools/v5_pmc.sh r4tree 64 > $GRAFT_REPO_ROOT/$OUT/pmc_r4.txt 2>&1)
DE_KERNEL=6 bash tools/v5_pmc.sh r5tree 64 > $OUT/pmc_r5.txt 2>&1
ls -la $OUT
