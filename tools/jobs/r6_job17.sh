#!/bin/bash
# round 6, GPU job 17: the scheduler's knobs again, now that a stage entry costs no scratch traffic and new paths are twice as fast (one knob at a time around the defaults)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6_job17; mkdir -p $OUT
python3 tools/v6_sweep.py --frames 10 "DE_V6_AREA_ST=60,80,100,130" "DE_V6_AREA_GAS=40,56,72,96" "DE_V6_AREA_CLOUD=60,75,90,110,130" "DE_V6_YIELD=40,48,56,60" "DE_V6_ELSEWHERE=32,48,64,96" "DE_V6_RETRY=3,6,10" "DE_V6_FLAT_AGAIN=16,32,48,64" 2>&1 | tee $OUT/sweep_cfg2.txt
python3 tools/v6_sweep.py --frames 6 --preset "config - sunset hurricane.txt" "DE_V6_AREA_CLOUD=60,90,130" "DE_V6_YIELD=40,56" "DE_V6_FLAT_AGAIN=16,32,64" 2>&1 | tee $OUT/sweep_sunset.txt
