#!/bin/bash
# round 5, GPU job 13: the sphere-trace stage with two records per lane (DE_V6_ST2=1): bits, then time on four cameras and the 1/8 share
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_job13; mkdir -p $OUT
SUN="config - sunset hurricane.txt"
ms() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f ms/step  single %.2f  kernel %.2f' % (j['ms_per_step'], j['single_frame_ms'], j['roofline']['kernel_ms']))"; }
DE_V6_ST2=1 timeout 900 python3 -m pytest tests/test_gpu_block_scheduler.py tests/test_gpu_round5.py -x -q -m gpu > $OUT/pytest_st2.txt 2>&1
tail -3 $OUT/pytest_st2.txt
DE_V6_ST2=1 python3 tools/v6_try.py > $OUT/try_st2.txt 2>&1
for rep in 1 2; do
  for v in 0 1; do
    echo -n "st2=$v cfg2: " >> $OUT/ab.txt; DE_V6_ST2=$v python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | ms >> $OUT/ab.txt
  done
done
for cam in "$SUN" "config - florida.txt" "config - Apollo 11.txt"; do
  for v in 0 1; do
    echo -n "st2=$v $cam: " >> $OUT/ab.txt; DE_V6_ST2=$v python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --preset "$cam" 2>/dev/null | ms >> $OUT/ab.txt
  done
done
for v in 0 1; do
  echo -n "st2=$v cfg2 1/8 share: " >> $OUT/ab.txt; DE_V6_ST2=$v python3 bench.py --no-cpu-baseline --steps 12 --warmup 3 --simulate-world 8 2>/dev/null | ms >> $OUT/ab.txt
done
for a in 60 100 150; do
  echo -n "st2=1 area_st $a cfg2: " >> $OUT/ab.txt; DE_V6_ST2=1 DE_V6_AREA_ST=$a python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | ms >> $OUT/ab.txt
done
DE_V6_ST2=1 python3 tools/v6_lanes.py --out $OUT/lanes_cfg2_st2.md > /dev/null 2> $OUT/lanes.err
cat $OUT/ab.txt
