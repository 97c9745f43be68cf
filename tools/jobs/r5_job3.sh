#!/bin/bash
# round 5, GPU job 3: tagged rings + single-instruction appends against the round-4 tree ON THE SAME BOX; thresholds re-tuned; new tests
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_job3; mkdir -p $OUT
SUN="config - sunset hurricane.txt"
ms() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f ms/step  single %.2f  kernel %.2f' % (j['ms_per_step'], j['single_frame_ms'], j['roofline']['kernel_ms']))"; }
timeout 1500 python3 -m pytest tests/test_gpu_block_scheduler.py tests/test_gpu_round5.py -x -q -m gpu > $OUT/pytest.txt 2>&1
for rep in 1 2; do
  echo -n "r4 tree cfg2: " >> $OUT/ab.txt; (cd build/r4_tree && python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | ms) >> $OUT/ab.txt
  echo -n "r5 tree cfg2: " >> $OUT/ab.txt; python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | ms >> $OUT/ab.txt
done
echo -n "r4 tree sunset: " >> $OUT/ab.txt; (cd build/r4_tree && python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --preset "$SUN" 2>/dev/null | ms) >> $OUT/ab.txt
echo -n "r5 tree sunset: " >> $OUT/ab.txt; python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --preset "$SUN" 2>/dev/null | ms >> $OUT/ab.txt
# thresholds (idle lanes at which a loop stage services), default 24 / 24 / 20
for cfg in "16 24 20" "24 16 20" "24 24 12" "24 24 16" "16 16 12" "16 16 16" "20 20 16" "12 12 12"; do
  set -- $cfg
  echo -n "svc $1 $2 $3 cfg2: " >> $OUT/svc.txt; DE_V6_SVC_ST=$1 DE_V6_SVC_GAS=$2 DE_V6_SVC_CLOUD=$3 python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | ms >> $OUT/svc.txt
  echo -n "svc $1 $2 $3 sunset: " >> $OUT/svc.txt; DE_V6_SVC_ST=$1 DE_V6_SVC_GAS=$2 DE_V6_SVC_CLOUD=$3 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --preset "$SUN" 2>/dev/null | ms >> $OUT/svc.txt
done
python3 tools/v6_lanes.py --out $OUT/lanes_cfg2.md > /dev/null 2> $OUT/lanes.err
python3 tools/v6_lanes.py --preset "$SUN" --out $OUT/lanes_sunset.md > /dev/null 2>> $OUT/lanes.err
ls -la $OUT
