#!/bin/bash
# round 5, GPU job 4: scalar stage constants + saddr map loads (the in-loop scratch reload gone) against the round-4 tree on the same box; the area rule
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_job4; mkdir -p $OUT
SUN="config - sunset hurricane.txt"
ms() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f ms/step  single %.2f  kernel %.2f' % (j['ms_per_step'], j['single_frame_ms'], j['roofline']['kernel_ms']))"; }
timeout 1500 python3 -m pytest tests/test_gpu_block_scheduler.py tests/test_gpu_round5.py -x -q -m gpu > $OUT/pytest.txt 2>&1
for rep in 1 2; do
  echo -n "r4 tree cfg2: " >> $OUT/ab.txt; (cd build/r4_tree && python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | ms) >> $OUT/ab.txt
  echo -n "r5 tree cfg2: " >> $OUT/ab.txt; python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | ms >> $OUT/ab.txt
done
echo -n "r4 tree sunset: " >> $OUT/ab.txt; (cd build/r4_tree && python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --preset "$SUN" 2>/dev/null | ms) >> $OUT/ab.txt
echo -n "r5 tree sunset: " >> $OUT/ab.txt; python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --preset "$SUN" 2>/dev/null | ms >> $OUT/ab.txt
# the area rule: idle lane-trips since the last service at which a loop stage services (sphere trace / gas / cloud)
for cfg in "72 36 90" "72 72 90" "100 72 90" "72 36 60" "72 36 130" "48 36 90" "48 24 60" "100 100 130" "72 36 45" "36 24 45"; do
  set -- $cfg
  echo -n "area $1 $2 $3 cfg2: " >> $OUT/area.txt; DE_V6_AREA_ST=$1 DE_V6_AREA_GAS=$2 DE_V6_AREA_CLOUD=$3 python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | ms >> $OUT/area.txt
  echo -n "area $1 $2 $3 sunset: " >> $OUT/area.txt; DE_V6_AREA_ST=$1 DE_V6_AREA_GAS=$2 DE_V6_AREA_CLOUD=$3 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --preset "$SUN" 2>/dev/null | ms >> $OUT/area.txt
done
for cam in "config - florida.txt" "config - Apollo 11.txt"; do
  echo -n "thresholds $cam: " >> $OUT/area.txt; python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --preset "$cam" 2>/dev/null | ms >> $OUT/area.txt
  echo -n "area 72 36 90 $cam: " >> $OUT/area.txt; DE_V6_AREA_ST=72 DE_V6_AREA_GAS=36 DE_V6_AREA_CLOUD=90 python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --preset "$cam" 2>/dev/null | ms >> $OUT/area.txt
done
python3 tools/v6_lanes.py --out $OUT/lanes_cfg2.md > /dev/null 2> $OUT/lanes.err
ls -la $OUT
