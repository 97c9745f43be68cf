#!/bin/bash
# round 5, GPU job 11: sweeps on the refactored default (bands 8, flat_again 32, area 100/72/90): gas table, service areas, yield, lanes tables
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_job11; mkdir -p $OUT
SUN="config - sunset hurricane.txt"
ms() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f ms/step  single %.2f  kernel %.2f' % (j['ms_per_step'], j['single_frame_ms'], j['roofline']['kernel_ms']))"; }
both() {  # label, env...
  label=$1; shift
  echo -n "$label cfg2: " >> $OUT/sweep.txt; env "$@" python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | ms >> $OUT/sweep.txt
  echo -n "$label sunset: " >> $OUT/sweep.txt; env "$@" python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --preset "$SUN" 2>/dev/null | ms >> $OUT/sweep.txt
}
both "default" DE_X=0
both "gas table" DE_LIB_PATH=$PWD/build/ab/gas_table.so
both "default again" DE_X=0
for a in "130 100 120" "80 60 70" "100 72 60" "100 72 130" "100 120 90" "100 40 90" "150 72 90" "60 72 90"; do
  set -- $a
  both "area $1 $2 $3" DE_V6_AREA_ST=$1 DE_V6_AREA_GAS=$2 DE_V6_AREA_CLOUD=$3
done
for y in 40 48 63; do both "yield $y" DE_V6_YIELD=$y; done
for e in 32 64; do both "elsewhere $e" DE_V6_ELSEWHERE=$e; done
for r in 3 10; do both "retry $r" DE_V6_RETRY=$r; done
for m in 16 32; do both "enter_min $m" DE_V6_ENTER_MIN=$m; done
python3 tools/v6_lanes.py --out $OUT/lanes_cfg2.md > /dev/null 2> $OUT/lanes.err
python3 tools/v6_lanes.py --preset "$SUN" --out $OUT/lanes_sunset.md > /dev/null 2>> $OUT/lanes.err
timeout 900 python3 -m pytest tests/test_gpu_round5.py -x -q -m gpu --durations=8 > $OUT/pytest_r5.txt 2>&1
timeout 1200 python3 -m pytest tests -q -m gpu --durations=25 -x > $OUT/pytest_all.txt 2>&1
