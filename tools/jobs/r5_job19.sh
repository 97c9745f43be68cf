#!/bin/bash
# round 5, GPU job 19: the whole GPU suite on the two-loop kernel, then the service constants and flat_again once more (the trips got cheaper)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_job19; mkdir -p $OUT
SUN="config - sunset hurricane.txt"
ms() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f ms/step  single %.2f  kernel %.2f' % (j['ms_per_step'], j['single_frame_ms'], j['roofline']['kernel_ms']))"; }
timeout 1500 python3 -m pytest tests -q -m gpu -x > $OUT/pytest_all.txt 2>&1
grep -E "passed|failed" $OUT/pytest_all.txt
both() {
  label=$1; shift
  echo -n "$label cfg2: " >> $OUT/sweep.txt; env "$@" python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | ms >> $OUT/sweep.txt
  echo -n "$label sunset: " >> $OUT/sweep.txt; env "$@" python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --preset "$SUN" 2>/dev/null | ms >> $OUT/sweep.txt
}
both "default (100 72 90, flat_again 32)" DE_X=0
for a in "70 50 60" "140 100 130" "100 72 60" "100 72 130" "100 110 90" "100 50 90" "150 72 90" "60 72 90" "200 150 180"; do
  set -- $a
  both "area $1 $2 $3" DE_V6_AREA_ST=$1 DE_V6_AREA_GAS=$2 DE_V6_AREA_CLOUD=$3
done
for fa in 16 48; do both "flat_again $fa" DE_V6_FLAT_AGAIN=$fa; done
for y in 48 63; do both "yield $y" DE_V6_YIELD=$y; done
both "default again" DE_X=0
cat $OUT/sweep.txt
