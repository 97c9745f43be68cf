#!/bin/bash
# round 5, GPU job 9: straight-line take / push, new defaults (service by idle lane-trips 100 / 72 / 90, further flat passes from 32 entries) against the round-4 tree
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_job9; mkdir -p $OUT
SUN="config - sunset hurricane.txt"
ms() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f ms/step  single %.2f  kernel %.2f' % (j['ms_per_step'], j['single_frame_ms'], j['roofline']['kernel_ms']))"; }
timeout 1200 python3 -m pytest tests/test_gpu_block_scheduler.py tests/test_gpu_round5.py tests/test_gpu_legacy_suite.py -x -q -m gpu > $OUT/pytest.txt 2>&1
echo -n "r4_tree cfg2: " >> $OUT/ab.txt; (cd build/r4_tree && python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | ms) >> $OUT/ab.txt
echo -n "r5 cfg2 (defaults): " >> $OUT/ab.txt; python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | ms >> $OUT/ab.txt
echo -n "r5 cfg2 thresholds, flat_again 1 (= round 4's rules): " >> $OUT/ab.txt; DE_V6_FLAT_AGAIN=1 DE_V6_AREA_ST=0 DE_V6_AREA_GAS=0 DE_V6_AREA_CLOUD=0 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | ms >> $OUT/ab.txt
for fa in 16 24 48 64; do
  echo -n "r5 cfg2 flat_again $fa: " >> $OUT/ab.txt; DE_V6_FLAT_AGAIN=$fa python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | ms >> $OUT/ab.txt
done
for fm in 16 32 48; do
  echo -n "r5 cfg2 flat_min $fm: " >> $OUT/ab.txt; DE_V6_FLAT_MIN=$fm python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | ms >> $OUT/ab.txt
done
for cam in "$SUN" "config - florida.txt" "config - Apollo 11.txt"; do
  echo -n "r4_tree $cam: " >> $OUT/ab.txt; (cd build/r4_tree && python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --preset "$cam" 2>/dev/null | ms) >> $OUT/ab.txt
  echo -n "r5 $cam: " >> $OUT/ab.txt; python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --preset "$cam" 2>/dev/null | ms >> $OUT/ab.txt
done
export TMPDIR=/tmp; cd /tmp
for t in r4_tree r5; do
  if [ $t = r5 ]; then R=$GRAFT_REPO_ROOT; else R=$GRAFT_REPO_ROOT/build/$t; fi
  DE_BIG_SLOTS=1 DE_KERNEL=6 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-include-regex "v6_kernel" --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$t -o pmc -- python3 $R/tools/frame.py 64 > $GRAFT_REPO_ROOT/$OUT/pmc_$t.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY' > $OUT/pmc.txt
import csv, glob, collections
for t in ("r4_tree", "r5"):
    acc = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/r5_job9/pmc_%s/**/*counter_collection.csv" % t, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(t, {k: "%.4e" % (sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
find $OUT -name "*.csv" -size +2M -delete
