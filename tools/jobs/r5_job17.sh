#!/bin/bash
# round 5, GPU job 14: the cloud step's tail as selects (default build) against the divergent form (build/ab/flat_loop.so): bits, instruction counts, time
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_job17; mkdir -p $OUT
SUN="config - sunset hurricane.txt"
ms() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f ms/step  single %.2f  kernel %.2f' % (j['ms_per_step'], j['single_frame_ms'], j['roofline']['kernel_ms']))"; }
timeout 900 python3 -m pytest tests/test_gpu_block_scheduler.py tests/test_gpu_round5.py -x -q -m gpu > $OUT/pytest.txt 2>&1
grep -E "passed|failed" $OUT/pytest.txt
B=$PWD/build/ab/flat_loop.so
for rep in 1 2; do
  echo -n "two loops cfg2: " >> $OUT/ab.txt; python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | ms >> $OUT/ab.txt
  echo -n "one loop cfg2: " >> $OUT/ab.txt; DE_LIB_PATH=$B python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | ms >> $OUT/ab.txt
done
for cam in "$SUN" "config - florida.txt" "config - Apollo 11.txt"; do
  echo -n "two loops $cam: " >> $OUT/ab.txt; python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --preset "$cam" 2>/dev/null | ms >> $OUT/ab.txt
  echo -n "one loop $cam: " >> $OUT/ab.txt; DE_LIB_PATH=$B python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --preset "$cam" 2>/dev/null | ms >> $OUT/ab.txt
done
export TMPDIR=/tmp; cd /tmp
for t in twoloops oneloop; do
  if [ $t = oneloop ]; then export DE_LIB_PATH=$B; else unset DE_LIB_PATH; fi
  DE_BIG_SLOTS=1 DE_KERNEL=6 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-include-regex "v6_kernel" --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$t -o pmc -- python3 $GRAFT_REPO_ROOT/tools/frame.py 64 > $GRAFT_REPO_ROOT/$OUT/pmc_$t.log 2>&1
done
unset DE_LIB_PATH
cd $GRAFT_REPO_ROOT
python3 - <<'PY' > $OUT/pmc.txt
import csv, glob, collections
for t in ("twoloops", "oneloop"):
    acc = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/r5_job17/pmc_%s/**/*counter_collection.csv" % t, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(t, {k: "%.4e" % (sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/ab.txt $OUT/pmc.txt
