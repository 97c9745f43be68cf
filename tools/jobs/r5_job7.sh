#!/bin/bash
# round 5, GPU job 7: instruction counts of one frame (r4 tree, this tree, gas densities from the table) + flat_again sweep
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_job7; mkdir -p $OUT
SUN="config - sunset hurricane.txt"
ms() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f ms/step  single %.2f  kernel %.2f' % (j['ms_per_step'], j['single_frame_ms'], j['roofline']['kernel_ms']))"; }
timeout 900 python3 -m pytest tests/test_gpu_block_scheduler.py -x -q -m gpu > $OUT/pytest.txt 2>&1
echo -n "r4 tree cfg2: " >> $OUT/ab.txt; (cd build/r4_tree && python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | ms) >> $OUT/ab.txt
echo -n "r5 tree cfg2: " >> $OUT/ab.txt; python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | ms >> $OUT/ab.txt
echo -n "r5 gas table cfg2: " >> $OUT/ab.txt; DE_LIB_PATH=$PWD/build/ab/gas_table.so python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | ms >> $OUT/ab.txt
for fa in 8 16 24 32 48; do
  echo -n "r5 flat_again $fa cfg2: " >> $OUT/ab.txt; DE_V6_FLAT_AGAIN=$fa python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | ms >> $OUT/ab.txt
done
echo -n "r5 area 100 72 90 flat_again 24 cfg2: " >> $OUT/ab.txt; DE_V6_FLAT_AGAIN=24 DE_V6_AREA_ST=100 DE_V6_AREA_GAS=72 DE_V6_AREA_CLOUD=90 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | ms >> $OUT/ab.txt
echo -n "r4 tree sunset: " >> $OUT/ab.txt; (cd build/r4_tree && python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --preset "$SUN" 2>/dev/null | ms) >> $OUT/ab.txt
echo -n "r5 tree sunset: " >> $OUT/ab.txt; python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --preset "$SUN" 2>/dev/null | ms >> $OUT/ab.txt
echo -n "r5 gas table sunset: " >> $OUT/ab.txt; DE_LIB_PATH=$PWD/build/ab/gas_table.so python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --preset "$SUN" 2>/dev/null | ms >> $OUT/ab.txt
echo -n "r5 tree sunset area 100 72 90: " >> $OUT/ab.txt; DE_V6_AREA_ST=100 DE_V6_AREA_GAS=72 DE_V6_AREA_CLOUD=90 python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --preset "$SUN" 2>/dev/null | ms >> $OUT/ab.txt
export TMPDIR=/tmp; cd /tmp
for t in r4 r5 gastab; do
  R=$GRAFT_REPO_ROOT; L=""
  if [ $t = r4 ]; then R=$GRAFT_REPO_ROOT/build/r4_tree; fi
  if [ $t = gastab ]; then export DE_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/gas_table.so; else unset DE_LIB_PATH; fi
  DE_BIG_SLOTS=1 DE_KERNEL=6 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-include-regex "v6_kernel" --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$t -o pmc -- python3 $R/tools/frame.py 64 > $GRAFT_REPO_ROOT/$OUT/pmc_$t.log 2>&1
done
unset DE_LIB_PATH
cd $GRAFT_REPO_ROOT
python3 - <<'PY' > $OUT/pmc.txt
import csv, glob, collections
for t in ("r4", "r5", "gastab"):
    acc = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/r5_job7/pmc_%s/**/*counter_collection.csv" % t, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(t, {k: "%.4e" % (sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
find $OUT -name "*.csv" -size +2M -delete
