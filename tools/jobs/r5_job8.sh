#!/bin/bash
# round 5, GPU job 8: which change costs what — the round-4 tree, the round-4 tree + lap-tagged ring cells only, this tree (refactored default library)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_job8; mkdir -p $OUT
SUN="config - sunset hurricane.txt"
ms() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f ms/step  single %.2f  kernel %.2f' % (j['ms_per_step'], j['single_frame_ms'], j['roofline']['kernel_ms']))"; }
timeout 900 python3 -m pytest tests/test_gpu_block_scheduler.py tests/test_gpu_round5.py -x -q -m gpu > $OUT/pytest.txt 2>&1
for t in r4_tree r4_tags; do
  echo -n "$t cfg2: " >> $OUT/ab.txt; (cd build/$t && python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | ms) >> $OUT/ab.txt
done
echo -n "r5 tree cfg2: " >> $OUT/ab.txt; python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | ms >> $OUT/ab.txt
echo -n "r5 tree cfg2 flat_again 32: " >> $OUT/ab.txt; DE_V6_FLAT_AGAIN=32 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | ms >> $OUT/ab.txt
echo -n "r5 tree cfg2 flat_again 32 area 100 72 90: " >> $OUT/ab.txt; DE_V6_FLAT_AGAIN=32 DE_V6_AREA_ST=100 DE_V6_AREA_GAS=72 DE_V6_AREA_CLOUD=90 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | ms >> $OUT/ab.txt
echo -n "r5 tree sunset flat_again 32 area 100 72 90: " >> $OUT/ab.txt; DE_V6_FLAT_AGAIN=32 DE_V6_AREA_ST=100 DE_V6_AREA_GAS=72 DE_V6_AREA_CLOUD=90 python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --preset "$SUN" 2>/dev/null | ms >> $OUT/ab.txt
export TMPDIR=/tmp; cd /tmp
for t in r4_tree r4_tags r5; do
  if [ $t = r5 ]; then R=$GRAFT_REPO_ROOT; else R=$GRAFT_REPO_ROOT/build/$t; fi
  DE_BIG_SLOTS=1 DE_KERNEL=6 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-include-regex "v6_kernel" --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$t -o pmc -- python3 $R/tools/frame.py 64 > $GRAFT_REPO_ROOT/$OUT/pmc_$t.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY' > $OUT/pmc.txt
import csv, glob, collections
for t in ("r4_tree", "r4_tags", "r5"):
    acc = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/r5_job8/pmc_%s/**/*counter_collection.csv" % t, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(t, {k: "%.4e" % (sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
find $OUT -name "*.csv" -size +2M -delete
