#!/bin/bash
# round 5, GPU job 10: one work counter per XCD (image bands) against one counter; the fast-math price; the whole GPU suite on the refactored library
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_job10; mkdir -p $OUT
SUN="config - sunset hurricane.txt"
ms() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f ms/step  single %.2f  kernel %.2f' % (j['ms_per_step'], j['single_frame_ms'], j['roofline']['kernel_ms']))"; }
timeout 1200 python3 -m pytest tests/test_gpu_block_scheduler.py tests/test_gpu_round5.py -x -q -m gpu > $OUT/pytest_quick.txt 2>&1
for b in 8 1 8 1; do
  echo -n "bands $b cfg2: " >> $OUT/ab.txt; DE_V6_BANDS=$b python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | ms >> $OUT/ab.txt
done
for cam in "$SUN" "config - florida.txt" "config - Apollo 11.txt"; do
  for b in 8 1; do
    echo -n "bands $b $cam: " >> $OUT/ab.txt; DE_V6_BANDS=$b python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --preset "$cam" 2>/dev/null | ms >> $OUT/ab.txt
  done
done
for b in 8 1; do
  echo -n "bands $b cfg2 1/8 share: " >> $OUT/ab.txt; DE_V6_BANDS=$b python3 bench.py --no-cpu-baseline --steps 12 --warmup 3 --simulate-world 8 2>/dev/null | ms >> $OUT/ab.txt
done
export TMPDIR=/tmp; cd /tmp
for b in 8 1; do
  DE_V6_BANDS=$b DE_BIG_SLOTS=1 DE_KERNEL=6 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --kernel-include-regex "v6_kernel" --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_l2_b$b -o pmc -- python3 $GRAFT_REPO_ROOT/tools/frame.py 64 > $GRAFT_REPO_ROOT/$OUT/pmc_l2_b$b.log 2>&1
  DE_V6_BANDS=$b DE_BIG_SLOTS=1 DE_KERNEL=6 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "v6_kernel" --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_fetch_b$b -o pmc -- python3 $GRAFT_REPO_ROOT/tools/frame.py 64 > $GRAFT_REPO_ROOT/$OUT/pmc_fetch_b$b.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY' > $OUT/pmc.txt
import csv, glob, collections
for t in ("l2_b8", "l2_b1", "fetch_b8", "fetch_b1"):
    acc = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/r5_job10/pmc_%s/**/*counter_collection.csv" % t, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(t, {k: "%.4e" % (sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
find $OUT -name "*.csv" -size +2M -delete
python3 tools/fast_math_price.py --quick > $OUT/fast_math_quick.md 2> $OUT/fast_math.err
timeout 2400 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_all.txt 2>&1
tail -5 $OUT/pytest_all.txt
