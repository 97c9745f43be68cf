#!/bin/bash
# usage: pmc_ab6.sh <lib.so | main> [...]  -> per library, for v6_kernel on BASELINE cfg2 (one frame at a time): kernel ms, fabric bytes written / read, L2 hit rate, VALU / SALU instructions
export TMPDIR=/tmp; cd /tmp
export DE_BIG_SLOTS=1
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for LIB in "$@"; do
  if [ "$LIB" = main ]; then unset DE_LIB_PATH; else export DE_LIB_PATH=$REPO/$LIB; fi
  MS=$(python3 $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f %.2f' % (j['roofline']['kernel_ms'], j['ms_per_step']))")
  D=/tmp/pmcab6_$$; rm -rf $D
  for PASS in "WRITE_SIZE" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU"; do
    rocprofv3 --pmc $PASS --kernel-include-regex "v6_kernel" --output-format csv -d $D/$(echo $PASS | cut -c1-8) -o pmc -- python3 $REPO/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-other-configs --sync-frames > $D.log 2>&1
  done
  python3 - "$LIB" "$MS" $D <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[3] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {}
for k, v in acc.items():
    big = [x for x in v if x > 0.5 * max(v)]      # the frame launches (the tail launch and the small priming calls are far below)
    m[k] = sum(big) / len(big)
hit, miss = m.get("TCC_HIT_sum", 0), m.get("TCC_MISS_sum", 0)
# WRITE_SIZE / FETCH_SIZE: 32-byte units as rocprofv3 reports them on this pool x the calibration of profiles/r2_fetch_calibration (FETCH x2)
print("%-24s kernel / step ms %s | WRITE_SIZE %.1f GB  FETCH_SIZE x2 %.1f GB | L2 hit %.3f misses %.3e | VALU %.3e SALU %.3e lane util %.3f" % (
    sys.argv[1], sys.argv[2], m.get("WRITE_SIZE", 0) * 1024 / 1e9, 2 * m.get("FETCH_SIZE", 0) * 1024 / 1e9, hit / max(hit + miss, 1), miss,
    m.get("SQ_INSTS_VALU", 0), m.get("SQ_INSTS_SALU", 0), m.get("SQ_THREAD_CYCLES_VALU", 0) / max(64 * m.get("SQ_ACTIVE_INST_VALU", 1), 1)))
PY
  rm -rf $D $D.log
done
