#!/bin/bash
# usage: pmc_ab.sh <lib.so> [more libs...]  -> per library: kernel ms of the bench frame (HIP events) + TCC hit / miss / requests of render_kernel_v2
export TMPDIR=/tmp; cd /tmp
export DE_BIG_SLOTS=1          # one launch at a time: clean kernel durations
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for LIB in "$@"; do
  export DE_LIB_PATH=$REPO/$LIB
  MS=$(python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.2f %.2f' % (j['roofline']['kernel_ms'], j['ms_per_step']))")
  rm -rf /tmp/pmcab_$$
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --kernel-include-regex "render_kernel_v2" --output-format csv -d /tmp/pmcab_$$ -o pmc -- python3 $REPO/bench.py --steps 2 --warmup 0 --no-cpu-baseline > /tmp/pmcab_$$.log 2>&1
  python3 - "$LIB" "$MS" /tmp/pmcab_$$ <<'PY'
import csv, glob, collections, sys, os
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[3] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {}
for k, v in acc.items():
    big = [x for x in v if x > 0.5 * max(v)]
    m[k] = sum(big) / len(big)
hit, miss = m.get("TCC_HIT_sum", 0), m.get("TCC_MISS_sum", 0)
print("%-28s kernel/step ms %s | L2 hit rate %.3f  misses %.3e  (x128 B = %.0f GB)  requests %.3e" % (sys.argv[1], sys.argv[2], hit / max(hit + miss, 1), miss, miss * 128 / 1e9, m.get("TCP_TCC_READ_REQ_sum", 0)))
PY
  rm -rf /tmp/pmcab_$$ /tmp/pmcab_$$.log
done
