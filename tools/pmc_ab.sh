#!/bin/bash
# usage: pmc_ab.sh <lib.so>  -> kernel ms + TCC hit/miss of the bench frame
export TMPDIR=/tmp; cd /tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
DE_LIB_PATH=$REPO/$1 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-include-regex "render_kernel_v2" --output-format csv -d /tmp/pmcab_$$ -o pmc -- python3 $REPO/bench.py --steps 2 --warmup 0 --no-cpu-baseline --frames-in-flight 1 > /tmp/pmcab_$$.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/pmcab_$$/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    big=[x for x in v if x > 0.5*max(v)]
    print("$1", k, "%.4g" % (sum(big)/len(big)))
PY
