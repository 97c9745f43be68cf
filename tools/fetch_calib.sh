#!/bin/bash
# Run ON THE GPU BOX: FETCH_SIZE / request counters of tools/micro/fetch_calib per kernel -> gpurun_out/fetch_calib/
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/fetch_calib
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $REPO/tools/micro/fetch_calib.hip -o /tmp/fetch_calib || exit 1
export TMPDIR=/tmp
cd /tmp
/tmp/fetch_calib > $OUT/timing.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- /tmp/fetch_calib > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pmc_req -o pmc -- /tmp/fetch_calib > $OUT/pmc_req.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("pmc_fetch", "pmc_req"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print("%-40s %-24s mean %.6g over %d launches" % (k[0][:40], k[1], sum(v) / len(v), len(v)))
PY
cat $OUT/timing.txt | tail -3
