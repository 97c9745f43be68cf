#!/bin/bash
# Run ON THE GPU BOX: PMC passes over the persistent stage-scheduler kernel (tools/frame.py).  usage: tools/v5_pmc.sh <tag> [spp] [preset]
TAG=${1:-v5}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT; export TMPDIR=/tmp; export DE_BIG_SLOTS=1; cd /tmp
RX="v5_kernel|v6_kernel|render_kernel_v2"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --kernel-include-regex "$RX" --output-format csv -d $OUT/sq1 -o pmc -- python3 $REPO/tools/frame.py "$@" > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "$RX" --output-format csv -d $OUT/sq2 -o pmc -- python3 $REPO/tools/frame.py "$@" > $OUT/sq2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --kernel-include-regex "$RX" --output-format csv -d $OUT/l2 -o pmc -- python3 $REPO/tools/frame.py "$@" > $OUT/l2.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --kernel-include-regex "$RX" --output-format csv -d $OUT/sq3 -o pmc -- python3 $REPO/tools/frame.py "$@" > $OUT/sq3.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    m = {}
    for c, v in sorted(d.items()):
        m[c] = sum(v) / len(v)
        print("   %-26s %.4e  (n=%d)" % (c, m[c], len(v)))
    if "SQ_THREAD_CYCLES_VALU" in m and "SQ_ACTIVE_INST_VALU" in m: print("   lane utilisation %.3f" % (m["SQ_THREAD_CYCLES_VALU"] / (64 * m["SQ_ACTIVE_INST_VALU"])))
    if "TCC_HIT_sum" in m: print("   L2 hit rate %.3f" % (m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])))
PY
grep frame $OUT/sq1.log
find $OUT -name "*.csv" -size +8M -delete
