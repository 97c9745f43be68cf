#!/bin/bash
export TMPDIR=/tmp; cd /tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for LIB in digital_earth_amd/libdigitalearth_hip.so build/ab/v3_lds16.so build/ab/v3_lds64.so; do
  export DE_LIB_PATH=$REPO/$LIB
  rm -rf /tmp/ns1a /tmp/ns1b
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ns1a -o s -- python3 $REPO/tools/ns1_lds_tile.py 2>/dev/null | grep staging
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VALU SQ_INSTS_LDS --kernel-include-regex "v3_primary" --output-format csv -d /tmp/ns1b -o p -- python3 $REPO/tools/ns1_lds_tile.py > /dev/null 2>&1
  python3 - "$LIB" <<'PY'
import csv, glob, sys
ms = None
for f in glob.glob("/tmp/ns1a/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "v3_primary" in r["Name"]: ms = float(r["AverageNs"]) / 1e6
c = {}
for f in glob.glob("/tmp/ns1b/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)): c[r["Counter_Name"]] = float(r["Counter_Value"])
print("%-42s v3_primary_kernel %.3f ms (4 spp) | L1->L2 read requests %.3e  L2 misses %.3e  VALU %.3e  LDS insts %.3e" % (sys.argv[1], ms, c.get("TCP_TCC_READ_REQ_sum", 0), c.get("TCC_MISS_sum", 0), c.get("SQ_INSTS_VALU", 0), c.get("SQ_INSTS_LDS", 0)))
PY
done
python3 -c "
import numpy as np
a=np.load('/tmp/ns1_libdigitalearth_hip.so.npy')
for n in ('v3_lds16.so','v3_lds64.so'):
    b=np.load('/tmp/ns1_%s.npy'%n); print(n, 'bit-identical to the dword tap:', float((a.view(np.uint32)==b.view(np.uint32)).mean()))"
