#!/bin/bash
# usage: tools/kernel_resources.sh [regex]  -> one line per kernel: VGPRs, SGPRs, spills, scratch, occupancy, LDS, code bytes (hipcc cross-compiles: no GPU needed)
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd $REPO/digital_earth_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-value -mllvm -enable-ipra -fno-optimize-sibling-calls ${DE_EXTRA_FLAGS} -S --cuda-device-only \
  -Rpass-analysis=kernel-resource-usage de_api.hip -o /tmp/de_res.s 2> /tmp/de_res.txt
python3 - "$1" <<'PY'
import re, sys
t = open('/tmp/de_res.txt').read()
pat = re.compile(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] else None
for m in re.finditer(r"Function Name: (\S+).*?TotalSGPRs: (\d+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?SGPRs Spill: (\d+).*?VGPRs Spill: (\d+).*?LDS Size \[bytes/block\]: (\d+)", t, re.S):
    name = m.group(1)
    if pat and not pat.search(name): continue
    print("%-70s VGPR %3s SGPR %3s scratch %3s occ %s sgpr-spill %3s vgpr-spill %3s LDS %5s" % (name[:70], m.group(3), m.group(2), m.group(4), m.group(5), m.group(6), m.group(7), m.group(8)))
PY
