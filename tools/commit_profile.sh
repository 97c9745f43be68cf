#!/bin/bash
# After `gpurun -- bash tools/profile_gpu.sh <tag>`: condense gpurun_out/prof_<tag>/ into profiles/<tag>_summary.md + profiles/hbm_traffic.json and
# copy the raw CSVs tests/test_profiles.py recomputes them from to profiles/<tag>_raw/.   usage: tools/commit_profile.sh <tag>
set -e
TAG=$1
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
python3 tools/summarize_profile.py $TAG
mkdir -p profiles/${TAG}_raw
for p in pmc_fetch pmc_l2 pmc_sq1 pmc_sq2 pmc_write; do cp gpurun_out/prof_$TAG/$p/pmc_counter_collection.csv profiles/${TAG}_raw/${p}_counter_collection.csv; done
cp gpurun_out/prof_$TAG/stats/stats_kernel_stats.csv gpurun_out/prof_$TAG/stats/stats_kernel_trace.csv profiles/${TAG}_raw/
ls -la profiles/${TAG}_raw
