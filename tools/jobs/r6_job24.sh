#!/bin/bash
# round 6, GPU job 24: counter passes of the cloud-heavy configs on the round-6 kernel (as r5_job40 for round 5), condensed on the box
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6_job24; mkdir -p $OUT/profiles
SUN="config - sunset hurricane.txt"
bash tools/profile_gpu.sh r6_sunset --preset "$SUN" > $OUT/prof_sunset.log 2>&1
python3 tools/summarize_profile.py r6_sunset 1920x1080_64spp_q2_sunset_hurricane nojson > $OUT/sum_sunset.log 2>&1
bash tools/profile_gpu.sh r6_cfg4 --config cfg4 > $OUT/prof_cfg4.log 2>&1
python3 tools/summarize_profile.py r6_cfg4 3840x2160_128spp_q2_cloud_heavy_sunset_hurricane nojson > $OUT/sum_cfg4.log 2>&1
for t in r6_sunset r6_cfg4; do
  mkdir -p $OUT/profiles/${t}_raw
  cp profiles/${t}_summary.md $OUT/profiles/
  for p in pmc_fetch pmc_l2 pmc_sq1 pmc_sq2 pmc_write; do cp gpurun_out/prof_$t/$p/pmc_counter_collection.csv $OUT/profiles/${t}_raw/${p}_counter_collection.csv; done
  cp gpurun_out/prof_$t/stats/stats_kernel_stats.csv $OUT/profiles/${t}_raw/
done
grep -E "frames in the trace|lane utilisation =|L2-miss read|write bytes|v6_kernel<false, false, false>. \|" $OUT/profiles/r6_sunset_summary.md $OUT/profiles/r6_cfg4_summary.md | cut -c1-260
