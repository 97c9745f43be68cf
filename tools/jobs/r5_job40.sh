#!/bin/bash
# round 5, GPU job 40: the counter passes of the cloud-heavy configs again, on the SHIPPED kernel (the committed r5_sunset / r5_cfg4 summaries are of
# the build before the instruction diet), condensed on the box
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_job40; mkdir -p $OUT/profiles
SUN="config - sunset hurricane.txt"
bash tools/profile_gpu.sh r5_sunset --preset "$SUN" > $OUT/prof_sunset.log 2>&1
python3 tools/summarize_profile.py r5_sunset 1920x1080_64spp_q2_sunset_hurricane nojson > $OUT/sum_sunset.log 2>&1
bash tools/profile_gpu.sh r5_cfg4 --width 3840 --height 2160 --spp 128 --cloud-heavy --preset "$SUN" > $OUT/prof_cfg4.log 2>&1
python3 tools/summarize_profile.py r5_cfg4 3840x2160_128spp_q2_cloud_heavy_sunset_hurricane nojson > $OUT/sum_cfg4.log 2>&1
for t in r5_sunset r5_cfg4; do
  mkdir -p $OUT/profiles/${t}_raw
  cp profiles/${t}_summary.md $OUT/profiles/
  for p in pmc_fetch pmc_l2 pmc_sq1 pmc_sq2 pmc_write; do cp gpurun_out/prof_$t/$p/pmc_counter_collection.csv $OUT/profiles/${t}_raw/${p}_counter_collection.csv; done
  cp gpurun_out/prof_$t/stats/stats_kernel_stats.csv $OUT/profiles/${t}_raw/
done
tail -30 $OUT/profiles/r5_sunset_summary.md
