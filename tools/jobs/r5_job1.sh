#!/bin/bash
# round 5, GPU job 1: counter evidence for the cloud-heavy configs under render_kernel_v6, the kernel's own statistics per camera
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_job1; mkdir -p $OUT
SUN="config - sunset hurricane.txt"
bash tools/profile_gpu.sh r5_sunset --preset "$SUN" > $OUT/prof_sunset.log 2>&1
bash tools/profile_gpu.sh r5_cfg4 --width 3840 --height 2160 --spp 128 --cloud-heavy --preset "$SUN" > $OUT/prof_cfg4.log 2>&1
for cam in default "config - florida.txt" "$SUN" "config - Apollo 11.txt"; do
  if [ "$cam" = default ]; then DE_V6_STATS=1 python3 tools/v6_try.py --time 64 --pipelined 6 > "$OUT/stats_default.txt" 2>&1
  else DE_V6_STATS=1 python3 tools/v6_try.py --time 64 --pipelined 6 --preset "$cam" > "$OUT/stats_$(echo $cam | tr ' ' '_').txt" 2>&1; fi
done
for c in 8 12 16 20; do
  echo "== DE_V6_SVC_CLOUD=$c sunset" >> $OUT/svc_cloud.txt
  DE_V6_SVC_CLOUD=$c python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --preset "$SUN" 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.readlines()[-1]); print(j['ms_per_step'], j['single_frame_ms'])" >> $OUT/svc_cloud.txt
done
ls -la $OUT
