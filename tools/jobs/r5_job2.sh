#!/bin/bash
# round 5, GPU job 2: tagged ring cells (bit-identity, small-pool soak), region statistics per camera, rocprofv3 passes of the cloud-heavy configs
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_job2; mkdir -p $OUT
SUN="config - sunset hurricane.txt"
timeout 900 python3 -m pytest tests/test_gpu_block_scheduler.py -x -q -m gpu > $OUT/pytest_block.txt 2>&1
python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 > $OUT/bench_default.json 2> $OUT/bench_default.err
for v in v6_p128 v6_p64 v6_p1120_cap1121; do
  echo "== $v" >> $OUT/soak.txt
  DE_LIB_PATH=build/ab/$v.so DE_SOAK_VARIANT=6 timeout 600 python3 tools/v5_soak.py 4 >> $OUT/soak.txt 2>&1
done
python3 tools/v6_lanes.py --out $OUT/lanes_cfg2.md > /dev/null 2> $OUT/lanes.err
python3 tools/v6_lanes.py --preset "$SUN" --out $OUT/lanes_sunset.md > /dev/null 2>> $OUT/lanes.err
python3 tools/v6_lanes.py --preset "config - florida.txt" --out $OUT/lanes_florida.md > /dev/null 2>> $OUT/lanes.err
python3 tools/v6_lanes.py --preset "config - Apollo 11.txt" --out $OUT/lanes_apollo.md > /dev/null 2>> $OUT/lanes.err
python3 tools/v6_lanes.py --preset "$SUN" --width 3840 --height 2160 --spp 128 --cloud-heavy --out $OUT/lanes_cfg4.md > /dev/null 2>> $OUT/lanes.err
python3 tools/v6_lanes.py --world 8 --out $OUT/lanes_cfg2_world8.md > /dev/null 2>> $OUT/lanes.err
bash tools/profile_gpu.sh r5_sunset --preset "$SUN" > $OUT/prof_sunset.log 2>&1
bash tools/profile_gpu.sh r5_cfg4 --width 3840 --height 2160 --spp 128 --cloud-heavy --preset "$SUN" > $OUT/prof_cfg4.log 2>&1
ls -la $OUT
