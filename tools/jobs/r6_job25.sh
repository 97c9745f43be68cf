#!/bin/bash
# round 6, GPU job 25: the kernel's statistics for the remaining cameras and BASELINE cfg4, and the price of the arithmetic contract (DE_FLAG_FAST_MATH) on the round-6 kernel
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6_job25; mkdir -p $OUT
python3 tools/v6_lanes.py --preset "config - florida.txt" --out $OUT/lanes_florida.md > /dev/null 2> $OUT/lanes.err
python3 tools/v6_lanes.py --preset "config - Apollo 11.txt" --out $OUT/lanes_apollo.md > /dev/null 2>> $OUT/lanes.err
python3 tools/v6_lanes.py --preset "config - sunset hurricane.txt" --width 3840 --height 2160 --spp 64 --cloud-heavy --out $OUT/lanes_cfg4.md > /dev/null 2>> $OUT/lanes.err
python3 tools/fast_math_price.py > $OUT/fast_math.md 2> $OUT/fast_math.err
tail -12 $OUT/fast_math.md
