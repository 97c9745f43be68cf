#!/bin/bash
# round 5, GPU job 33: the measurements of record for the build with the tail chain — rocprofv3 passes of cfg2, the bench line, simulated scaling at the configs' own sizes,
# the fast-math price, the kernel's statistics per camera, the whole GPU suite
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_job33; mkdir -p $OUT
SUN="config - sunset hurricane.txt"
bash tools/profile_gpu.sh r5f > $OUT/prof_r5f.log 2>&1
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_r5f.json 2> $OUT/bench_r5f.err
python3 tools/v6_lanes.py --out $OUT/lanes_cfg2.md > /dev/null 2> $OUT/lanes.err
python3 tools/v6_lanes.py --preset "$SUN" --out $OUT/lanes_sunset.md > /dev/null 2>> $OUT/lanes.err
python3 tools/v6_lanes.py --preset "config - florida.txt" --out $OUT/lanes_florida.md > /dev/null 2>> $OUT/lanes.err
python3 tools/v6_lanes.py --preset "config - Apollo 11.txt" --out $OUT/lanes_apollo.md > /dev/null 2>> $OUT/lanes.err
python3 tools/v6_lanes.py --preset "$SUN" --width 3840 --height 2160 --spp 64 --cloud-heavy --out $OUT/lanes_cfg4.md > /dev/null 2>> $OUT/lanes.err
python3 tools/v6_lanes.py --world 8 --out $OUT/lanes_cfg2_world8.md > /dev/null 2>> $OUT/lanes.err
ms() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f ms/step  single %.2f  kernel %.2f  value %.1f' % (j['ms_per_step'], j['single_frame_ms'], j['roofline']['kernel_ms'], j['value']))"; }
for cam in "$SUN" "config - florida.txt" "config - Apollo 11.txt"; do
  echo -n "$cam 1920x1080x64: " >> $OUT/cameras.txt; python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 --preset "$cam" 2>/dev/null | ms >> $OUT/cameras.txt
done
echo -n "cfg4 3840x2160x128 cloud-heavy sunset: " >> $OUT/cameras.txt; python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --width 3840 --height 2160 --spp 128 --cloud-heavy --preset "$SUN" 2>/dev/null | ms >> $OUT/cameras.txt
echo -n "window loop (tools/one_spp.py): " >> $OUT/cameras.txt; python3 tools/one_spp.py 2>&1 | tail -4 >> $OUT/cameras.txt
bash tools/simulated_world.sh > $OUT/simulated_world.log 2>&1
cp gpurun_out/simulated_world.json $OUT/ 2>/dev/null
python3 tools/fast_math_price.py > $OUT/fast_math.md 2> $OUT/fast_math.err
timeout 1500 python3 -m pytest tests -q -m gpu --durations=12 > $OUT/pytest_all.txt 2>&1
tail -4 $OUT/pytest_all.txt
