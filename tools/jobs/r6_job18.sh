#!/bin/bash
# round 6, GPU job 18: measurements of record for the round-6 kernel — rocprofv3 passes of cfg2 (+ the tail-exporting passes), the bench line, cameras, the kernel's
# statistics, the window loop, simulated scaling with the stand-in collective at the configs' own sizes
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6_job18; mkdir -p $OUT
bash tools/profile_gpu.sh r6f > $OUT/prof_r6f.log 2>&1
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_r6f.json 2> $OUT/bench_r6f.err
python3 tools/v6_lanes.py --out $OUT/lanes_cfg2.md > /dev/null 2> $OUT/lanes.err
python3 tools/v6_lanes.py --preset "config - sunset hurricane.txt" --out $OUT/lanes_sunset.md > /dev/null 2>> $OUT/lanes.err
python3 tools/v6_lanes.py --world 8 --out $OUT/lanes_cfg2_world8.md > /dev/null 2>> $OUT/lanes.err
python3 tools/v6_lanes.py --spp 1 --out $OUT/lanes_1spp.md > /dev/null 2>> $OUT/lanes.err
ms() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f ms/step  single %.2f  kernel %.2f  value %.1f' % (j['ms_per_step'], j['single_frame_ms'], j['roofline']['kernel_ms'], j['value']))"; }
for cam in "config - sunset hurricane.txt" "config - florida.txt" "config - Apollo 11.txt"; do
  echo -n "$cam 1920x1080x64: " >> $OUT/cameras.txt; python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 --preset "$cam" 2>/dev/null | ms >> $OUT/cameras.txt
done
for c in cfg3 cfg4 cfg5; do echo -n "$c (bench.py --config $c): " >> $OUT/cameras.txt; python3 bench.py --no-cpu-baseline --config $c --steps 3 --warmup 1 2>/dev/null | ms >> $OUT/cameras.txt; done
echo "window loop (tools/one_spp.py):" >> $OUT/cameras.txt; python3 tools/one_spp.py 2>&1 | tail -8 >> $OUT/cameras.txt
cat $OUT/cameras.txt
bash tools/simulated_world.sh > $OUT/simulated_world.log 2>&1
cp gpurun_out/simulated_world.json $OUT/ 2>/dev/null
tail -8 $OUT/simulated_world.log
