#!/bin/bash
# round 5, GPU job 21: the launch's drain, measured (population of the resident workgroups in 0.25 ms buckets after the work counter ran out), and the
# geometry question it raises: three workgroups of eight waves / four of six per CU (a draining workgroup holds a third / a quarter of the CU, not half)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_job21; mkdir -p $OUT
SUN="config - sunset hurricane.txt"
ms() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f ms/step  single %.2f  kernel %.2f  drain %s' % (j['ms_per_step'], j['single_frame_ms'], j['roofline']['kernel_ms'], j.get('drain_ms')))"; }
bash tools/profile_gpu.sh r5f > $OUT/prof_r5f.log 2>&1
python3 tools/v6_lanes.py --out $OUT/lanes_cfg2.md > /dev/null 2> $OUT/lanes.err
python3 tools/v6_lanes.py --world 8 --out $OUT/lanes_cfg2_world8.md > /dev/null 2>> $OUT/lanes.err
python3 tools/v6_lanes.py --preset "$SUN" --out $OUT/lanes_sunset.md > /dev/null 2>> $OUT/lanes.err
python3 tools/v6_lanes.py --preset "config - Apollo 11.txt" --out $OUT/lanes_apollo.md > /dev/null 2>> $OUT/lanes.err
for lib in main v6_w8x3 v6_w6x4 main; do
  if [ $lib = main ]; then unset DE_LIB_PATH; else export DE_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/$lib.so; fi
  echo -n "$lib cfg2: " >> $OUT/ab.txt; python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 2>$OUT/err_$lib.txt | ms >> $OUT/ab.txt
  echo -n "$lib cfg2 1/8 share: " >> $OUT/ab.txt; python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 --simulate-world 8 2>>$OUT/err_$lib.txt | ms >> $OUT/ab.txt
  echo -n "$lib sunset: " >> $OUT/ab.txt; python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --preset "$SUN" 2>>$OUT/err_$lib.txt | ms >> $OUT/ab.txt
done
unset DE_LIB_PATH
cat $OUT/ab.txt
tail -12 $OUT/lanes_cfg2.md
