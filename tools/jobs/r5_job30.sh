#!/bin/bash
# round 5, GPU job 30: how much a record slot is worth at the operating point: the same kernel with 1000 and 900 slots per workgroup (56-byte records) against 1120
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_job30; mkdir -p $OUT
SUN="config - sunset hurricane.txt"
ms() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f ms/step  single %.2f  kernel %.2f' % (j['ms_per_step'], j['single_frame_ms'], j['roofline']['kernel_ms']))"; }
for lib in main v6_p1000 v6_p900 main; do
  if [ $lib = main ]; then unset DE_LIB_PATH; else export DE_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/$lib.so; fi
  echo -n "$lib cfg2: " >> $OUT/ab.txt; timeout 300 python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 2>$OUT/err_$lib.txt | ms >> $OUT/ab.txt
  echo -n "$lib sunset: " >> $OUT/ab.txt; timeout 300 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --preset "$SUN" 2>>$OUT/err_$lib.txt | ms >> $OUT/ab.txt
done
cat $OUT/ab.txt
