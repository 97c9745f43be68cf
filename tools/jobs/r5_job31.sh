#!/bin/bash
# round 5, GPU job 31: 44-byte LDS records (the three pass-through scalars of a record in global memory, 1344 slots per workgroup) against the 56-byte / 1120-slot build
# of the commit before (build/ab/prev_r56.so): bit-exactness first (block scheduler, round 5, round 4 tests), then same-box timings on four views and a rank's share
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_job31; mkdir -p $OUT
SUN="config - sunset hurricane.txt"
ms() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f ms/step  single %.2f  kernel %.2f' % (j['ms_per_step'], j['single_frame_ms'], j['roofline']['kernel_ms']))"; }
timeout 1200 python3 -m pytest tests/test_gpu_block_scheduler.py tests/test_gpu_round5.py tests/test_gpu_round4.py tests/test_gpu_parity.py -x -q > $OUT/pytest.txt 2>&1
grep -E "passed|failed" $OUT/pytest.txt
for lib in main prev_r56 main prev_r56; do
  if [ $lib = main ]; then unset DE_LIB_PATH; else export DE_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/$lib.so; fi
  echo -n "$lib cfg2: " >> $OUT/ab.txt; timeout 300 python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 2>$OUT/err_$lib.txt | ms >> $OUT/ab.txt
  echo -n "$lib sunset: " >> $OUT/ab.txt; timeout 300 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --preset "$SUN" 2>>$OUT/err_$lib.txt | ms >> $OUT/ab.txt
  echo -n "$lib florida: " >> $OUT/ab.txt; timeout 300 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --preset "config - florida.txt" 2>>$OUT/err_$lib.txt | ms >> $OUT/ab.txt
  echo -n "$lib Apollo: " >> $OUT/ab.txt; timeout 300 python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 --preset "config - Apollo 11.txt" 2>>$OUT/err_$lib.txt | ms >> $OUT/ab.txt
  echo -n "$lib cfg2 1/8 share: " >> $OUT/ab.txt; timeout 300 python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 --simulate-world 8 2>>$OUT/err_$lib.txt | ms >> $OUT/ab.txt
done
unset DE_LIB_PATH
cat $OUT/ab.txt
