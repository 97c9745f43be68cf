#!/bin/bash
# round 6, GPU job 6: the scheduler loop without state (work items claimed 64 at a time after the free slots, band progress in LDS, new paths out of line, lane index from
# mbcnt) against the committed build (IPRA + scalar dispatch): bits, time, counters, same box; then the kernel's own test files
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6_job6; mkdir -p $OUT
export TMPDIR=/tmp
ms() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f ms/step  single %.2f  kernel %.2f' % (j['ms_per_step'], j['single_frame_ms'], j['roofline']['kernel_ms']))"; }
for lib in head main head main; do
  if [ $lib = main ]; then unset DE_LIB_PATH; else export DE_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/$lib.so; fi
  if [ $lib = main ]; then echo -n "$lib bits (of 5 cameras): " >> $OUT/ab.txt; python3 tools/v6_try.py --small 2>&1 | grep -c "fraction 1.000000" >> $OUT/ab.txt; fi
  echo -n "$lib cfg2: " >> $OUT/ab.txt; timeout 300 python3 bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 3 2>$OUT/err_$lib.txt | ms >> $OUT/ab.txt
  echo -n "$lib sunset: " >> $OUT/ab.txt; timeout 300 python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --preset "config - sunset hurricane.txt" 2>>$OUT/err_$lib.txt | ms >> $OUT/ab.txt
  echo -n "$lib Apollo: " >> $OUT/ab.txt; timeout 300 python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 --preset "config - Apollo 11.txt" 2>>$OUT/err_$lib.txt | ms >> $OUT/ab.txt
  echo -n "$lib cfg2 share 1/8: " >> $OUT/ab.txt; timeout 300 python3 bench.py --no-cpu-baseline --steps 60 --warmup 5 --simulate-world 8 2>>$OUT/err_$lib.txt | ms >> $OUT/ab.txt
  echo -n "$lib window loop: " >> $OUT/ab.txt; python3 tools/pipelined_trace.py run 90 3 2>&1 | tail -1 >> $OUT/ab.txt
done
unset DE_LIB_PATH
cat $OUT/ab.txt
bash tools/pmc_ab6.sh build/ab/head.so main 2>&1 | tee $OUT/pmc_ab.txt
timeout 2400 python3 -m pytest tests/test_gpu_block_scheduler.py tests/test_gpu_round5.py tests/test_gpu_round6.py tests/test_gpu_round4.py tests/test_gpu_round3.py tests/test_gpu_legacy_suite.py -q -x --durations=5 > $OUT/pytest.txt 2>&1; tail -12 $OUT/pytest.txt
