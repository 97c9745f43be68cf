#!/bin/bash
# round 6, GPU job 4: the calling convention attacked — internal-linkage stage functions + -mllvm -enable-ipra -fno-optimize-sibling-calls (LLVM's no-CSR
# optimisation: the 24 callee-saved VGPRs of a stage function are no longer saved / restored through scratch at each of the 13.9 M stage entries per frame) and a scalar
# dispatch of the scheduler's answer: bits, time, WRITE_SIZE / FETCH_SIZE / L2 / instruction counters, same box.  Then the pipelined loop at lag 1 / 2 / 3, and the stand-in's timeline.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6_job4; mkdir -p $OUT
export TMPDIR=/tmp
ms() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f ms/step  single %.2f  kernel %.2f' % (j['ms_per_step'], j['single_frame_ms'], j['roofline']['kernel_ms']))"; }
for lib in main ipra ipra_uni uni main ipra ipra_uni uni; do
  if [ $lib = main ]; then unset DE_LIB_PATH; else export DE_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/$lib.so; fi
  if [ $lib != main ]; then echo -n "$lib bits (of 5 cameras): " >> $OUT/ab.txt; python3 tools/v6_try.py --small 2>&1 | grep -c "fraction 1.000000" >> $OUT/ab.txt; fi
  echo -n "$lib cfg2: " >> $OUT/ab.txt; timeout 300 python3 bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 3 2>$OUT/err_$lib.txt | ms >> $OUT/ab.txt
  echo -n "$lib sunset: " >> $OUT/ab.txt; timeout 300 python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --preset "config - sunset hurricane.txt" 2>>$OUT/err_$lib.txt | ms >> $OUT/ab.txt
  echo -n "$lib cfg2 share 1/8: " >> $OUT/ab.txt; timeout 300 python3 bench.py --no-cpu-baseline --steps 60 --warmup 5 --simulate-world 8 2>>$OUT/err_$lib.txt | ms >> $OUT/ab.txt
done
unset DE_LIB_PATH
cat $OUT/ab.txt
bash tools/pmc_ab6.sh main build/ab/ipra.so build/ab/ipra_uni.so 2>&1 | tee $OUT/pmc_ab.txt
timeout 600 python3 -m pytest tests/test_gpu_round6.py -q > $OUT/pytest_r6.txt 2>&1; tail -3 $OUT/pytest_r6.txt
timeout 600 python3 tools/one_spp.py 2>&1 | tail -4 | tee $OUT/one_spp.txt
# where the stand-in collective's queue-to-finish time goes: kernel trace of a rank's 1/8 share with the stand-in, plain and with one CU per XCD withheld
for wh in 0 1; do
  ( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace_wh$wh -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 12 --warmup 2 --simulate-world 8 --standin-collective 0 --cu-withhold $wh ) > $OUT/trace_wh$wh.log 2>&1
  python3 tools/pipelined_trace.py tail $OUT/trace_wh$wh 70 > $OUT/trace_wh$wh.txt 2>&1
  find $OUT/trace_wh$wh -name "*.csv" -size +8M -delete
done
head -75 $OUT/trace_wh0.txt
