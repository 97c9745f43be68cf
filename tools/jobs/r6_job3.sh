#!/bin/bash
# round 6, GPU job 3: the pipelined loop at lag 1 / 2; the stand-in collective at low noise (100 steps); experiment: the primary ray's first gas steps inline in run_primary
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6_job3; mkdir -p $OUT
timeout 600 python3 -m pytest tests/test_gpu_round6.py -q > $OUT/pytest_r6.txt 2>&1; tail -5 $OUT/pytest_r6.txt
timeout 600 python3 tools/one_spp.py > $OUT/one_spp.txt 2>&1; tail -6 $OUT/one_spp.txt
ms() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f ms/step  single %.2f  kernel %.2f' % (j['ms_per_step'], j['single_frame_ms'], j['roofline']['kernel_ms']))"; }
for lib in main pgas4 pgas8 main pgas4 pgas8; do
  if [ $lib = main ]; then unset DE_LIB_PATH; else export DE_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/$lib.so; fi
  if [ $lib != main ]; then echo -n "$lib bits: " >> $OUT/ab.txt; python3 tools/v6_try.py --small 2>&1 | grep -c "fraction 1.000000" >> $OUT/ab.txt; fi
  echo -n "$lib cfg2: " >> $OUT/ab.txt; timeout 300 python3 bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 3 2>$OUT/err_$lib.txt | ms >> $OUT/ab.txt
  echo -n "$lib sunset: " >> $OUT/ab.txt; timeout 300 python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --preset "config - sunset hurricane.txt" 2>>$OUT/err_$lib.txt | ms >> $OUT/ab.txt
done
unset DE_LIB_PATH
cat $OUT/ab.txt
line() { python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
s=j.get('standin_collective',{})
print('%.2f ms/step  single %.2f  kernel %.2f  standin queue-to-finish %s ms' % (j['ms_per_step'], j['single_frame_ms'], j['roofline']['kernel_ms'], ('%.3f' % s['queue_to_finish_ms_mean_of_last_16']) if s else '-'))"; }
for rep in 1 2; do for wh in 0 1 2 4; do
  for st in -1 0 7; do
    echo -n "cfg2 share 1/8, cu_withhold $wh, standin $st: "; python3 bench.py --no-cpu-baseline --steps 100 --warmup 10 --simulate-world 8 --standin-collective $st --cu-withhold $wh 2>/dev/null | line
  done
done; done | tee $OUT/standin.txt
