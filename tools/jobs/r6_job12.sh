#!/bin/bash
# round 6, GPU job 12: one LDS round trip per scheduler round + idle back-off (main), the same without the back-off, the committed build (head)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6_job12; mkdir -p $OUT
ms() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f ms/step  single %.2f  kernel %.2f' % (j['ms_per_step'], j['single_frame_ms'], j['roofline']['kernel_ms']))"; }
for rep in 1 2; do for lib in head nobackoff main; do
  if [ $lib = main ]; then unset DE_LIB_PATH; else export DE_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/$lib.so; fi
  if [ $rep = 1 ]; then echo -n "$lib bits (of 5 cameras): "; python3 tools/v6_try.py --small 2>&1 | grep -c "fraction 1.000000"; fi
  echo -n "$lib window loop: "; python3 tools/pipelined_trace.py run 120 3 2>&1 | tail -1
  echo -n "$lib window loop: "; python3 tools/pipelined_trace.py run 120 1 2>&1 | tail -1
  echo -n "$lib share 1/8: "; timeout 300 python3 bench.py --no-cpu-baseline --steps 80 --warmup 5 --simulate-world 8 2>/dev/null | ms
  echo -n "$lib cfg2: "; timeout 300 python3 bench.py --no-cpu-baseline --no-other-configs --steps 30 --warmup 5 2>/dev/null | ms
  echo -n "$lib Apollo: "; timeout 300 python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 --preset "config - Apollo 11.txt" 2>/dev/null | ms
done; done | tee $OUT/ab.txt
unset DE_LIB_PATH
python3 tools/one_spp.py 2>&1 | tail -7 | tee $OUT/one_spp.txt
