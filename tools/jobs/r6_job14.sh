#!/bin/bash
# round 6, GPU job 14: why do overlapping SMALL launches run 1-5 % slower with the stateless scheduler loop?  Variants: 2 / 4 batches of new paths per visit, lane index from threadIdx
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6_job14; mkdir -p $OUT
ms() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f ms/step  single %.2f  kernel %.2f' % (j['ms_per_step'], j['single_frame_ms'], j['roofline']['kernel_ms']))"; }
for rep in 1 2 3; do for lib in head main pb2 pb4 tid; do
  if [ $lib = main ]; then unset DE_LIB_PATH; else export DE_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/$lib.so; fi
  if [ $rep = 1 ]; then echo -n "$lib bits (of 5 cameras): "; python3 tools/v6_try.py --small 2>&1 | grep -c "fraction 1.000000"; fi
  echo -n "$lib window loop: "; python3 tools/pipelined_trace.py run 120 3 2>&1 | tail -1
  echo -n "$lib share 1/8: "; timeout 300 python3 bench.py --no-cpu-baseline --steps 100 --warmup 5 --simulate-world 8 2>/dev/null | ms
  echo -n "$lib cfg2: "; timeout 300 python3 bench.py --no-cpu-baseline --no-other-configs --steps 30 --warmup 5 2>/dev/null | ms
done; done | tee $OUT/ab.txt
