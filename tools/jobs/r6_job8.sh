#!/bin/bash
# round 6, GPU job 8: committed build (head.so) against the stateless scheduler loop, four alternations per workload on one box
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6_job8; mkdir -p $OUT
ms() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f ms/step  single %.2f  kernel %.2f' % (j['ms_per_step'], j['single_frame_ms'], j['roofline']['kernel_ms']))"; }
for rep in 1 2 3 4; do for lib in head main; do
  if [ $lib = main ]; then unset DE_LIB_PATH; else export DE_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/$lib.so; fi
  echo -n "$lib cfg2: "; timeout 300 python3 bench.py --no-cpu-baseline --no-other-configs --steps 30 --warmup 5 2>/dev/null | ms
done; done | tee $OUT/ab_cfg2.txt
for rep in 1 2; do for lib in head main; do
  if [ $lib = main ]; then unset DE_LIB_PATH; else export DE_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/$lib.so; fi
  echo -n "$lib florida: "; timeout 300 python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 --preset "config - florida.txt" 2>/dev/null | ms
  echo -n "$lib share 1/8: "; timeout 300 python3 bench.py --no-cpu-baseline --steps 80 --warmup 5 --simulate-world 8 2>/dev/null | ms
  echo -n "$lib window loop: "; python3 tools/pipelined_trace.py run 120 3 2>&1 | tail -1
  echo -n "$lib window loop: "; python3 tools/pipelined_trace.py run 120 1 2>&1 | tail -1
done; done | tee $OUT/ab_other.txt
unset DE_LIB_PATH
