#!/bin/bash
# round 6, GPU job 19: VERDICT r5 #1b — the trip control of the loop stages: the busy flag of the inner loop as a 0 / 1 VGPR (9 scalar instructions fewer, 3 vector more per trip)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6_job19; mkdir -p $OUT
ms() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f ms/step  single %.2f  kernel %.2f' % (j['ms_per_step'], j['single_frame_ms'], j['roofline']['kernel_ms']))"; }
for rep in 1 2 3; do for lib in main busyint; do
  if [ $lib = main ]; then unset DE_LIB_PATH; else export DE_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/$lib.so; fi
  if [ $rep = 1 ] && [ $lib != main ]; then echo -n "$lib bits (of 5 cameras): "; python3 tools/v6_try.py --small 2>&1 | grep -c "fraction 1.000000"; fi
  echo -n "$lib cfg2: "; timeout 300 python3 bench.py --no-cpu-baseline --no-other-configs --steps 30 --warmup 5 2>/dev/null | ms
  echo -n "$lib sunset: "; timeout 300 python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 --preset "config - sunset hurricane.txt" 2>/dev/null | ms
done; done | tee $OUT/ab.txt
unset DE_LIB_PATH
bash tools/pmc_ab6.sh main build/ab/busyint.so 2>&1 | tee $OUT/pmc_ab.txt
