#!/bin/bash
# round 6, GPU job 10: with new paths twice as fast (64-item claims: the primary sphere trace hits in L2) waves come back to the scheduler sooner and enter shallow rings:
# busy lanes per trip fell 0.709 / 0.705 / 0.780 -> 0.681 / 0.692 / 0.765 on a rank's 1/8 share.  Sweep of the scheduler's enter_min (rings shallower than this are left to deepen).
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6_job10; mkdir -p $OUT
ms() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f ms/step  single %.2f  kernel %.2f' % (j['ms_per_step'], j['single_frame_ms'], j['roofline']['kernel_ms']))"; }
for rep in 1 2; do for em in 0 8 16 24 32 48 64; do
  echo -n "enter_min $em: share 1/8 "; DE_V6_ENTER_MIN=$em timeout 300 python3 bench.py --no-cpu-baseline --steps 80 --warmup 5 --simulate-world 8 2>/dev/null | ms
  echo -n "enter_min $em: window loop "; DE_V6_ENTER_MIN=$em python3 tools/pipelined_trace.py run 120 3 2>&1 | tail -1
  echo -n "enter_min $em: cfg2 "; DE_V6_ENTER_MIN=$em timeout 300 python3 bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 3 2>/dev/null | ms
done; done | tee $OUT/sweep.txt
