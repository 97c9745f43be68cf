#!/bin/bash
# round 6, GPU job 2: measurements before any kernel change — CU-mask mapping, the pipelined window loop's timeline and the tail threshold for one-sample launches,
# stage entries per stage (statistics build), the stand-in collective behind the persistent launch (a rank's 1/8 share of cfg2; cfg4), with and without withheld CUs
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6_job2; mkdir -p $OUT
export TMPDIR=/tmp
tools/micro/cu_mask > $OUT/cu_mask.txt 2>&1; cat $OUT/cu_mask.txt
timeout 600 python3 -m pytest tests/test_gpu_round6.py -q > $OUT/pytest_r6.txt 2>&1; tail -5 $OUT/pytest_r6.txt
# the pipelined loop: timeline, then the tail threshold
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/pipelined_trace.py run 40 ) > $OUT/trace_run.txt 2>&1; tail -2 $OUT/trace_run.txt
python3 tools/pipelined_trace.py show $OUT/trace > $OUT/trace_show.txt 2>&1; head -60 $OUT/trace_show.txt
for mp in 4194304 1048576 262144; do echo -n "DE_V6_TAIL_MIN_PATHS=$mp: "; DE_V6_TAIL_MIN_PATHS=$mp python3 tools/pipelined_trace.py run 90 2>&1 | tail -1; done | tee $OUT/pipelined_tail.txt
for sl in 2 3 4; do echo -n "DE_SLOTS=$sl: "; DE_SLOTS=$sl python3 tools/pipelined_trace.py run 90 2>&1 | tail -1; done | tee -a $OUT/pipelined_tail.txt
# stage entries
python3 tools/v6_lanes.py --out $OUT/lanes_cfg2.md > /dev/null 2> $OUT/lanes.err; grep -i "entries" $OUT/lanes_cfg2.md
# the stand-in collective
line() { python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
s=j.get('standin_collective',{})
print('%.2f ms/step  single %.2f  kernel %.2f  standin queue-to-finish %s ms' % (j['ms_per_step'], j['single_frame_ms'], j['roofline']['kernel_ms'], ('%.3f' % s['queue_to_finish_ms_mean_of_last_16']) if s else '-'))"; }
for wh in 0 1 2; do
  for st in -1 0 7; do
    echo -n "cfg2 share 1/8, cu_withhold $wh, standin $st: "; python3 bench.py --no-cpu-baseline --steps 30 --warmup 5 --simulate-world 8 --standin-collective $st --cu-withhold $wh 2>/dev/null | line
  done
done | tee $OUT/standin.txt
for wh in 0 1; do for st in -1 0; do
  echo -n "cfg2 whole frame, cu_withhold $wh, standin $st: "; python3 bench.py --no-cpu-baseline --no-other-configs --steps 12 --warmup 3 --standin-collective $st --cu-withhold $wh 2>/dev/null | line
done; done | tee -a $OUT/standin.txt
for st in -1 0; do
  echo -n "cfg4, standin $st: "; python3 bench.py --no-cpu-baseline --config cfg4 --steps 3 --warmup 1 --standin-collective $st 2>/dev/null | line
done | tee -a $OUT/standin.txt
