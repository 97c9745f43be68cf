#!/bin/bash
# round 6, GPU job 16: the stand-in collective on the final kernel — a rank's 1/8 share of cfg2 / cfg3 / cfg5 and of cfg4, CUs withheld 0 / 1 / 2, stand-in off / reduce / reduce + 7 copies
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6_job16; mkdir -p $OUT
line() { python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
s=j.get('standin_collective',{})
print('%.3f ms/step  single %.2f  standin queue-to-finish %s ms' % (j['ms_per_step'], j['single_frame_ms'], ('%.3f' % s['queue_to_finish_ms_mean_of_last_16']) if s else '-'))"; }
for rep in 1 2 3; do for wh in 0 1 2; do for st in -1 0 7; do
  echo -n "cfg2 1/8 share, cu_withhold $wh, standin $st: "; python3 bench.py --no-cpu-baseline --steps 120 --warmup 10 --simulate-world 8 --standin-collective $st --cu-withhold $wh 2>/dev/null | line
done; done; done | tee $OUT/standin_cfg2.txt
for wh in 0 1; do for st in -1 7; do
  echo -n "cfg2 whole frame, cu_withhold $wh, standin $st: "; python3 bench.py --no-cpu-baseline --no-other-configs --steps 30 --warmup 5 --standin-collective $st --cu-withhold $wh 2>/dev/null | line
  echo -n "cfg3 1/8 share (256 spp florida), cu_withhold $wh, standin $st: "; python3 bench.py --no-cpu-baseline --config cfg3 --steps 12 --warmup 2 --simulate-world 8 --standin-collective $st --cu-withhold $wh 2>/dev/null | line
  echo -n "cfg4 1/8 share, cu_withhold $wh, standin $st: "; python3 bench.py --no-cpu-baseline --config cfg4 --steps 4 --warmup 1 --simulate-world 8 --standin-collective $st --cu-withhold $wh 2>/dev/null | line
  echo -n "cfg5 1/8 share (progressive every 128), cu_withhold $wh, standin $st: "; python3 bench.py --no-cpu-baseline --config cfg5 --steps 6 --warmup 1 --simulate-world 8 --standin-collective $st --cu-withhold $wh 2>/dev/null | line
done; done | tee $OUT/standin_other.txt
