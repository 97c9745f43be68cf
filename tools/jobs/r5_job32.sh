#!/bin/bash
# round 5, GPU job 32: a launch exports only when another launch of the context is queued behind it on another stream: the whole GPU suite, then the step time (frames
# overlapping: as before) and the single-frame time (back to the one-launch figure) against the always-export setting and against no tail at all
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_job32; mkdir -p $OUT
SUN="config - sunset hurricane.txt"
ms() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f ms/step  single %.2f  kernel %.2f  drain %s' % (j['ms_per_step'], j['single_frame_ms'], j['roofline']['kernel_ms'], j.get('drain_ms')))"; }
timeout 1500 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_all.txt 2>&1
grep -E "passed|failed" $OUT/pytest_all.txt
three() {
  label=$1; shift
  echo -n "$label cfg2: " >> $OUT/ab.txt; env "$@" timeout 300 python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 2>$OUT/err.txt | ms >> $OUT/ab.txt
  echo -n "$label cfg2 1/8 share: " >> $OUT/ab.txt; env "$@" timeout 300 python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 --simulate-world 8 2>>$OUT/err.txt | ms >> $OUT/ab.txt
  echo -n "$label cfg2 serial frames: " >> $OUT/ab.txt; env "$@" timeout 300 python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --serial-frames 2>>$OUT/err.txt | ms >> $OUT/ab.txt
}
three "export when followed (default)" DE_X=0
three "export always" DE_V6_TAIL_ALONE=1
three "tail 0" DE_V6_TAIL=0
three "export when followed (default)" DE_X=0
cat $OUT/ab.txt
python3 tools/one_spp.py 2>&1 | tail -4
