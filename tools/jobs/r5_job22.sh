#!/bin/bash
# round 5, GPU job 22: the tail chain (a launch exports its workgroups' last paths to a pool, a small launch of the same kernel finishes them):
# bit-exactness first, then same-box timings against DE_V6_TAIL=0 (one launch per call, as before), the drain table, a first sweep of the thresholds
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_job22; mkdir -p $OUT
SUN="config - sunset hurricane.txt"
ms() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f ms/step  single %.2f  kernel %.2f  drain %s' % (j['ms_per_step'], j['single_frame_ms'], j['roofline']['kernel_ms'], j.get('drain_ms')))"; }
timeout 900 python3 -m pytest tests/test_gpu_block_scheduler.py tests/test_gpu_round5.py tests/test_gpu_round4.py -x -q > $OUT/pytest_quick.txt 2>&1
tail -3 $OUT/pytest_quick.txt
three() {
  label=$1; shift
  echo -n "$label cfg2: " >> $OUT/ab.txt; env "$@" timeout 300 python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 2>$OUT/err.txt | ms >> $OUT/ab.txt
  echo -n "$label cfg2 1/8 share: " >> $OUT/ab.txt; env "$@" timeout 300 python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 --simulate-world 8 2>>$OUT/err.txt | ms >> $OUT/ab.txt
  echo -n "$label sunset: " >> $OUT/ab.txt; env "$@" timeout 300 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --preset "$SUN" 2>>$OUT/err.txt | ms >> $OUT/ab.txt
}
three "tail 0" DE_V6_TAIL=0
three "tail 1 (96 -> 64 wg)" DE_V6_TAIL=1
three "tail 2 (96 -> 64 wg, 96 -> 8 wg)" DE_V6_TAIL=2
three "tail 0" DE_V6_TAIL=0
three "tail 1, export 64" DE_V6_TAIL=1 DE_V6_TAIL_EXPORT0=64
three "tail 1, export 160 -> 96 wg" DE_V6_TAIL=1 DE_V6_TAIL_EXPORT0=160 DE_V6_TAIL_GRID0=96
three "tail 1, export 96 -> 48 wg" DE_V6_TAIL=1 DE_V6_TAIL_GRID0=48
three "tail 2 (400 -> 192 wg, 96 -> 32 wg)" DE_V6_TAIL=2 DE_V6_TAIL_EXPORT0=400 DE_V6_TAIL_GRID0=192 DE_V6_TAIL_EXPORT1=96 DE_V6_TAIL_GRID1=32
three "tail 1 (96 -> 64 wg)" DE_V6_TAIL=1
cat $OUT/ab.txt
python3 tools/v6_lanes.py --out $OUT/lanes_cfg2.md > /dev/null 2> $OUT/lanes.err
sed -n '/## Drain/,$p' $OUT/lanes_cfg2.md | head -50
