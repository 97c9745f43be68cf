#!/bin/bash
# round 5, GPU job 26: the tail chain with the "no wave can still start paths" rule: the whole GPU suite (the settings test now forces the chain on with 0 / 1 / 2
# tail launches, the small-pool variants too), then same-box timings against DE_V6_TAIL=0
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_job26; mkdir -p $OUT
SUN="config - sunset hurricane.txt"
ms() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f ms/step  single %.2f  kernel %.2f  drain %s' % (j['ms_per_step'], j['single_frame_ms'], j['roofline']['kernel_ms'], j.get('drain_ms')))"; }
timeout 1500 python3 -m pytest tests -x -q -m gpu --durations=8 > $OUT/pytest_all.txt 2>&1
tail -14 $OUT/pytest_all.txt
four() {
  label=$1; shift
  echo -n "$label cfg2: " >> $OUT/ab.txt; env "$@" timeout 300 python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 2>$OUT/err.txt | ms >> $OUT/ab.txt
  echo -n "$label cfg2 1/8 share: " >> $OUT/ab.txt; env "$@" timeout 300 python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 --simulate-world 8 2>>$OUT/err.txt | ms >> $OUT/ab.txt
  echo -n "$label sunset: " >> $OUT/ab.txt; env "$@" timeout 300 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --preset "$SUN" 2>>$OUT/err.txt | ms >> $OUT/ab.txt
  echo -n "$label Apollo 1/8 share: " >> $OUT/ab.txt; env "$@" timeout 300 python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 --preset "config - Apollo 11.txt" --simulate-world 8 2>>$OUT/err.txt | ms >> $OUT/ab.txt
}
four "tail 0" DE_V6_TAIL=0
four "tail 1 (96 -> 64 wg)" DE_V6_TAIL=1
four "tail 1 (128 -> 64 wg)" DE_V6_TAIL=1 DE_V6_TAIL_EXPORT0=128
four "tail 1 (64 -> 32 wg)" DE_V6_TAIL=1 DE_V6_TAIL_EXPORT0=64 DE_V6_TAIL_GRID0=32
four "tail 2 (200 -> 96 wg, 64 -> 8 wg)" DE_V6_TAIL=2 DE_V6_TAIL_EXPORT0=200 DE_V6_TAIL_GRID0=96 DE_V6_TAIL_EXPORT1=64 DE_V6_TAIL_GRID1=8
four "tail 0" DE_V6_TAIL=0
four "tail 1 (96 -> 64 wg)" DE_V6_TAIL=1
cat $OUT/ab.txt
python3 tools/one_spp.py 2>&1 | tail -4
