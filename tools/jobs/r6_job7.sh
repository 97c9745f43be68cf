#!/bin/bash
# round 6, GPU job 7: the tail's export threshold / grid with the stateless scheduler loop (a workgroup now knows at once that nothing is left to start: it exports
# earlier, and more, than when its twelve waves had to find out one by one)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6_job7; mkdir -p $OUT
ms() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f ms/step  single %.2f  kernel %.2f  drain %.2f' % (j['ms_per_step'], j['single_frame_ms'], j['roofline']['kernel_ms'], j['drain_ms'] or 0))"; }
for rep in 1 2; do
for cfg in "128 64" "96 64" "64 64" "48 64" "32 64" "128 128" "96 128" "64 128" "64 96" "16 64"; do
  set -- $cfg
  echo -n "export $1 grid $2: cfg2 " ; DE_V6_TAIL_EXPORT0=$1 DE_V6_TAIL_GRID0=$2 timeout 300 python3 bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 3 2>/dev/null | ms
done
echo -n "tail off: cfg2 "; DE_V6_TAIL=0 timeout 300 python3 bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 3 2>/dev/null | ms
done | tee $OUT/sweep.txt
for cfg in "128 64" "64 64" "32 64"; do
  set -- $cfg
  echo -n "export $1 grid $2: share 1/8 "; DE_V6_TAIL_EXPORT0=$1 DE_V6_TAIL_GRID0=$2 timeout 300 python3 bench.py --no-cpu-baseline --steps 60 --warmup 5 --simulate-world 8 2>/dev/null | ms
  echo -n "export $1 grid $2: Apollo "; DE_V6_TAIL_EXPORT0=$1 DE_V6_TAIL_GRID0=$2 timeout 300 python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 --preset "config - Apollo 11.txt" 2>/dev/null | ms
  echo -n "export $1 grid $2: sunset "; DE_V6_TAIL_EXPORT0=$1 DE_V6_TAIL_GRID0=$2 timeout 300 python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --preset "config - sunset hurricane.txt" 2>/dev/null | ms
done | tee -a $OUT/sweep.txt
