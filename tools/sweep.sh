#!/bin/bash
# usage: tools/sweep.sh VAR v1 v2 ...   -> kernel ms of the bench workload for each value of env VAR
VAR=$1; shift
for v in "$@"; do
  export $VAR=$v
  python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', 'kernel_ms', round(d['roofline']['kernel_ms'],1), 'Msamples/s', round(d['value'],1))"
done
