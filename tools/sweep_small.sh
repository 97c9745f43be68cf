#!/bin/bash
VAR=$1; shift
for v in "$@"; do
  export $VAR=$v
  python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --width 960 --height 544 --spp 16 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v small frame (1/16 load)', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],2), 'ms')"
done
