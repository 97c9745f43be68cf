#!/bin/bash
# joint sweep of the v2 scheduler thresholds (env overrides read by de_api.hip)
for thr in 12 16; do for a in 10 12; do for g in 6 9; do for b in 24 40; do
  export DE_V2_THR=$thr DE_V2_A=$a DE_V2_G=$g DE_V2_B=$b
  python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --frames-in-flight 1 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('thr=$thr A=$a G=$g B=$b', round(d['roofline']['kernel_ms'],1))"
done; done; done; done
