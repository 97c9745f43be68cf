#!/bin/bash
# time the default frame with each library variant under build/flags/
for f in build/flags/*.so; do
  cp $f digital_earth_amd/libdigitalearth_hip.so
  python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --frames-in-flight 1 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', round(d['roofline']['kernel_ms'],1))"
done
