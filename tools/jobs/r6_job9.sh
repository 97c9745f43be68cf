#!/bin/bash
# round 6, GPU job 9: where the one-sample launch's time goes, committed build against the stateless scheduler loop (the kernel's statistics build)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6_job9; mkdir -p $OUT
for lib in head main; do
  if [ $lib = main ]; then unset DE_LIB_PATH; else export DE_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/$lib.so; fi
  python3 tools/v6_lanes.py --spp 1 --out $OUT/lanes_1spp_$lib.md > /dev/null 2> $OUT/err_$lib.txt
  python3 tools/v6_lanes.py --world 8 --out $OUT/lanes_w8_$lib.md > /dev/null 2>> $OUT/err_$lib.txt
  echo "== $lib, 1 spp"; head -16 $OUT/lanes_1spp_$lib.md | tail -12; grep -A12 "^## Drain" $OUT/lanes_1spp_$lib.md | head -3
  echo "== $lib, 1/8 share"; head -16 $OUT/lanes_w8_$lib.md | tail -12; grep "^## Drain" $OUT/lanes_w8_$lib.md
done
