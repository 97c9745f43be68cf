#!/bin/bash
# round 6, GPU job 11: kernel timelines of the pipelined window loop (lag 3), committed build against the stateless scheduler loop
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6_job11; mkdir -p $OUT
export TMPDIR=/tmp
for lib in head main; do
  if [ $lib = main ]; then unset DE_LIB_PATH; else export DE_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/$lib.so; fi
  ( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace_$lib -- python3 $GRAFT_REPO_ROOT/tools/pipelined_trace.py run 40 3 ) > $OUT/trace_$lib.log 2>&1; tail -1 $OUT/trace_$lib.log
  python3 tools/pipelined_trace.py show $OUT/trace_$lib > $OUT/show_$lib.txt 2>&1
  echo "== $lib"; head -14 $OUT/show_$lib.txt; grep "v6_kernel" $OUT/show_$lib.txt | head -14
done
