#!/bin/bash
# round 5, GPU job 29: the tail chain under calls in flight / partitions (new test), the threshold once more on a rank's share, the one-sample frame with the chain forced on
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_job29; mkdir -p $OUT
ms() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f ms/step  single %.2f  kernel %.2f  drain %s' % (j['ms_per_step'], j['single_frame_ms'], j['roofline']['kernel_ms'], j.get('drain_ms')))"; }
timeout 900 python3 -m pytest tests/test_gpu_round5.py tests/test_gpu_block_scheduler.py -x -q > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
two() {
  label=$1; shift
  echo -n "$label cfg2 1/8 share: " >> $OUT/ab.txt; env "$@" timeout 300 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 --simulate-world 8 2>>$OUT/err.txt | ms >> $OUT/ab.txt
  echo -n "$label cfg2: " >> $OUT/ab.txt; env "$@" timeout 300 python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 2>>$OUT/err.txt | ms >> $OUT/ab.txt
}
two "export 128 -> 64 wg" DE_X=0
two "export 160 -> 64 wg" DE_V6_TAIL_EXPORT0=160
two "export 192 -> 96 wg" DE_V6_TAIL_EXPORT0=192 DE_V6_TAIL_GRID0=96
two "export 128 -> 48 wg" DE_V6_TAIL_GRID0=48
two "export 128 -> 96 wg" DE_V6_TAIL_GRID0=96
two "tail 0" DE_V6_TAIL=0
two "export 128 -> 64 wg" DE_X=0
cat $OUT/ab.txt
echo "one-sample frames, chain as shipped (off below 4 Mi paths):" > $OUT/one_spp.txt; python3 tools/one_spp.py 2>&1 | tail -3 >> $OUT/one_spp.txt
echo "chain forced on:" >> $OUT/one_spp.txt; DE_V6_TAIL_MIN_PATHS=0 python3 tools/one_spp.py 2>&1 | tail -3 >> $OUT/one_spp.txt
cat $OUT/one_spp.txt
