#!/bin/bash
# round 6, GPU job 1: the new surface on hardware — round-6 tests, the 8-rank rehearsal, the default bench line with the other_configs legs, the window loop pipelined
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6_job1; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_round6.py -q -x > $OUT/pytest_r6.txt 2>&1; tail -15 $OUT/pytest_r6.txt
timeout 900 python3 bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; tail -c 3000 $OUT/bench.json; tail -5 $OUT/bench.err
timeout 600 python3 tools/one_spp.py > $OUT/one_spp.txt 2>&1; tail -12 $OUT/one_spp.txt
timeout 1500 python3 -m pytest tests/test_a_multi_rank_gpu.py -q -x -m gpu --durations=8 > $OUT/pytest_multi.txt 2>&1; tail -15 $OUT/pytest_multi.txt
timeout 1500 python3 -m pytest tests -q -m gpu --durations=12 --deselect tests/test_a_multi_rank_gpu.py --deselect tests/test_gpu_round6.py > $OUT/pytest_all.txt 2>&1; tail -6 $OUT/pytest_all.txt
