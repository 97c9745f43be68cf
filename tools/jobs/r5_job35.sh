#!/bin/bash
# round 5, GPU job 35: the multi-rank tests on one GPU (the progressive one now runs the gather of the owned pixels), then the two collectives of the tile partition with
# two and four ranks sharing the GPU over gloo (host staging: the bytes moved are what differs)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_job35; mkdir -p $OUT
timeout 1200 python3 -m pytest tests/test_a_multi_rank_gpu.py -x -q > $OUT/pytest.txt 2>&1
grep -E "passed|failed" $OUT/pytest.txt
for coll in reduce gather; do
  DE_DIST_BACKEND=gloo timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 4 --steps 6 --warmup 2 --no-cpu-baseline --collective $coll > $OUT/bench4_$coll.json 2> $OUT/bench4_$coll.err
  python3 -c "import json,sys; j=json.loads([l for l in open('$OUT/bench4_$coll.json') if l.startswith('{')][-1]); print('$coll', round(j['ms_per_step'],2), j['multi_gpu_parity'], [ (p['render_ms'], p['reduce_ms']) for p in j['per_rank']])"
done
