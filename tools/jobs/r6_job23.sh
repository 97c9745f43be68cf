#!/bin/bash
# round 6, GPU job 23: bench.py --gpus 2 at FULL size (BASELINE cfg2) with the two ranks sharing one GPU over gloo: the N > 1 line end to end (118.8 ms per step, assembled frame bit-identical)
cd $GRAFT_REPO_ROOT
DE_DIST_BACKEND=gloo DE_NO_LIB_LEG=1 timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 4 --warmup 1 > gpurun_out/mg2.json 2> gpurun_out/mg2.err; tail -3 gpurun_out/mg2.err
python3 -c "
import json
j=json.loads([l for l in open('gpurun_out/mg2.json') if l.startswith('{')][-1])
print(j['metric'], j['n_gpus'], j['value'], j['ms_per_step'], j['config']['parallelism'], j['multi_gpu_parity'], [ (p['render_ms'], p['reduce_ms'], p['tiles_owned']) for p in j['per_rank']], j['alt'])"
