#!/bin/bash
# round 6, GPU job 26: bench.py --gpus 8 at FULL size (BASELINE cfg2), the eight ranks sharing one GPU over gloo (eight contexts, eight sets of maps: 96 GB): the N = 8 line end to end
cd $GRAFT_REPO_ROOT
DE_DIST_BACKEND=gloo DE_NO_LIB_LEG=1 timeout 1500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29677 bench.py --gpus 8 --steps 4 --warmup 1 > gpurun_out/mg8.json 2> gpurun_out/mg8.err; tail -2 gpurun_out/mg8.err
python3 -c "
import json
j=json.loads([l for l in open('gpurun_out/mg8.json') if l.startswith('{')][-1])
print(j['metric'], j['n_gpus'], j['ranks_seen'], round(j['value'],1), round(j['ms_per_step'],2), j['config']['parallelism'], j['multi_gpu_parity']['bit_identical_fraction'], [(round(p['render_ms'],1), round(p['reduce_ms'],1), p['tiles_owned']) for p in j['per_rank']])"
