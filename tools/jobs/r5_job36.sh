#!/bin/bash
# round 5, GPU job 36: job 35 again with the gather's pixel order fixed for the device buffer ([H][W][3]); then the gas densities from the altitude table once more,
# now that the kernel sits at its issue ceiling (-DDE_GAS_TABLE: same bits; round 5 earlier: -7.8 % vector instructions, no time)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_job36; mkdir -p $OUT
SUN="config - sunset hurricane.txt"
ms() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f ms/step  single %.2f  kernel %.2f' % (j['ms_per_step'], j['single_frame_ms'], j['roofline']['kernel_ms']))"; }
timeout 1200 python3 -m pytest tests/test_a_multi_rank_gpu.py -x -q > $OUT/pytest.txt 2>&1
grep -E "passed|failed" $OUT/pytest.txt
for coll in reduce gather; do
  DE_DIST_BACKEND=gloo timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 4 --steps 6 --warmup 2 --no-cpu-baseline --collective $coll > $OUT/bench4_$coll.json 2> $OUT/bench4_$coll.err
  python3 -c "import json,sys; j=json.loads([l for l in open('$OUT/bench4_$coll.json') if l.startswith('{')][-1]); print('$coll', round(j['ms_per_step'],2), j['multi_gpu_parity']['bit_identical_fraction'], [ (round(p['render_ms'],1), round(p['reduce_ms'],1)) for p in j['per_rank']])"
done
for lib in main gas_table main gas_table; do
  if [ $lib = main ]; then unset DE_LIB_PATH; else export DE_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/$lib.so; fi
  echo -n "$lib cfg2: " >> $OUT/ab.txt; timeout 300 python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 2>$OUT/err_$lib.txt | ms >> $OUT/ab.txt
  echo -n "$lib sunset: " >> $OUT/ab.txt; timeout 300 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --preset "$SUN" 2>>$OUT/err_$lib.txt | ms >> $OUT/ab.txt
  echo -n "$lib Apollo: " >> $OUT/ab.txt; timeout 300 python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 --preset "config - Apollo 11.txt" 2>>$OUT/err_$lib.txt | ms >> $OUT/ab.txt
done
unset DE_LIB_PATH
cat $OUT/ab.txt
