#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace stats + PMC passes of the bench workload.
# Usage: tools/profile_gpu.sh <tag> [bench args...]     -> gpurun_out/prof_<tag>/
# Counter passes are separate runs with --pmc only (never combined with other trace domains).
set -u
TAG=${1:-r1}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export DE_BIG_SLOTS=1    # one launch at a time: with several in flight the small kernels of one frame queue behind the persistent waves of the next and their durations show that wait
cd /tmp
BENCH=(python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --frames-in-flight 1 --sync-frames "$@")   # one frame at a time: with two in flight the small kernels of one frame queue behind the persistent waves of the other and their durations show that wait
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- "${BENCH[@]}" > $OUT/stats.log 2>&1
# SQ pass 1: issue / occupancy
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU \
  --kernel-include-regex "render_kernel|ray_march_kernel|v3_|v5_kernel|v6_kernel" --output-format csv -d $OUT/pmc_sq1 -o pmc -- "${BENCH[@]}" > $OUT/pmc_sq1.log 2>&1
# SQ pass 2: stalls
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
  --kernel-include-regex "render_kernel|ray_march_kernel|v3_|v5_kernel|v6_kernel" --output-format csv -d $OUT/pmc_sq2 -o pmc -- "${BENCH[@]}" > $OUT/pmc_sq2.log 2>&1
# TCC passes: HBM traffic (FETCH_SIZE costs 3 of 4 TCC slots, WRITE_SIZE 2: separate passes)
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "render_kernel|ray_march_kernel|v3_|v5_kernel|v6_kernel" --output-format csv -d $OUT/pmc_fetch -o pmc -- "${BENCH[@]}" > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "render_kernel|ray_march_kernel|v3_|v5_kernel|v6_kernel" --output-format csv -d $OUT/pmc_write -o pmc -- "${BENCH[@]}" > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --kernel-include-regex "render_kernel|ray_march_kernel|v3_|v5_kernel|v6_kernel" --output-format csv -d $OUT/pmc_l2 -o pmc -- "${BENCH[@]}" > $OUT/pmc_l2.log 2>&1
# The same frame with the tail chain EXPORTING (one frame at a time no launch has a successor, every workgroup finishes in place and the passes above never see
# the export / import path that the timed region of bench.py — frames back to back — runs): DE_V6_TAIL_ALONE=1 = export regardless.  Time + the two traffic counters + instructions.
export DE_V6_TAIL_ALONE=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tail_stats -o stats -- "${BENCH[@]}" > $OUT/tail_stats.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "v6_kernel" --output-format csv -d $OUT/tail_pmc_write -o pmc -- "${BENCH[@]}" > $OUT/tail_pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "v6_kernel" --output-format csv -d $OUT/tail_pmc_fetch -o pmc -- "${BENCH[@]}" > $OUT/tail_pmc_fetch.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --kernel-include-regex "v6_kernel" --output-format csv -d $OUT/tail_pmc_sq -o pmc -- "${BENCH[@]}" > $OUT/tail_pmc_sq.log 2>&1
unset DE_V6_TAIL_ALONE
# keep only the small files
find $OUT -name "*.csv" -size +24M -delete
ls -R $OUT | head -50
