#!/bin/bash
# Build a variant of the library for A/B timing WITHOUT touching the product .so:
#   tools/ab_build.sh <name> [extra hipcc flags, e.g. -DDE_V2_MIN_WAVES=4]   -> build/ab/<name>.so
# (the product's own flags — digital_earth_amd/build.py FLAGS — are the base; DE_AB_BASE_FLAGS="" builds without the round-6 additions)
# Use it with DE_LIB_PATH=build/ab/<name>.so python tools/sweep_knobs.py ...
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/build/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -shared -Wno-unused-value ${DE_AB_BASE_FLAGS--mllvm -enable-ipra -fno-optimize-sibling-calls} "$@" \
  $ROOT/digital_earth_amd/csrc/de_api.hip $ROOT/digital_earth_amd/csrc/de_fast.hip -o $ROOT/build/ab/$NAME.so
echo built $ROOT/build/ab/$NAME.so
