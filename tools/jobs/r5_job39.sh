#!/bin/bash
# round 5, GPU job 39: the legacy library's whole cross-check suite (variants 1 / 3 / 5 against the oracle and the product), which the default GPU suite runs in its quick form
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_job39; mkdir -p $OUT
DE_TEST_LEGACY=full timeout 2400 python3 -m pytest tests/test_gpu_legacy_suite.py -x -q > $OUT/pytest_legacy_full.txt 2>&1
tail -5 $OUT/pytest_legacy_full.txt
DE_LIB_PATH=$GRAFT_REPO_ROOT/digital_earth_amd/libdigitalearth_hip_legacy.so timeout 2400 python3 -m pytest tests/legacy -q -m gpu -p no:cacheprovider --durations=5 > $OUT/pytest_legacy_direct.txt 2>&1
tail -12 $OUT/pytest_legacy_direct.txt
