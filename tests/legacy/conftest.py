"""tests/legacy — cross-checks against the kernel families the product no longer runs (per-lane loops, the wavefront pipeline and the launch
policy of rounds 3-4, the HBM-queue scheduler): three further statements of path_tracer that must give the product's bits.  They live in
libdigitalearth_hip_legacy.so (the product's sources + -DDE_LEGACY_VARIANTS) and a process loads ONE library, so these tests run in a child
process with DE_LIB_PATH pointing at it — tests/test_gpu_legacy_suite.py starts it (a quick subset by default, everything with
DE_TEST_LEGACY=full) — and skip themselves anywhere else."""
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))          # helpers.py and the round tests' shared code


def pytest_configure(config):
    config.addinivalue_line("markers", "legacy_quick: the subset of the legacy cross-checks the default GPU suite runs (in a child process)")


def pytest_collection_modifyitems(config, items):
    from digital_earth_amd import _native
    if os.path.basename(_native.LIB_PATH) == "libdigitalearth_hip_legacy.so":
        return
    skip = pytest.mark.skip(reason="needs the legacy library: DE_LIB_PATH=digital_earth_amd/libdigitalearth_hip_legacy.so (tests/test_gpu_legacy_suite.py runs these in a child process)")
    for item in items:
        if HERE in str(item.fspath):
            item.add_marker(skip)
