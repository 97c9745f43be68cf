"""The kernel families the product no longer runs — per-lane loops (variant 1), the wavefront pipeline through HBM queues with the launch policy
of rounds 3-4 (variants 3 / 4 with DE_AUTO_V6=0), the HBM-queue stage scheduler (variant 5) — are kept as independent statements of path_tracer
in libdigitalearth_hip_legacy.so (the product's sources + -DDE_LEGACY_VARIANTS).  A process loads one library, so their cross-checks
(tests/legacy/) run here in a child process with DE_LIB_PATH pointing at it: the `legacy_quick` subset by default — each family against the
oracle on four cameras — and all of them with DE_TEST_LEGACY=full."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(3000)
def test_legacy_kernel_families_give_the_same_bits():
    lib = os.path.join(ROOT, "digital_earth_amd", "libdigitalearth_hip_legacy.so")
    assert os.path.exists(lib), "legacy library missing: run __graft_entry__.build()"
    full = os.environ.get("DE_TEST_LEGACY", "") == "full"
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "legacy"), "-x", "-q", "-m", "gpu" if full else "gpu and legacy_quick", "-p", "no:cacheprovider"]
    out = subprocess.run(cmd, env=dict(os.environ, DE_LIB_PATH=lib), cwd=ROOT, capture_output=True, text=True, timeout=2900)
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0, tail
    assert " passed" in tail and " skipped" not in tail.split("\n")[-2], tail
