"""The product's N > 1 path on real hardware: bench.py launched as 2 and as 4 ranks (torch.distributed.run, subprocess children)
sharing ONE GPU.  RCCL refuses two ranks on one device, so the ranks talk gloo and parallel.reduce_hdr stages the HDR
buffer through host memory; everything else — DistributedFrame (torch-owned HDR bound into the context, side stream),
tile partition in the kernel, reduce to rank 0, display on rank 0 — is the code the 8-GPU run uses.

This module is named to run FIRST: its children are started before the pytest process itself has touched the GPU.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--steps", "2", "--warmup", "1", "--width", "256", "--height", "128", "--spp", "3", "--tex-size", "1024", "512",
          "--no-cpu-baseline"]


def _run(cmd, env_extra=None, timeout=900):
    env = dict(os.environ)
    env.update(env_extra or {})
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, text=True)
    return p


def _json_line(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert lines, out
    return json.loads(lines[-1])


def test_gpus_flag_without_launcher_fails_loudly():
    # no GPU needed: the check precedes every import
    p = _run([sys.executable, "bench.py", "--gpus", "2"], timeout=120)
    assert p.returncode != 0 and "torch.distributed.run" in (p.stderr + p.stdout)


@pytest.mark.gpu
@pytest.mark.timeout(2400)
@pytest.mark.parametrize("world", [2, 4])
def test_ranks_on_one_gpu_match_one_rank(tmp_path, world):
    solo, multi = str(tmp_path / "solo.npy"), str(tmp_path / "multi.npy")
    p1 = _run([sys.executable, "bench.py", "--gpus", "1", "--dump-hdr", solo] + COMMON)
    assert p1.returncode == 0, p1.stderr[-3000:]
    j1 = _json_line(p1.stdout)
    assert j1["n_gpus"] == 1 and j1["ranks_seen"] == 1
    port = 29600 + (os.getpid() % 1500) + world
    p2 = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(port), "bench.py", "--gpus", str(world), "--dump-hdr", multi] + COMMON,
              env_extra={"DE_DIST_BACKEND": "gloo"})
    assert p2.returncode == 0, (p2.stdout[-2000:], p2.stderr[-4000:])
    j2 = _json_line(p2.stdout)
    assert j2["n_gpus"] == world and j2["ranks_seen"] == world and j2["scaling"] == "strong"
    assert j2["config"]["samples_per_step"] == 256 * 128 * 3
    # the bench's own check on the box it runs on: assembled frame == 1-rank frame
    assert j2["multi_gpu_parity"]["bit_identical_fraction"] == 1.0
    a, b = np.load(solo), np.load(multi)
    assert a.shape == (256, 128, 3) and np.isfinite(a).all() and a.max() > 0
    assert (a.view(np.uint32) == b.view(np.uint32)).all()
