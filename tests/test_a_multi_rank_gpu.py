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
    # what makes an N-GPU line interpretable: per-rank render / collective times and tile counts, the launch
    # settings, and the same steps under the other setting (serial <-> overlapping frames)
    assert [pr["rank"] for pr in j2["per_rank"]] == list(range(world))
    assert sum(pr["tiles_owned"] for pr in j2["per_rank"]) == (256 // 8) * (128 // 8)
    assert all(pr["render_ms"] > 0 and pr["reduce_ms"] >= 0 for pr in j2["per_rank"])
    assert j2["config"]["waves_per_cu"] == 24 and j2["config"]["launch_slots"] == 3      # render_kernel_v6: two workgroups of twelve waves per CU
    assert all(pr["kernel"] == "per-CU stage scheduler" for pr in j2["per_rank"])
    assert j2["alt"]["serial_frames"] is True and j2["alt"]["value"] > 0
    assert j1["config"]["waves_per_cu"] == 24 and "alt" not in j1


def test_tile_ownership_is_balanced_at_eight_ranks():
    # host logic only: the diagonal interleave gives every rank the same number of 8x8 tiles within 1 % at the BASELINE sizes
    from digital_earth_amd import parallel
    for (w, h) in ((1920, 1080), (3840, 2160)):
        t = parallel.tiles_per_rank(w, h, 8)
        assert t.sum() == (w // 8) * (h // 8) and t.max() / t.min() <= 1.01, (w, h, t)


@pytest.mark.gpu
@pytest.mark.timeout(2400)
@pytest.mark.parametrize("mode", ["tiles", "samples", "gather"])
def test_eight_ranks_rehearsal_of_the_driver_run(tmp_path, mode):
    """Exactly what the driver launches for the 8-GPU line (`python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8`), rehearsed with the
    eight ranks sharing this box's one GPU over gloo, for both partitions and the gather collective: eight ranks seen by the collective, the assembled
    frame equal to the 1-rank frame (tiles: bit for bit; samples: rank-ordered partial sums, relative L2 <= 1e-6), the tile counts balanced, and a line that
    parses with `roofline`, `per_rank` and the config's name in it."""
    solo, multi = str(tmp_path / "solo.npy"), str(tmp_path / "multi.npy")
    common = [a if a != "3" else "8" for a in COMMON]                     # 8 spp: one sample index per rank under the sample partition
    p1 = _run([sys.executable, "bench.py", "--gpus", "1", "--dump-hdr", solo] + common)
    assert p1.returncode == 0, p1.stderr[-3000:]
    extra = {"tiles": ["--split", "tiles"], "samples": ["--split", "samples"], "gather": ["--split", "tiles", "--collective", "gather"]}[mode]
    port = 29300 + (os.getpid() % 1500) + {"tiles": 0, "samples": 1, "gather": 2}[mode]
    p2 = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
               "--master-port", str(port), "bench.py", "--gpus", "8", "--dump-hdr", multi] + extra + common,
              env_extra={"DE_DIST_BACKEND": "gloo", "DE_NO_LIB_LEG": "1"}, timeout=2000)
    assert p2.returncode == 0, (p2.stdout[-2000:], p2.stderr[-4000:])
    j = _json_line(p2.stdout)
    assert j["n_gpus"] == 8 and j["ranks_seen"] == 8 and j["scaling"] == "strong" and j["value"] > 0
    assert j["roofline"]["frac"] > 0 and j["roofline"]["bound"] == "valu_issue" and j["unit"] == "Msamples/s"
    assert [pr["rank"] for pr in j["per_rank"]] == list(range(8))
    a, b = np.load(solo), np.load(multi)
    assert a.max() > 0 and np.isfinite(b).all()
    if mode == "samples":
        assert all(pr["samples_per_pixel"] == 1 for pr in j["per_rank"])
        rel = float(np.sqrt(((a.astype(np.float64) - b) ** 2).sum()) / np.sqrt((a.astype(np.float64) ** 2).sum()))
        assert rel <= 1e-6 and j["multi_gpu_parity"]["rel_l2"] <= 1e-6, (rel, j["multi_gpu_parity"])
    else:
        tiles = [pr["tiles_owned"] for pr in j["per_rank"]]
        assert sum(tiles) == (256 // 8) * (128 // 8) and max(tiles) / min(tiles) <= 1.01
        assert j["multi_gpu_parity"]["bit_identical_fraction"] == 1.0
        assert (a.view(np.uint32) == b.view(np.uint32)).all()


@pytest.mark.gpu
@pytest.mark.timeout(1200)
def test_progressive_frames_on_two_ranks(tmp_path):
    """BASELINE cfg5's mode: out-of-place reduce + display after every K spp.  The assembled frame after the last
    reduce equals the 1-rank frame bit for bit — an in-place reduce would have counted the other ranks' early samples twice."""
    solo, multi = str(tmp_path / "solo.npy"), str(tmp_path / "multi.npy")
    common = [a if a != "3" else "6" for a in COMMON]                     # 6 spp, reduced after every 2
    p1 = _run([sys.executable, "bench.py", "--gpus", "1", "--dump-hdr", solo] + common)
    assert p1.returncode == 0, p1.stderr[-3000:]
    port = 29700 + (os.getpid() % 1500)
    p2 = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), "bench.py", "--gpus", "2", "--progressive", "2", "--collective", "gather", "--dump-hdr", multi] + common,      # the gather of the owned pixels, out of place (the in-place sum-reduce: the test above)
              env_extra={"DE_DIST_BACKEND": "gloo"})
    assert p2.returncode == 0, (p2.stdout[-2000:], p2.stderr[-4000:])
    j2 = _json_line(p2.stdout)
    assert j2["config"]["progressive_every"] == 2
    a, b = np.load(solo), np.load(multi)
    assert a.max() > 0 and (a.view(np.uint32) == b.view(np.uint32)).all()


@pytest.mark.gpu
@pytest.mark.timeout(1200)
def test_progressive_frames_on_one_rank(tmp_path):
    """with ONE rank the out-of-place reduce is a copy on the torch stream; it must be ordered after the launch slots
    (`DistributedFrame._reduce` flushes unconditionally), or the assembled frame that gets displayed and dumped is a partial sum."""
    plain, prog = str(tmp_path / "plain.npy"), str(tmp_path / "prog.npy")
    common = [a if a != "3" else "6" for a in COMMON]
    p1 = _run([sys.executable, "bench.py", "--gpus", "1", "--dump-hdr", plain] + common)
    assert p1.returncode == 0, p1.stderr[-3000:]
    p2 = _run([sys.executable, "bench.py", "--gpus", "1", "--progressive", "2", "--dump-hdr", prog] + common)
    assert p2.returncode == 0, p2.stderr[-3000:]
    a, b = np.load(plain), np.load(prog)
    assert a.max() > 0 and (a.view(np.uint32) == b.view(np.uint32)).all()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_library_communicator_refuses_two_ranks_on_one_gpu_cleanly():
    """`bench.py --reduce lib` (de_comm_init / de_reduce inside the library) with two ranks on ONE device: RCCL refuses a
    duplicate GPU at ncclCommInitRank; the library must turn that into an error message and a non-zero exit, not a crash or a
    hang.  (Between two GPUs this path is unverified: no multi-GPU box was available to the builder.)"""
    port = 29800 + (os.getpid() % 1500)
    p = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", str(port), "bench.py", "--gpus", "2", "--reduce", "lib"] + COMMON,
             env_extra={"DE_DIST_BACKEND": "gloo", "NCCL_DEBUG": "WARN"}, timeout=420)
    text = p.stdout + p.stderr
    assert p.returncode != 0
    assert "ncclCommInitRank" in text or "libdigitalearth_hip error" in text, text[-3000:]


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_in_library_reduce_leg_reports_instead_of_hanging(tmp_path):
    """bench.py --gpus N gives the library's own communicator one untimed frame after the measurement (round 4) and reports it as
    `lib_reduce`.  Two ranks on ONE device: RCCL refuses the communicator — the leg must turn that into an `error` entry of the line (the
    measurement intact, exit code 0), not into a hang or a lost line.  Between two GPUs the same leg reports bit_identical_fraction."""
    port = 29500 + (os.getpid() % 400)
    p = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", str(port), "bench.py", "--gpus", "2"] + COMMON,
             env_extra={"DE_DIST_BACKEND": "gloo", "DE_FORCE_LIB_LEG": "1"}, timeout=600)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    j = _json_line(p.stdout)
    assert j["value"] > 0 and j["multi_gpu_parity"]["bit_identical_fraction"] == 1.0
    assert "lib_reduce" in j and ("error" in j["lib_reduce"] or j["lib_reduce"].get("bit_identical_fraction") == 1.0), j.get("lib_reduce")
    assert all("kernel" in pr for pr in j["per_rank"]) and j["config"]["primed_frames"] == 4


def _visible_gpus():
    """Counted in a child process: this pytest process must not load torch's HIP runtime before the library's own."""
    try:
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], stdout=subprocess.PIPE,
                             stderr=subprocess.DEVNULL, timeout=300, text=True).stdout.strip().splitlines()
        return int(out[-1]) if out else 0
    except Exception:
        return 0


@pytest.mark.gpu
@pytest.mark.timeout(1800)
@pytest.mark.skipif(_visible_gpus() < 2, reason="needs two MI355X: the builder's box has one (this is the first test to run on a multi-GPU node)")
@pytest.mark.parametrize("reduce_impl", ["torch", "lib", "torch-gather", "torch-samples", "lib-samples"])
def test_two_gpus_rccl_reduce_matches_one_rank(tmp_path, reduce_impl):
    """RCCL between two devices: torch.distributed's reduce on the bound tensor, and the library's own communicator
    (de_comm_init / de_reduce: hand-declared ncclCommInitRank ABI, ncclFloat32 / ncclSum enums).  The assembled frame equals the
    1-rank frame bit for bit."""
    solo, multi = str(tmp_path / "solo.npy"), str(tmp_path / "multi.npy")
    p1 = _run([sys.executable, "bench.py", "--gpus", "1", "--dump-hdr", solo] + COMMON)
    assert p1.returncode == 0, p1.stderr[-3000:]
    port = 29900 + (os.getpid() % 1500)
    p2 = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), "bench.py", "--gpus", "2", "--dump-hdr", multi] +
              {"torch": ["--reduce", "torch"], "lib": ["--reduce", "lib"], "torch-gather": ["--collective", "gather"],
               "torch-samples": ["--split", "samples", "--reduce", "torch"], "lib-samples": ["--split", "samples", "--reduce", "lib"]}[reduce_impl] + COMMON)
    assert p2.returncode == 0, (p2.stdout[-2000:], p2.stderr[-4000:])
    j2 = _json_line(p2.stdout)
    a, b = np.load(solo), np.load(multi)
    if reduce_impl.endswith("samples"):
        # the sample partition's collective (de_reduce_ordered's ncclSend / ncclRecv group and ordered_sum_kernel with two parts had never met a second rank):
        # rank-ordered partial sums, f32 re-association against the one-rank frame only
        rel = float(np.sqrt(((a.astype(np.float64) - b) ** 2).sum()) / np.sqrt((a.astype(np.float64) ** 2).sum()))
        assert j2["devices_seen"] == 2 and rel <= 1e-6 and j2["multi_gpu_parity"]["rel_l2"] <= 1e-6, (rel, j2["multi_gpu_parity"])
        return
    assert j2["devices_seen"] == 2 and j2["multi_gpu_parity"]["bit_identical_fraction"] == 1.0
    assert (a.view(np.uint32) == b.view(np.uint32)).all()
