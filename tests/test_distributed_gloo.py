"""N > 1 path on CPU: two processes (gloo), each renders its own tiles — with the oracle standing in for the GPU —
into a full-size zero-initialised HDR tensor; parallel.reduce_hdr sums them onto rank 0, which must equal the
single-process frame bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP, SEED = 64, 32, 2, 5


def _frame(rank, world):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import make_oracle
    from digital_earth_amd import luts
    names, crf = luts.load_crfs()
    o = make_oracle(W, H, (luts.load_cie(), luts.load_srgb2spec(), luts.load_o3(), crf, names), threads=2)
    p = o.get_params(); p.fov = 0.45; o.set_params(p)
    o.accumulate(SPP, SEED, rank, world)
    return o.fetch_hdr()


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from digital_earth_amd import parallel
    part = _frame(rank, world)
    mask = parallel.owned_pixel_mask(W, H, rank, world)
    assert (part[~mask] == 0).all()                    # a rank touches only its own tiles
    hdr = torch.from_numpy(part.copy())
    parallel.reduce_hdr(hdr, dst=0)
    if rank == 0:
        np.save(out_path, hdr.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_tile_split_and_reduce(tmp_path):
    out = str(tmp_path / "reduced.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    reduced = np.load(out)
    full = _frame(0, 1)
    assert (reduced.view(np.uint32) == full.view(np.uint32)).all()


def _frame_steps(rank, world, steps):
    """The running sums of a rank after each of `steps` accumulate(1) calls (oracle standing in for the GPU)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import make_oracle
    from digital_earth_amd import luts
    names, crf = luts.load_crfs()
    o = make_oracle(W, H, (luts.load_cie(), luts.load_srgb2spec(), luts.load_o3(), crf, names), threads=2)
    p = o.get_params(); p.fov = 0.45; o.set_params(p)
    out = []
    for _ in range(steps):
        o.accumulate(1, SEED, rank, world)
        out.append(o.fetch_hdr())
    return out


def _progressive_worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from digital_earth_amd import parallel
    sums = _frame_steps(rank, world, 3)
    shown = []
    work = torch.zeros(W, H, 3)
    for k in range(3):                                   # progressive display: reduce after every sample, out of place
        hdr = torch.from_numpy(sums[k].copy())
        before = hdr.clone()
        parallel.reduce_hdr(hdr, dst=0, out=work)
        assert torch.equal(hdr, before)                  # the rank's own running sums are untouched, on every rank
        if rank == 0:
            shown.append(work.numpy().copy())
    if rank == 0:
        np.save(out_path, np.stack(shown))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_progressive_out_of_place_reduce(tmp_path):
    """SURVEY §8e "reduce every K spp" (BASELINE cfg5): after each reduce the root shows the frame a single process would have
    after the same number of samples — which an in-place reduce cannot give (it would add the other ranks' running sums into the
    root's own buffer again at the next reduce)."""
    out = str(tmp_path / "shown.npy")
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_progressive_worker, args=(2, port, out), nprocs=2, join=True)
    shown = np.load(out)
    full = _frame_steps(0, 1, 3)
    for k in range(3):
        assert (shown[k].view(np.uint32) == full[k].view(np.uint32)).all(), k
