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


def _sample_share(rank, world, spp, steps=1):
    """A rank's buffer under the SAMPLE partition after `steps` calls of accumulate(spp) (oracle standing in for the GPU)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import make_oracle
    from digital_earth_amd import luts
    names, crf = luts.load_crfs()
    o = make_oracle(W, H, (luts.load_cie(), luts.load_srgb2spec(), luts.load_o3(), crf, names), threads=2)
    p = o.get_params(); p.fov = 0.45; o.set_params(p)
    o.set_sample_partition(rank, world)
    out = []
    for _ in range(steps):
        o.accumulate(spp, SEED)
        out.append(o.fetch_hdr())
    return out


def _sample_worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from digital_earth_amd import parallel
    sums = _sample_share(rank, world, 3, steps=2)            # two progressive displays of 3 sample indices each
    shown = []
    work = torch.zeros(W, H, 3)
    for k in range(2):
        hdr = torch.from_numpy(sums[k].copy())
        before = hdr.clone()
        parallel.reduce_hdr_ordered(hdr, dst=0, out=work)    # out of place: every rank keeps its running sums
        assert torch.equal(hdr, before)
        if rank == 0:
            shown.append(work.numpy().copy())
    hdr = torch.from_numpy(sums[1].copy())
    parallel.reduce_hdr_ordered(hdr, dst=0)                  # in place: the root's buffer becomes the assembled frame
    if rank == 0:
        shown.append(hdr.numpy().copy())
        np.save(out_path, np.stack(shown))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sample_split_and_ordered_reduce(tmp_path):
    """SURVEY §8e's alternative split (bench.py --split samples): rank r renders the sample indices = r (mod world) of every pixel
    (reference renderer.py:371-380: one accumulate() per sample index, dealt round robin), the root gathers the ranks' partial sums and
    adds them in rank order.  The assembled frame equals those partial sums added the same way bit for bit — it does not depend on the
    collective's algorithm — and differs from the one-process frame (one running sum in sample order) by f32 re-association only."""
    out = str(tmp_path / "sample_split.npy")
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_sample_worker, args=(2, port, out), nprocs=2, join=True)
    shown = np.load(out)
    p0, p1 = _sample_share(0, 2, 3, steps=2), _sample_share(1, 2, 3, steps=2)
    for k in range(2):
        want = p0[k] + p1[k]                                  # rank order
        assert (shown[k].view(np.uint32) == want.view(np.uint32)).all(), k
    assert (shown[2].view(np.uint32) == (p0[1] + p1[1]).view(np.uint32)).all()
    full = _frame_steps(0, 1, 6)[5]
    err = np.sqrt(((shown[2].astype(np.float64) - full) ** 2).sum()) / np.sqrt((full.astype(np.float64) ** 2).sum())
    assert err <= 1e-6, err
    # a rank's share alone: the indices it owns, nothing else
    assert (p0[0] != full).any() and (p0[1] + p1[1] != 0).any()


def _gather_worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from digital_earth_amd import parallel
    part = _frame(rank, world)
    hdr = torch.from_numpy(part.copy())
    work = torch.full((W, H, 3), 7.0)                    # out of place first: the rank's own buffer must stay as it is
    before = hdr.clone()
    parallel.gather_hdr_tiles(hdr, W, H, dst=0, out=work)
    assert torch.equal(hdr, before)
    parallel.gather_hdr_tiles(hdr, W, H, dst=0)            # in place: the root's buffer becomes the frame
    dev = torch.from_numpy(np.ascontiguousarray(part.transpose(1, 0, 2))).reshape(-1)     # the library's device buffer: [H][W][3], flat (what a bound tensor is)
    parallel.gather_hdr_tiles(dev, W, H, dst=0, layout="HW")
    if rank == 0:
        np.save(out_path, np.stack([work.numpy(), hdr.numpy(), dev.reshape(H, W, 3).numpy().transpose(1, 0, 2)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_three_rank_tile_gather_equals_the_single_process_frame(tmp_path):
    """SURVEY §8e's bandwidth-optimal collective for the tile partition: each rank sends only the pixels it owns (unequal counts: 32 tiles over
    three ranks), the root scatters them into place — out of place and in place, the single-process frame bit for bit."""
    out = str(tmp_path / "gathered.npy")
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_gather_worker, args=(3, port, out), nprocs=3, join=True)
    got = np.load(out)
    full = _frame(0, 1)
    assert (got[0].view(np.uint32) == full.view(np.uint32)).all()
    assert (got[1].view(np.uint32) == full.view(np.uint32)).all()
    assert (got[2].view(np.uint32) == full.view(np.uint32)).all()      # the device buffer's pixel order
