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
