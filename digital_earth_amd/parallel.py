"""Multi-GPU partition of one frame (no reference counterpart; BASELINE.json north_star): tiles (default) or samples.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  The image grid is cut into 8x8-pixel
tiles (one wavefront each); tile (tx, ty) belongs to rank (tx + ty) % world — a diagonal interleave, so that every
rank gets the same mix of black sky, limb and cloudy day side.  Each rank accumulates its tiles into a full-size,
zero-initialised HDR buffer; one sum-reduce of that buffer to rank 0 per fetch assembles the frame (the tiles are
disjoint, so every pixel is x + 0 + ... + 0: exact).  The RNG stream of a sample depends only on
(seed, absolute pixel, sample index), so the assembled frame is bit-identical to a 1-GPU render.

Ways to run the collective:
  * `reduce_hdr` — torch.distributed on a torch-owned HDR tensor (default; what bench.py times);
  * `gather_hdr_tiles` — the same frame from a gather of the owned pixels only (1/world of the bytes; `DistributedFrame(collective="gather")`,
    `bench.py --collective gather`);
  * `LibraryComm` — the library's own RCCL communicator behind the C ABI (`de_comm_init` / `de_reduce`,
    include/digital_earth.h), for hosts without PyTorch; only the 128-byte unique id has to travel between ranks.

The SAMPLE partition (`split="samples"`, SURVEY §8e's alternative for the 1024-spp progressive config; the reference's loop is one
accumulate() per sample index, renderer.py:371-380): every rank renders the whole image for the sample indices = rank (mod world) —
perfect balance, no per-tile launch tail.  Every pixel then has a partial sum on every rank and f32 addition is not associative, so the
collective is `reduce_hdr_ordered`: gather on the root, add in rank order.  The assembled frame is bit-identical to those partial sums
added that way on any machine; against the 1-rank frame it differs by re-association only (relative L2 <= 1e-6, tests).

The functions take torch tensors on any device, which is how the world_size-2 gloo tests exercise them on CPU.
"""
import numpy as np

TILE = 8


def tile_owner(tx, ty, world):
    return (tx + ty) % world


def owned_pixel_mask(width, height, rank, world):
    """Boolean (W, H) mask of the pixels rank `rank` renders — the host-side statement of the kernel's ownership rule."""
    tx = np.arange(width) // TILE
    ty = np.arange(height) // TILE
    return ((tx[:, None] + ty[None, :]) % world) == rank


def tiles_per_rank(width, height, world):
    tx, ty = width // TILE, height // TILE
    owner = (np.arange(tx)[:, None] + np.arange(ty)[None, :]) % world
    return np.bincount(owner.ravel(), minlength=world)


def reduce_hdr(hdr, dst=0, group=None, out=None):
    """Sum the per-rank HDR buffers onto rank `dst`.  One collective per fetch, not per sample: 24.9 MB at 1080p, 99.5 MB at 4K.

    out=None: in place (the root's buffer becomes the assembled frame — the one reduce at the end of a frame).
    out=tensor: OUT OF PLACE — `hdr` is left untouched on every rank (it keeps accumulating: progressive display, "reduce every
    K spp") and the root finds the assembled frame in `out`; on the other ranks `out` is scratch.

    RCCL ("nccl") reduces device tensors on the current stream.  gloo has no reduce for device tensors: there the buffer is
    staged through host memory (the 2-ranks-on-one-GPU dry run and the CPU tests)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) <= 1:
        if out is not None:
            out.copy_(hdr)
            return out
        return hdr
    if hdr.is_cuda and dist.get_backend(group) != "nccl":
        host = hdr.cpu()                                   # synchronises the current stream
        dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM, group=group)
        if dist.get_rank(group) == dst:
            (out if out is not None else hdr).copy_(host)
        return out if out is not None else hdr
    if out is not None:
        out.copy_(hdr)
        dist.reduce(out, dst=dst, op=dist.ReduceOp.SUM, group=group)
        return out
    dist.reduce(hdr, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return hdr


_TILE_INDEX_CACHE = {}


def _owned_pixel_index(width, height, rank, world, device, layout):
    """Linear pixel indices of the pixels `rank` owns, as an int64 tensor on `device` (cached).  layout "WH": x * H + y — the (W, H, 3) arrays the API hands
    out (fetch_hdr, the reference's field); "HW": y * W + x — the library's accumulation buffer on the device (a bound HDR tensor)."""
    import torch
    key = (width, height, rank, world, str(device), layout)
    if key not in _TILE_INDEX_CACHE:
        mask = owned_pixel_mask(width, height, rank, world)
        _TILE_INDEX_CACHE[key] = torch.from_numpy(np.flatnonzero((mask if layout == "WH" else mask.T).ravel())).to(device)
    return _TILE_INDEX_CACHE[key]


def gather_hdr_tiles(hdr, width, height, dst=0, group=None, out=None, layout="WH"):
    """The other collective of a TILE partition ("a direct 7-peer gather of 1/8-size tiles ... is the bandwidth-optimal variant"):
    every rank packs the pixels it owns (12 W H / world bytes: 3.1 MB at 1080p and 8 ranks instead of the 24.9 MB a sum-reduce of the whole buffer
    moves per rank), one gather onto `dst`, which scatters each rank's pixels into place.  The assembled frame holds every rank's bits UNCHANGED
    (a copy instead of x + 0 + ... + 0) — identical to `reduce_hdr`'s except that a -0.0 stays -0.0.  out=None: the root's own buffer becomes the
    frame; out=tensor: out of place (progressive display).  gloo + device tensors: staged through host memory, as in reduce_hdr.  `layout`: the pixel order
    of `hdr` (see _owned_pixel_index)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) <= 1:
        if out is not None:
            out.copy_(hdr)
            return out
        return hdr
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    staged = hdr.is_cuda and dist.get_backend(group) != "nccl"
    counts = [int(n) * TILE * TILE for n in tiles_per_rank(width, height, world)]
    n_max = max(counts)
    mine = _owned_pixel_index(width, height, rank, world, hdr.device, layout)
    send = torch.zeros(n_max, 3, dtype=hdr.dtype, device=hdr.device)
    send[:counts[rank]] = hdr.view(-1, 3)[mine]
    if staged:
        send = send.cpu()
    parts = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, gather_list=parts, dst=dst, group=group)
    if rank == dst:
        target = out if out is not None else hdr
        tv = target.view(-1, 3)
        for r in range(world):
            if r == rank and out is None:
                continue                                     # in place: the root's own pixels are where they belong
            idx = _owned_pixel_index(width, height, r, world, hdr.device, layout)
            tv[idx] = parts[r][:counts[r]].to(hdr.device)
    return out if out is not None else hdr


def reduce_hdr_ordered(hdr, dst=0, group=None, out=None):
    """The collective of a SAMPLE partition: the per-rank buffers (partial sums of every pixel) are gathered on `dst` and added in RANK
    order — ((h0 + h1) + h2) + ... — so the result does not depend on how RCCL would have ordered a sum-reduce.  out=None: the root's
    `hdr` becomes the assembled frame; out=tensor: out of place (progressive display: every rank keeps its running sums).
    12 W H bytes per peer into the root: 25 MB x 7 over seven xGMI links at 1080p."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) <= 1:
        if out is not None:
            out.copy_(hdr)
            return out
        return hdr
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    staged = hdr.is_cuda and dist.get_backend(group) != "nccl"       # gloo: through host memory (several ranks on ONE GPU, CPU tests)
    src = hdr.cpu() if staged else hdr
    parts = [torch.empty_like(src) for _ in range(world)] if rank == dst else None
    dist.gather(src, gather_list=parts, dst=dst, group=group)
    if rank == dst:
        acc = parts[0].clone()
        for r in range(1, world):
            acc += parts[r]
        (out if out is not None else hdr).copy_(acc)
    return out if out is not None else hdr


class DistributedFrame:
    """Binds a Renderer to this process's rank: torch owns the HDR buffer (so RCCL can reduce it in place) and the
    renderer uses a torch stream as its context stream (so the collective is ordered after the launches, see `flush`).

    Lifetime: the Renderer is handed references to the tensor and the stream (`Renderer._bound`), so neither can be
    collected while the context still points at them; `close()` gives the context back its own buffer and stream."""

    def __init__(self, renderer, rank, world, device_index, stream=None, split="tiles", collective="reduce"):
        import torch
        if split not in ("tiles", "samples"):
            raise ValueError(split)
        if collective not in ("reduce", "gather"):
            raise ValueError(collective)
        self.collective = collective                 # tile partition only: sum-reduce of the whole buffer (default) or gather of the owned pixels
        self.torch = torch
        self.r = renderer
        self.rank, self.world, self.split = rank, world, split
        self.device = torch.device("cuda", device_index)
        self.stream = stream if stream is not None else torch.cuda.current_stream(self.device)
        W, H = renderer.image_res
        with torch.cuda.stream(self.stream):
            # allocated AND zero-filled on the frame's stream: every later reset / accumulate / reduce is enqueued on
            # the same stream, so the fill is ordered before them (a fill on torch's default stream would not be)
            self.hdr = torch.zeros(H * W * 3, dtype=torch.float32, device=self.device)
        self.stream.synchronize()
        self.assembled = None                       # root's frame of the out-of-place (progressive) reduce
        self._ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        self._timed = False
        self.standin_extra = None                   # one-GPU experiments: an int makes _reduce enqueue the stand-in collective (Renderer.debug_standin_reduce) when world == 1
        renderer.bind_hdr(self.hdr.data_ptr(), self.hdr.numel())
        renderer.set_stream(self.stream.cuda_stream)
        renderer._bound = (self.hdr, self.stream)
        if split == "samples":
            renderer.set_tile_partition(0, 1)
            renderer.set_sample_partition(rank, world)
        else:
            renderer.set_sample_partition(0, 1)
            renderer.set_tile_partition(rank, world)

    def _reduce(self, out=None):
        # what follows is torch's own work on this stream (the collective, or with one rank the copy into `out`): order it
        # after the launches — they run on the context's launch slots, not on this stream
        self.r.flush()
        self._ev[0].record(self.stream)
        if self.world == 1 and self.standin_extra is not None:
            self.r.debug_standin_reduce(self.standin_extra)      # where the collective of an N-rank run goes: after the launches, before the display
        if self.split == "samples":
            reduce_hdr_ordered(self.hdr, dst=0, out=out)
        elif self.collective == "gather":
            gather_hdr_tiles(self.hdr, self.r.image_res[0], self.r.image_res[1], dst=0, out=out, layout="HW")      # the bound tensor IS the device buffer: [H][W][3]
        else:
            reduce_hdr(self.hdr, dst=0, out=out)
        self._ev[1].record(self.stream)
        self._timed = True

    def render_frame(self, spp):
        """reset -> accumulate(spp) on this rank's tiles -> reduce to rank 0 -> display transform on rank 0 (device).
        Everything is enqueued; with RCCL nothing here waits for the GPU."""
        with self.torch.cuda.stream(self.stream):
            self.r.set_display_source(None)
            self.r.reset_framebuffer()        # zeroes the bound HDR buffer on the frame's stream
            self.r.accumulate(spp)
            self._reduce()
            if self.rank == 0:
                self.r.render_to_image_device()

    def render_frame_progressive(self, spp, every):
        """BASELINE cfg5 (1024 spp progressive): reset, then `every` samples at a time -> out-of-place reduce -> display on
        rank 0.  Each rank's buffer keeps its own running sums; rank 0 shows the assembled frame after every reduce."""
        with self.torch.cuda.stream(self.stream):
            if self.assembled is None:
                self.assembled = self.torch.zeros_like(self.hdr)
            self.r.reset_framebuffer()
            done = 0
            while done < spp:
                n = min(every, spp - done)
                self.r.accumulate(n)
                self._reduce(out=self.assembled)
                if self.rank == 0:
                    self.r.set_display_source(self.assembled.data_ptr())
                    self.r.render_to_image_device()
                done += n

    def last_reduce_ms(self):
        if not self._timed:
            return 0.0
        self._ev[1].synchronize()
        return float(self._ev[0].elapsed_time(self._ev[1]))

    def fetch_hdr(self):
        """The (reduced, on rank 0) HDR buffer as a (W, H, 3) array."""
        return self.r.fetch_hdr()

    def synchronize(self):
        self.r.synchronize()
        self.stream.synchronize()

    def close(self):
        if getattr(self, "r", None) is not None and getattr(self.r, "_bound", None) is not None:
            self.r.synchronize()
            self.stream.synchronize()
            self.r.set_display_source(None)
            self.r.unbind()
        self.r = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LibraryComm:
    """The library's own RCCL communicator (C ABI: de_comm_unique_id / de_comm_init / de_reduce / de_comm_destroy).

    `exchange(id_bytes_or_None) -> id_bytes` is any host-side broadcast of the 128-byte unique id from rank 0
    (a file, a socket, MPI, torch.distributed.broadcast_object_list ...): the only thing the host framework provides.
    One communicator per rank: the launches of consecutive frames overlap inside the context (launch slots), so frames in
    flight need no communicators of their own.  UNVERIFIED between two GPUs (no multi-GPU box was available to the builder;
    tests/test_gpu_round2.py runs it with one rank, tests/test_a_multi_rank_gpu.py asserts RCCL's clean refusal of two ranks on
    one device)."""

    def __init__(self, renderer, rank, world, exchange, split="tiles"):
        if split not in ("tiles", "samples"):
            raise ValueError(split)
        self.r = renderer
        self.rank, self.world, self.split = rank, world, split
        uid = renderer.comm_unique_id() if rank == 0 else None
        uid = exchange(uid)
        renderer.comm_init(uid, rank, world)
        if split == "samples":
            renderer.set_tile_partition(0, 1)
            renderer.set_sample_partition(rank, world)
        else:
            renderer.set_sample_partition(0, 1)
            renderer.set_tile_partition(rank, world)

    def render_frame(self, spp, root=0):
        self.r.reset_framebuffer()
        self.r.accumulate(spp)
        if self.split == "samples":
            self.r.reduce_ordered(root)
        else:
            self.r.reduce(root)
        if self.rank == root:
            self.r.render_to_image_device()

    def render_frame_progressive(self, spp, every, root=0):
        self.r.reset_framebuffer()
        done = 0
        while done < spp:
            n = min(every, spp - done)
            self.r.accumulate(n)
            if self.split == "samples":
                self.r.reduce_ordered(root, out_of_place=True)
            else:
                self.r.reduce_progressive(root)
            if self.rank == root:
                self.r.render_to_image_device()
            done += n

    def last_reduce_ms(self):
        return self.r.last_reduce_ms()

    def fetch_hdr(self):
        return self.r.fetch_hdr()

    def synchronize(self):
        self.r.synchronize()

    def close(self):
        if self.r is not None:
            self.r.comm_destroy()
            self.r = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
