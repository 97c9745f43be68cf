#!/usr/bin/env python3
"""bench.py — Msamples/s of the spectral path-tracing hot path on N MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One "step" = one frame of the workload: reset_framebuffer, accumulate(spp) on this rank's 8x8 tiles, (N > 1) one
RCCL sum-reduce of the HDR buffer to rank 0, and the display transform on rank 0.  Textures and LUTs are resident
in HBM before the timed region (the reference uploads them once, renderer.py:136-145).  The frame is FIXED and split
over the ranks (tile partition), so scaling is "strong".  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1]): 1920x1080, 64 spp, full spectral Rayleigh+Mie+O3+clouds, quality-2 maps
(21600x10800 / 16200x8100, synthetic stand-ins: the reference ships no maps), default camera and sun.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
FP32_VECTOR_PEAK_TFLOPS = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--tex-quality", type=int, default=2)
    ap.add_argument("--preset", default=None, help="camera preset file name under digital_earth_amd/data/configs")
    ap.add_argument("--cloud-heavy", action="store_true")
    ap.add_argument("--frames-in-flight", type=int, default=2, help="independent frame contexts alternated step by step (tail overlap)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--simulate-world", type=int, default=0, help="1-GPU experiment: render only rank 0's share of an N-rank tile partition (no collective); not a bench line")
    ap.add_argument("--cpu-spp", type=int, default=8, help="upper bound on the spp of the bounded CPU-baseline sample (~15 s)")
    return ap.parse_args()


def apply_preset(r, name):
    from digital_earth_amd.earth_viewer import load_config
    cfg = load_config(os.path.join(ROOT, "digital_earth_amd", "data", "configs", name))
    cfg.apply(r)


def cpu_baseline(r, args):
    """The CPU oracle (a port of the reference algorithm, NOT Taichi's ti.cpu) on a bounded sample of the same workload:
    the full frame at --cpu-spp samples per pixel, same maps / camera / seed, all host threads."""
    import ctypes
    from digital_earth_amd import luts
    from oracle import oracle_binding as ob
    W, H = r.image_res
    o = ob.Oracle(W, H)
    names, crf = luts.load_crfs()
    o.upload_luts(luts.load_cie(), luts.load_srgb2spec(), luts.load_o3(), crf)
    for s in range(7):
        o.upload_texture(s, r.download_texture(s))
    p = ob.DeParams()
    ctypes.memmove(ctypes.byref(p), ctypes.byref(r._params), ctypes.sizeof(p))
    o.set_params(p)
    # bounded sample: 1 spp first; then as many more samples per pixel as fit in ~15 s (at most --cpu-spp, default 8)
    t0 = time.perf_counter()
    o.accumulate(1, r.seed)
    t1 = time.perf_counter() - t0
    extra = int(min(max(args.cpu_spp - 1, 0), max(0.0, 15.0 / max(t1, 1e-3) - 1.0)))
    if extra > 0:
        o.accumulate(extra, r.seed)
    dt = time.perf_counter() - t0
    args.cpu_spp = 1 + extra
    n = W * H * args.cpu_spp
    out = {"value": n / dt / 1e6, "unit": "Msamples/s", "cores": o.threads, "kind": "port",
           "sample": "full %dx%d frame at %d spp (%d samples, %.1f s), same maps/camera/seed; C++ restatement of the "
                     "reference algorithm (oracle/), OpenMP over 16x8 pixel blocks" % (W, H, args.cpu_spp, n, dt)}
    # the same sample doubles as a full-size parity check of the GPU image (1 spp at matched seed)
    return out, o


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist
    from digital_earth_amd.renderer import Renderer
    from digital_earth_amd import parallel

    n_dev = max(torch.cuda.device_count(), 1)
    device_index = local_rank % n_dev          # one rank per GPU on the driver's node; modulo only matters for the 1-GPU dry run
    torch.cuda.set_device(device_index)
    if world > 1:
        # "nccl" is RCCL on ROCm.  DE_DIST_BACKEND=gloo lets the N > 1 code path be exercised on a single-GPU box.
        backend = os.environ.get("DE_DIST_BACKEND", "nccl")
        kw = {"device_id": torch.device("cuda", device_index)} if backend == "nccl" else {}
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)

    W, H, spp = args.width, args.height, args.spp
    # Frames in flight: each has its own context (maps, buffers) and HIP stream; consecutive steps alternate between them,
    # so the tail of one frame (a few very long paths finishing) overlaps the start of the next.  Every step is still a
    # complete frame (reset, spp samples per pixel, reduce, display).
    frames = []
    for k in range(max(1, args.frames_in_flight)):
        rk = Renderer((W, H), (0, 1, 0), device=device_index, texture_source="synthetic", texture_quality=args.tex_quality,
                      cloud_heavy=args.cloud_heavy, seed=0)
        if args.preset:
            apply_preset(rk, args.preset)
        rk.copy_textures()
        frames.append(parallel.DistributedFrame(rk, rank, world, device_index, stream=torch.cuda.Stream(device=device_index)))
        if args.simulate_world > 1 and world == 1:
            rk.set_tile_partition(0, args.simulate_world)
    r, frame = frames[0].r, frames[0]

    def barrier():
        for f in frames:
            f.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # algorithmic work per sample, measured by the kernel's own counters on this camera/seed (untimed)
    r.enable_counters(True)
    r.reset_framebuffer()
    r.accumulate(2)
    r.synchronize()
    cnt = r.counters()
    r.enable_counters(False)
    n_r8 = cnt["taps_r8"] / max(cnt["samples"], 1)
    n_rgb = cnt["taps_rgb8"] / max(cnt["samples"], 1)

    for i in range(args.warmup):
        frames[i % len(frames)].render_frame(spp)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        frames[i % len(frames)].render_frame(spp)
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0

    # an extra, untimed pass to collect the HIP-event duration of every launch (events are re-recorded per call)
    per_launch = []
    for _ in range(max(1, min(args.steps, 3))):
        frame.render_frame(spp)
        r.synchronize()
        per_launch.append(r.last_accumulate_ms())
    avg_kernel_ms = float(np.mean(per_launch))

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        k = torch.tensor([avg_kernel_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(k, op=dist.ReduceOp.MAX)
        avg_kernel_ms = float(k.item())

    total_samples = W * H * spp * args.steps
    value = total_samples / elapsed / 1e6
    ms_per_step = elapsed / args.steps * 1e3

    # roofline of the dominant kernel (render_kernel): algorithmic bytes per launch / measured launch duration.
    # B_sample = 8 B contribution record + 4 B per r8 tap + 16 B per colour tap
    # (SURVEY §8d); samples per launch on this rank = its share of the frame.
    my_samples = int(parallel.owned_pixel_mask(W, H, rank, world).sum()) * spp
    # render_kernel_v2 writes one 8-byte (radiance, wavelength node) record per sample; the HDR read-modify-write
    # (24 B per pixel per launch) happens in accumulate_kernel
    b_sample = 8.0 + 4.0 * n_r8 + 16.0 * n_rgb
    achieved_gbps = b_sample * my_samples / (avg_kernel_ms * 1e-3) / 1e9
    traffic = None
    valu_insts = lane_util = None
    traffic_note = "not collected in this run (see profiles/ for the rocprofv3 --pmc pass)"
    prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(prof):
        try:
            with open(prof) as f:
                pj = json.load(f)
            if pj.get("workload") == "%dx%d_%dspp_q%d" % (W, H, spp, args.tex_quality) and world == 1:
                traffic = pj.get("hbm_bytes_per_launch")
                traffic_note = pj.get("note", "")
                valu_insts, lane_util = pj.get("valu_insts_per_launch"), pj.get("valu_lane_utilisation")
        except Exception:
            pass
    # VALU view of the same launch: iterations x a per-iteration f32 op estimate (DESIGN.md "Algorithmic flops")
    ops_per_sample = cnt["sphere_steps"] / max(cnt["samples"], 1) * 190.0 + cnt["tracking_steps"] / max(cnt["samples"], 1) * 150.0 \
        + cnt["vertices"] / max(cnt["samples"], 1) * 600.0
    achieved_tflops = ops_per_sample * my_samples / (avg_kernel_ms * 1e-3) / 1e12

    out = {
        "metric": "Msamples/sec (1920x1080, 64 spp) spectral path tracing", "value": value, "unit": "Msamples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%dx%d, %d spp, full spectral Rayleigh+Mie+O3+clouds, quality-%d synthetic maps, %s, "
                               "BASELINE configs[1]" % (W, H, spp, args.tex_quality, args.preset or "default camera"),
                   "parallelism": "tile-partition x%d + RCCL reduce" % world if world > 1 else "single GPU",
                   "frames_in_flight": len(frames),
                   "samples_per_step": W * H * spp},
        "roofline": {"bound": "hbm", "achieved": achieved_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved_gbps / HBM_PEAK_GBPS, "traffic": traffic, "traffic_note": traffic_note,
                     "kernel": "render_kernel", "kernel_ms": avg_kernel_ms, "bytes_per_sample": b_sample,
                     "taps_r8_per_sample": n_r8, "taps_rgb_per_sample": n_rgb,
                     "note": "the path is VALU/latency bound, not HBM bound (SURVEY §8d): see valu"},
        "valu": {"achieved": achieved_tflops, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s (f32 vector, estimated op count)",
                 "frac": achieved_tflops / FP32_VECTOR_PEAK_TFLOPS, "ops_per_sample": ops_per_sample,
                 "sphere_steps_per_sample": cnt["sphere_steps"] / max(cnt["samples"], 1),
                 "tracking_steps_per_sample": cnt["tracking_steps"] / max(cnt["samples"], 1),
                 "vertices_per_sample": cnt["vertices"] / max(cnt["samples"], 1)},
    }
    if valu_insts:
        # The bound that actually holds: VALU issue.  Peak = one wave64 f32 instruction per 1.11 ns per SIMD, measured on
        # this chip with tools/micro/dep_latency.hip (independent of chain count, reached with >= 3 waves per SIMD);
        # instructions per launch = SQ_INSTS_VALU of the committed rocprofv3 --pmc pass (same workload).
        peak_ginst = 256 * 4 / 1.11                     # 1e9 wave-instructions / s
        ach_ginst = valu_insts / (avg_kernel_ms * 1e-3) / 1e9
        out["valu"]["issue"] = {"achieved": ach_ginst, "peak": peak_ginst, "unit": "1e9 wave64 VALU instructions/s",
                                "frac": ach_ginst / peak_ginst, "insts_per_launch": valu_insts, "lane_utilisation": lane_util,
                                "note": "peak measured by tools/micro/dep_latency.hip; instruction count from profiles/ (rocprofv3 SQ_INSTS_VALU)"}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base, o = cpu_baseline(r, args)
        out["cpu_baseline"] = base
        # full-size parity at matched seed: GPU 1-spp frame vs the CPU frame just computed
        r.set_tile_partition(0, 1)
        r.reset_framebuffer()
        r.accumulate(args.cpu_spp)
        g = r.fetch_hdr().astype(np.float64)
        c = o.fetch_hdr().astype(np.float64)
        out["parity"] = {"rel_l2_vs_cpu_port": float(np.sqrt(((g - c) ** 2).sum()) / max(np.sqrt((c ** 2).sum()), 1e-300)),
                         "bit_identical_fraction": float((g == c).mean()), "spp": args.cpu_spp}
        out["speedup_vs_cpu_port"] = value / base["value"]
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
