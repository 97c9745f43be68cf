"""Round 5: the sample partition (SURVEY §8e's alternative split; reference renderer.py:371-380 dealt round robin) against the oracle,
the rank-ordered assembly, the ring protocol of render_kernel_v6 on pools small enough that every cell is reused thousands of times,
the launch's drain clock, the zero-copy image's lifetime."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import make_oracle, rel_l2

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def Renderer():
    from digital_earth_amd.renderer import Renderer as R
    return R


def _pair(Renderer, lut_arrays, W=128, H=64, fov=0.42, seed=21):
    r = Renderer((W, H), (0, 1, 0), texture_source="synthetic", texture_size=(2048, 1024), seed=seed)
    r.set_fov(fov)
    r.copy_textures()
    o = make_oracle(W, H, lut_arrays, [r.download_texture(s) for s in range(7)], r._params)
    return r, o


@pytest.mark.timeout(900)
def test_sample_partition_against_the_oracle(Renderer, lut_arrays):
    """Rank k of a 3-way sample partition renders sample indices k, k + 3, ... of every pixel; the frame's counter advances by the whole
    call on every rank; calls of 2 + 5 + 1 samples split the same indices as one call of 8.  Each rank's buffer equals the oracle's for
    the same partition bit for bit (8192 paths per sample index: render_kernel_v6; the last rank also on the state machine)."""
    WORLD, SEED = 3, 21
    parts = []
    for rank in range(WORLD):
        r, o = _pair(Renderer, lut_arrays)
        r.set_sample_partition(rank, WORLD)
        for n in (2, 5, 1):
            r.accumulate(n)
        assert r.current_spp == 8
        o.set_sample_partition(rank, WORLD)
        o.accumulate(8, SEED)
        g, c = r.fetch_hdr(), o.fetch_hdr()
        assert (g.view(np.uint32) == c.view(np.uint32)).all()
        if rank == WORLD - 1:
            r.set_kernel_variant(2)
            r.reset_framebuffer()
            r.accumulate(8)
            assert (r.fetch_hdr().view(np.uint32) == c.view(np.uint32)).all()
        parts.append(g)
        r.close()
    # the assembled frame: partial sums in rank order; against one running sum in sample order it is re-association only
    assembled = (parts[0] + parts[1]) + parts[2]
    r, o = _pair(Renderer, lut_arrays)
    r.accumulate(8)
    solo = r.fetch_hdr()
    assert rel_l2(assembled, solo) <= 1e-6
    assert np.abs(assembled - solo).max() <= 1e-6 * np.abs(solo).max()      # per element: a few ulps of the pixel's largest partial sum (RGB channels cancel)
    r.close()


@pytest.mark.timeout(900)
def test_sample_partition_with_an_empty_share(Renderer, lut_arrays):
    """A call that holds no sample index of this rank launches nothing and still advances the frame's counter."""
    r, o = _pair(Renderer, lut_arrays)
    r.set_sample_partition(3, 4)
    r.accumulate(2)                       # indices 0, 1: none of them = 3 (mod 4)
    assert r.current_spp == 2 and not r.fetch_hdr().any()
    r.accumulate(3)                       # indices 2, 3, 4: index 3
    o.set_sample_partition(3, 4)
    o.accumulate(5, 21)
    assert (r.fetch_hdr().view(np.uint32) == o.fetch_hdr().view(np.uint32)).all()
    r.close()


@pytest.mark.timeout(600)
def test_library_ordered_reduce_with_one_rank(Renderer):
    """de_reduce_ordered through a 1-rank communicator of the library (RCCL refuses two ranks on one device): the root's own buffer is the
    only part, in place and out of place."""
    r = Renderer((64, 32), (0, 1, 0), texture_source="synthetic", texture_size=(512, 256))
    r.copy_textures()
    r.comm_init(r.comm_unique_id(), 0, 1)
    r.set_sample_partition(0, 1)
    r.accumulate(2)
    before = r.fetch_hdr()
    r.reduce_ordered(0)
    assert (r.fetch_hdr().view(np.uint32) == before.view(np.uint32)).all()
    r.reduce_ordered(0, out_of_place=True)
    assert (r.fetch_hdr().view(np.uint32) == before.view(np.uint32)).all()      # the display source is the assembled buffer now
    assert r.last_reduce_ms() >= 0.0
    r.comm_destroy()
    r.close()


VARIANTS = [("v6_p64", "-DDE_V6_P=64u -DDE_V6_CAP=65u"), ("v6_p128", "-DDE_V6_P=128u -DDE_V6_CAP=129u"), ("v6_cap_p1", "-DDE_V6_CAP=1121u")]
_CHILD = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
from digital_earth_amd.renderer import Renderer
from digital_earth_amd.earth_viewer import load_config
out = {}
for k, (W, H, spp, preset, heavy) in enumerate(((256, 128, 6, None, False), (192, 96, 5, "config - sunset hurricane.txt", True), (128, 64, 7, "config - florida.txt", False))):
    r = Renderer((W, H), (0, 1, 0), texture_source="synthetic", texture_size=(2048, 1024), cloud_heavy=heavy, seed=3)
    if preset: load_config(preset).apply(r)
    r.set_kernel_variant(6)
    for rep in range(int(os.environ.get("DE_TEST_REPS", "1"))):
        r.reset_framebuffer(); r.accumulate(spp)
        out["f%%d_%%d" %% (k, rep)] = r.fetch_hdr()
    r.close()
np.savez(sys.argv[1], **out)
"""


@pytest.mark.timeout(3000)
def test_ring_protocol_on_small_pools(tmp_path):
    """VERDICT r4 weak #7 / ADVICE: the rings of render_kernel_v6 at DE_V6_CAP = DE_V6_P + 1 (no slack between what can be queued and the
    ring's size) and with pools of 64 and 128 slots per workgroup (a frame recycles every slot and every ring cell thousands of times; 12
    waves compete for one or two wave-loads of slots, so claims, takes and appends interleave as tightly as they can).  The variant
    libraries are built by __graft_entry__.build() (tools/ab_build.sh: the same sources with -D flags); every frame must equal the
    shipped geometry's bit for bit — a lost, duplicated or stale slot number is a wrong pixel or the abort word, never silent."""
    env = dict(os.environ)
    code = _CHILD % {"root": ROOT}
    ref_path = str(tmp_path / "ref.npz")
    subprocess.run([sys.executable, "-c", code, ref_path], check=True, env=env, timeout=900)
    ref = np.load(ref_path)
    for name, _flags in VARIANTS:
        lib = os.path.join(ROOT, "build", "ab", name + ".so")
        assert os.path.exists(lib), "variant library %s missing: run __graft_entry__.build()" % lib
        env2 = dict(env, DE_LIB_PATH=lib, DE_TEST_REPS="3", DE_V6_TAIL_MIN_PATHS="0", DE_V6_TAIL_ALONE="1")      # the tail chain too (its export threshold, 96 paths, is above these pools' size)
        path = str(tmp_path / (name + ".npz"))
        subprocess.run([sys.executable, "-c", code, path], check=True, env=env2, timeout=1800)
        got = np.load(path)
        for key in got.files:
            want = ref[key.rsplit("_", 1)[0] + "_0"]
            assert (got[key].view(np.uint32) == want.view(np.uint32)).all(), "%s %s" % (name, key)


@pytest.mark.timeout(600)
def test_launch_phases_and_zero_copy_image_lifetime(Renderer):
    r = Renderer((256, 128), (0, 1, 0), texture_source="synthetic", texture_size=(1024, 512))
    r.copy_textures()
    r.accumulate(4)
    launch_ms, drain_ms = r.last_launch_phases()
    assert r.last_call_info()["variant"] == 6
    assert 0.0 < drain_ms <= launch_ms < 1e4
    view = r.fetch_image(copy=False)
    assert not view.flags.writeable and view._owner is r
    with pytest.raises(RuntimeError):
        r.close()                         # the view IS the context's staging buffer
    keep = np.array(view)
    del view
    r.close()
    assert keep.max() <= 1.0


@pytest.mark.timeout(900)
def test_fast_math_flag_is_statistically_the_same_image(Renderer):
    """DE_FLAG_FAST_MATH (opt-in): the same integrator on the hardware's transcendental units.  Not bit-reproducible — a sample whose collision
    test lands on the other side follows another path (a path takes ~150 such tests: measured 5-14 % of the samples per camera,
    profiles/r5_fast_math.md) — so the tolerance is statistical: at 1 spp fewer than 25 % of the samples take another path, the others agree
    to 1e-3 relative (roundings); the mean radiance of a 64-spp frame agrees to 1 %, and the frame differs from the contract's by less than
    two renders of the contract's kernel with different seeds differ (the flag moves samples, not the estimate).
    The default (flag off) stays the arithmetic contract: bit-identical to the oracle (every other test)."""
    kw = dict(texture_source="synthetic", texture_size=(2048, 1024))
    frames = {}
    for name, fast, seed in (("contract", False, 5), ("fast", True, 5), ("other seed", False, 6)):
        r = Renderer((256, 128), (0, 1, 0), seed=seed, **kw)
        r.set_fov(0.42)
        r.copy_textures()
        if seed == 6:
            r.seed = 6
        r.set_fast_math(fast)
        r.accumulate(1)
        one = r.fetch_hdr().astype(np.float64).sum(axis=2)
        r.accumulate(63)
        assert r.last_call_info()["variant"] == 6
        frames[name] = (one, r.fetch_hdr().astype(np.float64))
        r.close()
    one_c, f_c = frames["contract"]
    one_f, f_f = frames["fast"]
    lit = np.maximum(one_c, one_f) > 0
    moved = np.abs(one_f - one_c) > 1e-3 * np.maximum(np.abs(one_c), np.abs(one_f))
    assert lit.sum() > 5000 and 0 < (moved & lit).sum() < 0.25 * lit.sum()
    assert abs(f_f.sum() / f_c.sum() - 1.0) < 0.01
    assert rel_l2(f_f, f_c) < rel_l2(frames["other seed"][1], f_c)


@pytest.mark.timeout(900)
def test_tail_chain_with_calls_in_flight_and_partitions(Renderer):
    """render_kernel_v6's tail chain (a launch exports its workgroups' last paths to a pool, a small launch of the same kernel finishes them) set through
    de_set_tuning on a live context: many calls of different sizes in flight on all launch slots (every slot's pool and control block in use at once, tail
    launches of one call queued behind the persistent workgroups of the next), a tile partition and a sample partition — the frame equals the one-launch-per-
    call frame bit for bit, with one and with two tail levels."""
    kw = dict(seed=11, texture_source="synthetic", texture_size=(2048, 1024))
    W, H = 512, 256

    def frame(levels, export0, grid0, alone=1):
        r = Renderer((W, H), (0, 1, 0), **kw)
        r.set_fov(0.42)
        r.copy_textures()
        t = r.tuning()
        t.v6_tail_levels = levels; t.v6_tail_export[0] = export0; t.v6_tail_grid[0] = grid0
        t.v6_tail_export[1] = 24; t.v6_tail_grid[1] = 4; t.v6_tail_min_paths = 0
        t.v6_tail_when_alone = alone      # 0 = as shipped: a launch exports only when another call is queued behind it on another launch slot (most of the twelve below are)
        r.set_tuning(t)
        r.set_kernel_variant(6)
        out = []
        for spp in (3, 1, 5, 2, 7, 1, 1, 4, 6, 2, 3, 5):      # twelve calls, nothing waits in between
            r.accumulate(spp)
        assert r.last_call_info()["variant"] == 6
        out.append(r.fetch_hdr())
        r.reset_framebuffer(); r.set_tile_partition(2, 5); r.accumulate(6); r.accumulate(3)
        out.append(r.fetch_hdr())
        r.set_tile_partition(0, 1); r.reset_framebuffer(); r.set_sample_partition(1, 3); r.accumulate(9)
        out.append(r.fetch_hdr())
        launch_ms, drain_ms = r.last_launch_phases()
        assert 0.0 < drain_ms <= launch_ms < 1e4
        r.close()
        return out

    ref = frame(0, 96, 64)
    assert ref[0].max() > 0 and ref[1].max() > 0 and ref[2].max() > 0
    for levels, export0, grid0, alone in ((1, 128, 64, 0), (1, 128, 64, 1), (2, 400, 128, 1), (1, 1024, 8, 0)):
        got = frame(levels, export0, grid0, alone)
        for a, b in zip(ref, got):
            assert (a.view(np.uint32) == b.view(np.uint32)).all(), (levels, export0, grid0, alone)
