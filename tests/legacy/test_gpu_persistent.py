"""render_kernel_v5 — path_tracer as ONE persistent kernel whose waves pick the fullest of seven chunked HBM queues and run that
stage's body (csrc/render_kernel_v5.hip) — is the same function as the wave-level state machine, the wavefront pipeline and the
oracle, bit for bit: the stage bodies are shared with render_kernel_v3, every path runs the same arithmetic on its own RNG stream,
only where and when changes.  What is new — and what these tests lean on — is the hand-over of records between workgroups INSIDE a
launch (write-through stores, agent-scope ring cells, chunk recycling): small pools, one-chunk claims and eight queue shards push
every path through as many hand-overs and recycled chunks as possible."""
import numpy as np
import pytest

from helpers import bits_equal_fraction, make_oracle, load_preset

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Renderer():
    from digital_earth_amd.renderer import Renderer as R
    return R


def _apply(r, preset):
    p = load_preset(preset)
    r.set_camera_pos(*p["pos"]); r.set_look_at(*p["look_at"]); r.set_up(*p["up"]); r.set_fov(p["fov"])
    r.set_aspect_scale(p["aspect_scale"]); r.set_sun_angle(p["sun_angle"]); r.set_sun_path_rot(p["sun_path_rot"])


@pytest.mark.legacy_quick
@pytest.mark.parametrize("preset", [None, "config - florida.txt", "config - sunset hurricane.txt", "config - Apollo 11.txt"])
def test_persistent_scheduler_matches_the_oracle(Renderer, lut_arrays, preset):
    W, H, SPP = 96, 48, 5
    r = Renderer((W, H), (0, 1, 0), seed=9, texture_source="synthetic", texture_size=(2048, 1024))
    if preset:
        _apply(r, preset)
    else:
        r.set_fov(0.42)
    r.set_kernel_variant(5)
    r.accumulate(SPP)
    got = r.fetch_hdr()
    o = make_oracle(W, H, lut_arrays, [r.download_texture(s) for s in range(7)], r._params)
    o.accumulate(SPP, 9)
    assert bits_equal_fraction(got, o.fetch_hdr()) == 1.0
    assert got.max() > 0


@pytest.mark.parametrize("shards,pool_log2,claim_k,wpc", [(1, 18, 2, 16), (8, 18, 1, 16), (1, 14, 4, 1), (8, 15, 1, 2), (2, 18, 2, 20)])
def test_persistent_scheduler_settings_do_not_move_bits(Renderer, monkeypatch, shards, pool_log2, claim_k, wpc):
    """Queue shards (1 / one per XCD), a pool so small that every chunk is recycled hundreds of times (2^14 chunks for a launch of
    1.6 M paths on 256 waves: each wave may hold ~24 chunks), chunk claims of 1..4, 4..20 waves per CU: the frame is the state machine's, bit for bit."""
    monkeypatch.setenv("DE_V5_SHARDS", str(shards)); monkeypatch.setenv("DE_V5_POOL_LOG2", str(pool_log2))
    monkeypatch.setenv("DE_V5_CLAIM_K", str(claim_k)); monkeypatch.setenv("DE_V5_WPC", str(wpc))
    monkeypatch.setenv("DE_V5_RESERVE_PCT", "50")
    W, H, SPP = 640, 360, 12
    kw = dict(seed=5, texture_source="synthetic", texture_size=(4096, 2048))
    out = {}
    for variant in (2, 5):
        r = Renderer((W, H), (0, 1, 0), **kw)
        r.set_kernel_variant(variant)
        r.accumulate(SPP - 5); r.accumulate(5)
        out[variant] = r.fetch_hdr()
    assert bits_equal_fraction(out[2], out[5]) == 1.0
    assert out[5].max() > 0


def test_persistent_scheduler_variants_and_modes(Renderer):
    """Variant 5 against variant 2 with everything that changes the launch: progressive calls, a reset in between, the CLAMP sampler,
    the fixed-wavelength constant-map configuration (BASELINE cfg1), a tile partition."""
    from digital_earth_amd._native import DE_FLAG_CLAMP_SAMPLER
    kw = dict(seed=4, texture_source="synthetic", texture_size=(1024, 512))
    a = Renderer((128, 64), (0, 1, 0), **kw); a.set_fov(0.45)
    b = Renderer((128, 64), (0, 1, 0), **kw); b.set_fov(0.45); b.set_kernel_variant(5)
    for r in (a, b):
        r.accumulate(2); r.accumulate(1); r.accumulate(4)
    assert bits_equal_fraction(a.fetch_hdr(), b.fetch_hdr()) == 1.0
    assert (a.fetch_image() == b.fetch_image()).all()
    for r in (a, b):
        r.reset_framebuffer(); r.set_flag(DE_FLAG_CLAMP_SAMPLER, True); r.accumulate(3)
    assert bits_equal_fraction(a.fetch_hdr(), b.fetch_hdr()) == 1.0
    for r in (a, b):
        r.set_flag(DE_FLAG_CLAMP_SAMPLER, False); r.reset_framebuffer(); r.set_tile_partition(1, 3); r.accumulate(3)
    part = b.fetch_hdr()
    assert bits_equal_fraction(a.fetch_hdr(), part) == 1.0
    from digital_earth_amd import parallel
    assert (part[~parallel.owned_pixel_mask(128, 64, 1, 3)] == 0).all()
    c1 = Renderer((64, 64), (0, 1, 0), texture_source="constant"); c1.set_fixed_wavelength(550.0); c1.set_topo_res_override(21600)
    c5 = Renderer((64, 64), (0, 1, 0), texture_source="constant"); c5.set_fixed_wavelength(550.0); c5.set_topo_res_override(21600)
    c5.set_kernel_variant(5)
    c1.accumulate(4); c5.accumulate(4)
    assert bits_equal_fraction(c1.fetch_hdr(), c5.fetch_hdr()) == 1.0


def test_persistent_scheduler_deep_paths(Renderer):
    """A close look straight down at a cloud-heavy day side: 4.6 vertices per sample on average, paths to the 25-vertex limit — every
    path crosses ~40 chunk hand-overs; Russian roulette and the sc > 9 extinction change happen in whatever wave holds the record."""
    kw = dict(seed=21, texture_source="synthetic", texture_size=(1024, 512), cloud_heavy=True)
    out = []
    for variant in (2, 5):
        r = Renderer((128, 64), (0, 1, 0), **kw)
        r.set_camera_pos(-9.0e6, 0.0, 9.0e6); r.set_fov(0.25)
        r.set_kernel_variant(variant)
        r.accumulate(16)
        out.append(r.fetch_hdr())
    assert bits_equal_fraction(out[0], out[1]) == 1.0
    assert out[1].max() > 0


def test_persistent_scheduler_overlapping_calls(Renderer):
    """24 calls of one sample per pixel overlap on the launch slots (each slot owns a pool, rings and cold records): the sums keep call
    order and equal one call of 24."""
    kw = dict(seed=3, texture_source="synthetic", texture_size=(2048, 1024))
    a = Renderer((256, 128), (0, 1, 0), **kw); a.set_kernel_variant(5)
    b = Renderer((256, 128), (0, 1, 0), **kw); b.set_kernel_variant(5)
    c = Renderer((256, 128), (0, 1, 0), **kw)
    for _ in range(24):
        a.accumulate(1)
    b.accumulate(24); c.accumulate(24)
    fa, fb, fc = a.fetch_hdr(), b.fetch_hdr(), c.fetch_hdr()
    assert bits_equal_fraction(fa, fb) == 1.0 and bits_equal_fraction(fb, fc) == 1.0


def test_lone_small_calls_of_the_earlier_policy_run_the_persistent_scheduler(Renderer, monkeypatch):
    """The automatic variant's policy before render_kernel_v6 (DE_AUTO_V6=0): a small call that finds nothing of its context in flight —
    the reference's window loop, accumulate() then fetch_image() (earth_viewer.py:241-243) — runs render_kernel_v5 (14.8 against 20.4 ms
    per 1920x1080 frame); small calls issued while others are in flight overlap on the state machine.  Same bits either way.  (What the
    default does today: tests/test_gpu_block_scheduler.py.)"""
    monkeypatch.setenv("DE_AUTO_V6", "0")
    kw = dict(seed=6, texture_source="synthetic", texture_size=(4096, 2048))
    W, H = 1024, 512                                              # 524 288 paths per sample: above the 2^18 floor, below the pipeline's 8 M
    a = Renderer((W, H), (0, 1, 0), **kw)                          # the loop: every call is alone
    kinds = []
    for _ in range(5):
        a.accumulate(1)
        kinds.append(a.last_call_info()["variant"])
        a.fetch_image()
    assert kinds == [5] * 5, kinds
    b = Renderer((W, H), (0, 1, 0), **kw)                          # calls back to back: only the first is alone
    kinds = []
    for _ in range(5):
        b.accumulate(1)
        kinds.append(b.last_call_info()["variant"])
    assert kinds[0] == 5 and kinds[1:] == [2] * 4, kinds
    c = Renderer((W, H), (0, 1, 0), **kw); c.set_kernel_variant(2); c.accumulate(5)
    fa, fb, fc = a.fetch_hdr(), b.fetch_hdr(), c.fetch_hdr()
    assert bits_equal_fraction(fa, fc) == 1.0 and bits_equal_fraction(fb, fc) == 1.0 and fc.max() > 0
