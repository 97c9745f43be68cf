"""render_kernel_v6 — path_tracer as ONE persistent kernel whose stage queues live in LDS, one scheduler per workgroup
(csrc/render_kernel_v6.hip; the automatic variant's choice for every call of at least 4096 paths since round 4) — is the same
function as the wave-level state machine, the wavefront pipeline, the HBM-queue scheduler and the oracle, bit for bit: the stage
bodies are shared with render_kernel_v3, every path runs the same arithmetic on its own RNG stream, only where and when changes.
What is new — and what these tests lean on — is (i) the hand-over of 16-bit slot numbers between the waves of a workgroup through
LDS rings (tickets, entries taken when seen), and (ii) SUSPENDED loop stages: a wave left with a few long records writes the
loop's state (t, transmittance, the running point, the step count) to the spare words of the cold record and hands the record
back to its own ring; whoever takes it resumes the loop exactly where it stopped.  The yield / service thresholds below force as
many suspensions, refills and re-entries as possible."""
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


@pytest.mark.parametrize("preset", [None, "config - florida.txt", "config - sunset hurricane.txt", "config - Apollo 11.txt"])
def test_block_scheduler_matches_the_oracle(Renderer, lut_arrays, preset):
    W, H, SPP = 96, 48, 5
    r = Renderer((W, H), (0, 1, 0), seed=9, texture_source="synthetic", texture_size=(2048, 1024))
    if preset:
        _apply(r, preset)
    else:
        r.set_fov(0.42)
    r.set_kernel_variant(6)
    r.accumulate(SPP)
    assert r.last_call_info()["variant"] == 6
    got = r.fetch_hdr()
    o = make_oracle(W, H, lut_arrays, [r.download_texture(s) for s in range(7)], r._params)
    o.accumulate(SPP, 9)
    assert bits_equal_fraction(got, o.fetch_hdr()) == 1.0
    assert got.max() > 0


@pytest.mark.parametrize("yield_max,svc,area,retry,elsewhere,enter_min,flat_again,bands,tail",
                         [(0, 16, 0, 6, 48, 0, 1, 1, (1, 96, 64)), (63, 1, 1, 1, 1, 0, 1, 8, (2, 300, 64)), (63, 48, 5000, 2, 1, 0, 64, 8, (1, 8, 16)), (32, 8, 0, 12, 200, 64, 16, 1, (0, 96, 64)),
                          (56, 63, 400, 1, 16, 32, 32, 8, (2, 96, 64))])
def test_block_scheduler_settings_do_not_move_bits(Renderer, monkeypatch, yield_max, svc, area, retry, elsewhere, enter_min, flat_again, bands, tail):
    """No yields at all; a wave that hands its records back whenever its ring is dry and services after every trip with an idle lane
    (each record is suspended and resumed many times); services by idle lane-trips (the shipped rule: area > 0) at both extremes and by
    idle lanes (round 4's rule: area = 0); further straight-line passes from 1 / 16 / 32 / 64 entries; one work counter or one per XCD; a
    scheduler that leaves shallow rings alone: the frame is the state machine's, bit for bit.  1.4 M paths on 512 workgroups of 1120 slots:
    every slot is reused ~3 times per launch.  (The knobs travel as ONE de_set_tuning call: the Python layer reads these variables, the
    library reads none.)  `tail` = (levels, export threshold, workgroups of the first tail launch): the frame's last paths finished by 0 / 1 / 2
    further launches from pools in HBM (render_kernel_v6.hip, "The tail"), forced on for these small frames; with yields off (first row) the
    records in the loop stages' lanes only come back through the export's own yield."""
    monkeypatch.setenv("DE_V6_TAIL", str(tail[0])); monkeypatch.setenv("DE_V6_TAIL_EXPORT0", str(tail[1])); monkeypatch.setenv("DE_V6_TAIL_GRID0", str(tail[2]))
    monkeypatch.setenv("DE_V6_TAIL_EXPORT1", "40"); monkeypatch.setenv("DE_V6_TAIL_GRID1", "8"); monkeypatch.setenv("DE_V6_TAIL_MIN_PATHS", "0"); monkeypatch.setenv("DE_V6_TAIL_ALONE", "1")      # export also when no launch is queued behind (the default exports only then)
    monkeypatch.setenv("DE_V6_YIELD", str(yield_max)); monkeypatch.setenv("DE_V6_RETRY", str(retry))
    monkeypatch.setenv("DE_V6_ELSEWHERE", str(elsewhere)); monkeypatch.setenv("DE_V6_ENTER_MIN", str(enter_min))
    monkeypatch.setenv("DE_V6_FLAT_AGAIN", str(flat_again)); monkeypatch.setenv("DE_V6_BANDS", str(bands))
    for k in ("ST", "GAS", "CLOUD"):
        monkeypatch.setenv("DE_V6_SVC_" + k, str(svc))
        monkeypatch.setenv("DE_V6_AREA_" + k, str(area))
    W, H, SPP = 640, 360, 12
    kw = dict(seed=5, texture_source="synthetic", texture_size=(4096, 2048))
    out = {}
    for variant in (2, 6):
        r = Renderer((W, H), (0, 1, 0), **kw)
        r.set_kernel_variant(variant)
        r.accumulate(SPP - 5); r.accumulate(5)
        out[variant] = r.fetch_hdr()
    assert bits_equal_fraction(out[2], out[6]) == 1.0
    assert out[6].max() > 0


def test_block_scheduler_variants_and_modes(Renderer):
    """Variant 6 against variant 2 with everything that changes the launch: progressive calls, a reset in between, the CLAMP sampler,
    the fixed-wavelength constant-map configuration (BASELINE cfg1), a tile partition."""
    from digital_earth_amd._native import DE_FLAG_CLAMP_SAMPLER
    kw = dict(seed=4, texture_source="synthetic", texture_size=(1024, 512))
    a = Renderer((128, 64), (0, 1, 0), **kw); a.set_fov(0.45); a.set_kernel_variant(2)
    b = Renderer((128, 64), (0, 1, 0), **kw); b.set_fov(0.45); b.set_kernel_variant(6)
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
    c1 = Renderer((64, 64), (0, 1, 0), texture_source="constant"); c1.set_fixed_wavelength(550.0); c1.set_topo_res_override(21600); c1.set_kernel_variant(2)
    c6 = Renderer((64, 64), (0, 1, 0), texture_source="constant"); c6.set_fixed_wavelength(550.0); c6.set_topo_res_override(21600); c6.set_kernel_variant(6)
    c1.accumulate(4); c6.accumulate(4)
    assert bits_equal_fraction(c1.fetch_hdr(), c6.fetch_hdr()) == 1.0


def test_block_scheduler_deep_paths(Renderer, monkeypatch):
    """A close look straight down at a cloud-heavy day side: 4.6 vertices per sample on average, paths to the 25-vertex limit; with
    the yield threshold at 63 every cloud segment is suspended and resumed several times in whatever wave takes the record, the
    Russian roulette and the sc > 9 extinction change included."""
    monkeypatch.setenv("DE_V6_YIELD", "63"); monkeypatch.setenv("DE_V6_ELSEWHERE", "1"); monkeypatch.setenv("DE_V6_RETRY", "1")
    kw = dict(seed=21, texture_source="synthetic", texture_size=(1024, 512), cloud_heavy=True)
    out = []
    for variant in (2, 6):
        r = Renderer((128, 64), (0, 1, 0), **kw)
        r.set_camera_pos(-9.0e6, 0.0, 9.0e6); r.set_fov(0.25)
        r.set_kernel_variant(variant)
        r.accumulate(16)
        out.append(r.fetch_hdr())
    assert bits_equal_fraction(out[0], out[1]) == 1.0
    assert out[1].max() > 0


def test_block_scheduler_overlapping_calls(Renderer):
    """24 calls of one sample per pixel on the launch slots (each slot owns its control words and cold records): the sums keep call
    order and equal one call of 24, on this kernel and on the state machine."""
    kw = dict(seed=3, texture_source="synthetic", texture_size=(2048, 1024))
    a = Renderer((256, 128), (0, 1, 0), **kw); a.set_kernel_variant(6)
    b = Renderer((256, 128), (0, 1, 0), **kw); b.set_kernel_variant(6)
    c = Renderer((256, 128), (0, 1, 0), **kw); c.set_kernel_variant(2)
    for _ in range(24):
        a.accumulate(1)
    b.accumulate(24); c.accumulate(24)
    fa, fb, fc = a.fetch_hdr(), b.fetch_hdr(), c.fetch_hdr()
    assert bits_equal_fraction(fa, fb) == 1.0 and bits_equal_fraction(fb, fc) == 1.0


def test_the_default_variant_chooses_by_call_size(Renderer, monkeypatch):
    """The automatic variant (round 4): calls of at least 4096 paths run render_kernel_v6, in the reference's window loop (accumulate()
    then fetch_image(), earth_viewer.py:241-243) as well as back to back (tools/small_calls.py: it ends a call sooner than the state
    machine from 4096 paths up); smaller calls run the state machine (the earlier policy — lone small calls on render_kernel_v5 — lives in the legacy library:
    tests/legacy/test_gpu_persistent.py).  Same bits every way."""
    kw = dict(seed=6, texture_source="synthetic", texture_size=(4096, 2048))
    W, H = 1024, 512                                              # 524 288 paths per sample
    a = Renderer((W, H), (0, 1, 0), **kw)
    kinds = []
    for _ in range(3):
        a.accumulate(1); kinds.append(a.last_call_info()["variant"]); a.fetch_image()
    for _ in range(2):
        a.accumulate(1); kinds.append(a.last_call_info()["variant"])
    assert kinds == [6] * 5, kinds
    assert a.memory_use() < (1 << 30)                              # 37 MB of cold records per launch slot, no queues
    s = Renderer((32, 32), (0, 1, 0), **kw); s.accumulate(2)      # 2048 paths: the state machine
    assert s.last_call_info()["variant"] == 2
    m = Renderer((64, 64), (0, 1, 0), **kw); m.accumulate(1)      # 4096 paths: the scheduler
    assert m.last_call_info()["variant"] == 6
    c = Renderer((W, H), (0, 1, 0), **kw); c.set_kernel_variant(2); c.accumulate(5)
    fa, fc = a.fetch_hdr(), c.fetch_hdr()
    assert bits_equal_fraction(fa, fc) == 1.0 and fc.max() > 0


def test_block_scheduler_full_frame(Renderer):
    """BASELINE configs[1] at full size (1920x1080, 16 of its 64 spp to keep the test short): 33 M paths through 512 workgroups, every
    slot reused ~70 times; against the state machine, bit for bit."""
    kw = dict(seed=2, texture_source="synthetic")
    out = []
    for variant in (2, 6):
        r = Renderer((1920, 1080), (0, 1, 0), **kw)
        r.copy_textures(); r.trim_textures()
        r.set_kernel_variant(variant)
        r.accumulate(16)
        out.append(r.fetch_hdr())
    assert bits_equal_fraction(out[0], out[1]) == 1.0 and out[1].max() > 0


def test_default_variant_on_odd_shapes_call_patterns_and_partitions(Renderer):
    """The automatic variant against the state machine on shapes that stress the work-item mapping (one tile row / column, 1936x1088,
    a single 16x8 frame below the 4096-path threshold), on call patterns that mix both kernels in one accumulation, on 130 spp in one
    call, and as the sum of the ranks of 3-, 5- and 7-way tile partitions: bit for bit.  (tools/edge_sizes.py prints the same.)"""
    kw = dict(seed=31, texture_source="synthetic", texture_size=(2048, 1024))
    for (W, H, calls) in ((16, 8, [1]), (80, 40, [3, 1]), (208, 104, [7]), (1008, 8, [2, 2, 2]), (16, 1000, [5]), (1936, 1088, [2]), (640, 360, [1] * 9), (320, 200, [130])):
        frames = []
        for variant in (2, 4):
            r = Renderer((W, H), (0, 1, 0), **kw)
            r.set_kernel_variant(variant)
            for s in calls:
                r.accumulate(s)
            frames.append(r.fetch_hdr())
        assert bits_equal_fraction(frames[0], frames[1]) == 1.0, (W, H, calls)
        assert frames[1].max() > 0
    for world in (3, 5, 7):
        full = Renderer((256, 128), (0, 1, 0), **kw); full.set_kernel_variant(2); full.accumulate(6)
        acc = np.zeros_like(full.fetch_hdr())
        for rank in range(world):
            r = Renderer((256, 128), (0, 1, 0), **kw)
            r.set_tile_partition(rank, world); r.accumulate(6)
            assert r.last_call_info()["variant"] == 6
            acc += r.fetch_hdr()
        assert bits_equal_fraction(acc, full.fetch_hdr()) == 1.0, world


def test_fetch_image_view_is_the_image(Renderer):
    """fetch_image(copy=False) — de_fetch_image_view: the display image in the library's pinned staging buffer, without the last host copy —
    holds the same floats as fetch_image(), is read-only, and follows the accumulation."""
    r = Renderer((256, 128), (0, 1, 0), seed=8, texture_source="synthetic", texture_size=(2048, 1024))
    r.accumulate(3)
    a = r.fetch_image()
    v = r.fetch_image(copy=False)
    assert v.shape == a.shape == (256, 128, 3) and v.dtype == np.float32 and not v.flags.writeable
    assert (np.asarray(v) == a).all() and a.max() > 0
    r.accumulate(2)
    b = r.fetch_image()
    v2 = r.fetch_image(copy=False)
    assert (np.asarray(v2) == b).all() and not (b == a).all()
