"""Legacy cross-checks that lived in the round tests: the per-lane loop kernel (variant 1) against the state machine, and the memory budget of
the wavefront pipeline's queues (the launch policy of rounds 3-4, DE_AUTO_V6=0)."""
import numpy as np
import pytest

from helpers import bits_equal_fraction, load_preset

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Renderer():
    from digital_earth_amd.renderer import Renderer as R
    return R


@pytest.mark.legacy_quick
@pytest.mark.parametrize("preset", [None, "config - sunset hurricane.txt"])
def test_kernel_variants_agree(Renderer, preset):
    """The wave-level state machine (variant 2, default) and the per-lane loop kernel (variant 1) are the same function."""
    r = Renderer((128, 64), (0, 1, 0), texture_source="synthetic", texture_size=(2048, 1024), seed=3)
    if preset:
        p = load_preset(preset)
        r.set_camera_pos(*p["pos"]); r.set_look_at(*p["look_at"]); r.set_up(*p["up"]); r.set_fov(p["fov"])
        r.set_sun_angle(p["sun_angle"]); r.set_sun_path_rot(p["sun_path_rot"])
    else:
        r.set_fov(0.4)
    out = {}
    for variant in (1, 2):
        r.set_kernel_variant(variant)
        r.reset_framebuffer()
        r.enable_counters(True)
        r.accumulate(3)
        cnt = r.counters(); cnt.pop("sched", None)
        out[variant] = (r.fetch_hdr(), cnt, r.debug_samples(1))
    assert (out[1][0].view(np.uint32) == out[2][0].view(np.uint32)).all()
    for k in ("samples", "tracking_steps", "vertices", "rng_draws", "taps_rgb8"):
        assert out[1][1][k] == out[2][1][k], k
    assert out[2][1]["sphere_steps"] <= out[1][1]["sphere_steps"]      # variant 2 stops tracing rays that have escaped
    assert (out[1][2].view(np.uint32) == out[2][2].view(np.uint32)).all()


def test_memory_budget_cuts_the_batches_not_the_bits(Renderer, monkeypatch):
    """de_set_memory_budget on the wavefront pipeline — the automatic variant's choice for large calls until the per-CU
    stage scheduler took over (which holds 37 MB per launch slot whatever the call: tests/test_gpu_block_scheduler.py); DE_AUTO_V6=0
    selects that policy.  1920x1088 x 64 spp = 134 M paths in two batches on two pipes, 512 B per path and pipe plus
    each pipe's room for holes (1.9 GB): 72 GB.  With 60 GB the pipes get 30 GB each: the batches shrink (more launches), the memory
    held stays inside the budget, the frame is the state machine's bit for bit; with a budget that leaves a pipe room for less than a
    worthwhile batch the call runs on the state machine, in one launch."""
    from helpers import bits_equal_fraction
    monkeypatch.setenv("DE_AUTO_V6", "0")
    W, H, SPP = 1920, 1088, 64
    kw = dict(seed=17, texture_source="synthetic", texture_size=(4096, 2048))
    ref = Renderer((W, H), (0, 1, 0), **kw); ref.set_kernel_variant(2); ref.accumulate(SPP)
    want = ref.fetch_hdr()
    r = Renderer((W, H), (0, 1, 0), **kw)
    r.accumulate(SPP); r.synchronize()
    free_launches, free_bytes = r.last_call_info()["launches"], r.memory_use()
    assert r.last_call_info()["variant"] == 3 and free_bytes > 0
    assert bits_equal_fraction(r.fetch_hdr(), want) == 1.0
    budget = 60 << 30
    r.set_memory_budget(budget)
    assert r.memory_use() == 0                                   # setting it releases what was held
    r.reset_framebuffer(); r.accumulate(SPP); r.synchronize()
    info = r.last_call_info()
    assert info["variant"] == 3 and info["launches"] > free_launches, (info, free_launches)
    assert 0 < r.memory_use() <= budget
    assert bits_equal_fraction(r.fetch_hdr(), want) == 1.0
    r.set_memory_budget(8 << 30)                                 # 4 GB per pipe: 4 M paths, not worth a pipeline launch
    r.reset_framebuffer(); r.accumulate(SPP); r.synchronize()
    assert r.last_call_info()["variant"] == 2 and r.last_call_info()["launches"] == 1 and r.memory_use() == 0
    assert bits_equal_fraction(r.fetch_hdr(), want) == 1.0
    r.set_memory_budget(0)
    r.reset_framebuffer(); r.accumulate(SPP); r.synchronize()
    assert r.last_call_info()["launches"] == free_launches

