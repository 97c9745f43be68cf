"""Round 4: BASELINE configs[3] and [4] on the map tier they name, against the oracle at the FULL sample count (until
now they ran with 4096x2048 maps and faced the oracle for 2 spp), and what the first N > 1 run leans on inside one process."""
import numpy as np
import pytest

from helpers import make_oracle, load_preset

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Renderer():
    from digital_earth_amd.renderer import Renderer as R
    return R


def _apply(r, name, crf_by_name=None):
    p = load_preset(name)
    r.set_camera_pos(*p["pos"]); r.set_look_at(*p["look_at"]); r.set_up(*p["up"])
    r.set_fov(p["fov"]); r.set_aspect_scale(p["aspect_scale"]); r.set_exposure(p["exposure"])
    r.set_gamma(p["gamma"]); r.set_sun_angle(p["sun_angle"]); r.set_sun_path_rot(p["sun_path_rot"])
    r.set_crf(p["crf"] if crf_by_name is None else r.crf_names.index(crf_by_name))
    return p


@pytest.mark.timeout(3000)
def test_cfg4_quality2_cloud_heavy_full_sample_count(Renderer, lut_arrays):
    """BASELINE configs[3]: 3840x2160, 128 spp, `config - sunset hurricane.txt`, the quality-2 tier (lib/textures.py:65-79:
    21600x10800 / 16200x8100) with the cloud-heavy stand-in maps.  The oracle renders every 293rd 8x8 tile of the same frame at all
    128 spp (1.06 G paths on the GPU, 3.6 M on the host: the oracle is what this test waits for); those pixels match bit for bit."""
    from digital_earth_amd import parallel
    W, H, SPP = 3840, 2160, 128
    r = Renderer((W, H), (0, 1, 0), texture_source="synthetic", texture_quality=2, cloud_heavy=True, seed=7)
    r.copy_textures()
    _apply(r, "config - sunset hurricane.txt")
    for _ in range(4):
        r.accumulate(32)
    g = r.fetch_hdr()
    assert r.current_spp == SPP and np.isfinite(g).all() and g.max() > 0
    o = make_oracle(W, H, lut_arrays, [r.download_texture(s) for s in range(7)], r._params)
    o.accumulate(SPP, 7, 23, 293)
    mask = parallel.owned_pixel_mask(W, H, 23, 293)
    assert mask.sum() >= 27000
    assert (o.fetch_hdr()[mask].view(np.uint32) == g[mask].view(np.uint32)).all()
    img = r.fetch_image()
    assert img.min() >= 0.0 and img.max() <= 1.0 and img.std() > 0.01


@pytest.mark.timeout(3000)
def test_cfg5_quality2_progressive_with_the_film_curve_by_name(Renderer, lut_arrays):
    """BASELINE configs[4]: 1920x1080, 1024 spp PROGRESSIVE (a display after every 128 samples), `config - Apollo 11.txt`, quality-2
    maps, the camera-response curve selected BY NAME — the preset's index 12 counts os.listdir order on the author's machine (Q10):
    the curve is taken by the name the executed-reference fixtures carry for this preset, not by the index.
    Oracle: every 193rd tile at all 1024 spp, HDR bit for bit, and the displayed pixels through the same film curve."""
    import os
    from digital_earth_amd import parallel
    from test_ref_fixtures import GOLDEN
    W, H, SPP, EVERY = 1920, 1080, 1024, 128
    name = str(np.load(os.path.join(GOLDEN, "ref_q2.npz"))["apollo_crf_name"])
    r = Renderer((W, H), (0, 1, 0), texture_source="synthetic", texture_quality=2, seed=13)
    r.copy_textures()
    p = _apply(r, "config - Apollo 11.txt", crf_by_name=name)
    # (the preset's index 12 is another curve under this repo's pinned order: which one the author saw is not knowable, DESIGN §2 Q10;
    # the fixtures carry the name the executed reference resolved it to in the build container)
    assert name in r.crf_names and name != "Neutral.rf" and r._params.selected_crf == r.crf_names.index(name)
    shown = []
    while r.current_spp < SPP:
        r.accumulate(EVERY)
        shown.append(float(r.fetch_image().mean()))          # the progressive display (renderer.py:382-384)
    g, img = r.fetch_hdr(), r.fetch_image()
    assert len(shown) == SPP // EVERY and np.isfinite(g).all() and g.max() > 0
    assert abs(shown[-1] - shown[-2]) < 0.005 * shown[-1] and max(shown) < 1.05 * min(shown)   # the display converges; it does not brighten with the sample count
    o = make_oracle(W, H, lut_arrays, [r.download_texture(s) for s in range(7)], r._params)
    o.accumulate(SPP, 13, 41, 193)
    mask = parallel.owned_pixel_mask(W, H, 41, 193)
    assert mask.sum() >= 10000
    assert (o.fetch_hdr()[mask].view(np.uint32) == g[mask].view(np.uint32)).all()
    assert np.abs(o.fetch_image()[mask] - img[mask]).max() <= 1e-5
    assert img.min() >= 0.0 and img.max() <= 1.0 and img.std() > 0.01
