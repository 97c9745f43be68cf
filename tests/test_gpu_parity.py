"""GPU parity: libdigitalearth_hip.so (through the C ABI) against the CPU oracle on identical inputs.

Bar (BASELINE.json north_star): per-pixel radiance within 1e-3 relative L2 at matched seed.  Because both sides
implement the same arithmetic contract, the tests additionally report — and for the leaf math require — exact
bit equality.
"""
import numpy as np
import pytest

from helpers import rel_l2, bits_equal_fraction, make_oracle, copy_params, load_preset
from oracle import oracle_binding as ob

pytestmark = pytest.mark.gpu

TOL = 1e-3   # relative L2 of HDR radiance, BASELINE.json north_star


@pytest.fixture(scope="module")
def Renderer():
    from digital_earth_amd.renderer import Renderer as R
    return R


def _inputs(fn, n, rng):
    if fn == "exp":
        return rng.uniform(-90, 90, n), None
    if fn == "log":
        return np.exp(rng.uniform(-85, 85, n)), None
    if fn in ("sin", "cos"):
        return rng.uniform(-20, 20, n), None
    if fn == "atan2":
        return rng.normal(size=n), rng.normal(size=n)
    if fn == "asin":
        return rng.uniform(-1.001, 1.001, n), None
    if fn == "pow":
        return rng.uniform(0, 100, n), rng.uniform(-3, 3, n)
    if fn == "div":
        return rng.normal(size=n) * 1e3, rng.normal(size=n)
    return rng.uniform(0, 1e14, n), None   # sqrt


@pytest.mark.parametrize("fid,fn", list(enumerate(["exp", "log", "sin", "cos", "atan2", "asin", "pow", "div", "sqrt"])))
def test_math_bit_exact(Renderer, fid, fn):
    r = Renderer((16, 8), (0, 1, 0), texture_source="constant")
    rng = np.random.default_rng(1234 + fid)
    a, b = _inputs(fn, 1 << 20, rng)
    a = a.astype(np.float32)
    specials = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-40, 88.5, -87.5, 0.5, 2.0 ** -24], np.float32)
    a[:specials.size] = specials
    if b is not None:
        b = b.astype(np.float32)
        b[:specials.size] = specials[::-1]
    if fn == "atan2":     # stated domain of the contract-2 atan2: zero or 2^-60 <= |v| <= 2^60 (components of a unit vector)
        for v in (a, b):
            v[~np.isfinite(v)] = 3.0
            v[(np.abs(v) < 2.0 ** -60) & (v != 0)] = 2.0 ** -59
    got = r.debug_math(fid, a, b)
    x = a[:, None] if b is None else np.stack([a, b], 1)
    want = ob.evaluate(fn, x)[:, 0]
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    assert same.all(), "%s: %d of %d differ, first at a=%r" % (fn, (~same).sum(), same.size, a[~same][:4])


def test_rgb_unorm8_decode_exact(Renderer):
    # (float)b / 255.0f on the device must be the correctly rounded quotient for all 256 bytes
    r = Renderer((16, 8), (0, 1, 0), texture_source="constant")
    b = np.arange(256, dtype=np.float32)
    got = r.debug_math(7, b, np.full(256, 255.0, np.float32))
    assert (got == (b / np.float32(255.0))).all()


def _render_pair(Renderer, lut_arrays, W, H, spp, configure, texture_kw, seed=0):
    r = Renderer((W, H), (0, 1, 0), seed=seed, **texture_kw)
    configure(r)
    r.copy_textures()
    texels = [r.download_texture(s) for s in range(7)]
    o = make_oracle(W, H, lut_arrays, texels, r._params)
    return r, o


def _compare(r, o, spp, seed=0, what="hdr"):
    r.reset_framebuffer()
    o.reset()
    r.accumulate(spp)
    o.accumulate(spp, seed)
    g, c = r.fetch_hdr(), o.fetch_hdr()
    return g, c, rel_l2(g, c), bits_equal_fraction(g, c)


def test_cfg1_flat_earth_single_wavelength(Renderer, lut_arrays):
    """BASELINE configs[0]: 256x256, 4 spp, single wavelength, flat-albedo Earth, no textures."""
    def configure(r):
        r.set_fixed_wavelength(550.0)
        r.set_topo_res_override(21600)
    r, o = _render_pair(Renderer, lut_arrays, 256, 256, 4, configure, dict(texture_source="constant"))
    # per-sample trace first: radiance, wavelength, draw count and vertex count of sample 0 of every pixel
    dg, dc = r.debug_samples(0), o.debug_samples(0, 0)
    assert (dg[..., 1] == dc[..., 1]).all()
    assert (dg[..., 2] == dc[..., 2]).mean() > 0.999, "RNG draw counts diverge"
    g, c, err, same = _compare(r, o, 4)
    print("cfg1: rel L2 %.3e, bit-identical pixels %.6f" % (err, same))
    assert np.isfinite(g).all()
    assert err <= TOL
    assert same > 0.999
    # display path on the same accumulation
    assert np.abs(r.fetch_image() - o.fetch_image()).max() <= 1e-5


@pytest.mark.parametrize("preset,size", [(None, (128, 128)), ("config - florida.txt", (128, 72)),
                                         ("config - sunset hurricane.txt", (128, 72)), ("config - Apollo 11.txt", (128, 72))])
def test_textured_spectral_parity(Renderer, lut_arrays, preset, size):
    """Full spectral Rayleigh+Mie+O3+clouds with synthetic maps (4K tier), default camera and the three presets."""
    W, H = size
    def configure(r):
        if preset:
            p = load_preset(preset)
            r.set_camera_pos(*p["pos"]); r.set_look_at(*p["look_at"]); r.set_up(*p["up"])
            r.set_fov(p["fov"]); r.set_aspect_scale(p["aspect_scale"]); r.set_exposure(p["exposure"])
            r.set_crf(p["crf"]); r.set_gamma(p["gamma"]); r.set_sun_angle(p["sun_angle"]); r.set_sun_path_rot(p["sun_path_rot"])
    r, o = _render_pair(Renderer, lut_arrays, W, H, 2, configure, dict(texture_source="synthetic", texture_size=(2048, 1024)))
    g, c, err, same = _compare(r, o, 2)
    print("%s: rel L2 %.3e, bit-identical %.6f" % (preset, err, same))
    assert np.isfinite(g).all()
    assert err <= TOL
    assert same > 0.999
    assert np.abs(r.fetch_image() - o.fetch_image()).max() <= 1e-5


def test_clamp_sampler_and_ray_marcher(Renderer, lut_arrays):
    from digital_earth_amd import _native
    for flag, name in ((_native.DE_FLAG_CLAMP_SAMPLER, "clamp"), (_native.DE_FLAG_RAY_MARCHER, "ray_marcher")):
        def configure(r):
            r.set_flag(flag, True)
            r.set_fov(0.45)
        r, o = _render_pair(Renderer, lut_arrays, 64, 64, 1, configure, dict(texture_source="synthetic", texture_size=(1024, 512)))
        g, c, err, same = _compare(r, o, 1)
        print("%s: rel L2 %.3e, bit-identical %.6f" % (name, err, same))
        assert err <= TOL and same > 0.999


def test_synthetic_maps_match_cpu_statement(Renderer):
    """GPU generator vs the oracle's CPU statement of the same specification, byte for byte."""
    r = Renderer((16, 8), (0, 1, 0), texture_source="synthetic", texture_size=(768, 384))
    r.copy_textures()
    o = ob.Oracle(16, 8)
    for s in range(7):
        o.generate_texture(s, 768, 384, 20240127, 0)
        g = r.download_texture(s)
        c = o.download_texture(s, 768, 384, g.shape[2])
        assert (g == c).all(), "slot %d: %d bytes differ" % (s, (g != c).sum())


def test_display_transform_random_hdr(Renderer, lut_arrays):
    """_render_to_image is deterministic: random HDR in, every CRF, gamma != 1 (<= 1e-5 abs)."""
    W, H = 64, 32
    r = Renderer((W, H), (0, 1, 0), texture_source="constant")
    r.copy_textures()
    rng = np.random.default_rng(7)
    hdr = (rng.lognormal(-1.0, 2.0, size=(W, H, 3)) * rng.uniform(0, 1, size=(W, H, 1))).astype(np.float32)
    hdr[0, 0] = 0.0
    hdr[1, 1] = [-0.5, 2.0, 1e-6]
    for crf, gamma, exposure in ((0, 1.0, 2.5), (5, 0.8, 0.0), (12, 1.001, 2.432), (15, 2.2, -1.0)):
        r.set_crf(crf); r.set_gamma(gamma); r.set_exposure(exposure)
        r.upload_hdr(hdr, 16)
        o = make_oracle(W, H, lut_arrays, None, r._params)
        o.set_hdr(hdr, 16)
        g, c = r.fetch_image(), o.fetch_image()
        assert np.abs(g - c).max() <= 1e-5, (crf, gamma)
        assert g.min() >= 0.0 and g.max() <= 1.0


def test_tile_partition_is_bit_exact(Renderer, lut_arrays):
    """1-GPU image == N-rank tiled image, bit for bit: RNG keyed on absolute pixel and sample index.
    Virtual ranks run one after the other on the one device and write disjoint tiles."""
    W, H = 128, 64
    r = Renderer((W, H), (0, 1, 0), texture_source="synthetic", texture_size=(1024, 512))
    r.set_fov(0.4)
    r.accumulate(3)
    full = r.fetch_hdr()
    for world in (2, 8):
        r.reset_framebuffer()
        for rank in range(world):
            r.set_tile_partition(rank, world)
            r.set_current_spp(0)
            r.accumulate(3)
        r.set_tile_partition(0, 1)
        tiled = r.fetch_hdr()
        assert (tiled.view(np.uint32) == full.view(np.uint32)).all(), world


def test_progressive_accumulation_matches_single_launch(Renderer):
    """accumulate() x 4 == accumulate(4): the per-pixel association ((c+s0)+s1)+... does not depend on launch shape."""
    r = Renderer((64, 64), (0, 1, 0), texture_source="synthetic", texture_size=(1024, 512))
    r.set_fov(0.4)
    r.accumulate(4)
    a = r.fetch_hdr()
    r.reset_framebuffer()
    for _ in range(4):
        r.accumulate()
    b = r.fetch_hdr()
    assert r.current_spp == 4
    assert (a.view(np.uint32) == b.view(np.uint32)).all()


def test_counters_match_oracle(Renderer, lut_arrays):
    def configure(r):
        r.set_fov(0.4)
    r, o = _render_pair(Renderer, lut_arrays, 64, 64, 1, configure, dict(texture_source="synthetic", texture_size=(1024, 512)))
    r.enable_counters(True)
    r.reset_framebuffer(); o.reset()
    r.accumulate(1); o.accumulate(1, 0)
    g, c = r.counters(), o.counters()
    print(g, c)
    for k in ("samples", "tracking_steps", "vertices", "rng_draws", "taps_rgb8"):
        assert g[k] == c[k], k
    # the GPU skips the cloud taps the reference multiplies by zero and the sphere-trace steps of escaped rays
    assert g["taps_r8"] <= c["taps_r8"] and g["sphere_steps"] <= c["sphere_steps"]


def test_errors_are_reported(Renderer):
    from digital_earth_amd._native import DigitalEarthError
    with pytest.raises(DigitalEarthError):
        Renderer((100, 50), (0, 1, 0), texture_source="constant")   # not a multiple of (16, 8), renderer.py:46



def test_fast_sqrt_is_correctly_rounded(Renderer):
    """de_sqrt_nr (v_rsq_f32 + one coupled Newton step) == IEEE sqrt for EVERY significand at both exponent parities
    (all 128 exponents of its stated range: tools/exhaustive_math.py)."""
    r = Renderer((16, 8), (0, 1, 0), texture_source="constant")
    m = np.arange(1 << 23, dtype=np.uint32)
    for expo in (127, 128, 172, 173, 97, 110, 111, 64, 65, 190, 191):   # x in [1,2), [2,4), ~4e13 (|pos|^2), ~1e-9, asin's (1-|x|)/2 range, the ends of the stated range
        x = ((np.uint32(expo) << np.uint32(23)) | m).view(np.float32)
        got = r.debug_math(9, x)
        want = np.sqrt(x.astype(np.float64)).astype(np.float32)     # correctly rounded: f64 sqrt of an f32 rounds once more, exact for 24-bit inputs
        assert (got.view(np.uint32) == want.view(np.uint32)).all(), expo
    assert r.debug_math(9, np.array([0.0], np.float32))[0] == 0.0
    assert np.isnan(r.debug_math(9, np.array([-1.0, np.nan], np.float32))).all()


def test_fast_reciprocal_is_correctly_rounded(Renderer):
    """de_rcp_nr (v_rcp_f32 + one Newton step) == 1 / x (IEEE) for EVERY significand, at several exponents."""
    r = Renderer((16, 8), (0, 1, 0), texture_source="constant")
    m = np.arange(1 << 23, dtype=np.uint32)
    one = np.ones(1 << 23, np.float32)
    for expo in (127, 126, 149, 150, 100, 180):       # [1,2), [0.5,1), |position| ~ 6.4e6 m, small, large
        x = ((np.uint32(expo) << np.uint32(23)) | m).view(np.float32)
        got = r.debug_math(13, x)
        want = r.debug_math(7, one, x)
        assert (got.view(np.uint32) == want.view(np.uint32)).all(), expo
        assert (want == (1.0 / x.astype(np.float64)).astype(np.float32)).all()


def test_exp_of_a_non_positive_argument(Renderer):
    """de_exp_nonpos == de_exp on arguments <= 0 (dense sample down to underflow, -0, -inf) and on NaN."""
    r = Renderer((16, 8), (0, 1, 0), texture_source="constant")
    rng = np.random.default_rng(3)
    x = np.concatenate([-np.abs(rng.standard_normal(1 << 22)).astype(np.float32) * np.float32(40.0),
                        -np.linspace(0.0, 120.0, 1 << 20, dtype=np.float32),
                        np.array([0.0, -0.0, -np.inf, np.nan, -87.0, -87.00001, -1e-30], np.float32)])
    a, b = r.debug_math(14, x), r.debug_math(0, x)
    same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    assert same.all()


def test_log_of_a_random_draw(Renderer):
    """de_log_unit (contract 2: integer mantissa reduction, 6-coefficient polynomial) against the oracle's independent
    statement on EVERY value rng_next can return (k * 2^-24, k = 0 .. 2^24 - 1), and against the f64 logarithm."""
    r = Renderer((16, 8), (0, 1, 0), texture_source="constant")
    x = (np.arange(1 << 24, dtype=np.float64) * 2.0 ** -24).astype(np.float32)
    a = r.debug_math(12, x)
    b = ob.evaluate("log_unit", x)[:, 0]
    assert (a.view(np.uint32) == b.view(np.uint32)).all()
    assert np.isneginf(a[0]) and a[1] < -16.6
    assert np.max(np.abs(a[1:].astype(np.float64) / np.log(x[1:].astype(np.float64)) - 1)) < 6e-7


def test_three_operation_division_by_literals(Renderer):
    """de_div_rc_ng(x, c) == x / c for every significand of x, for the literal divisors used by the kernels
    (the device-side twin of tools/verify_exact_div.c)."""
    r = Renderer((16, 8), (0, 1, 0), texture_source="constant")
    m = np.arange(1 << 23, dtype=np.uint32)
    x = ((np.uint32(127) << np.uint32(23)) | m).view(np.float32)
    for c in (255.0, np.float32(np.pi), 49.0, np.float32(532307548.4168), np.float32(1.225), np.float32(8136.646), 6000.0,
              np.float32(0.1) * np.float32(0.029), np.float32(0.02) * np.float32(0.029)):
        cc = np.full_like(x, np.float32(c))
        assert (r.debug_math(10, x, cc).view(np.uint32) == r.debug_math(7, x, cc).view(np.uint32)).all(), c


def test_real_texture_files_and_orientation(Renderer, lut_arrays, tmp_path):
    """`texture_source="files"`: PNG/JPG maps named as lib/textures.py expects, read with Taichi's imread orientation
    ([x][y], y up: image row 0 is the NORTH edge = last texel row)."""
    from PIL import Image
    from digital_earth_amd import textures as T
    rng = np.random.default_rng(5)
    table = T.texture_table(0)
    expect = {}
    for slot, (name, _) in table.items():
        ch = T.SLOT_CHANNELS[slot]
        img = rng.integers(0, 256, size=(32, 64, 3), dtype=np.uint8)        # image rows top-down
        if ch == 1:
            img[..., 1] = img[..., 0]; img[..., 2] = img[..., 0]
        if name.endswith(".jpg"):
            name_out = name                                                    # lossy: only check it loads
        Image.fromarray(img).save(str(tmp_path / name))
        expect[slot] = img[::-1, :, :ch]
    with pytest.raises(FileNotFoundError):
        Renderer((64, 32), (0, 1, 0), texture_source="files", texture_dir=str(tmp_path / "missing"), texture_quality=0)
    r = Renderer((64, 32), (0, 1, 0), texture_source="files", texture_dir=str(tmp_path), texture_quality=0)
    r.set_fov(0.45)
    r.copy_textures()
    texels = [r.download_texture(s) for s in range(7)]
    for slot in range(7):
        if not table[slot][0].endswith(".jpg"):
            assert (texels[slot] == expect[slot]).all(), slot
    o = make_oracle(64, 32, lut_arrays, texels, r._params)
    r.accumulate(2); o.accumulate(2, 0)
    assert (r.fetch_hdr().view(np.uint32) == o.fetch_hdr().view(np.uint32)).all()


def test_resume_from_uploaded_accumulation(Renderer):
    """Progressive state = (HDR sums, sample count): a render resumed from a saved state continues bit-identically."""
    kw = dict(texture_source="synthetic", texture_size=(1024, 512))
    a = Renderer((64, 64), (0, 1, 0), **kw); a.set_fov(0.4)
    a.accumulate(5)
    full = a.fetch_hdr()
    b = Renderer((64, 64), (0, 1, 0), **kw); b.set_fov(0.4)
    b.accumulate(2)
    saved, spp = b.fetch_hdr(), b.current_spp
    c = Renderer((64, 64), (0, 1, 0), **kw); c.set_fov(0.4)
    c.upload_hdr(saved, spp)
    c.accumulate(3)
    assert c.current_spp == 5
    assert (c.fetch_hdr().view(np.uint32) == full.view(np.uint32)).all()


def test_headless_earth_viewer(tmp_path):
    from digital_earth_amd.earth_viewer import EarthViewer
    v = EarthViewer(config="config - florida.txt", screen_res=(128, 72), texture_source="synthetic", texture_size=(1024, 512))
    img = v.render(spp=4)
    assert img.shape == (128, 72, 3) and img.dtype == np.float32 and 0.0 <= img.min() and img.max() <= 1.0 and img.mean() > 0.01
    out = tmp_path / "florida.png"
    v.save(str(out))
    assert out.stat().st_size > 1000
    assert v.renderer.current_spp == 4 and v.renderer.crf_names[0] == "Neutral.rf"


def test_reciprocal_product_matches_the_oracles_statement(Renderer):
    """Contract 2 writes a quotient by a shared divisor as a * RN(1/b): the device's 3-operation reciprocal times a, against
    numpy's IEEE statement of the same two roundings, on 2^27 operand pairs incl. the normalisation pattern."""
    r = Renderer((16, 8), (0, 1, 0), texture_source="constant")
    rng = np.random.default_rng(99)
    for rep in range(8):
        n = 1 << 24
        a = (rng.standard_normal(n) * 10.0 ** rng.uniform(-6, 8, n)).astype(np.float32)
        b = (np.abs(rng.standard_normal(n)) * 10.0 ** rng.uniform(-6, 8, n) + 1e-12).astype(np.float32)
        if rep % 2:     # the real use: |component| <= length, length ~ 6.4e6
            b = (6.371e6 + rng.uniform(0, 1.2e5, n)).astype(np.float32)
            a = (rng.uniform(-1, 1, n) * b).astype(np.float32)
        want = a * (np.float32(1.0) / b)
        assert (r.debug_math(11, a, b).view(np.uint32) == want.view(np.uint32)).all(), rep


def test_full_size_frame_properties(Renderer, lut_arrays):
    """BASELINE.json's full frame size (1920x1080): properties that do not need the oracle to render the whole frame.
    (a) the oracle renders a sparse subset of 8x8 tiles (rank 7 of 61 in the tile partition) of the SAME full frame and
        those pixels must match the GPU frame bit for bit;  (b) the 8-rank tiled frame equals the 1-rank frame;
    (c) every pixel is rendered by exactly one rank (sum of the rank frames == frame, x + 0 exact)."""
    from digital_earth_amd import parallel
    W, H, SPP = 1920, 1080, 2
    r = Renderer((W, H), (0, 1, 0), texture_source="synthetic", texture_size=(4096, 2048), seed=11)
    r.copy_textures()
    r.accumulate(SPP)
    full = r.fetch_hdr()
    assert np.isfinite(full).all() and full.max() > 0
    # (a)
    texels = [r.download_texture(s) for s in range(7)]
    o = make_oracle(W, H, lut_arrays, texels, r._params)
    o.accumulate(SPP, 11, 7, 61)
    part = o.fetch_hdr()
    mask = parallel.owned_pixel_mask(W, H, 7, 61)
    assert mask.sum() > 30000
    assert (part[~mask] == 0).all()
    assert (part[mask].view(np.uint32) == full[mask].view(np.uint32)).all()
    # (b), (c)
    total = np.zeros_like(full)
    for rank in range(8):
        r.reset_framebuffer()
        r.set_tile_partition(rank, 8)
        r.accumulate(SPP)
        f = r.fetch_hdr()
        m = parallel.owned_pixel_mask(W, H, rank, 8)
        assert (f[~m] == 0).all()
        total += f
    assert (total.view(np.uint32) == full.view(np.uint32)).all()


def test_edge_cases_of_accumulate(Renderer):
    from digital_earth_amd._native import DigitalEarthError
    r = Renderer((256, 256), (0, 1, 0), texture_source="constant")
    r.accumulate(0)                                   # nothing to do: state unchanged
    assert r.current_spp == 0 and (r.fetch_hdr() == 0).all()
    with pytest.raises(DigitalEarthError):
        r._lib.de_accumulate  # noqa: B018  (large requests are split inside de_accumulate: tests/test_gpu_round2.py)
        check_rc = r._lib.de_accumulate(r._h, -1, 0, 0, 1)
        from digital_earth_amd._native import check
        check(check_rc)
    with pytest.raises(ValueError):
        r.set_tile_partition(3, 2)
    r.set_tile_partition(1, 2)
    r.accumulate(1)
    h = r.fetch_hdr()
    assert (h[0:8, 0:8] == 0).all() and (h[8:16, 0:8] != 0).any()      # tile (0,0) belongs to rank 0, tile (1,0) to rank 1


def test_degenerate_camera_terminates(Renderer, lut_arrays):
    """look_at == camera_pos makes every ray direction NaN (normalized() of a zero vector, renderer.py:272).  The reference's
    loops all end on NaN comparisons and the NaN guard zeroes the radiance (pathtracer.py:466-467); the kernel must
    terminate too, and agree with the oracle."""
    r = Renderer((64, 32), (0, 1, 0), texture_source="synthetic", texture_size=(512, 256))
    r.set_look_at(*[float(x) for x in r.camera_pos[None]])
    r.copy_textures()
    o = make_oracle(64, 32, lut_arrays, [r.download_texture(s) for s in range(7)], r._params)
    r.accumulate(1); o.accumulate(1, 0)
    g, c = r.fetch_hdr(), o.fetch_hdr()
    assert (g == 0).all() and (c == 0).all()


def test_agx_display_transform(Renderer, lut_arrays):
    """DE_FLAG_AGX: srgb_transfer(agx.display_transform(linear)) (renderer.py:356, lib/AgX.py) vs the oracle."""
    W, H = 64, 32
    r = Renderer((W, H), (0, 1, 0), texture_source="constant")
    r.copy_textures()
    r.set_display_transform("agx")
    rng = np.random.default_rng(17)
    hdr = (rng.lognormal(-1.0, 2.0, size=(W, H, 3)) * rng.uniform(0, 1, size=(W, H, 1))).astype(np.float32)
    hdr[0, 0] = 0.0; hdr[1, 1] = [-0.5, 2.0, 1e-6]; hdr[2, 2] = 0.18 * 4 / 2 ** 2.5
    r.upload_hdr(hdr, 4)
    o = make_oracle(W, H, lut_arrays, None, r._params)
    o.set_hdr(hdr, 4)
    g, c = r.fetch_image(), o.fetch_image()
    assert np.isfinite(g).all()
    assert np.abs(g - c).max() <= 1e-5
    # middle grey at the centre of the vignette-free pixel maps to AgX's y_pivot 0.5 -> sRGB 0.7354
    r.vignette_strength = 0.0
    o2 = make_oracle(W, H, lut_arrays, None, r._params); o2.set_hdr(hdr, 4)
    g2 = r.fetch_image()
    assert np.abs(g2 - o2.fetch_image()).max() <= 1e-5
    assert abs(g2[2, 2, 1] - 0.7354) < 2e-3
    r.set_display_transform("opendrt")
    assert np.abs(r.fetch_image() - g2).max() > 1e-3


def test_checkpoint_file_round_trip(Renderer, tmp_path):
    kw = dict(texture_source="synthetic", texture_size=(1024, 512))
    a = Renderer((64, 64), (0, 1, 0), seed=21, **kw); a.set_fov(0.4); a.set_exposure(1.5)
    a.accumulate(3)
    a.save_checkpoint(str(tmp_path / "state.npz"))
    a.accumulate(2)
    want = a.fetch_hdr()
    b = Renderer((64, 64), (0, 1, 0), **kw)
    b.load_checkpoint(str(tmp_path / "state.npz"))
    assert b.current_spp == 3 and b.seed == 21 and abs(b.exposure[None] - 1.5) < 1e-7
    b.accumulate(2)
    assert (b.fetch_hdr().view(np.uint32) == want.view(np.uint32)).all()
    with pytest.raises(ValueError):
        Renderer((32, 32), (0, 1, 0), **kw).load_checkpoint(str(tmp_path / "state.npz"))


def _apply_preset(r, name):
    p = load_preset(name)
    r.set_camera_pos(*p["pos"]); r.set_look_at(*p["look_at"]); r.set_up(*p["up"])
    r.set_fov(p["fov"]); r.set_aspect_scale(p["aspect_scale"]); r.set_exposure(p["exposure"])
    r.set_crf(p["crf"]); r.set_gamma(p["gamma"]); r.set_sun_angle(p["sun_angle"]); r.set_sun_path_rot(p["sun_path_rot"])


@pytest.mark.parametrize("cfg", ["cfg3", "cfg4", "cfg5"])
def test_baseline_configs_at_full_size(Renderer, lut_arrays, cfg):
    """BASELINE.json configs[2..4] at their full frame size and sample count, through properties that need no full oracle
    render: (a) N spp in one launch == the same N spp accumulated progressively, bit for bit (sample index continues
    from current_spp; the per-pixel sum keeps sample order);  (b) the oracle renders every 97th 8x8 tile of the SAME full
    frame for the first 2 spp and those pixels match the GPU bit for bit;  (c) the frame is finite (linear-sRGB components of a
    single wavelength may be negative, their sum over the spectrum is what the display transform sees);
    (d) the display transform of the preset (CRF of Apollo 11 included) stays in [0, 1]."""
    from digital_earth_amd import parallel
    W, H, SPP, CHUNK, preset, heavy = {
        "cfg3": (1920, 1080, 256, 64, "config - florida.txt", False),
        "cfg4": (3840, 2160, 128, 32, "config - sunset hurricane.txt", True),
        "cfg5": (1920, 1080, 1024, 128, "config - Apollo 11.txt", False)}[cfg]
    r = Renderer((W, H), (0, 1, 0), texture_source="synthetic", texture_size=(4096, 2048), cloud_heavy=heavy, seed=5)
    r.copy_textures()
    _apply_preset(r, preset)
    # (b) first: 2 spp against the oracle on a sparse tile subset
    r.accumulate(2)
    first = r.fetch_hdr()
    texels = [r.download_texture(s) for s in range(7)]
    o = make_oracle(W, H, lut_arrays, texels, r._params)
    o.accumulate(2, 5, 3, 97)
    mask = parallel.owned_pixel_mask(W, H, 3, 97)
    assert mask.sum() > 10000
    assert (o.fetch_hdr()[mask].view(np.uint32) == first[mask].view(np.uint32)).all()
    # (a) progressive: continue in chunks up to SPP, then one launch of SPP
    while r.current_spp < SPP:
        r.accumulate(min(CHUNK, SPP - r.current_spp))
    assert r.current_spp == SPP
    progressive = r.fetch_hdr()
    image = r.fetch_image()
    r.reset_framebuffer()
    r.accumulate(SPP)
    one = r.fetch_hdr()
    assert (one.view(np.uint32) == progressive.view(np.uint32)).all()
    # (c), (d)
    assert np.isfinite(one).all() and one.max() > 0.0
    assert image.shape == (W, H, 3) and image.min() >= 0.0 and image.max() <= 1.0 and image.std() > 0.01


def test_random_cameras_match_the_oracle(Renderer, lut_arrays):
    """Fuzz: cameras from 2 m above the ground to 100 000 km out, looking anywhere (up from below the cloud deck, along
    the limb, away from the planet), any sun position, several terrain scales and both sampler address modes.  Every
    configuration must match the oracle bit for bit — the proofs behind the shortened division / reciprocal / square
    root sequences and the escaped-ray early-out are stated for ranges, and this is where a range that does not hold
    would show."""
    from digital_earth_amd import _native
    W, H, SPP = 64, 32, 2
    rng = np.random.default_rng(20260101)
    r = Renderer((W, H), (0, 1, 0), texture_source="synthetic", texture_size=(1024, 512), seed=9)
    r.copy_textures()
    texels = {False: [r.download_texture(s) for s in range(7)]}
    R = 6371e3
    worst = 1.0
    for k in range(28):
        alt = float(10.0 ** rng.uniform(0.3, 8.0))                      # 2 m .. 100 000 km
        u = rng.standard_normal(3); u /= np.linalg.norm(u)
        pos = u * (R + 8000.0 * 1.0 + alt) if alt < 2e4 else u * (R + alt)   # low cameras sit above the highest terrain
        mode = k % 4
        if mode == 0:   look = np.zeros(3)                               # at the planet
        elif mode == 1: look = pos + rng.standard_normal(3) * 1e6        # anywhere
        elif mode == 2:                                                  # along the limb
            t = np.cross(u, rng.standard_normal(3)); t /= np.linalg.norm(t)
            look = pos + t * 1e6 - u * rng.uniform(0.0, 2e5)
        else:           look = pos + u * 1e6 + rng.standard_normal(3) * 3e5   # up and away
        r.set_camera_pos(*[float(x) for x in pos]); r.set_look_at(*[float(x) for x in look])
        up = rng.standard_normal(3); r.set_up(*[float(x) for x in up])
        r.set_fov(float(rng.uniform(0.02, 0.9))); r.set_aspect_scale(float(rng.uniform(0.8, 1.2)))
        r.set_sun_angle(float(rng.uniform(0.0, 2 * np.pi))); r.set_sun_path_rot(float(rng.uniform(-1.5, 1.5)))
        r.land_height_scale = float(rng.choice([0.0, 8000.0, 12000.0, 30000.0]))
        clamp = bool(k % 5 == 4)
        r.set_flag(_native.DE_FLAG_CLAMP_SAMPLER, clamp)
        r._push_params()
        o = make_oracle(W, H, lut_arrays, texels[False], r._params)
        r.reset_framebuffer(); r.accumulate(SPP); o.accumulate(SPP, 9)
        g, c = r.fetch_hdr(), o.fetch_hdr()
        same = bits_equal_fraction(g, c)
        worst = min(worst, same)
        assert same == 1.0, (k, alt, mode, same, rel_l2(g, c))
    assert worst == 1.0


@pytest.mark.parametrize("case", ["centre", "inside", "north_pole_down", "south_axis_up", "on_x_axis", "grazing_inside_atmosphere",
                                  "negative_terrain_scale", "wide_fov", "marcher_low", "fixed_wavelength_low"])
def test_special_cameras_match_the_oracle(Renderer, lut_arrays, case):
    """Hand-picked geometry: camera at the planet's centre / inside the planet (every SDF value negative), on the polar
    axis (atan2(0, 0), asin(+-1)), on a coordinate axis, just above the ground inside the atmosphere looking along it,
    a negative terrain scale, a near-fisheye field of view; plus the ray marcher and the single-wavelength mode from a
    low camera.  All must terminate and match the oracle bit for bit."""
    W, H, SPP = 64, 32, 2
    R = 6371e3
    r = Renderer((W, H), (0, 1, 0), texture_source="synthetic", texture_size=(1024, 512), seed=4)
    r.copy_textures()
    cfg = {
        "centre": dict(pos=(0.0, 0.0, 0.0), look=(1.0, 0.0, 0.0)),
        "inside": dict(pos=(R * 0.5, 1000.0, -2000.0), look=(R * 2, 0.0, 0.0)),
        "north_pole_down": dict(pos=(0.0, R + 4.0e5, 0.0), look=(0.0, 0.0, 0.0), up=(0.0, 0.0, 1.0)),
        "south_axis_up": dict(pos=(0.0, -(R + 9.0e3), 0.0), look=(0.0, -(R + 1.0e6), 0.0), up=(1.0, 0.0, 0.0)),
        "on_x_axis": dict(pos=(R + 2.0e6, 0.0, 0.0), look=(0.0, 0.0, 0.0)),
        "grazing_inside_atmosphere": dict(pos=(R + 9000.0, 0.0, 0.0), look=(R + 9000.0, 0.0, 1.0e6), fov=0.3),
        "negative_terrain_scale": dict(pos=(-15e6, 0.0, 15e6), look=(0.0, 0.0, 0.0), scale=-8000.0),
        "wide_fov": dict(pos=(R + 3.0e5, 1.0e5, 0.0), look=(0.0, 0.0, 0.0), fov=1.5),
        "marcher_low": dict(pos=(R + 2.0e4, 0.0, 5.0e3), look=(0.0, 5.0e6, 0.0), marcher=True),
        "fixed_wavelength_low": dict(pos=(0.0, 2.0e3, R + 1.2e4), look=(1.0e6, 0.0, R), fixed=610.0),
    }[case]
    r.set_camera_pos(*cfg["pos"]); r.set_look_at(*cfg["look"]); r.set_up(*cfg.get("up", (0.0, 1.0, 0.0)))
    if "fov" in cfg: r.set_fov(cfg["fov"])
    if "scale" in cfg: r.land_height_scale = cfg["scale"]
    if cfg.get("marcher"): r.set_integrator("ray_marcher")
    if "fixed" in cfg: r.set_fixed_wavelength(cfg["fixed"])
    r.set_sun_angle(1.1); r.set_sun_path_rot(-0.4)
    r._push_params()
    o = make_oracle(W, H, lut_arrays, [r.download_texture(s) for s in range(7)], r._params)
    r.accumulate(SPP); o.accumulate(SPP, 4)
    g, c = r.fetch_hdr(), o.fetch_hdr()
    assert np.isfinite(g).all()
    assert bits_equal_fraction(g, c) == 1.0, (case, rel_l2(g, c))
