"""GPU parity, second batch: the map tiers the bench actually runs, the headless viewer through the PRODUCT's preset
parser, checkpoint / resume against the oracle, and the plumbing added for multi-GPU runs (shared maps, the library's
own RCCL communicator, split launches)."""
import os

import numpy as np
import pytest

from helpers import rel_l2, bits_equal_fraction, make_oracle, copy_params
from oracle import oracle_binding as ob

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Renderer():
    from digital_earth_amd.renderer import Renderer as R
    return R


def _oracle_for(r, lut_arrays, W, H):
    return make_oracle(W, H, lut_arrays, [r.download_texture(s) for s in range(7)], r._params)


def test_cfg2_full_size_quality2_maps(Renderer, lut_arrays):
    """BASELINE configs[1] as bench.py runs it: 1920x1080, 64 spp, default camera, quality-2 maps (21600x10800 and
    16200x8100: 233 M-entry footprint-packed arrays, 8100 % 4 = 0 but 16200/8 = 2025 tiles per row).  The oracle renders
    every 97th 8x8 tile of the same frame at all 64 spp; those pixels must match bit for bit."""
    from digital_earth_amd import parallel
    W, H, SPP = 1920, 1080, 64
    r = Renderer((W, H), (0, 1, 0), texture_source="synthetic", texture_quality=2, seed=0)
    r.copy_textures()
    r.accumulate(SPP)
    g = r.fetch_hdr()
    assert np.isfinite(g).all() and g.max() > 0
    o = _oracle_for(r, lut_arrays, W, H)
    o.accumulate(SPP, 0, 3, 97)
    mask = parallel.owned_pixel_mask(W, H, 3, 97)
    assert mask.sum() > 20000
    c = o.fetch_hdr()
    assert (c[mask].view(np.uint32) == g[mask].view(np.uint32)).all(), rel_l2(g[mask], c[mask])
    img = r.fetch_image()
    assert img.min() >= 0.0 and img.max() <= 1.0 and img.std() > 0.01


@pytest.mark.parametrize("clamp", [False, True])
def test_mixed_map_sizes_and_partial_tiles(Renderer, lut_arrays, clamp):
    """The quality-1 tier gives the cloud map another size than the topography (lib/textures.py:57-64: 8100x4050 vs
    10800x5400), which takes the kernel's per-lane map-geometry select; and 8100 % 8 = 4, 4050 % 4 = 2 leave partial 8x4
    footprint tiles at the east and north edges.  Scaled-down maps with the same properties (810 % 8 = 2, 405 % 4 = 1,
    1080 / 540 full tiles), REPEAT and CLAMP, a camera that sees the seam and the pole."""
    from digital_earth_amd import _native, textures as T
    W, H, SPP = 128, 64, 2
    sizes = {T.ALBEDO: (1080, 540), T.TOPOGRAPHY: (1080, 540), T.OCEAN: (810, 405), T.CLOUDS: (810, 405),
             T.BATHYMETRY: (1080, 540), T.EMISSIVE: (1082, 541), T.STARS: (810, 405)}
    r = Renderer((W, H), (0, 1, 0), texture_source="synthetic", texture_size=sizes, seed=2)
    r.set_flag(_native.DE_FLAG_CLAMP_SAMPLER, clamp)
    lit = 0
    for pos, look, sun in (((-15e6, 0.0, 15e6), (0, 0, 0), 1.047), ((2.0e6, 9.0e6, 0.0), (0, 6.0e6, 0), 0.1), ((8.0e6, 1.0e5, 1.0e4), (0, 0, 0), 4.6)):
        r.set_camera_pos(*pos); r.set_look_at(*look); r.set_sun_angle(sun)
        r.copy_textures() if not r._textures_copied else None
        o = _oracle_for(r, lut_arrays, W, H)
        r.reset_framebuffer()
        r.accumulate(SPP); o.accumulate(SPP, 2)
        g, c = r.fetch_hdr(), o.fetch_hdr()
        lit += int(g.max() > 0)
        assert bits_equal_fraction(g, c) == 1.0, (clamp, pos, rel_l2(g, c))
    assert lit >= 2


def test_earth_viewer_preset_through_product_parser(lut_arrays):
    """f1: EarthViewer(config=...) reads the preset with the product's load_config; the oracle is configured from the
    same file through the same parser's Config object (not through the renderer), so a parser that drops or swaps a
    line shows up as a mismatch against the values written in the file."""
    from digital_earth_amd.earth_viewer import EarthViewer, load_config
    from digital_earth_amd import luts
    W, H = 128, 72
    name = "config - Apollo 11.txt"
    v = EarthViewer(config=name, screen_res=(W, H), texture_source="synthetic", texture_size=(1024, 512))
    # what the file says, read here without the product parser
    with open(os.path.join(luts.CONFIG_DIR, name)) as f:
        tok = f.read().split()
    want = [float(x) for x in tok]
    cfg = load_config(name)
    got = list(cfg.position) + list(cfg.look_at) + list(cfg.up) + [cfg.fov, cfg.aspect_scale, cfg.exposure, cfg.selected_crf,
                                                                  cfg.gamma, cfg.sun_angle, cfg.sun_path_rot]
    assert got == want
    r = v.renderer
    p = ob.DeParams()
    o = make_oracle(W, H, lut_arrays, [r.download_texture(s) for s in range(7)])
    p = o.get_params()
    for i in range(3):
        p.camera_pos[i] = want[i]; p.look_at[i] = want[3 + i]
    up = np.array(want[6:9], np.float32); up = (np.float32(1.0) / np.sqrt((up[0] * up[0] + up[1] * up[1]) + up[2] * up[2])) * up   # ti.Vector.normalized(): invlen * v
    for i in range(3):
        p.up[i] = float(up[i])
    p.fov, p.aspect_scale, p.exposure, p.selected_crf, p.gamma, p.sun_angle, p.sun_path_rot = want[9], want[10], want[11], int(want[12]), want[13], want[14], want[15]
    o.set_params(p)
    img = v.render(spp=3)
    o.accumulate(3, 0)
    assert (r.fetch_hdr().view(np.uint32) == o.fetch_hdr().view(np.uint32)).all()
    assert np.abs(img - o.fetch_image()).max() <= 1e-5
    assert r.selected_crf[None] == 12 and len(r.crf_names) == 16 and r.crf_names[0] == "Neutral.rf"   # order: Neutral first, then by name (Q10)


def test_earth_viewer_frame_loop_reset_semantics(lut_arrays):
    """earth_viewer.py:203-317: exposure / camera response / gamma changes keep the accumulation, camera / sun / fov /
    aspect changes reset it — AFTER the frame that was rendered with the old values."""
    from digital_earth_amd.earth_viewer import EarthViewer
    W, H = 64, 32
    v = EarthViewer(config="config - florida.txt", screen_res=(W, H), texture_source="synthetic", texture_size=(1024, 512))
    r = v.renderer
    o = _oracle_for(r, lut_arrays, W, H)
    img = v.frame(spp=2)
    o.accumulate(2, 0)
    assert r.current_spp == 2 and np.abs(img - o.fetch_image()).max() <= 1e-5
    # display-only sliders: no reset, the NEXT frame shows them on 4 accumulated samples
    img = v.frame(spp=2, exposure=1.25, gamma=0.9, selected_crf=5)
    assert r.current_spp == 4
    o.accumulate(2, 0)
    assert np.abs(img - o.fetch_image()).max() <= 1e-5          # still the old display parameters
    img = v.frame(spp=1)
    o.set_params(copy_params(r._params)); o.accumulate(1, 0)
    assert r.current_spp == 5 and np.abs(img - o.fetch_image()).max() <= 1e-5
    assert (r.fetch_hdr().view(np.uint32) == o.fetch_hdr().view(np.uint32)).all()
    # fov: the frame is rendered with the old fov, then the accumulation is dropped
    img = v.frame(spp=1, fov=0.31)
    assert r.current_spp == 0 and abs(r.fov[None] - 0.31) < 1e-7
    o.accumulate(1, 0)
    assert np.abs(img - o.fetch_image()).max() <= 1e-5
    o2 = _oracle_for(r, lut_arrays, W, H)
    img = v.frame(spp=2)
    o2.accumulate(2, 0)
    assert r.current_spp == 2 and np.abs(img - o2.fetch_image()).max() <= 1e-5
    # sun and camera reset too; an unchanged slider value does not
    v.frame(spp=1, sun_angle=r.sun_angle[None]); assert r.current_spp == 3
    v.frame(spp=1, sun_path_rot=0.2); assert r.current_spp == 0
    v.frame(spp=1); assert r.current_spp == 1
    v.camera.rotate(0.01, 0.0)
    v.frame(spp=1)
    assert r.current_spp == 0 and np.allclose(r.look_at[None], v.camera.look_at.astype(np.float32))
    with pytest.raises(TypeError):
        v.frame(spp=1, zoom=2.0)


def test_checkpoint_resume_against_the_oracle(Renderer, lut_arrays, tmp_path):
    """f4: 2 spp -> save -> (new renderer) load -> 3 spp must equal the ORACLE's accumulate(5), not just another GPU run."""
    kw = dict(texture_source="synthetic", texture_size=(1024, 512))
    a = Renderer((64, 64), (0, 1, 0), seed=13, **kw); a.set_fov(0.4); a.set_sun_angle(0.8)
    a.accumulate(2)
    a.save_checkpoint(str(tmp_path / "s.npz"))
    b = Renderer((64, 64), (0, 1, 0), **kw)
    b.load_checkpoint(str(tmp_path / "s.npz"))
    b.accumulate(3)
    o = _oracle_for(b, lut_arrays, 64, 64)
    o.accumulate(5, 13)
    assert b.current_spp == 5
    assert (b.fetch_hdr().view(np.uint32) == o.fetch_hdr().view(np.uint32)).all()
    assert np.abs(b.fetch_image() - o.fetch_image()).max() <= 1e-5


def test_shared_maps_between_contexts(Renderer):
    """de_share_textures: a second context renders from the first one's maps and LUTs (frames in flight without a second
    copy); same bits as a context with its own copies; the borrower survives re-uploading maps of its own."""
    kw = dict(texture_source="synthetic", texture_size=(1024, 512))
    a = Renderer((64, 32), (0, 1, 0), seed=3, **kw); a.set_fov(0.4)
    a.accumulate(2)
    want = a.fetch_hdr()
    b = Renderer((64, 32), (0, 1, 0), seed=3, **kw); b.set_fov(0.4)
    b.share_textures_from(a)
    b.accumulate(2)
    assert (b.fetch_hdr().view(np.uint32) == want.view(np.uint32)).all()
    assert (b.download_texture(1) == a.download_texture(1)).all()
    img_a, img_b = a.fetch_image(), b.fetch_image()
    assert (img_a == img_b).all()
    b.copy_textures()                       # back to maps of its own
    b.reset_framebuffer(); b.accumulate(2)
    assert (b.fetch_hdr().view(np.uint32) == want.view(np.uint32)).all()
    b.close(); a.close()


def test_trim_textures_keeps_rendering(Renderer):
    from digital_earth_amd._native import DigitalEarthError
    from digital_earth_amd import _native
    r = Renderer((64, 32), (0, 1, 0), texture_source="synthetic", texture_size=(512, 256), seed=1); r.set_fov(0.4)
    r.accumulate(2)
    want = r.fetch_hdr()
    r.trim_textures()
    r.reset_framebuffer(); r.accumulate(2)
    assert (r.fetch_hdr().view(np.uint32) == want.view(np.uint32)).all()
    with pytest.raises(DigitalEarthError):
        r.download_texture(1)
    r.set_flag(_native.DE_FLAG_CLAMP_SAMPLER, True)
    with pytest.raises(DigitalEarthError):
        r.accumulate(1)                       # repacking for another address mode needs the released copies
    r.copy_textures()                         # generating them again makes the context whole
    r.accumulate(1)


def test_library_communicator_single_rank(Renderer):
    """de_comm_unique_id / de_comm_init / de_reduce / de_comm_destroy with a 1-rank RCCL communicator: binds librccl at
    run time and runs ncclReduce in place on the context's stream (a 1-GPU box cannot hold two RCCL ranks)."""
    from digital_earth_amd._native import DigitalEarthError
    r = Renderer((64, 32), (0, 1, 0), texture_source="synthetic", texture_size=(512, 256)); r.set_fov(0.4)
    with pytest.raises(DigitalEarthError):
        r.reduce(0)                          # no communicator yet
    uid = r.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    r.comm_init(uid, 0, 1)
    with pytest.raises(DigitalEarthError):
        r.comm_init(uid, 0, 1)               # already has one
    r.accumulate(2)
    before = r.fetch_hdr()
    r.reduce(0)
    r.synchronize()
    assert (r.fetch_hdr().view(np.uint32) == before.view(np.uint32)).all() and before.max() > 0
    with pytest.raises(DigitalEarthError):
        r.reduce(1)
    r.comm_destroy()
    r.comm_destroy()                         # idempotent


def test_library_comm_wrapper_single_rank(Renderer):
    """parallel.LibraryComm (the torch-free host side of de_comm_* / de_reduce, `bench.py --reduce lib`) with one rank: the
    frame loop reset -> accumulate -> reduce -> display gives the plain renderer's frame."""
    from digital_earth_amd import parallel
    kw = dict(texture_source="synthetic", texture_size=(512, 256))
    a = Renderer((64, 32), (0, 1, 0), seed=2, **kw); a.set_fov(0.4)
    a.accumulate(2)
    want_hdr, want_img = a.fetch_hdr(), a.fetch_image()
    b = Renderer((64, 32), (0, 1, 0), seed=2, **kw); b.set_fov(0.4)
    lc = parallel.LibraryComm(b, 0, 1, exchange=lambda uid: uid)
    lc.render_frame(2)
    assert (b.fetch_hdr().view(np.uint32) == want_hdr.view(np.uint32)).all()
    assert (b.fetch_image() == want_img).all()
    lc.render_frame(2)                      # a second frame starts from a reset buffer
    assert (b.fetch_hdr().view(np.uint32) == want_hdr.view(np.uint32)).all()
    lc.close()


def test_split_launches_give_the_same_bits(Renderer, monkeypatch):
    """de_accumulate splits a request that exceeds render_kernel_v2's per-launch limits; forced here at 3 spp per launch."""
    kw = dict(texture_source="synthetic", texture_size=(1024, 512))
    a = Renderer((64, 64), (0, 1, 0), seed=8, **kw); a.set_fov(0.4)
    a.accumulate(8)
    want = a.fetch_hdr()
    monkeypatch.setenv("DE_V2_MAX_SPP", "3")
    b = Renderer((64, 64), (0, 1, 0), seed=8, **kw); b.set_fov(0.4)
    monkeypatch.delenv("DE_V2_MAX_SPP")
    b.enable_counters(True)
    b.accumulate(8)
    assert b.current_spp == 8 and b.counters()["samples"] == 64 * 64 * 8
    assert (b.fetch_hdr().view(np.uint32) == want.view(np.uint32)).all()


def test_parameter_changes_rebuild_the_right_tables(Renderer, lut_arrays):
    """Camera / display changes rebuild the frame constants only, LUT-dependent changes (fixed wavelength, address mode)
    the wavelength table too: after each kind of change one long-lived context must match the oracle."""
    from digital_earth_amd import _native
    r = Renderer((64, 32), (0, 1, 0), texture_source="synthetic", texture_size=(512, 256), seed=6); r.set_fov(0.4)
    r.copy_textures()
    texels = [r.download_texture(s) for s in range(7)]
    steps = [lambda: r.set_camera_pos(-9e6, 2e6, 9e6), lambda: r.set_fixed_wavelength(480.0), lambda: r.set_exposure(0.5),
             lambda: r.set_fixed_wavelength(610.0), lambda: r.set_fixed_wavelength(None),
             lambda: r.set_flag(_native.DE_FLAG_CLAMP_SAMPLER, True), lambda: r.set_sun_angle(2.0)]
    for k, step in enumerate(steps):
        step()
        o = make_oracle(64, 32, lut_arrays, texels, r._params)
        r.reset_framebuffer(); r.accumulate(1); o.accumulate(1, 6)
        assert (r.fetch_hdr().view(np.uint32) == o.fetch_hdr().view(np.uint32)).all(), k
        assert np.abs(r.fetch_image() - o.fetch_image()).max() <= 1e-5, k
