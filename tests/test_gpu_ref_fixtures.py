"""The HIP path, through the C ABI, against the REFERENCE'S OWN TEXT executed (tests/golden/ref_*.npz; see tests/test_ref_fixtures.py
and tools/ref_fixtures/make.py for provenance and tolerances).  No oracle in this file: the fixtures are the reference's numbers."""
import os

import numpy as np
import pytest

from test_ref_fixtures import CAMERAS, GOLDEN, compare_marcher_with_fixture, compare_with_fixture, per_sample

pytestmark = pytest.mark.gpu


def apply_fixture_camera(r, z, cam, crf_names):
    c = z[cam + "_camera"]
    r.set_camera_pos(*c[0:3]); r.set_look_at(*c[3:6]); r.set_up(*c[6:9])
    r.set_fov(c[9]); r.set_aspect_scale(c[10]); r.set_exposure(c[11]); r.set_gamma(c[13])
    r.set_sun_angle(c[14]); r.set_sun_path_rot(c[15])
    r.set_crf(list(crf_names).index(str(z[cam + "_crf_name"])))       # the preset's index counts os.listdir order (Q10): by name
    # the scalars as the reference's set_* kernels stored them (f32), bit for bit — `up` normalised like ti.Vector.normalized()
    f = z[cam + "_fields"]
    got = np.array([r.fov[None], r.aspect_scale[None], r.exposure[None], r.gamma[None], r.sun_angle[None], r.sun_path_rot[None]]
                   + list(r.up[None]), dtype=np.float32)
    assert (got.view(np.uint32) == f.view(np.uint32)).all(), (cam, got, f)


@pytest.mark.parametrize("mode", ["numpy", "oracle"])
def test_hip_frames_match_the_executed_reference(mode, lut_arrays):
    from digital_earth_amd.renderer import Renderer
    z = np.load(os.path.join(GOLDEN, "ref_frames_%s.npz" % mode))
    W, H = (int(x) for x in z["image_res"])
    spp = int(z["spp"])
    r = Renderer((W, H), (0, 1, 0), seed=int(z["rng_seed"]), texture_source="synthetic", texture_size=tuple(int(x) for x in z["tier"]),
                 synth_seed=int(z["synth_seed"]))
    for cam in CAMERAS:
        apply_fixture_camera(r, z, cam, lut_arrays[4])
        rad, wl, draws, verts = per_sample(z, cam, [r.debug_samples(s) for s in range(spp)])
        same = compare_with_fixture(z, cam, rad, wl, draws, verts, mode)
        ok = np.ones((W, H), dtype=bool)
        ok[z[cam + "_u"][~same], z[cam + "_v"][~same]] = False
        r.reset_framebuffer()
        r.accumulate(spp)
        hdr, ref_hdr = r.fetch_hdr().astype(np.float64), z[cam + "_hdr"].astype(np.float64)
        l2 = lambda a, b: float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))
        assert l2(hdr[ok], ref_hdr[ok]) <= 1e-4, (cam, l2(hdr[ok], ref_hdr[ok]))     # north-star: 1e-3
        assert l2(hdr, ref_hdr) <= 1e-2
        assert np.abs(r.fetch_image() - z[cam + "_image"])[ok].max() <= 3e-4


def test_hip_quality2_tier_matches_the_executed_reference(lut_arrays):
    """The reference's own tier (TEXTURE_QUALITY = 2: 21600x10800 / 16200x8100 maps), 1920x1080, four cameras, one sample on a
    lattice of 1798 pixels over the frame — BASELINE configs[1] / [2] / [4] as the reference's text computes them."""
    from digital_earth_amd.renderer import Renderer
    z = np.load(os.path.join(GOLDEN, "ref_q2.npz"))
    W, H = (int(x) for x in z["image_res"])
    r = Renderer((W, H), (0, 1, 0), seed=int(z["rng_seed"]), texture_source="synthetic", texture_quality=2, synth_seed=int(z["synth_seed"]))
    import ctypes
    r.copy_textures()
    sizes = []
    for s in range(7):
        w, h = ctypes.c_int(), ctypes.c_int()
        assert r._lib.de_texture_info(r._h, s, ctypes.byref(w), ctypes.byref(h), None) == 0
        sizes.append((w.value, h.value))
    assert sizes == [tuple(int(x) for x in s) for s in z["sizes"]]       # lib/textures.py:65-79 (quality 2)
    for cam in CAMERAS:
        apply_fixture_camera(r, z, cam, lut_arrays[4])
        rad, wl, draws, verts = per_sample(z, cam, [r.debug_samples(0)])
        same = compare_with_fixture(z, cam, rad, wl, draws, verts, "numpy", min_same=0.998)    # measured: 0, 1, 2, 0 of 1798 differ
        # colour contribution of each sample: xyzToRGB @ (radiance * response * rcp_pdf) (renderer.py:329-330)
        r.reset_framebuffer(); r.accumulate(1)
        hdr = r.fetch_hdr()
        rgb = hdr[z[cam + "_u"], z[cam + "_v"]].astype(np.float64)
        ref = z[cam + "_rgb"].astype(np.float64)
        assert np.sqrt(((rgb - ref)[same] ** 2).sum() / (ref[same] ** 2).sum()) <= 1e-4


def test_hip_cloud_heavy_4k_matches_the_executed_reference(lut_arrays):
    """BASELINE configs[3] as the reference's text computes it (tools/ref_fixtures/make.py q2c): the quality-2 tier with the cloud-heavy
    maps, 3840x2160, `config - sunset hurricane.txt`, one sample on a lattice of 1782 pixels over the frame."""
    from digital_earth_amd.renderer import Renderer
    z = np.load(os.path.join(GOLDEN, "ref_q2c.npz"))
    W, H = (int(x) for x in z["image_res"])
    assert (W, H) == (3840, 2160) and int(z["synth_variant"]) == 1
    r = Renderer((W, H), (0, 1, 0), seed=int(z["rng_seed"]), texture_source="synthetic", texture_quality=2, synth_seed=int(z["synth_seed"]),
                 cloud_heavy=True)
    r.copy_textures()
    cam = "sunset"
    apply_fixture_camera(r, z, cam, lut_arrays[4])
    rad, wl, draws, verts = per_sample(z, cam, [r.debug_samples(0)])
    same = compare_with_fixture(z, cam, rad, wl, draws, verts, "numpy", min_same=0.995)
    r.reset_framebuffer(); r.accumulate(1)
    hdr = r.fetch_hdr()
    rgb = hdr[z[cam + "_u"], z[cam + "_v"]].astype(np.float64)
    ref = z[cam + "_rgb"].astype(np.float64)
    assert np.sqrt(((rgb - ref)[same] ** 2).sum() / (ref[same] ** 2).sum()) <= 1e-4


def test_hip_ray_marcher_matches_the_executed_reference(lut_arrays):
    """ray_march_kernel against the reference's ray_marcher (pathtracer.py:471-685) executed under the stand-in."""
    from digital_earth_amd.renderer import Renderer
    z = np.load(os.path.join(GOLDEN, "ref_marcher.npz"))
    W, H = (int(x) for x in z["image_res"])
    r = Renderer((W, H), (0, 1, 0), seed=int(z["rng_seed"]), texture_source="synthetic", texture_size=tuple(int(x) for x in z["tier"]),
                 synth_seed=int(z["synth_seed"]))
    r.set_integrator("ray_marcher")
    for cam in ("default", "florida"):
        apply_fixture_camera(r, z, cam, lut_arrays[4])
        rad, wl, draws, _ = per_sample(z, cam, [r.debug_samples(0)])
        compare_marcher_with_fixture(z, cam, rad, wl, draws)
        r.reset_framebuffer(); r.accumulate(1)
        hdr, ref_hdr = r.fetch_hdr().astype(np.float64), z[cam + "_hdr"].astype(np.float64)
        assert np.sqrt(((hdr - ref_hdr) ** 2).sum() / (ref_hdr ** 2).sum()) <= 1e-4
