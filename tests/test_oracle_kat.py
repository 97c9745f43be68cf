"""Known-answer tests of the CPU oracle.  The reference holds no tests or golden vectors for this path,
so the expected values are closed forms evaluated in float64 here — not outputs of the reference."""
import numpy as np
import pytest

from oracle import oracle_binding as ob

ev = ob.evaluate


def ulp_err(got, ref64):
    ref32 = ref64.astype(np.float32)
    ulp = np.spacing(np.abs(ref32)).astype(np.float64)
    ulp[ulp == 0] = 1e-45
    return np.max(np.abs(got.astype(np.float64) - ref64) / ulp)


def test_elementary_functions_accuracy():
    rng = np.random.default_rng(0)
    x = rng.uniform(-87, 88, 400000).astype(np.float32)
    assert ulp_err(ev("exp", x)[:, 0], np.exp(x.astype(np.float64))) < 1.6
    x = np.exp(rng.uniform(-80, 80, 400000)).astype(np.float32)
    assert ulp_err(ev("log", x)[:, 0], np.log(x.astype(np.float64))) < 1.6
    x = rng.uniform(0, 2 * np.pi, 400000).astype(np.float32)
    assert np.abs(ev("sin", x)[:, 0] - np.sin(x.astype(np.float64))).max() < 1.5e-7
    assert np.abs(ev("cos", x)[:, 0] - np.cos(x.astype(np.float64))).max() < 1.5e-7
    # contract 2: atan2 and asin feed the sphere UV map only; their stated accuracy is absolute —
    # 8e-7 rad (0.003 texel of a 21600-wide map; 3.5e-7 from the polynomial, the rest f32 rounding near pi) and 2.5e-7 rad
    yx = rng.normal(size=(400000, 2)).astype(np.float32)
    assert np.abs(ev("atan2", yx)[:, 0] - np.arctan2(yx[:, 0].astype(np.float64), yx[:, 1].astype(np.float64))).max() < 8e-7
    x = rng.uniform(-1, 1, 400000).astype(np.float32)
    assert np.abs(ev("asin", x)[:, 0] - np.arcsin(x.astype(np.float64))).max() < 2.5e-7
    # log of a random draw (free-flight distances): every value k * 2^-24, relative error below 6e-7
    k = np.concatenate([rng.integers(1, 1 << 24, 400000), [1, 2, 3, (1 << 24) - 1, 1 << 23, (1 << 23) + 1, 11863283, 11863284]])
    x = (k.astype(np.float64) * 2.0 ** -24).astype(np.float32)
    got = ev("log_unit", x)[:, 0].astype(np.float64)
    assert np.max(np.abs(got / np.log(x.astype(np.float64)) - 1)) < 6e-7
    ab = np.stack([rng.uniform(1e-3, 100, 200000), rng.uniform(-3, 3, 200000)], 1).astype(np.float32)
    ref = ab[:, 0].astype(np.float64) ** ab[:, 1].astype(np.float64)
    assert np.max(np.abs(ev("pow", ab)[:, 0] / ref - 1)) < 3e-6


def test_elementary_functions_special_values():
    assert ev("log", [0.0])[0, 0] == -np.inf
    assert ev("log_unit", [0.0])[0, 0] == -np.inf                  # -log(0) = +inf ends a tracking loop (pathtracer.py:92-96)
    assert np.isnan(ev("log", [-1.0])[0, 0])
    assert ev("exp", [-100.0])[0, 0] == 0.0 and ev("exp", [0.0])[0, 0] == 1.0 and np.isinf(ev("exp", [89.0])[0, 0])
    assert ev("atan2", [[0.0, 0.0]])[0, 0] == 0.0
    assert ev("asin", [1.0000001])[0, 0] == np.float32(np.pi / 2)   # clamped argument
    assert ev("pow", [[0.37, 1.0]])[0, 0] == np.float32(0.37) and ev("pow", [[5.0, 0.0]])[0, 0] == 1.0
    assert ev("pow", [[0.0, 0.4167]])[0, 0] == 0.0


def test_f16_quantisation_matches_numpy():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.normal(size=50000) * 10.0 ** rng.uniform(-9, 5, 50000), [0.0, 65504.0, 65520.0, 1e-8, 6e-8, 113.042]]).astype(np.float32)
    got = ev("quantize_f16", x)[:, 0]
    with np.errstate(over="ignore"):
        want = x.astype(np.float16).astype(np.float32)
    assert (got.view(np.uint32) == want.view(np.uint32)).all()


def test_rsi_hit_miss_inside():
    R = 6371e3
    # from outside, head on: near = d - R, far = d + R
    r = ev("rsi", [[0, 0, -2 * R, 0, 0, 1, R]], 2)[0]
    assert abs(r[0] - R) < 2 and abs(r[1] - 3 * R) < 2
    # inside: near < 0 < far
    r = ev("rsi", [[0, 0, 0, 1, 0, 0, R]], 2)[0]
    assert r[0] == -R and r[1] == R
    # miss: NaN, NaN (sqrt of a negative discriminant; never (-1, -1))
    r = ev("rsi", [[0, 3 * R, -2 * R, 0, 0, 1, R]], 2)[0]
    assert np.isnan(r).all()


def test_sphere_uv_map():
    uv = ev("sphere_UV_map", [[-1, 0, 0], [1, 0, 0], [0, 0, 1], [0, 1, 0], [0, -1, 0]], 2)
    assert np.allclose(uv[0], [0.5, 0.5]) and np.allclose(uv[1], [1.0, 0.5], atol=1e-6) and np.allclose(uv[2], [0.75, 0.5], atol=1e-6)
    assert uv[3][1] == 1.0 and 0.0 <= uv[4][1] < 1e-8     # fma(-pi/2, RN(1/pi), 0.5): the south pole lands 6e-9 above 0


def _integrate_phase(fn, extra=()):
    c = np.linspace(-1, 1, 2000001)
    x = np.stack([c] + [np.full_like(c, e) for e in extra], 1)
    p = ev(fn, x)[:, 0].astype(np.float64)
    return 2 * np.pi * np.trapezoid(p, c)


@pytest.mark.parametrize("fn,extra,tol", [("rayleigh_phase", (), 1e-5), ("hg_phase", (0.8,), 1e-4), ("hg_phase", (0.91,), 2e-4),
                                          ("draine_phase", (0.541064, 20.3257), 1e-4), ("cloud_phase", (1.0,), 3e-4)])
def test_phase_functions_integrate_to_one(fn, extra, tol):
    assert abs(_integrate_phase(fn, extra) - 1.0) < tol


def test_klein_nishina_phase_values_and_norm():
    # KN phase (e = 3000) at cos = 1 is 54.883, at cos = 0 is 0.018288
    p = ev("mie_phase", [1.0, 0.0])[:, 0]
    assert abs(p[0] - 54.883) < 5e-3 and abs(p[1] - 0.018288) < 2e-6
    # analytic integral over the sphere: e/(2 pi ln(2e+1)) * 2 pi * ln(2e+1)/e = 1  (substitution, f64 check of the f32 samples near the peak)
    c = 1 - np.logspace(-9, np.log10(2), 400001)[::-1]
    p = ev("mie_phase", c.astype(np.float32))[:, 0].astype(np.float64)
    assert abs(2 * np.pi * np.trapezoid(p, c.astype(np.float32).astype(np.float64)) - 1.0) < 2e-3


def test_cloud_phase_constants():
    k = ev("cloud_consts", [0.0], 4)[0]
    assert np.allclose(k, [0.984469, 0.541064, 20.3257, 0.473642], rtol=2e-6)


def test_extinction_spectra_and_densities():
    assert abs(ev("extinction_rayleigh", [550.0])[0, 0] / 1.167e-5 - 1) < 2e-3
    assert abs(ev("extinction_rayleigh", [400.0])[0, 0] / 4.320e-5 - 1) < 2e-3
    assert abs(ev("extinction_mie", [550.0])[0, 0] / 2.049e-5 - 1) < 2e-3
    d0 = ev("density", [0.0], 3)[0]
    assert abs(d0[0] - 0.99637) < 2e-5 and abs(d0[1] - 1.06) < 1e-6
    # Rayleigh profile in f64
    h = np.array([0.0, 5e3, 2e4, 8e4], np.float32)
    ref = 3.68082 * np.exp(-(h.astype(np.float64) + 24239.99) ** 2 / 532307548.4168) / 1.225
    assert np.allclose(ev("density", h, 3)[:, 0], ref, rtol=3e-6)
    # ozone: peak value at 25 km is 1 (plus nothing from the cubic tail)
    assert abs(ev("density", [25000.0], 3)[0, 2] - 1.0) < 1e-6
    # negative elevations clamp to 0 (get_density: h = max(h, 0))
    assert (ev("density", [-50.0], 3)[0] == d0).all()


def test_planck_and_sun():
    assert abs(ev("plancks", [[5778.0, 550.0]])[0, 0] / 28042.73 - 1) < 2e-5
    assert abs(ev("plancks", [[2700.0, 550.0]])[0, 0] / 187.53 - 1) < 5e-5
    # sun solid angle 6.78131e-5 sr, computed in f32 from 1 - cos(4.646e-3): quantised to ~0.3 %
    assert abs(ev("solid_angle", [6.95e8 / 1.4959e11])[0, 0] / 6.78131e-5 - 1) < 6e-3


def test_opendrt_constants_and_behaviour():
    k = ev("opendrt_consts", [0.0], 7)[0]
    assert np.allclose(k, [64.0, 0.11696, 0.1217628, 1.0049752, 1.3366673, 1.0259646, 0.2618453], rtol=2e-6)
    rgb = ev("opendrt", [[0.18, 0.18, 0.18], [0, 0, 0], [1e4, 1e4, 1e4], [5.0, 0.1, 0.1]], 3)
    assert np.allclose(rgb[0], 0.11696, atol=2e-4)          # middle grey maps to gy
    assert (rgb[1] >= 0).all() and rgb[1].max() < 1e-6   # maxf3(1e-5, rgb) keeps a 1e-8 floor (lib/OpenDRT.py:371)
    assert (rgb[2] <= 1.0).all() and rgb[2].min() > 0.99    # clamp_max = 1
    assert rgb[3][0] > rgb[3][1] and (rgb[3] >= 0).all() and (rgb[3] <= 1).all()
    mono = ev("opendrt", np.repeat(np.logspace(-3, 3, 200)[:, None], 3, 1), 3)[:, 1]
    assert (np.diff(mono) >= -1e-7).all()                   # tonescale is monotonic


def test_srgb_transfer():
    x = np.array([0.0, 0.001, 0.0031308, 0.2, 1.0], np.float32)
    ref = np.where(x <= 0.0031308, x * 12.92, 1.055 * x.astype(np.float64) ** (1 / 2.4) - 0.055)
    assert np.allclose(ev("srgb_transfer", x)[:, 0], ref, atol=2e-6)


def test_rng_uniform_and_independent():
    u = ev("rng", np.stack([np.zeros(4096), np.arange(4096), np.zeros(4096)], 1), 64)
    assert u.min() >= 0.0 and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 3e-3 and abs(u.var() - 1 / 12) < 2e-3
    assert abs(np.corrcoef(u[:, :-1].ravel(), u[:, 1:].ravel())[0, 1]) < 1.2e-2       # lag-1 within a stream (sigma = 2e-3 at this sample size)
    assert abs(np.corrcoef(u[:-1, 0], u[1:, 0])[0, 1]) < 6e-2                         # neighbouring pixels
    # the value set is 24-bit
    assert (u * 2 ** 24 == np.round(u * 2 ** 24)).all()
    # streams differ by sample index and seed
    a = ev("rng", [[0, 7, 0]], 8); b = ev("rng", [[0, 7, 1]], 8); c = ev("rng", [[1, 7, 0]], 8)
    assert not (a == b).any() and not (a == c).any()


def _dirs(kind, n, view=(0.3, -0.5, 0.81), p5=0.0, p6=0.0, seed=5):
    v = np.array(view, np.float64); v /= np.linalg.norm(v)
    x = np.tile(np.array([kind, seed, v[0], v[1], v[2], p5, p6], np.float32), (n, 1))
    return ev("sample_dir", x, 3).astype(np.float64), v


def test_sampled_directions_are_unit_and_follow_their_pdfs():
    n = 200000
    d, v = _dirs(0, n)                                   # sample_sphere: uniform
    assert np.allclose(np.linalg.norm(d, axis=1), 1, atol=1e-5) and np.abs(d.mean(0)).max() < 6e-3
    d, v = _dirs(1, n)                                   # cosine hemisphere: E[cos] = 2/3
    c = d @ v
    assert c.min() > -1e-4 and abs(c.mean() - 2 / 3) < 3e-3
    cos_max = np.cos(6.95e8 / 1.4959e11)
    d, v = _dirs(2, n, p5=np.float32(cos_max))           # sun cone: stays inside the cone (f32 slack)
    assert (d @ v).min() > cos_max - 2e-7
    for g in (0.8, 0.91):                                # HG: E[cos] = g
        d, v = _dirs(3, n, p5=g)
        assert abs((d @ v).mean() - g) < 3e-3
    d, v = _dirs(5, n, p5=0.541064, p6=20.3257)          # Draine: chi-square against its own pdf
    c = d @ v
    edges = np.linspace(-1, 1, 41)
    hist = np.histogram(c, edges)[0]
    cc = np.linspace(-1, 1, 400001)
    pdf = ev("draine_phase", np.stack([cc, np.full_like(cc, 0.541064), np.full_like(cc, 20.3257)], 1))[:, 0].astype(np.float64) * 2 * np.pi
    cdf = np.concatenate([[0], np.cumsum((pdf[1:] + pdf[:-1]) / 2 * np.diff(cc))])
    expect = np.diff(np.interp(edges, cc, cdf)) * n
    chi2 = ((hist - expect) ** 2 / np.maximum(expect, 1)).sum()
    assert chi2 < 100, chi2                              # 39 dof
    d, v = _dirs(4, n)                                   # Klein-Nishina e=3000: strongly forward
    assert abs(np.median(d @ v) - (1 - (np.sqrt(6001.0) - 1) / 3000.0)) < 2e-3   # CDF = ln(1+e(1-c))/ln(2e+1)


def test_earth_brdf_energy_bound_and_reciprocity_of_terms():
    rng = np.random.default_rng(3)
    n = np.array([0, 0, 1.0])
    def hemi(k):
        d = rng.normal(size=(k, 3)); d /= np.linalg.norm(d, axis=1)[:, None]; d[:, 2] = np.abs(d[:, 2]); return d
    k = 20000
    l = hemi(k)
    for v in (np.array([0, 0, 1.0]), np.array([0.6, 0, 0.8]), np.array([0.995, 0, 0.0998749])):
        for ocean, bathy in ((0.0, 0.5), (1.0, 0.2), (1.0, 0.9)):
            x = np.concatenate([np.tile([1.0, ocean, bathy], (k, 1)), np.tile(v, (k, 1)), np.tile(n, (k, 1)), l], 1)
            o = ev("earth_brdf", x, 2).astype(np.float64)
            assert np.isfinite(o).all() and (o[:, 0] >= 0).all()
            # white furnace bound: uniform-hemisphere estimate of the albedo integral of brdf * cos stays below 1
            est = (o[:, 0] * o[:, 1]).mean() * 2 * np.pi
            assert est < 1.0, (v, ocean, est)


def test_agx_known_points():
    """lib/AgX.py: scene middle grey (0.18) sits on the curve's pivot (y = 0.5); black -> 0; the curve is monotonic and
    bounded by [0, 1] for achromatic input."""
    a = ev("agx", [[0.18, 0.18, 0.18], [0.0, 0.0, 0.0], [1e4, 1e4, 1e4]], 3)
    assert np.allclose(a[0], 0.5, atol=1e-5) and (a[1] == 0).all() and np.allclose(a[2], 1.0, atol=1e-4)
    g = np.logspace(-4, 2, 200)
    mono = ev("agx", np.repeat(g[:, None], 3, 1), 3)[:, 1]
    assert (np.diff(mono) >= -1e-6).all() and mono.min() >= 0 and mono.max() <= 1
