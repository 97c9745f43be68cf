"""The oracle's display path against tests/display_f64.py (an independent float64 / numpy statement of the same reference
formulas, constants derived there from the module constants).  VERDICT r1 #7a: the C++ host constants of the product
(de_api.hip) and of the oracle were the same text; this third statement shares nothing with either."""
import numpy as np

import display_f64 as D
from helpers import make_oracle
from oracle import oracle_binding as ob


def test_tonescale_constants_from_an_independent_derivation():
    m, s, ds, clamp_max, dch_s = D.tonescale_constants()
    k = ob.evaluate("opendrt_consts", [0.0], 7)[0]           # px, gy, s0, m0, s, m, dch_s
    assert abs(k[4] / s - 1) < 2e-7 and abs(k[5] / m - 1) < 2e-7 and abs(k[6] / dch_s - 1) < 2e-7
    assert ds == 1.0 and clamp_max == 1.0


def test_opendrt_pixels_against_float64():
    rng = np.random.default_rng(11)
    rgb = (rng.lognormal(-1.5, 2.0, size=(20000, 3)) * rng.uniform(0, 1, size=(20000, 1))).astype(np.float32)
    rgb[:4] = [[0, 0, 0], [0.18, 0.18, 0.18], [50, 0.01, 0.01], [1e-3, 2e-3, 5e-4]]
    got = ob.evaluate("opendrt", rgb, 3).astype(np.float64)
    want = D.opendrt(rgb.astype(np.float64))
    err = np.abs(got - want).max(1)
    assert np.quantile(err, 0.999) < 3e-6, np.quantile(err, 0.999)
    assert err.max() < 2e-4          # the |b| < 1e-4 cut-offs of sdivf are discontinuities: f32 and f64 may sit on either side


def test_full_display_transform_against_float64(lut_arrays):
    W, H = 96, 48
    rng = np.random.default_rng(3)
    hdr = (rng.lognormal(-1.0, 2.0, size=(W, H, 3)) * rng.uniform(0, 1, size=(W, H, 1))).astype(np.float32)
    crf = lut_arrays[3]
    for sel, gamma, exposure in ((0, 1.0, 2.5), (7, 0.8, 0.3), (12, 1.001, 2.432)):
        o = make_oracle(W, H, lut_arrays)
        p = o.get_params(); p.selected_crf = sel; p.gamma = gamma; p.exposure = exposure; o.set_params(p)
        o.set_hdr(hdr, 8)
        got = o.fetch_image().astype(np.float64)
        want = D.render_to_image(hdr, 8, crf.astype(np.float64), sel, float(np.float32(exposure)), float(np.float32(gamma)))
        err = np.abs(got - want)
        assert np.quantile(err, 0.999) < 3e-5, (sel, np.quantile(err, 0.999))
        assert err.max() < 2e-3, (sel, err.max())
