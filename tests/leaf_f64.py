"""Independent float64 / numpy statements of three leaf formulas of the hot path, written from the reference's formulas
(file:line below) and sharing no code with oracle/ or csrc/ (VERDICT r1 #7b):
  earth_brdf and its terms          lib/surface_rendering_models.py:9-122, 146-152, 169-171
  the colour grade of get_land_material   pathtracer.py:292-306
  the Draine phase function's CDF   lib/volume_rendering_models.py:121-123 (integrated numerically here)"""
import numpy as np

LUMA = np.array([0.2126729, 0.7151522, 0.0721750])


def _mix(a, b, t):
    return a * (1.0 - t) + b * t


def _smoothstep(e0, e1, x):
    t = np.clip((x - e0) / (e1 - e0), 0.0, 1.0)
    return t * t * (3.0 - 2.0 * t)


def _sat(x):
    return np.clip(x, 0.0, 1.0)


def fresnel_dielectric(c, f0):
    n = (1.0 + np.sqrt(f0)) / (1.0 - np.sqrt(f0))
    sin_i = np.sqrt(_sat(1.0 - c * c))
    sin_t = sin_i / max(n, 1e-8)
    cos_t = np.sqrt(1.0 - sin_t * sin_t)
    rs = ((c - n * cos_t) / max(c + n * cos_t, 1e-8)) ** 2
    rp = ((cos_t - n * c) / max(cos_t + n * c, 1e-8)) ** 2
    return _sat(0.5 * (rs + rp))


def ggx_specular(rough, f0, ndl, ndv, ldh, ndh):
    a2 = rough * rough
    d = a2 / (np.pi * ((a2 - 1.0) * ndh * ndh + 1.0) ** 2)
    lam = lambda c: 0.5 * (-1.0 + np.sqrt(a2 * (1.0 - c * c) / (c * c) + 1.0))
    g = 1.0 / (1.0 + lam(ndv) + lam(ndl))
    return d * g * fresnel_dielectric(ldh, f0) / max(4.0 * ndl * ndv, 1e-5)


def beckmann_specular(rough, f0, ndl, ndv, ldh, ndh):
    alpha = 2.0 * rough * rough
    c2 = ndh * ndh
    d = np.exp(-(1.0 - c2) / (alpha * alpha * c2)) / max(np.pi * alpha * alpha * c2 * c2, 1e-5)
    vis = min(1.0, 2.0 * ndv * ndh / ldh, 2.0 * ndl * ndh / ldh)
    return d * vis * fresnel_dielectric(ldh, f0)


def disney_diffuse(rough, ndl, ndv, ldh):
    rr = 2.0 * rough * ldh * ldh
    fl, fv = (1.0 - ndl) ** 5, (1.0 - ndv) ** 5
    return (1.0 - 0.5 * fl) * (1.0 - 0.5 * fv) / np.pi + rr * (fl + fv + fl * fv * (rr - 1.0)) / np.pi


def earth_brdf(albedo, ocean, bathy, v, n, l):
    v, n, l = (np.asarray(x, np.float64) for x in (v, n, l))
    h = (v + l) / np.linalg.norm(v + l)
    ndl, ndv, ldh, ndh = _sat(n @ l), _sat(n @ v), _sat(l @ h), _sat(n @ h)
    land_rough = 0.73
    ocean_rough = _mix(0.25, 0.19, _smoothstep(0.3, 0.7, bathy))
    diffuse = disney_diffuse(land_rough, ndl, ndv, ldh)
    land = ggx_specular(land_rough, 0.04, ndl, ndv, ldh, ndh)
    o_ggx = ggx_specular(ocean_rough, 0.02, ndl, ndv, ldh, ndh)
    o_beck = 0.65 * beckmann_specular(ocean_rough, 0.02, ndl, ndv, ldh, ndh)
    ocean_spec = _mix(o_beck, o_ggx, np.clip(_smoothstep(0.2, 0.95, ndv), 0.05, 0.94))
    return albedo * diffuse * 0.28 + _mix(land, ocean_spec, _smoothstep(0.6, 1.0, ocean)) * 0.5, ndl


def grade_land_albedo(tex, ocean):
    tex = np.asarray(tex, np.float64)
    grey = lambda c: np.full(3, c @ LUMA)
    land = _mix(grey(tex), tex, 6.5)
    green = _smoothstep(1.5, 1.9, (land[1] / (land @ LUMA)) ** 2)
    land = tex / (green * 0.7 + 1.0)
    land = _mix(grey(land), land, 1.4 - green * 0.45)
    land = _mix(land, land * np.array([255.0, 128.0, 64.0]) / 255.0, 0.2 * (1.0 - green))
    sea = _mix(grey(tex), tex, 0.75) * 0.9
    return _mix(land, sea, ocean)


def draine_cdf(cos_theta, g, alpha, n=400001):
    """P(cos <= cos_theta) for the Draine phase function, by trapezoid quadrature of 2 pi p(c) over [-1, cos_theta]."""
    c = np.linspace(-1.0, 1.0, n)
    p = (1 - g * g) * (1 + alpha * c * c) / (4.0 * (1 + alpha * (1 + 2 * g * g) / 3.0) * np.pi * (1 + g * g - 2 * g * c) ** 1.5)
    cdf = np.concatenate([[0.0], np.cumsum((p[1:] + p[:-1]) * 0.5 * np.diff(c))]) * 2.0 * np.pi
    return np.interp(cos_theta, c, cdf)
