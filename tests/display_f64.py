"""An INDEPENDENT float64 / numpy statement of the display path — Renderer._render_to_image (renderer.py:346-365),
camera_response (:333-344), openDR_transform with the live module constants (lib/OpenDRT.py:39-55, 221-485) and
srgb_transfer (lib/colour.py:74-79) — written array-at-once from the formulas, sharing no code with oracle/ or csrc/.
Every constant is derived here from the module constants.  It pins the oracle's (and through it the kernel's) display
arithmetic and host-side constants to a second, differently structured statement (VERDICT r1 #7a)."""
import numpy as np

LP, GB, CONTRAST, FLARE, RW, BW, DCH, DCH_TOE = 100.0, 0.12, 1.0, 0.005, 0.25, 0.35, 0.35, 0.0
HS = np.array([0.3, -0.1, -0.2])          # hue shift r, g, b
V_P = 0.5
REC709_TO_XYZ = np.array([[0.412390917540, 0.357584357262, 0.180480793118], [0.212639078498, 0.715168714523, 0.072192311287],
                          [0.019330825657, 0.119194783270, 0.950532138348]])
XYZ_TO_REC709 = np.array([[3.2409699419, -1.53738317757, -0.498610760293], [-0.969243636281, 1.87596750151, 0.041555057407],
                          [0.055630079697, -0.203976958889, 1.05697151424]])


def inverse_flare(x, fl):
    return (x + np.sqrt(x * (4.0 * fl + x))) / 2.0


def tonescale_constants():
    """(m, s, ds, clamp_max, dch_s) for EOTF = linear, Lp = 100 (lib/OpenDRT.py:257-319, 404)."""
    ds = 100.0 / LP
    clamp_max = ds * LP / 100.0
    px = 128.0 * np.log10(LP) / np.log10(100.0) - 64.0
    py = LP / 100.0
    gx = 0.18
    gy = 11.696 / 100.0 * (1.0 + GB * np.log2(py))
    s0, m0 = inverse_flare(gy, FLARE), inverse_flare(py, FLARE)
    ip = 1.0 / CONTRAST
    s = (px * gx * (m0 ** ip - s0 ** ip)) / (px * s0 ** ip - gx * m0 ** ip)
    m = m0 ** ip * (s + px) / px
    return m, s, ds, clamp_max, DCH / s


def safe_div(a, b):
    """sdivf: 0 where |b| < 1e-4 (the reference's `result == 0.0` typo leaves the initial 0)."""
    b = np.broadcast_to(b, np.shape(a)) if np.ndim(a) else b
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(np.abs(b) < 1e-4, 0.0, a / np.where(b == 0, 1.0, b))


def safe_pow(a, b):
    with np.errstate(invalid="ignore"):
        return np.where(a <= 0.0, a, np.power(np.maximum(a, 1e-300), b))


def opendrt(rgb):
    """rgb: (..., 3) linear Rec.709 -> display-linear Rec.709 in [0, 1]."""
    m, s, ds, clamp_max, dch_s = tonescale_constants()
    rgb = np.asarray(rgb, np.float64)
    rgb = rgb @ REC709_TO_XYZ        # vdot(m, v) = v @ m: the row-vector product, as the reference writes it (transposed twice: cancels)
    rgb = rgb @ XYZ_TO_REC709
    mx = rgb.max(-1, keepdims=True)
    mn = rgb.min(-1, keepdims=True)
    hue = safe_div(rgb - mn, mx)
    others = hue.sum(-1, keepdims=True) - hue
    hue = np.clip(hue - others, 0.0, 2.0)                       # narrow_hue_angles
    w = np.array([RW, 1.0, BW]); w = w / np.sqrt((w * w).sum())
    lum = np.sqrt(((w * np.maximum(rgb, 1e-5)) ** 2).sum(-1, keepdims=True))
    rats = safe_div(rgb, lum)
    ts = safe_pow(m * lum / (lum + s), CONTRAST)
    ts = safe_pow(ts, 2.0) / (ts + FLARE)
    ts = ts * ds
    ccf = safe_div(1.0, lum * dch_s + 1.0)
    toe_ccf = (DCH_TOE + 1.0) * safe_div(lum, lum + DCH_TOE) * ccf
    hw = (1.0 - ccf) * hue
    shift = np.stack([hw[..., 2] * HS[2] - hw[..., 1] * HS[1], hw[..., 0] * HS[0] - hw[..., 2] * HS[2],
                      hw[..., 1] * HS[1] - hw[..., 0] * HS[0]], -1)
    rats = rats + shift
    rats = np.maximum(1.0 - toe_ccf + rats * toe_ccf, 0.0)
    rmx = rats.max(-1, keepdims=True)
    rmn = rats.min(-1, keepdims=True)
    chroma = safe_div(rmx - rmn, rmx)
    chf = safe_pow(chroma * ts, V_P)
    rats = safe_div(rats, rmx) * chf + rats * (1.0 - chf)
    return np.minimum(rats * ts, clamp_max)


def bilinear_crf(crf, u, slice_index, clamp):
    """crf: (1024, n, 3); sample_lod at (u, (slice + 0.5) / n), texel centres at (i + 0.5) / N; returns (..., 3)."""
    n = crf.shape[1]
    v = (slice_index + 0.5) / n
    x = u * 1024.0 - 0.5
    y = v * n - 0.5
    x0 = np.floor(x); fx = x - x0
    y0 = np.floor(y); fy = y - y0

    def wrap(i, size):
        i = i.astype(np.int64) if isinstance(i, np.ndarray) else int(i)
        return np.clip(i, 0, size - 1) if clamp else np.mod(i, size)
    i0, i1 = wrap(x0, 1024), wrap(x0 + 1, 1024)
    j0, j1 = wrap(np.array(y0), n), wrap(np.array(y0 + 1), n)
    a = crf[i0, j0] * (1 - fx)[..., None] + crf[i1, j0] * fx[..., None]
    b = crf[i0, j1] * (1 - fx)[..., None] + crf[i1, j1] * fx[..., None]
    return a * (1 - fy) + b * fy


def srgb_transfer(x):
    x = np.asarray(x, np.float64)
    return np.where(x <= 0.0031308, x * 12.92, 1.055 * np.abs(x) ** (1 / 2.4) - 0.055)


def render_to_image(hdr_sum, samples, crf, selected_crf, exposure, gamma, vignette_strength=0.9, vignette_radius=0.0,
                    vignette_center=(0.5, 0.5), clamp_sampler=False):
    """hdr_sum: (W, H, 3) accumulated linear radiance; returns the (W, H, 3) displayed image."""
    W, H = hdr_sum.shape[:2]
    u = (np.arange(W) / W)[:, None]
    v = (np.arange(H) / H)[None, :]
    darken = 1.0 - vignette_strength * np.maximum(np.sqrt((u - vignette_center[0]) ** 2 + (v - vignette_center[1]) ** 2) - vignette_radius, 0.0)
    linear = hdr_sum.astype(np.float64) / samples * darken[..., None] * 2.0 ** exposure
    t = np.clip(opendrt(linear), 0.0, 1.0)
    lookup = np.minimum(t + 0.5 / 1024.0, 1.0 - 0.5 / 1024.0)
    cam = np.stack([bilinear_crf(crf, lookup[..., k], selected_crf, clamp_sampler)[..., k] for k in range(3)], -1)
    cam = np.clip(cam, 0.0, 1.0)
    return srgb_transfer(np.power(cam, gamma))
