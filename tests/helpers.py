"""Shared helpers of the test-suite: build an oracle that mirrors a product Renderer, error metrics."""
import ctypes

import numpy as np

from oracle import oracle_binding as ob
from digital_earth_amd import luts, textures as T


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.sqrt(np.sum((a - b) ** 2)) / max(np.sqrt(np.sum(b ** 2)), 1e-300))


def bits_equal_fraction(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    b = np.ascontiguousarray(b, dtype=np.float32).view(np.uint32)
    return float(np.mean(a == b))


def make_oracle(width, height, lut_arrays, texels=None, params=None, threads=0):
    """Oracle with the reference LUTs; texels = list of 7 uint8 arrays [h][w][c] or None for BASELINE cfg1 constants."""
    o = ob.Oracle(width, height, threads)
    cie, s2s, o3, crf, _ = lut_arrays
    o.upload_luts(cie, s2s, o3, crf)
    for s in range(7):
        t = texels[s] if texels is not None else T.constant_texels(s, (128, 128, 128) if s == T.ALBEDO else 0)
        o.upload_texture(s, t)
    if params is not None:
        o.set_params(copy_params(params))
    return o


def copy_params(src):
    """Copy a de_params ctypes struct between the product's and the oracle's (identical) definitions."""
    dst = ob.DeParams()
    ctypes.memmove(ctypes.byref(dst), ctypes.byref(src), ctypes.sizeof(dst))
    return dst


def load_preset(name):
    """camera/sun preset from digital_earth_amd/data/configs (the reference's `config - *.txt` files), read with the
    PRODUCT's parser (digital_earth_amd.earth_viewer.load_config; checked against the file's tokens in
    tests/test_host_logic.py and tests/test_gpu_round2.py)."""
    from digital_earth_amd.earth_viewer import load_config
    c = load_config(name)
    return dict(pos=[float(x) for x in c.position], look_at=[float(x) for x in c.look_at], up=[float(x) for x in c.up], fov=c.fov,
                aspect_scale=c.aspect_scale, exposure=c.exposure, crf=c.selected_crf, gamma=c.gamma, sun_angle=c.sun_angle,
                sun_path_rot=c.sun_path_rot)
