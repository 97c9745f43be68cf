"""LUT loading — host side of renderer.py:96-134 (CIE / srgb2spec / O3) and Renderer.load_crfs (renderer.py:147-167).

The files under data/LUT are the reference's own numeric artefacts (LUT/CIE.dat, LUT/srgb2spec.dat,
LUT/ozone_cross_section.dat, LUT/camera_response_functions/*.rf), byte-identical copies (sha256 pinned in
tests/test_luts.py).  Arrays are returned exactly as read; quantisation to the texture formats of the reference
(rgba16f for CIE) happens inside the native library, where the reference does it in its copy_* kernels.
"""
import os
import numpy as np

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
LUT_DIR = os.path.join(DATA_DIR, "LUT")
CRF_DIR = os.path.join(LUT_DIR, "camera_response_functions")
CONFIG_DIR = os.path.join(DATA_DIR, "configs")

CIE_LUT_RES = (441, 2)          # lib/textures.py:7
O3_CROSSEC_LUT_RES = 441        # lib/textures.py:8
CIE_LUT_FILE = os.path.join(LUT_DIR, "CIE.dat")                      # lib/textures.py:29
SRGB2SPEC_LUT_FILE = os.path.join(LUT_DIR, "srgb2spec.dat")          # lib/textures.py:30
O3_CROSSEC_LUT_FILE = os.path.join(LUT_DIR, "ozone_cross_section.dat")  # lib/textures.py:31


def load_cie():
    """renderer.py:99-100 — f32[(x + y*441)*3 + c]; row y=0 normalised CDF, row y=1 CIE 1931 colour matching functions."""
    a = np.fromfile(CIE_LUT_FILE, dtype=np.float32, count=CIE_LUT_RES[0] * CIE_LUT_RES[1] * 3)
    if a.size != 441 * 2 * 3:
        raise IOError("CIE.dat truncated")
    return np.ascontiguousarray(a)


def load_srgb2spec():
    """renderer.py:110-111 — 300 x 3 binary16 values, returned as their uint16 bit patterns."""
    a = np.fromfile(SRGB2SPEC_LUT_FILE, dtype=np.uint16, count=300 * 3)
    if a.size != 900:
        raise IOError("srgb2spec.dat truncated")
    return np.ascontiguousarray(a)


def load_o3():
    """renderer.py:120-121 — ozone absorption cross-section, cm^2, 390..830 nm."""
    a = np.fromfile(O3_CROSSEC_LUT_FILE, dtype=np.float32, count=O3_CROSSEC_LUT_RES)
    if a.size != 441:
        raise IOError("ozone_cross_section.dat truncated")
    return np.ascontiguousarray(a)


def crf_filenames(directory=CRF_DIR):
    """renderer.py:154-158 lists the directory in os.listdir order (filesystem dependent) with
    Neutral.rf moved to the front.  Here the order is pinned: Neutral.rf first, the rest sorted by name."""
    names = [f for f in os.listdir(directory)
             if (f.endswith(".txt") or f.endswith(".rf")) and "README" not in f]
    names.sort()
    names.insert(0, names.pop(names.index("Neutral.rf")))
    return names


def load_crfs(directory=CRF_DIR):
    """renderer.py:147-167 — returns (names, array of shape (1024, n, 3) f32); the irradiance column is dropped."""
    names = crf_filenames(directory)
    data = []
    for name in names:
        with open(os.path.join(directory, name), "r") as fh:
            rows = [list(map(float, line.split()))[1:] for line in fh.readlines()]
        data.append(rows)
    arr = np.array(data, dtype=np.float32).transpose(1, 0, 2)
    return names, np.ascontiguousarray(arr)
