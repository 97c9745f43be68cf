"""ctypes binding of the CPU oracle (oracle/libde_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under digital_earth_amd/ imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libde_oracle.so")


class DeParams(ctypes.Structure):
    """Mirror of `de_params` in include/digital_earth.h."""
    _fields_ = [
        ("camera_pos", ctypes.c_float * 3), ("look_at", ctypes.c_float * 3), ("up", ctypes.c_float * 3),
        ("fov", ctypes.c_float), ("aspect_scale", ctypes.c_float), ("sun_angle", ctypes.c_float),
        ("sun_path_rot", ctypes.c_float), ("land_height_scale", ctypes.c_float), ("exposure", ctypes.c_float),
        ("gamma", ctypes.c_float), ("selected_crf", ctypes.c_int32), ("vignette_strength", ctypes.c_float),
        ("vignette_radius", ctypes.c_float), ("vignette_center", ctypes.c_float * 2), ("flags", ctypes.c_uint32),
        ("fixed_wavelength", ctypes.c_float), ("topo_res_override", ctypes.c_int32), ("reserved", ctypes.c_int32 * 7),
    ]


class DeCounters(ctypes.Structure):
    _fields_ = [("samples", ctypes.c_uint64), ("taps_r8", ctypes.c_uint64), ("taps_rgb8", ctypes.c_uint64),
                ("sphere_steps", ctypes.c_uint64), ("tracking_steps", ctypes.c_uint64), ("vertices", ctypes.c_uint64),
                ("rng_draws", ctypes.c_uint64), ("reserved", ctypes.c_uint64 * 9)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_ if k != "reserved"}


def build(force=False):
    """Compile the oracle with oracle/Makefile (gcc).  Building the checker is not using it."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "libde_oracle.so"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.deo_last_error.restype = ctypes.c_char_p
        L.deo_arithmetic_contract.restype = ctypes.c_int
        L.deo_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        L.deo_destroy.argtypes = [ctypes.c_void_p]
        L.deo_set_threads.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.deo_get_threads.argtypes = [ctypes.c_void_p]
        L.deo_upload_texture.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.deo_generate_texture.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_int]
        L.deo_download_texture.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64]
        L.deo_upload_luts.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 4 + [ctypes.c_int]
        L.deo_set_params.argtypes = [ctypes.c_void_p, ctypes.POINTER(DeParams)]
        L.deo_get_params.argtypes = [ctypes.c_void_p, ctypes.POINTER(DeParams)]
        L.deo_reset.argtypes = [ctypes.c_void_p]
        L.deo_accumulate.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_int, ctypes.c_int]
        L.deo_accumulate_strided.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_int, ctypes.c_int]
        L.deo_debug_samples.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p]
        L.deo_fetch_image.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.deo_fetch_hdr.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.deo_set_hdr.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.deo_set_sample_partition.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.deo_current_spp.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        L.deo_get_counters.argtypes = [ctypes.c_void_p, ctypes.POINTER(DeCounters)]
        L.deo_eval.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.deo_eval_transmittance.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_float, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise RuntimeError("oracle error %d: %s" % (rc, lib().deo_last_error().decode()))


def evaluate(fn, x, n_out=1):
    """Evaluate a leaf function of the restatement on rows of x (deo_eval)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    if x.ndim == 1:
        x = x[:, None]
    out = np.zeros((x.shape[0], n_out), dtype=np.float32)
    _check(lib().deo_eval(fn.encode(), x.ctypes.data, x.shape[1], out.ctypes.data, n_out, x.shape[0]))
    return out


class Oracle:
    """CPU restatement of the reference Renderer hot path (same call sequence as the HIP library)."""

    def __init__(self, width, height, threads=0):
        self.W, self.H = int(width), int(height)
        self._h = ctypes.c_void_p()
        _check(lib().deo_create(self.W, self.H, ctypes.byref(self._h)))
        if threads:
            lib().deo_set_threads(self._h, threads)

    def close(self):
        if self._h:
            lib().deo_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def threads(self):
        return lib().deo_get_threads(self._h)

    def upload_texture(self, slot, texels):
        t = np.ascontiguousarray(texels, dtype=np.uint8)
        h, w, ch = t.shape
        _check(lib().deo_upload_texture(self._h, slot, t.ctypes.data, w, h, ch))

    def generate_texture(self, slot, w, h, seed, variant=0):
        _check(lib().deo_generate_texture(self._h, slot, w, h, seed, variant))

    def download_texture(self, slot, w, h, ch):
        out = np.zeros((h, w, ch), dtype=np.uint8)
        _check(lib().deo_download_texture(self._h, slot, out.ctypes.data, out.nbytes))
        return out

    def upload_luts(self, cie, srgb2spec_bits, o3, crf):
        cie = np.ascontiguousarray(cie, np.float32)
        s = np.ascontiguousarray(srgb2spec_bits, np.uint16)
        o3 = np.ascontiguousarray(o3, np.float32)
        crf = np.ascontiguousarray(crf, np.float32)
        _check(lib().deo_upload_luts(self._h, cie.ctypes.data, s.ctypes.data, o3.ctypes.data, crf.ctypes.data, crf.shape[1]))

    def get_params(self):
        p = DeParams()
        _check(lib().deo_get_params(self._h, ctypes.byref(p)))
        return p

    def set_params(self, p):
        _check(lib().deo_set_params(self._h, ctypes.byref(p)))

    def reset(self):
        _check(lib().deo_reset(self._h))

    def accumulate(self, spp=1, seed=0, tile_rank=0, tile_world=1):
        _check(lib().deo_accumulate(self._h, spp, seed, tile_rank, tile_world))

    def accumulate_strided(self, spp, seed, first, stride):
        """The checker's sparse subset: only the 8x8 tiles whose raster index = first (mod stride) — spread over the whole frame."""
        _check(lib().deo_accumulate_strided(self._h, spp, seed, first, stride))

    @staticmethod
    def strided_mask(width, height, first, stride):
        """(W, H) boolean mask of the pixels accumulate_strided renders."""
        tx = np.arange(width) // 8
        ty = np.arange(height) // 8
        return ((ty[None, :] * (width // 8) + tx[:, None]) % stride) == first

    def set_sample_partition(self, rank, world):
        """Of the frame's sample indices this oracle renders those = rank (mod world); accumulate(spp) still advances the index by spp."""
        _check(lib().deo_set_sample_partition(self._h, rank, world))

    def debug_samples(self, seed, sample_index):
        out = np.zeros((self.H, self.W, 4), dtype=np.float32)
        _check(lib().deo_debug_samples(self._h, seed, sample_index, out.ctypes.data))
        return out

    def fetch_image(self):
        out = np.zeros((self.W, self.H, 3), dtype=np.float32)
        _check(lib().deo_fetch_image(self._h, out.ctypes.data))
        return out

    def fetch_hdr(self):
        out = np.zeros((self.W, self.H, 3), dtype=np.float32)
        _check(lib().deo_fetch_hdr(self._h, out.ctypes.data))
        return out

    def set_hdr(self, hdr, spp):
        hdr = np.ascontiguousarray(hdr, np.float32)
        assert hdr.shape == (self.W, self.H, 3)
        _check(lib().deo_set_hdr(self._h, hdr.ctypes.data, spp))

    def counters(self):
        c = DeCounters()
        _check(lib().deo_get_counters(self._h, ctypes.byref(c)))
        return c.as_dict()

    def mean_transmittance(self, seed, pos, direction, wavelength, n):
        pos = np.ascontiguousarray(pos, np.float32)
        direction = np.ascontiguousarray(direction, np.float32)
        out = ctypes.c_float()
        _check(lib().deo_eval_transmittance(self._h, seed, pos.ctypes.data, direction.ctypes.data, wavelength, n, ctypes.byref(out)))
        return out.value
