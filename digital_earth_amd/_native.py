"""ctypes binding of libdigitalearth_hip.so (C ABI in include/digital_earth.h).

There is no CPU implementation behind this module: if the HIP library is missing, cannot be loaded, or finds no
gfx950 device, the error is raised — nothing falls back to another code path.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DE_LIB_PATH: A/B experiments load another build of the same ABI (tools/ab_build.sh) without touching the product library
LIB_PATH = os.environ.get("DE_LIB_PATH") or os.path.join(_HERE, "libdigitalearth_hip.so")


class DeParams(ctypes.Structure):
    """`de_params` (include/digital_earth.h)."""
    _fields_ = [
        ("camera_pos", ctypes.c_float * 3), ("look_at", ctypes.c_float * 3), ("up", ctypes.c_float * 3),
        ("fov", ctypes.c_float), ("aspect_scale", ctypes.c_float), ("sun_angle", ctypes.c_float),
        ("sun_path_rot", ctypes.c_float), ("land_height_scale", ctypes.c_float), ("exposure", ctypes.c_float),
        ("gamma", ctypes.c_float), ("selected_crf", ctypes.c_int32), ("vignette_strength", ctypes.c_float),
        ("vignette_radius", ctypes.c_float), ("vignette_center", ctypes.c_float * 2), ("flags", ctypes.c_uint32),
        ("fixed_wavelength", ctypes.c_float), ("topo_res_override", ctypes.c_int32), ("reserved", ctypes.c_int32 * 7),
    ]


class DeCounters(ctypes.Structure):
    """`de_counters` (include/digital_earth.h)."""
    _fields_ = [("samples", ctypes.c_uint64), ("taps_r8", ctypes.c_uint64), ("taps_rgb8", ctypes.c_uint64),
                ("sphere_steps", ctypes.c_uint64), ("tracking_steps", ctypes.c_uint64), ("vertices", ctypes.c_uint64),
                ("rng_draws", ctypes.c_uint64), ("reserved", ctypes.c_uint64 * 9)]

    def as_dict(self):
        d = {k: int(getattr(self, k)) for k, _ in self._fields_ if k != "reserved"}
        d["sched"] = [int(x) for x in self.reserved]
        return d


DE_FLAG_FIXED_WAVELENGTH = 1 << 0
DE_FLAG_CLAMP_SAMPLER = 1 << 1
DE_FLAG_RAY_MARCHER = 1 << 2
DE_FLAG_AGX = 1 << 3
DE_FLAG_FAST_MATH = 1 << 5

# name -> (restype, argtypes): every symbol include/digital_earth.h declares (the binder's header)
_P = ctypes.c_void_p
SYMBOLS = {
    "de_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_P)]),
    "de_destroy": (ctypes.c_int, [_P]),
    "de_upload_texture": (ctypes.c_int, [_P, ctypes.c_int, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "de_generate_texture": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_int]),
    "de_share_textures": (ctypes.c_int, [_P, _P]),
    "de_trim_textures": (ctypes.c_int, [_P]),
    "de_upload_luts": (ctypes.c_int, [_P, _P, _P, _P, _P, ctypes.c_int]),
    "de_set_params": (ctypes.c_int, [_P, ctypes.POINTER(DeParams)]),
    "de_get_params": (ctypes.c_int, [_P, ctypes.POINTER(DeParams)]),
    "de_reset": (ctypes.c_int, [_P]),
    "de_accumulate": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_uint64, ctypes.c_int, ctypes.c_int]),
    "de_fetch_image": (ctypes.c_int, [_P, _P]),
    "de_fetch_image_view": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.POINTER(ctypes.c_float))]),
    "de_fetch_image_begin": (ctypes.c_int, [_P]),
    "de_fetch_image_end": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.POINTER(ctypes.c_float))]),
    "de_render_to_image": (ctypes.c_int, [_P, ctypes.POINTER(_P)]),
    "de_fetch_hdr": (ctypes.c_int, [_P, _P]),
    "de_upload_hdr": (ctypes.c_int, [_P, _P, ctypes.c_int]),
    "de_current_spp": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_int)]),
    "de_set_current_spp": (ctypes.c_int, [_P, ctypes.c_int]),
    "de_hdr_device_ptr": (ctypes.c_int, [_P, ctypes.POINTER(_P), ctypes.POINTER(ctypes.c_uint64)]),
    "de_bind_hdr": (ctypes.c_int, [_P, _P, ctypes.c_uint64]),
    "de_set_stream": (ctypes.c_int, [_P, _P]),
    "de_use_own_stream": (ctypes.c_int, [_P]),
    "de_flush": (ctypes.c_int, [_P]),
    "de_comm_unique_id": (ctypes.c_int, [_P]),
    "de_comm_init": (ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_int]),
    "de_comm_destroy": (ctypes.c_int, [_P]),
    "de_reduce": (ctypes.c_int, [_P, _P, ctypes.c_int]),
    "de_reduce_progressive": (ctypes.c_int, [_P, _P, ctypes.c_int]),
    "de_set_sample_partition": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int]),
    "de_reduce_ordered": (ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_int]),
    "de_set_display_source": (ctypes.c_int, [_P, _P]),
    "de_synchronize": (ctypes.c_int, [_P]),
    "de_get_tuning": (ctypes.c_int, [_P, _P]),
    "de_set_tuning": (ctypes.c_int, [_P, _P]),
    "de_last_error": (ctypes.c_char_p, []),
    "de_abi_version": (ctypes.c_int, []),
    "de_arithmetic_contract": (ctypes.c_int, []),
}

# measurement, experiment and test hooks: include/digital_earth_debug.h (same library)
DEBUG_SYMBOLS = {
    "de_download_texture": (ctypes.c_int, [_P, ctypes.c_int, _P, ctypes.c_uint64]),
    "de_texture_info": (ctypes.c_int, [_P, ctypes.c_int] + [ctypes.POINTER(ctypes.c_int)] * 3),
    "de_last_reduce_ms": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_float)]),
    "de_set_launch_slots": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int]),
    "de_set_wave_budget": (ctypes.c_int, [_P, ctypes.c_int]),
    "de_last_accumulate_ms": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_float)]),
    "de_get_counters": (ctypes.c_int, [_P, ctypes.POINTER(DeCounters)]),
    "de_enable_counters": (ctypes.c_int, [_P, ctypes.c_int]),
    "de_set_kernel_variant": (ctypes.c_int, [_P, ctypes.c_int]),
    "de_debug_samples": (ctypes.c_int, [_P, ctypes.c_uint64, ctypes.c_int, _P]),
    "de_debug_sched_stats": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]),
    "de_set_memory_budget": (ctypes.c_int, [_P, ctypes.c_uint64]),
    "de_get_memory_use": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_uint64)]),
    "de_last_call_info": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "de_last_launch_phases": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_float)]),
    "de_debug_v6_stats": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]),
    "de_debug_math": (ctypes.c_int, [_P, ctypes.c_int, _P, _P, _P, ctypes.c_uint64]),
    "de_debug_ordered_sum": (ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    "de_debug_standin_reduce": (ctypes.c_int, [_P, ctypes.c_int]),
}

# entry points of the legacy library only (include/digital_earth_legacy.h): bound when present
LEGACY_SYMBOLS = {
    "de_debug_v5_stats": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]),
}


class DeTuning(ctypes.Structure):
    """de_tuning of include/digital_earth.h"""
    _fields_ = [("struct_bytes", ctypes.c_uint32), ("kernel_variant", ctypes.c_int32), ("launch_slots", ctypes.c_int32), ("big_launch_slots", ctypes.c_int32),
                ("v6_min_paths", ctypes.c_uint64), ("v6_service_area", ctypes.c_int32 * 3), ("v6_service_lanes", ctypes.c_int32 * 3),
                ("v6_yield_max", ctypes.c_int32), ("v6_elsewhere_min", ctypes.c_int32), ("v6_retry", ctypes.c_int32), ("v6_enter_min", ctypes.c_int32),
                ("v6_flat_min", ctypes.c_int32), ("v6_flat_again", ctypes.c_int32), ("v6_bands", ctypes.c_int32),
                ("v6_tail_levels", ctypes.c_int32), ("v6_tail_export", ctypes.c_int32 * 2), ("v6_tail_min_paths", ctypes.c_uint32), ("v6_tail_when_alone", ctypes.c_int32), ("v6_tail_grid", ctypes.c_int32 * 2), ("v6_stats", ctypes.c_int32),
                ("v2_pend", ctypes.c_int32), ("v2_heavy", ctypes.c_int32), ("v2_b", ctypes.c_int32), ("v2_gas", ctypes.c_int32), ("v2_chunk", ctypes.c_int32),
                ("v2_waves_per_cu", ctypes.c_int32), ("v2_max_spp", ctypes.c_int32), ("trace", ctypes.c_int32), ("v6_cu_withhold", ctypes.c_int32)]


# Experiment overrides: the LIBRARY reads no environment variable for its tuning (de_set_tuning); this layer does, once per context, for the
# tools and tests that sweep knobs from the shell.  environment name -> (de_tuning field, index or None)
ENV_TUNING = {
    "DE_KERNEL": ("kernel_variant", None), "DE_SLOTS": ("launch_slots", None), "DE_BIG_SLOTS": ("big_launch_slots", None),
    "DE_AUTO_V6_MIN_ITEMS": ("v6_min_paths", None), "DE_V6_STATS": ("v6_stats", None), "DE_AUTO_TRACE": ("trace", None),
    "DE_V6_AREA_ST": ("v6_service_area", 0), "DE_V6_AREA_GAS": ("v6_service_area", 1), "DE_V6_AREA_CLOUD": ("v6_service_area", 2),
    "DE_V6_SVC_ST": ("v6_service_lanes", 0), "DE_V6_SVC_GAS": ("v6_service_lanes", 1), "DE_V6_SVC_CLOUD": ("v6_service_lanes", 2),
    "DE_V6_YIELD": ("v6_yield_max", None), "DE_V6_ELSEWHERE": ("v6_elsewhere_min", None), "DE_V6_RETRY": ("v6_retry", None),
    "DE_V6_ENTER_MIN": ("v6_enter_min", None), "DE_V6_FLAT_MIN": ("v6_flat_min", None), "DE_V6_FLAT_AGAIN": ("v6_flat_again", None), "DE_V6_BANDS": ("v6_bands", None),
    "DE_V6_TAIL": ("v6_tail_levels", None), "DE_V6_TAIL_EXPORT0": ("v6_tail_export", 0), "DE_V6_TAIL_EXPORT1": ("v6_tail_export", 1),
    "DE_V6_TAIL_MIN_PATHS": ("v6_tail_min_paths", None), "DE_V6_TAIL_ALONE": ("v6_tail_when_alone", None), "DE_V6_TAIL_GRID0": ("v6_tail_grid", 0), "DE_V6_TAIL_GRID1": ("v6_tail_grid", 1),
    "DE_V2_THR": ("v2_pend", None), "DE_V2_A": ("v2_heavy", None), "DE_V2_B": ("v2_b", None), "DE_V2_G": ("v2_gas", None),
    "DE_V6_CU_WITHHOLD": ("v6_cu_withhold", None), "DE_V2_CHUNK": ("v2_chunk", None), "DE_V2_WPC": ("v2_waves_per_cu", None), "DE_V2_MAX_SPP": ("v2_max_spp", None),
}


def apply_env_tuning(handle):
    """Read the experiment overrides above from the environment and hand them to the context in one de_set_tuning call (nothing is set: no call)."""
    found = {k: os.environ[k] for k in ENV_TUNING if os.environ.get(k, "") != ""}
    if not found:
        return
    L = load()
    t = DeTuning()
    check(L.de_get_tuning(handle, ctypes.byref(t)))
    for name, text in found.items():
        field, i = ENV_TUNING[name]
        value = 1 if (name == "DE_AUTO_TRACE" and not text.lstrip("-").isdigit()) else int(text)
        if i is None:
            setattr(t, field, value)
        else:
            getattr(t, field)[i] = value
    if "DE_SLOTS" in found and "DE_BIG_SLOTS" not in found:
        t.big_launch_slots = min(t.big_launch_slots, t.launch_slots)
    check(L.de_set_tuning(handle, ctypes.byref(t)))


_lib = None
ABI_VERSION = 6   # DE_ABI_VERSION of include/digital_earth.h


class NativeLibraryError(RuntimeError):
    pass


def load():
    """Load libdigitalearth_hip.so and type every entry point.  Raises NativeLibraryError if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # The launch slots of a context (de_accumulate) are HIP streams; the runtime multiplexes streams onto GPU_MAX_HW_QUEUES
    # hardware queues (default 4) and launches that share a queue run one after the other.  16 queues let the 8 slots of a
    # context run side by side (measured, tools/one_spp.py: 64 x accumulate(1) takes 2.3x the time of accumulate(64) with 4
    # queues, 1.3x with 16).  Only a default: an exported value wins, and it has no effect once HIP is initialised.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    try:
        L = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise NativeLibraryError("cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in list(SYMBOLS.items()) + list(DEBUG_SYMBOLS.items()):
        try:
            fn = getattr(L, name)
        except AttributeError:
            raise NativeLibraryError("%s does not export %s" % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in LEGACY_SYMBOLS.items():
        if hasattr(L, name):
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
    if L.de_abi_version() != ABI_VERSION:
        raise NativeLibraryError("%s has ABI version %d, this binding expects %d: rebuild it" % (LIB_PATH, L.de_abi_version(), ABI_VERSION))
    _lib = L
    return L


class DigitalEarthError(RuntimeError):
    def __init__(self, code, message):
        RuntimeError.__init__(self, "libdigitalearth_hip error %d: %s" % (code, message))
        self.code = code


def check(rc):
    if rc != 0:
        raise DigitalEarthError(rc, load().de_last_error().decode(errors="replace"))
