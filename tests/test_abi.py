"""The C-ABI shared library loads and exports every symbol include/digital_earth.h declares (no compute without a GPU),
and the product path fails loudly — never falls back — when no device is present."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="digital_earth.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(de_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from digital_earth_amd import _native
    assert declared_symbols() == sorted(_native.SYMBOLS.keys())
    # the measurement / experiment / test hooks have a header of their own: same library, same discipline
    assert declared_symbols("digital_earth_debug.h") == sorted(_native.DEBUG_SYMBOLS.keys())
    assert not set(_native.SYMBOLS) & set(_native.DEBUG_SYMBOLS)


def test_the_binders_header_stays_small():
    """What a binder of the hot path has to read: at most 40 entry points, none of them a debug hook or a knob that only experiments turn."""
    names = declared_symbols()
    assert len(names) <= 40, len(names)
    assert not [n for n in names if n.startswith("de_debug_") or n in ("de_set_wave_budget", "de_set_memory_budget", "de_get_memory_use")]


def test_library_exports_every_declared_symbol():
    from digital_earth_amd import _native, build
    build.build()
    lib = _native.load()
    for name in declared_symbols() + declared_symbols("digital_earth_debug.h"):
        assert hasattr(lib, name), name
    # the library, the header and the binding carry the same ABI version (a stale .so of another round is refused by load())
    text = open(os.path.join(ROOT, "include", "digital_earth.h")).read()
    header_version = int(re.search(r"#define\s+DE_ABI_VERSION\s+(\d+)", text).group(1))
    assert lib.de_abi_version() == header_version == _native.ABI_VERSION == 6


def test_struct_layouts_match_header():
    from digital_earth_amd import _native
    from oracle import oracle_binding as ob
    # de_params: 9+7 floats, int, 4 floats, uint, float, int, 7 ints = 31 x 4 bytes; de_counters: 16 x 8 bytes
    assert ctypes.sizeof(_native.DeParams) == 31 * 4 == ctypes.sizeof(ob.DeParams)
    assert ctypes.sizeof(_native.DeCounters) == 16 * 8
    assert _native.DeParams.flags.offset == 84 and _native.DeParams.selected_crf.offset == 64


def test_tuning_struct_matches_header():
    """de_tuning (the one way to tune the library since ABI 5): the ctypes mirror has the header's fields, in the header's order, with the header's types — a
    field added on one side only would shift every later knob silently (struct_bytes only catches a size change)."""
    from digital_earth_amd import _native
    text = open(os.path.join(ROOT, "include", "digital_earth.h")).read()
    body = re.search(r"typedef struct de_tuning \{(.*?)\} de_tuning;", text, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    ctype = {"uint32_t": ctypes.c_uint32, "int32_t": ctypes.c_int32, "uint64_t": ctypes.c_uint64}
    want = []
    for decl in body.split(";"):
        m = re.match(r"\s*(uint32_t|int32_t|uint64_t)\s+(.*)", decl.strip(), re.S)
        if not m:
            continue
        for name in m.group(2).split(","):
            a = re.match(r"\s*(\w+)(?:\[(\d+)\])?\s*$", name)
            want.append((a.group(1), ctype[m.group(1)] * int(a.group(2)) if a.group(2) else ctype[m.group(1)]))
    got = list(_native.DeTuning._fields_)
    assert [n for n, _ in got] == [n for n, _ in want]
    for (n, a), (_, b) in zip(got, want):
        assert ctypes.sizeof(a) == ctypes.sizeof(b) and getattr(a, "_length_", 1) == getattr(b, "_length_", 1), n
    # every environment override of the Python layer names a field of the struct
    fields = {n for n, _ in got}
    assert all(f in fields for f, _ in _native.ENV_TUNING.values())


def test_no_silent_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from digital_earth_amd import _native
    from digital_earth_amd.renderer import Renderer
    with pytest.raises(_native.DigitalEarthError) as e:
        Renderer((64, 64), (0, 1, 0), texture_source="constant")
    assert e.value.code == -2 and "no CPU path" in str(e.value)


def test_product_does_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "digital_earth_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(d, f), errors="replace").read()
                assert "oracle_binding" not in text and "libde_oracle" not in text and "deo_" not in text, os.path.join(d, f)
