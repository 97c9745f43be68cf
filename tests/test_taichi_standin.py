"""The Taichi stand-in that executes the reference's Python for tests/golden/ref_*.npz (tools/ref_fixtures/standin/): its statements
about Taichi's LANGUAGE semantics, written down as executable checks.  These are the assumptions a reader may want to dispute
(tools/ref_fixtures/README.md); each test names the Taichi behaviour it encodes.  No reference code is needed here: the kernels
below are written for the test (as a module file: like Taichi, the stand-in reads a kernel's source and does not do closures)."""
import importlib.util
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STANDIN = os.path.join(ROOT, "tools", "ref_fixtures", "standin")

KERNELS = '''
import taichi as ti
from taichi.math import *

@ti.func
def typed_locals(a: ti.f32):
    x = 0.1
    y = x * 9.0                       # f32(0.1) * 9 in f32: 0.90000004
    k = 2
    k = 3.9                           # stays i32: truncates
    z = 0.1 * 9.0 + a                 # 0.1 * 9.0 folded in double (0.9), then added to an f32
    return x, y, k, z

@ti.func
def callee(r: vec2, s: ti.f32):
    r.x *= 2.0
    r += vec2(1.0, 1.0)
    return r, s

@ti.func
def caller():
    v = vec2(1.0, 2.0)
    out, s = callee(v, 0.1)
    return v, out, s

@ti.func
def helper(x):
    return x + 1.0

@ti.func
def shadowing(x):
    helper = helper(x)                # pathtracer.py:405 `land_normal = land_normal(...)`
    return helper * 2.0

@ti.func
def count(res):
    n = 0
    for _ in range(0, log2(res)):     # lib/colour.py:26
        n += 1
    return n
'''


@pytest.fixture(scope="module")
def env(tmp_path_factory):
    sys.path.insert(0, STANDIN)
    try:
        import taichi as t
        assert os.path.dirname(t.__file__).startswith(STANDIN)      # the stand-in, not a real Taichi
        path = str(tmp_path_factory.mktemp("standin") / "standin_kernels.py")
        open(path, "w").write(KERNELS)
        spec = importlib.util.spec_from_file_location("standin_kernels", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        yield t, mod
    finally:
        sys.path.remove(STANDIN)
        for k in [k for k in sys.modules if k == "taichi" or k.startswith("taichi.")]:
            del sys.modules[k]


def test_locals_take_the_default_types_and_keep_them(env):
    """default_fp = f32: `x = 0.1` is an f32 variable, arithmetic on it rounds to f32; re-assignment casts to the variable's type
    (an i32 variable assigned a float truncates); Python-float-only sub-expressions are folded in double first."""
    _, k = env
    x, y, kk, z = k.typed_locals(0.0)
    assert isinstance(x, np.float32) and x == np.float32(0.1)
    assert isinstance(y, np.float32) and y == np.float32(0.1) * np.float32(9.0)
    assert kk == 3 and not isinstance(kk, (float, np.floating))
    assert z == np.float32(0.1 * 9.0) and z != y


def test_annotated_arguments_are_passed_by_value(env):
    """A vecN-annotated argument is a copy: `rand.x *= ...` inside the callee (lib/sampling.py:42-43) must not touch the caller's
    vector; ti.f32 arguments are cast on entry."""
    _, k = env
    v, out, s = k.caller()
    assert [float(c) for c in v] == [1.0, 2.0] and [float(c) for c in out] == [3.0, 3.0]
    assert isinstance(s, np.float32) and s == np.float32(0.1)


def test_vector_methods_follow_taichi_lang_matrix(env):
    """sum() is a left fold, dot = (a*b).sum(), norm = sqrt(norm_sqr), normalized = (1 / norm) * v (NOT v / norm), mat3(v, v, v) =
    rows, M @ v = matrix . column, v @ M = row . matrix, vec3(scalar, vec2) concatenates, swizzles."""
    t, _ = env
    vec3, mat3 = t.math.vec3, t.math.mat3
    a = vec3(1e8, 1.0, -1e8)
    assert a.sum() == np.float32(np.float32(np.float32(1e8) + np.float32(1.0)) + np.float32(-1e8))      # 0, not 1
    v = vec3(3.0, 5.0, 7.0)
    inv = np.float32(1.0) / np.sqrt(np.float32(np.float32(np.float32(9.0) + np.float32(25.0)) + np.float32(49.0)))
    assert [float(x) for x in v.normalized()] == [float(inv * np.float32(c)) for c in (3.0, 5.0, 7.0)]
    m = mat3(vec3(1.0, 2.0, 3.0), vec3(4.0, 5.0, 6.0), vec3(7.0, 8.0, 10.0))
    assert [float(x) for x in (m @ vec3(1.0, 0.0, 0.0))] == [1.0, 4.0, 7.0]
    assert [float(x) for x in (vec3(1.0, 0.0, 0.0) @ m)] == [1.0, 2.0, 3.0]
    assert [float(x) for x in vec3(1.0, vec3(2.0, 3.0, 4.0).xy)] == [1.0, 2.0, 3.0]
    assert [float(x) for x in vec3(2.0, 3.0, 4.0).zyx] == [4.0, 3.0, 2.0]


def test_a_global_function_shadowed_later_resolves_like_taichi(env):
    """In Taichi's AST builder the name is the global function until the assignment creates the local; plain Python would raise
    UnboundLocalError on `helper = helper(x)`."""
    _, k = env
    assert k.shadowing(np.float32(1.5)) == np.float32(5.0)


def test_range_truncates_a_float_bound_and_mix_does_not_clamp(env):
    """`for x in range(0, log2(res))` with res = 441 runs 8 times; taichi.math.mix is x*(1-a) + y*a for any a; step(edge, x) = 1 if
    x >= edge; max(0.0, NaN) drops the NaN (maxnum, what the LLVM backends give)."""
    t, k = env
    assert k.count(441) == 8
    mix, step, vec3 = t.math.mix, t.math.step, t.math.vec3
    assert mix(np.float32(2.0), np.float32(4.0), 6.5) == np.float32(2.0) * np.float32(1.0 - 6.5) + np.float32(4.0) * np.float32(6.5)
    assert [float(x) for x in step(vec3(0.5, 0.5, 0.5), vec3(0.4, 0.5, 0.6))] == [0.0, 1.0, 1.0]
    assert t.max(0.0, np.float32("nan")) == 0.0


def test_the_random_stream_is_the_oracles(env):
    """ti.random() is Taichi-internal: the stand-in uses the oracle's xoroshiro64* stream per (seed, pixel, sample) — checked
    here against the oracle itself."""
    t, _ = env
    from oracle import oracle_binding as ob
    want = ob.evaluate("rng", np.array([[7.0, 1234.0, 3.0]], np.float32), 16)[0]
    t.rng.seed(7, 1234, 3)
    got = np.array([t.random() for _ in range(16)], dtype=np.float32)
    assert (got.view(np.uint32) == want.view(np.uint32)).all() and t.rng.draws == 16
