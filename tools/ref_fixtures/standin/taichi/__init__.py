"""A minimal stand-in for the `taichi` package — FIXTURE GENERATION ONLY (tools/ref_fixtures/make.py).

Purpose: execute the reference's OWN Python text (/root/reference/pathtracer.py, renderer.py, lib/*.py) once, in the build
container, without the third-party `taichi` compiler (not installed, no network), and record what it computes.  Nothing in
the product or in the tests imports this package; only the .npz files it helped to write travel (tests/golden/ref_*.npz).

What Taichi does to a @ti.func / @ti.kernel body is re-stated here as a small AST pass (`_rewrite`) plus plain Python/numpy:

  * default_fp = f32, default_ip = i32: every local assigned from a Python literal becomes an f32 / i32 variable and KEEPS its
    type on re-assignment (`x = 1.0` -> np.float32; `x += v` casts to x's type); arguments annotated ti.f32 / float / vecN are
    cast / copied on entry (by-value), un-annotated and ti.template() arguments pass by reference.
  * arithmetic on np.float32 scalars is IEEE binary32 (+ - * / sqrt correctly rounded); Python-float-only sub-expressions are
    evaluated by Python in double and rounded when they meet an f32 value — as Taichi's AST builder does (numpy's NEP-50
    "weak scalar" rule gives exactly this).
  * Vector / Matrix methods follow taichi/lang/matrix.py: sum() is a left fold, dot = (a*b).sum(), norm = sqrt(norm_sqr),
    normalized = (1 / norm) * v, cross and matmul in the textbook order; taichi.math's mix / clamp / smoothstep / step / fract /
    log2 are its one-line definitions.
  * a name that is a global function and is later assigned in the same body (`land_normal = land_normal(...)`,
    pathtracer.py:405) resolves like in Taichi's builder: global before the assignment, local after it.
  * `range(0, log2(441))` truncates its float bound (8 iterations).

What is NOT Taichi's and is DEFINED here, the same way the repo's oracle defines it: ti.random() is the
oracle's xoroshiro64* stream keyed by (seed, pixel, sample index); Texture.sample_lod is bilinear with texel centres at
(i + 1/2) / N and REPEAT addressing, evaluated with the Vulkan-spec weight formula in f32; exp / log / sin / cos / atan2 /
asin / pow come from numpy's float32 routines (MATH = "numpy", <= a few ulp, independent of the oracle) or from the oracle's
fixed sequences (MATH = "oracle").
"""
import ast
import inspect
import itertools
import math as _math
import textwrap
import types as _pytypes

import numpy as np

np.seterr(all="ignore")

MATH = "numpy"             # or "oracle": set by make.py before the reference modules are imported
_oracle_eval = None        # callable(name, [args]) -> float, installed by make.py for MATH == "oracle"

F32 = np.float32


# ------------------------------------------------------------------------------------------------ dtypes
class DType:
    def __init__(self, name, np_type):
        self.name, self.np = name, np_type

    def __call__(self, x=0):
        return self.np(x)

    def __repr__(self):
        return "ti." + self.name


f32 = DType("f32", np.float32)
f64 = DType("f64", np.float64)
f16 = DType("f16", np.float16)
i32 = DType("i32", np.int32)
u8 = DType("u8", np.uint8)
u32 = DType("u32", np.uint32)


def _np_type(dt):
    if isinstance(dt, DType):
        return dt.np
    if dt is float:
        return np.float32
    if dt is int:
        return np.int32
    return dt


def _is_pyscalar(x):
    return isinstance(x, (int, float, bool)) and not isinstance(x, np.generic)


def _is_f32(x):
    return isinstance(x, np.floating)


# ------------------------------------------------------------------------------------------------ vectors / matrices
_SWZ = {c: i for s in ("xyzw", "rgba", "stpq") for i, c in enumerate(s)}


class Vec:
    """ti.Vector of f32 (or i32 / bool) entries."""
    __slots__ = ("a",)
    __array_ufunc__ = None          # np.float32 * Vec defers to Vec.__rmul__ instead of broadcasting over the entries

    def __init__(self, entries):
        flat = []
        for e in entries:
            if isinstance(e, Vec):
                flat.extend(e.a.tolist() if e.a.dtype != np.float32 else list(e.a))
            else:
                flat.append(e)
        if all(isinstance(e, (bool, np.bool_)) for e in flat):
            arr = np.array(flat, dtype=bool)
        elif all(isinstance(e, (int, np.integer)) and not isinstance(e, (bool, np.bool_)) for e in flat):
            arr = np.array(flat, dtype=np.int32)
        else:
            arr = np.array([np.float32(e) for e in flat], dtype=np.float32)
        object.__setattr__(self, "a", arr)

    @staticmethod
    def _wrap(arr):
        v = Vec.__new__(Vec)
        object.__setattr__(v, "a", arr)
        return v

    def copy(self):
        return Vec._wrap(self.a.copy())

    @property
    def n(self):
        return self.a.shape[0]

    # ---- element / swizzle access
    def __getitem__(self, i):
        return self.a[int(i)]

    def __setitem__(self, i, v):
        self.a[int(i)] = v

    def __getattr__(self, name):
        try:
            idx = [_SWZ[c] for c in name]
        except KeyError:
            raise AttributeError(name)
        if len(idx) == 1:
            return self.a[idx[0]]
        return Vec._wrap(self.a[idx].copy())

    def __setattr__(self, name, value):
        idx = [_SWZ[c] for c in name]
        if len(idx) == 1:
            self.a[idx[0]] = value
        else:
            vals = value.a if isinstance(value, Vec) else [value] * len(idx)
            for k, i in enumerate(idx):
                self.a[i] = vals[k]

    def __len__(self):
        return self.a.shape[0]

    def __iter__(self):
        return iter(self.a)

    # ---- arithmetic (elementwise; scalars broadcast)
    @staticmethod
    def _o(other):
        if isinstance(other, Vec):
            return other.a
        assert not isinstance(other, np.float64), "a float64 leaked into f32 arithmetic"
        return other

    def __add__(self, o): return Vec._wrap(self.a + Vec._o(o))
    def __radd__(self, o): return Vec._wrap(Vec._o(o) + self.a)
    def __sub__(self, o): return Vec._wrap(self.a - Vec._o(o))
    def __rsub__(self, o): return Vec._wrap(Vec._o(o) - self.a)
    def __mul__(self, o): return Vec._wrap(self.a * Vec._o(o))
    def __rmul__(self, o): return Vec._wrap(Vec._o(o) * self.a)
    def __truediv__(self, o): return Vec._wrap((self.a.astype(np.float32) if self.a.dtype != np.float32 else self.a) / Vec._o(o))
    def __rtruediv__(self, o): return Vec._wrap(Vec._o(o) / self.a)
    def __neg__(self): return Vec._wrap(-self.a)
    def __abs__(self): return Vec._wrap(np.abs(self.a))
    def __pow__(self, o): return pow(self, o)
    def __ge__(self, o): return Vec._wrap(self.a >= Vec._o(o))
    def __gt__(self, o): return Vec._wrap(self.a > Vec._o(o))
    def __le__(self, o): return Vec._wrap(self.a <= Vec._o(o))
    def __lt__(self, o): return Vec._wrap(self.a < Vec._o(o))

    def __matmul__(self, m):                       # row vector @ matrix: r[j] = sum_k v[k] * M[k, j]
        assert isinstance(m, Mat)
        out = []
        for j in range(m.a.shape[1]):
            acc = self.a[0] * m.a[0, j]
            for k in range(1, self.n):
                acc = acc + self.a[k] * m.a[k, j]
            out.append(acc)
        return Vec(out)

    # ---- taichi/lang/matrix.py
    def sum(self):
        ret = self.a[0]
        for i in range(1, self.n):
            ret = ret + self.a[i]
        return ret

    def dot(self, o):
        return (self * o).sum()

    def norm_sqr(self):
        return (self * self).sum()

    def norm(self, eps=0):
        return sqrt(self.norm_sqr() + eps) if eps else sqrt(self.norm_sqr())

    def normalized(self, eps=0):
        invlen = 1 / (self.norm() + eps) if eps else 1 / self.norm()
        return invlen * self

    def cross(self, o):
        a, b = self.a, o.a
        return Vec([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]])

    def __repr__(self):
        return "Vec(%s)" % ", ".join(repr(float(x)) for x in self.a)


class Mat:
    """ti.Matrix (row-major entries)."""
    __slots__ = ("a",)
    __array_ufunc__ = None

    def __init__(self, rows):
        object.__setattr__(self, "a", np.array([[np.float32(e) for e in r] for r in rows], dtype=np.float32))

    def copy(self):
        m = Mat.__new__(Mat)
        object.__setattr__(m, "a", self.a.copy())
        return m

    def __getitem__(self, ij):
        return self.a[int(ij[0]), int(ij[1])]

    def __setitem__(self, ij, v):
        self.a[int(ij[0]), int(ij[1])] = v

    def transpose(self):
        m = Mat.__new__(Mat)
        object.__setattr__(m, "a", self.a.T.copy())
        return m

    def __matmul__(self, v):                       # matrix @ column vector: r[i] = sum_k M[i, k] * v[k]
        if isinstance(v, Vec):
            out = []
            for i in range(self.a.shape[0]):
                acc = self.a[i, 0] * v.a[0]
                for k in range(1, self.a.shape[1]):
                    acc = acc + self.a[i, k] * v.a[k]
                out.append(acc)
            return Vec(out)
        raise TypeError("Mat @ %r" % type(v))


def _vec_ctor(n):
    def ctor(*args):
        if len(args) == 1 and not isinstance(args[0], Vec):
            return Vec([args[0]] * n)                     # vec3(s) broadcasts
        v = Vec(list(args))                              # concatenation: vec3(s, vec2), vec4(v.xyz, s), vec3(p.xyx)
        assert v.n == n, "vec%d from %d components" % (n, v.n)
        return v
    ctor._ti_vec = n
    ctor.__name__ = "vec%d" % n
    return ctor


vec2, vec3, vec4 = _vec_ctor(2), _vec_ctor(3), _vec_ctor(4)


def mat3(*args):
    if len(args) == 1 and isinstance(args[0], (list, tuple)):
        args = tuple(args[0])
    if len(args) == 1 and _is_pyscalar(args[0]):
        return Mat([[args[0]] * 3] * 3)                   # taichi.math.mat3(s): every entry s
    if len(args) == 3 and all(isinstance(r, Vec) for r in args):
        return Mat([list(r.a) for r in args])            # rows
    if len(args) == 9:
        return Mat([args[0:3], args[3:6], args[6:9]])
    raise TypeError("mat3%r" % (args,))


mat3._ti_mat = 3


# ------------------------------------------------------------------------------------------------ elementary functions
def _f32fun(name, *xs):
    """One f32 elementary function under the selected MATH mode."""
    if MATH == "oracle" and name in ("exp", "log", "sin", "cos", "atan2", "asin", "pow"):
        # pow with an integer-valued exponent stays with numpy: there the oracle multiplies (x*x*x, also for x < 0 — libm's pow
        # semantics, what the ti.cpu / LLVM backends give); its general pow_ is exp(y log x) and has no value for x < 0
        if not (name == "pow" and float(xs[1]) == round(float(xs[1]))):
            return np.float32(_oracle_eval(name, [float(np.float32(x)) for x in xs]))
    fn = {"exp": np.exp, "log": np.log, "sin": np.sin, "cos": np.cos, "atan2": np.arctan2, "asin": np.arcsin, "acos": np.arccos,
          "pow": np.power, "tan": np.tan, "tanh": np.tanh}[name]
    return np.float32(fn(*[np.float32(x) for x in xs]))


_PYFUN = {"exp": _math.exp, "log": _math.log, "sin": _math.sin, "cos": _math.cos, "atan2": _math.atan2, "asin": _math.asin,
          "acos": _math.acos, "pow": _math.pow, "tan": _math.tan, "tanh": _math.tanh}


def _elementwise(name, *xs):
    if any(isinstance(x, Vec) for x in xs):
        n = next(x.n for x in xs if isinstance(x, Vec))
        return Vec([_elementwise(name, *[(x[i] if isinstance(x, Vec) else x) for x in xs]) for i in range(n)])
    if all(_is_pyscalar(x) for x in xs):
        return _PYFUN[name](*[float(x) for x in xs])      # Python scope: double, like Taichi's ops on constants
    return _f32fun(name, *xs)


def exp(x): return _elementwise("exp", x)
def log(x): return _elementwise("log", x)
def sin(x): return _elementwise("sin", x)
def cos(x): return _elementwise("cos", x)
def tan(x): return _elementwise("tan", x)
def tanh(x): return _elementwise("tanh", x)
def asin(x): return _elementwise("asin", x)
def acos(x): return _elementwise("acos", x)
def atan2(y, x): return _elementwise("atan2", y, x)


def pow(x, y):                                           # noqa: A001 - taichi.math exports pow
    return _elementwise("pow", x, y)


def sqrt(x):
    if isinstance(x, Vec):
        return Vec._wrap(np.sqrt(x.a))
    if _is_pyscalar(x):
        return _math.sqrt(x)
    return np.sqrt(np.float32(x))


def floor(x):
    if isinstance(x, Vec):
        return Vec._wrap(np.floor(x.a))
    if _is_pyscalar(x):
        return float(_math.floor(x))
    return np.floor(np.float32(x))


def ceil(x):
    if isinstance(x, Vec):
        return Vec._wrap(np.ceil(x.a))
    if _is_pyscalar(x):
        return float(_math.ceil(x))
    return np.ceil(np.float32(x))


def _minmax(np_fn, py_fn, a, b):
    if isinstance(a, Vec) or isinstance(b, Vec):
        aa = a.a if isinstance(a, Vec) else a
        bb = b.a if isinstance(b, Vec) else b
        return Vec._wrap(np_fn(aa, bb).astype(np.float32))
    if _is_pyscalar(a) and _is_pyscalar(b):
        return py_fn(a, b)
    if isinstance(a, (int, np.integer)) and isinstance(b, (int, np.integer)):
        return py_fn(a, b)
    return np.float32(np_fn(np.float32(a), np.float32(b)))    # fmax / fmin: a NaN operand is dropped (maxnum)


def max(*args):                                           # noqa: A001
    r = args[0]
    for x in args[1:]:
        r = _minmax(np.fmax, lambda p, q: p if p >= q else q, r, x)
    return r


def min(*args):                                           # noqa: A001
    r = args[0]
    for x in args[1:]:
        r = _minmax(np.fmin, lambda p, q: p if p <= q else q, r, x)
    return r


def isinf(x):
    return bool(np.isinf(np.float32(x)))


def isnan(x):
    return bool(np.isnan(np.float32(x)))


def select(cond, a, b):
    if isinstance(cond, Vec):
        aa = a.a if isinstance(a, Vec) else a
        bb = b.a if isinstance(b, Vec) else b
        return Vec._wrap(np.where(cond.a, aa, bb).astype(np.float32))
    return a if cond else b


def cast(x, dt):
    t = _np_type(dt)
    if isinstance(x, Vec):
        return Vec._wrap(x.a.astype(t))
    if t in (np.int32, np.uint8, np.uint32):
        return int(x)                                     # truncation toward zero
    return t(x)


# taichi/math/mathimpl.py
pi = _math.pi
e = _math.e
inf = float("inf")
nan = float("nan")


def mix(x, y, a):
    return x * (1.0 - a) + y * a


def clamp(x, xmin, xmax):
    return max(xmin, min(xmax, x))


def smoothstep(edge0, edge1, x):
    t = clamp((x - edge0) / (edge1 - edge0), 0.0, 1.0)
    return t * t * (3.0 - 2.0 * t)


def step(edge, x):
    return cast(x >= edge, float)


def fract(x):
    return x - floor(x)


def log2(x):
    return log(x) / log(2.0)


def length(x):
    return x.norm()


def normalize(v):
    return v.normalized()


def dot(a, b):
    return a.dot(b)


def cross(a, b):
    return a.cross(b)


def distance(a, b):
    return (a - b).norm()


def sign(x):
    return np.float32(np.sign(np.float32(x)))


def radians(x):
    return x * (pi / 180.0)


def degrees(x):
    return x * (180.0 / pi)


# ------------------------------------------------------------------------------------------------ RNG (oracle's definition)
class _Rng:
    """xoroshiro64* keyed by (seed, absolute pixel, sample index): oracle/oracle_lib.h `Rng` (ti.random() is Taichi-internal)."""
    M = 0xFFFFFFFF

    def __init__(self):
        self.s0, self.s1, self.draws = 1, 0, 0

    @staticmethod
    def _fmix32(h):
        h ^= h >> 16; h = (h * 0x85EBCA6B) & 0xFFFFFFFF; h ^= h >> 13; h = (h * 0xC2B2AE35) & 0xFFFFFFFF; h ^= h >> 16
        return h

    @staticmethod
    def _rotl(x, k):
        return ((x << k) | (x >> (32 - k))) & 0xFFFFFFFF

    def seed(self, seed, pixel, sample):
        lo, hi = seed & self.M, (seed >> 32) & self.M
        a = self._fmix32(((pixel * 0x9E3779B1) & self.M) ^ lo)
        b = self._fmix32(((sample * 0x85EBCA77) & self.M) ^ hi ^ a)
        self.s0 = self._fmix32(a ^ self._rotl(b, 16) ^ 0x68E31DA4)
        self.s1 = self._fmix32((b + 0xB5297A4D) & self.M)
        if (self.s0 | self.s1) == 0:
            self.s0 = 1
        self.draws = 0

    def next_f32(self):
        r = (self.s0 * 0x9E3779BB) & self.M
        t = self.s1 ^ self.s0
        self.s0 = self._rotl(self.s0, 26) ^ t ^ ((t << 9) & self.M)
        self.s1 = self._rotl(t, 13)
        self.draws += 1
        return np.float32(r >> 8) * np.float32(2.0 ** -24)


rng = _Rng()


def random(dtype=None):
    return rng.next_f32()


# ------------------------------------------------------------------------------------------------ fields, textures
class _Field:
    def __init__(self, n, dtype, shape):
        self.n, self.np_type = n, _np_type(dtype)
        self.data = None
        self.shape = None
        self.iter_indices = None        # make.py: the subset of indices a struct-for visits, and a per-index hook
        self.iter_hook = None
        if shape is not None:
            self._alloc(shape)

    def _alloc(self, shape):
        if isinstance(shape, (int, np.integer)):
            shape = (int(shape),)
        self.shape = tuple(int(s) for s in shape)
        full = self.shape + ((self.n,) if self.n else ())
        self.data = np.zeros(full, dtype=self.np_type)

    def _idx(self, key):
        if key is None:
            return ()
        if isinstance(key, Vec):
            return tuple(int(k) for k in key.a)
        if isinstance(key, tuple):
            return tuple(int(k) for k in key)
        return (int(key),)

    def __getitem__(self, key):
        v = self.data[self._idx(key)]
        if self.n:
            return Vec._wrap(v.astype(np.float32)) if self.np_type in (np.float32, np.float16) else Vec._wrap(v.astype(np.int32))
        return np.float32(v) if self.np_type in (np.float32, np.float16) else int(v)

    def __setitem__(self, key, value):
        self.data[self._idx(key)] = value.a if isinstance(value, Vec) else value

    def __iter__(self):                                    # struct-for: `for u, v in field`
        idx = self.iter_indices if self.iter_indices is not None else itertools.product(*[range(s) for s in self.shape])
        for ij in idx:
            if self.iter_hook is not None:
                self.iter_hook(ij)
            yield ij

    def from_numpy(self, arr):
        arr = np.asarray(arr)
        assert arr.shape == self.data.shape, (arr.shape, self.data.shape)
        self.data = arr.astype(self.np_type)

    def to_numpy(self):
        return self.data.copy()

    def fill(self, v):
        self.data[...] = v


def field(dtype, shape=None):
    return _Field(0, dtype, shape)


class _VectorNS:
    def __call__(self, entries, dt=None):
        return Vec(list(entries))

    @staticmethod
    def field(n, dtype, shape=None):
        return _Field(n, dtype, shape)


Vector = _VectorNS()


class _SNode:
    def __init__(self, dims=()):
        self.dims = dims

    def dense(self, axes, sizes):
        if isinstance(sizes, (int, np.integer)):
            sizes = (sizes,)
        cur = self.dims or (1,) * len(sizes)
        return _SNode(tuple(int(a) * int(b) for a, b in zip(cur, sizes)))

    def place(self, *fields):
        for f in fields:
            f._alloc(self.dims)


root = _SNode()
ij = "ij"
i = "i"
j = "j"


class Format:
    r8, rgba8, rgba16f, rgba32f = "r8", "rgba8", "rgba16f", "rgba32f"


class Texture:
    """ti.Texture: texel (i, j) = data[j][i]; sample_lod = bilinear, centres at (i + 1/2) / N, REPEAT (or CLAMP) addressing."""
    clamp_mode = False

    def __init__(self, fmt, res):
        self.fmt, self.res = fmt, (int(res[0]), int(res[1]))
        self.data = None              # [h][w][c]: float32, or the bytes of a unorm8 format (texel value = byte / 255 in f32,
        self.unorm8 = False           # correctly rounded — what tex.store of `u8 / 255.0` leaves, renderer.py:171-210)

    def load_unorm8(self, texels_hwc):
        t = np.asarray(texels_hwc)
        assert t.dtype == np.uint8 and t.shape[:2] == (self.res[1], self.res[0]), (t.shape, self.res)
        self.data, self.unorm8 = t, True

    def load_float(self, texels_hwc):
        t = np.asarray(texels_hwc, dtype=np.float32)
        assert t.shape[:2] == (self.res[1], self.res[0]), (t.shape, self.res)
        if self.fmt == Format.rgba16f:
            t = t.astype(np.float16).astype(np.float32)
        self.data = t

    def _wrap(self, k, n):
        if Texture.clamp_mode:
            return 0 if k < 0 else (n - 1 if k >= n else k)
        return k % n

    def sample_lod(self, uv, lod):
        w, h = self.res
        x = np.float32(uv[0]) * np.float32(w) - np.float32(0.5)
        y = np.float32(uv[1]) * np.float32(h) - np.float32(0.5)
        x0, y0 = np.floor(x), np.floor(y)
        fx, fy = x - x0, y - y0
        i0, i1 = self._wrap(int(x0), w), self._wrap(int(x0) + 1, w)
        j0, j1 = self._wrap(int(y0), h), self._wrap(int(y0) + 1, h)
        one = np.float32(1.0)
        t00, t10, t01, t11 = self.data[j0, i0], self.data[j0, i1], self.data[j1, i0], self.data[j1, i1]
        if self.unorm8:
            k255 = np.float32(255.0)
            t00, t10, t01, t11 = (t.astype(np.float32) / k255 for t in (t00, t10, t01, t11))
        # Vulkan spec 16.9.3: tau = (1-a)(1-b) t00 + a(1-b) t10 + (1-a)b t01 + ab t11
        out = ((one - fx) * (one - fy)) * t00 + (fx * (one - fy)) * t10 + ((one - fx) * fy) * t01 + (fx * fy) * t11
        c = out.shape[0]
        vals = [out[k] if k < c else np.float32(0.0) for k in range(4)]
        return Vec(vals)


class _Types:
    @staticmethod
    def texture(**kw):
        return "texture"

    @staticmethod
    def rw_texture(**kw):
        return "rw_texture"


types = _Types()


def template():
    return "template"


class _Tools:
    images = {}                 # file name -> array [x][y][c] (ti.tools.imread orientation), registered by make.py

    @staticmethod
    def imread(filename, channels=0):
        return _Tools.images[filename]


tools = _Tools()


def ndrange(*dims):
    return itertools.product(*[range(int(d)) for d in dims])


def static(x):
    return x


def loop_config(**kw):
    pass


def init(*a, **kw):
    pass


vulkan = cpu = gpu = "arch"


# ------------------------------------------------------------------------------------------------ struct types
def dataclass(cls):
    ann = dict(cls.__dict__.get("__annotations__", {}))

    def __init__(self, **kw):
        for name, t in ann.items():
            n = getattr(t, "_ti_vec", None)
            object.__setattr__(self, name, Vec([0.0] * n) if n else np.float32(0.0))
        for k, v in kw.items():
            setattr(self, k, v)

    def __setattr__(self, name, value):
        t = ann[name]
        if getattr(t, "_ti_vec", None):
            object.__setattr__(self, name, value.copy())
        else:
            object.__setattr__(self, name, np.float32(value))

    return type(cls.__name__, (), {"__init__": __init__, "__setattr__": __setattr__, "_ti_struct": True})


def data_oriented(cls):
    return cls


# ------------------------------------------------------------------------------------------------ Taichi-scope semantics
def _ti_var(v):
    """A new local variable: Python literals take the default types (f32 / i32), compound values are copied."""
    if isinstance(v, (Vec, Mat)):
        return v.copy()
    if isinstance(v, tuple):
        return tuple(_ti_var(x) for x in v)
    if isinstance(v, bool) or isinstance(v, np.bool_):
        return bool(v)
    if isinstance(v, float):
        return np.float32(v)
    assert not isinstance(v, np.float64), "a float64 leaked into Taichi scope"
    return v


def _ti_assign(old, v):
    """Re-assignment keeps the variable's type (Taichi casts the value, with a warning when precision is lost)."""
    if isinstance(old, np.floating):
        return np.float32(v)
    if isinstance(old, Vec) and isinstance(v, Vec):
        return Vec._wrap(v.a.astype(old.a.dtype))
    if isinstance(old, (bool, np.bool_)):
        return bool(v)
    if isinstance(old, (int, np.integer)) and not isinstance(v, (Vec, Mat)):
        return int(v)
    return _ti_var(v)


def _ti_arg(v, kind):
    if kind == "f32":
        return np.float32(v)
    if kind == "i32":
        return int(v)
    if kind == "copy" and isinstance(v, (Vec, Mat)):
        return v.copy()
    return v


def _ti_range(*a):
    return range(*[int(x) for x in a])


_HELPERS = {"_ti_var": _ti_var, "_ti_assign": _ti_assign, "_ti_arg": _ti_arg, "_ti_range": _ti_range}


def _ann_kind(node):
    if node is None:
        return None
    src = ast.unparse(node)
    if src in ("ti.f32", "float", "ti.f64"):
        return "f32"
    if src in ("ti.i32", "int", "ti.u8", "ti.u32"):
        return "i32"
    if src in ("vec2", "vec3", "vec4", "mat3", "ti.math.vec3", "ti.math.vec2", "ti.math.vec4", "ti.math.mat3"):
        return "copy"
    return None


class _Rewriter:
    """The AST pass described in the module docstring."""

    def __init__(self, fdef, global_names):
        self.fdef = fdef
        self.global_names = global_names
        self.tmp = 0
        self.rename = set()

    # ---- pass 1: names that are used as a global before they are assigned (source order)
    def _ordered(self, node):
        if isinstance(node, ast.Assign):
            yield from self._ordered(node.value)
            for t in node.targets:
                yield from self._ordered(t)
        elif isinstance(node, ast.AugAssign):
            yield from self._ordered(node.value)
            yield from self._ordered(node.target)
        elif isinstance(node, ast.For):
            yield from self._ordered(node.iter)
            yield from self._ordered(node.target)
            for s in node.body + node.orelse:
                yield from self._ordered(s)
        else:
            yield node
            for ch in ast.iter_child_nodes(node):
                yield from self._ordered(ch)

    def find_collisions(self):
        params = {a.arg for a in self.fdef.args.args}
        assigned_anywhere = {n.id for n in ast.walk(self.fdef) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Store)}
        seen_store = set(params)
        self.local_from = {}
        for stmt in self.fdef.body:
            for n in self._ordered(stmt):
                if not isinstance(n, ast.Name):
                    continue
                if isinstance(n.ctx, ast.Store):
                    seen_store.add(n.id)
                elif n.id in assigned_anywhere and n.id not in seen_store and n.id in self.global_names:
                    self.rename.add(n.id)
        if self.rename:
            seen = set()
            for stmt in self.fdef.body:
                for n in self._ordered(stmt):
                    if isinstance(n, ast.Name) and n.id in self.rename:
                        if isinstance(n.ctx, ast.Store):
                            seen.add(n.id)
                        if n.id in seen:
                            n.id = n.id + "__local"

    # ---- pass 2: assignments
    def _call(self, fn, *args):
        return ast.Call(func=ast.Name(id=fn, ctx=ast.Load()), args=list(args), keywords=[])

    def _store_name(self, name, value, scopes):
        if any(name in s for s in scopes):
            rhs = self._call("_ti_assign", ast.Name(id=name, ctx=ast.Load()), value)
        else:
            scopes[-1].add(name)
            rhs = self._call("_ti_var", value)
        return ast.Assign(targets=[ast.Name(id=name, ctx=ast.Store())], value=rhs)

    def block(self, stmts, scopes):
        out = []
        for s in stmts:
            out.extend(self.stmt(s, scopes))
        return out

    def stmt(self, s, scopes):
        if isinstance(s, ast.Assign) and len(s.targets) == 1:
            t = s.targets[0]
            if isinstance(t, ast.Name):
                return [ast.copy_location(self._store_name(t.id, s.value, scopes), s)]
            if isinstance(t, ast.Tuple):
                self.tmp += 1
                tmp = "_ti_tmp%d" % self.tmp
                res = [ast.copy_location(ast.Assign(targets=[ast.Name(id=tmp, ctx=ast.Store())], value=s.value), s)]
                for k, el in enumerate(t.elts):
                    item = ast.Subscript(value=ast.Name(id=tmp, ctx=ast.Load()), slice=ast.Constant(value=k), ctx=ast.Load())
                    if isinstance(el, ast.Name):
                        res.append(ast.copy_location(self._store_name(el.id, item, scopes), s))
                    else:
                        res.append(ast.copy_location(ast.Assign(targets=[el], value=item), s))
                return res
            return [s]
        if isinstance(s, ast.AnnAssign) and isinstance(s.target, ast.Name) and s.value is not None:
            return [ast.copy_location(self._store_name(s.target.id, s.value, scopes), s)]
        if isinstance(s, ast.AugAssign) and isinstance(s.target, ast.Name):
            name = s.target.id
            val = ast.BinOp(left=ast.Name(id=name, ctx=ast.Load()), op=s.op, right=s.value)
            return [ast.copy_location(ast.Assign(targets=[ast.Name(id=name, ctx=ast.Store())],
                                                 value=self._call("_ti_assign", ast.Name(id=name, ctx=ast.Load()), val)), s)]
        if isinstance(s, ast.If):
            s.body = self.block(s.body, scopes + [set()])
            s.orelse = self.block(s.orelse, scopes + [set()])
            return [s]
        if isinstance(s, (ast.For, ast.While)):
            inner = set()
            if isinstance(s, ast.For):
                for n in ast.walk(s.target):
                    if isinstance(n, ast.Name):
                        inner.add(n.id)
            s.body = self.block(s.body, scopes + [inner])
            s.orelse = self.block(s.orelse, scopes + [set()])
            return [s]
        return [s]

    def run(self):
        f = self.fdef
        f.decorator_list = []
        f.returns = None
        self.find_collisions()
        prologue = []
        scope = set()
        for a in f.args.args:
            kind = _ann_kind(a.annotation)
            a.annotation = None
            scope.add(a.arg)
            if kind:
                prologue.append(ast.Assign(targets=[ast.Name(id=a.arg, ctx=ast.Store())],
                                           value=self._call("_ti_arg", ast.Name(id=a.arg, ctx=ast.Load()), ast.Constant(value=kind))))
        body = self.block(f.body, [scope])
        f.body = prologue + body

        class _Range(ast.NodeTransformer):
            def visit_Call(self, node):
                self.generic_visit(node)
                if isinstance(node.func, ast.Name) and node.func.id == "range":
                    node.func = ast.Name(id="_ti_range", ctx=ast.Load())
                return node
        _Range().visit(f)
        return f


def _rewrite(fn):
    src = textwrap.dedent(inspect.getsource(fn))
    tree = ast.parse(src)
    fdef = tree.body[0]
    assert isinstance(fdef, ast.FunctionDef), fn
    g = fn.__globals__
    _Rewriter(fdef, set(g.keys())).run()
    ast.fix_missing_locations(tree)
    ast.increment_lineno(tree, fn.__code__.co_firstlineno - 1)
    code = compile(tree, inspect.getsourcefile(fn), "exec")
    for k, v in _HELPERS.items():
        g.setdefault(k, v)
    ns = {}
    exec(code, g, ns)
    new = ns[fdef.name]
    new.__defaults__ = fn.__defaults__
    new._ti_original = fn
    return new


def func(fn):
    return _rewrite(fn)


def kernel(fn):
    return _rewrite(fn)


import builtins as _builtins  # noqa: E402

abs = _builtins.abs                                          # noqa: A001 - ti.abs

# `import taichi as ti; ti.math.vec3(...)`
from . import math  # noqa: E402,F401
