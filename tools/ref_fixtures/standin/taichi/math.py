"""`from taichi.math import *` of the stand-in (see taichi/__init__.py): the names taichi/math/mathimpl.py exports that the
reference uses.  `abs` is NOT exported by Taichi either: the reference's abs() is the Python builtin (Vec.__abs__)."""
from . import (acos, asin, atan2, ceil, clamp, cos, cross, degrees, distance, dot, e, exp, floor, fract, inf, isinf, isnan,
               length, log, log2, mat3, max, min, mix, nan, normalize, pi, pow, radians, sign, sin, smoothstep, sqrt, step, tan,
               tanh, vec2, vec3, vec4)

__all__ = ["acos", "asin", "atan2", "ceil", "clamp", "cos", "cross", "degrees", "distance", "dot", "e", "exp", "floor", "fract",
           "inf", "isinf", "isnan", "length", "log", "log2", "mat3", "max", "min", "mix", "nan", "normalize", "pi", "pow", "radians",
           "sign", "sin", "smoothstep", "sqrt", "step", "tan", "tanh", "vec2", "vec3", "vec4"]
