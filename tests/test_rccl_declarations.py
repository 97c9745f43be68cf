"""The library binds RCCL at run time with hand-declared prototypes (csrc/de_rccl.h: no link-time dependency, one RCCL instance per process).  No RCCL call
between two GPUs has run on the builder's boxes (one GPU each), so the declarations are checked here against the header the image ships: a translation unit
that includes <rccl/rccl.h> and calls every bound entry point THROUGH the hand-declared pointer types with the header's own argument types — a parameter that
is missing, in another position or of another width is a compile error; the enum values and the by-value 128-byte id are static_asserts."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = "/opt/rocm/include/rccl/rccl.h"

SRC = r"""
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstddef>
#include <type_traits>
struct UniqueId128 { char internal[128]; };
%(rccl_struct)s
static_assert(sizeof(ncclUniqueId) == sizeof(UniqueId128) && std::is_trivially_copyable<ncclUniqueId>::value, "ncclUniqueId is 128 plain bytes, passed by value");
static_assert((int)ncclFloat32 == %(f32)d && (int)ncclSum == %(sum)d, "the enum values de_rccl.h passes as ints");
static_assert(sizeof(ncclDataType_t) == sizeof(int) && sizeof(ncclRedOp_t) == sizeof(int) && sizeof(ncclResult_t) == sizeof(int), "enums travel as ints");
static_assert(sizeof(ncclComm_t) == sizeof(void*), "communicators travel as void*");
// every hand-declared pointer type must be CALLABLE like the header's function with the header's argument types, and the real function convertible to a
// pointer of the same arity (reinterpret_cast would hide an arity error: compare parameter counts through a template)
template <typename R, typename... A> constexpr int arity(R (*)(A...)) { return (int)sizeof...(A); }
#define SAME_ARITY(member, fn) static_assert(arity(Rccl{}.member) == arity(&fn), #fn)
SAME_ARITY(GetUniqueId, ncclGetUniqueId); SAME_ARITY(CommInitRank, ncclCommInitRank); SAME_ARITY(Reduce, ncclReduce); SAME_ARITY(Send, ncclSend);
SAME_ARITY(Recv, ncclRecv); SAME_ARITY(GroupStart, ncclGroupStart); SAME_ARITY(GroupEnd, ncclGroupEnd); SAME_ARITY(CommCount, ncclCommCount);
SAME_ARITY(CommDestroy, ncclCommDestroy); SAME_ARITY(CommUserRank, ncclCommUserRank); SAME_ARITY(GetErrorString, ncclGetErrorString);
template <typename A, typename B> constexpr bool same_size() { return sizeof(A) == sizeof(B); }
template <typename R1, typename... A1, typename R2, typename... A2> constexpr bool same_widths(R1 (*)(A1...), R2 (*)(A2...)) { return (same_size<A1, A2>() && ...) && sizeof(R1) == sizeof(R2); }
#define SAME_WIDTHS(member, fn) static_assert(same_widths(Rccl{}.member, &fn), #fn)
SAME_WIDTHS(GetUniqueId, ncclGetUniqueId); SAME_WIDTHS(CommInitRank, ncclCommInitRank); SAME_WIDTHS(Reduce, ncclReduce); SAME_WIDTHS(Send, ncclSend);
SAME_WIDTHS(Recv, ncclRecv); SAME_WIDTHS(CommCount, ncclCommCount); SAME_WIDTHS(CommDestroy, ncclCommDestroy); SAME_WIDTHS(CommUserRank, ncclCommUserRank);
SAME_WIDTHS(GetErrorString, ncclGetErrorString);
int main() { return 0; }
"""


@pytest.mark.skipif(not os.path.exists(HEADER) or not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")), reason="rccl.h / hipcc not in this image")
def test_hand_declared_rccl_prototypes_match_the_header(tmp_path):
    text = open(os.path.join(ROOT, "digital_earth_amd", "csrc", "de_rccl.h")).read()
    m = re.search(r"struct Rccl \{.*?\n\};", text, re.S)
    assert m, "struct Rccl not found in de_rccl.h"
    f32 = int(re.search(r"(?:DE_NCCL_FLOAT32|ncclFloat32)\D+(\d+)", text).group(1)) if re.search(r"(?:DE_NCCL_FLOAT32|ncclFloat32)\D+(\d+)", text) else 7
    ssum = int(re.search(r"(?:DE_NCCL_SUM|ncclSum)\D+(\d+)", text).group(1)) if re.search(r"(?:DE_NCCL_SUM|ncclSum)\D+(\d+)", text) else 0
    src = tmp_path / "rccl_decl.cpp"
    src.write_text(SRC % {"rccl_struct": m.group(0), "f32": f32, "sum": ssum})
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")
    out = subprocess.run([hipcc, "-std=c++17", "-fsyntax-only", "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", str(src)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
