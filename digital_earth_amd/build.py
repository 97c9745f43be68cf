"""Build libdigitalearth_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
OUT = os.path.join(_HERE, "libdigitalearth_hip.so")
SOURCES = ["de_api.hip"]   # single translation unit: includes render_kernel.hip and aux_kernels.hip
DEPS = ["de_api.hip", "render_kernel.hip", "render_kernel_v2.hip", "render_kernel_v3.hip", "render_kernel_v5.hip", "render_kernel_v6.hip", "aux_kernels.hip", "de_kernels.h", "de_device.h", "de_math.h",
        os.path.join("..", "..", "include", "digital_earth.h")]
# -ffp-contract=off is part of the arithmetic contract (de_math.h): fused operations only where written.
# -fno-slp-vectorize: the SLP vectoriser packs pairs of scalar f32 operations into v_pk_* instructions; on gfx950 the
# moves that build the register pairs cost more than the packing saves in this kernel (measured: 358 -> 334 ms per frame).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-fPIC",
         "-shared", "-Wno-unused-value"]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("DE_EXTRA_FLAGS", "").split()      # experiments only (-DDE_V6_WAVES=16 ...): the shipped library is built without
    cmd = [hipcc] + FLAGS + extra + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return OUT


if __name__ == "__main__":
    build(force=True, verbose=True)
