"""Build libdigitalearth_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
OUT = os.path.join(_HERE, "libdigitalearth_hip.so")
# The same sources with -DDE_LEGACY_VARIANTS: the product plus the kernel families it no longer runs (per-lane loops, the wavefront pipeline and its
# launch policy, the HBM-queue scheduler) — three more statements of path_tracer for the cross-check tests (tests/legacy/, loaded through DE_LIB_PATH).
OUT_LEGACY = os.path.join(_HERE, "libdigitalearth_hip_legacy.so")
SOURCES = ["de_api.hip", "de_fast.hip"]   # de_fast.hip: render_kernel_v6 once more, on the hardware transcendentals (DE_FLAG_FAST_MATH).  de_api.hip is one translation unit: de_api.hip -> de_rccl.h -> de_launch.h -> de_host_consts.h -> de_context.h -> the kernels
DEPS = ["de_api.hip", "de_fast.hip", "de_rccl.h", "de_launch.h", "de_host_consts.h", "de_context.h", "render_kernel.hip", "render_kernel_v2.hip", "de_stages.h", "render_kernel_v6.hip",
        "aux_kernels.hip", "de_kernels.h", "de_device.h", "de_math.h", os.path.join("..", "..", "include", "digital_earth.h"), os.path.join("..", "..", "include", "digital_earth_debug.h")]
DEPS_LEGACY = DEPS + [os.path.join("legacy", f) for f in ("render_kernel_v3.hip", "render_kernel_v5.hip", "de_launch_legacy.h", "de_ctx_legacy_members.inc")]
# -ffp-contract=off is part of the arithmetic contract (de_math.h): fused operations only where written.
# -fno-slp-vectorize: the SLP vectoriser packs pairs of scalar f32 operations into v_pk_* instructions; on gfx950 the
# moves that build the register pairs cost more than the packing saves in this kernel (measured: 358 -> 334 ms per frame).
# -mllvm -enable-ipra -fno-optimize-sibling-calls (round 6): LLVM's interprocedural register allocation includes the "no callee-saved registers"
# optimisation for functions with internal linkage that are never tail-called — the stage functions of render_kernel_v6 (static, called from the kernel's
# scheduler loop only).  Without it every stage entry saved and restored up to 24 callee-saved VGPRs through scratch: 48 scratch operations x 13.9 M stage
# entries per BASELINE cfg2 frame = 140 GB of fabric traffic (a third of all of it) for values nobody keeps.  -fno-optimize-sibling-calls keeps the `tail`
# marker off the calls (the optimisation refuses tail-called functions).  Measured, same box: cfg2 104.7 -> 100.9 ms per step, WRITE_SIZE 94.6 -> 49.2 GB,
# L2 misses 2.61e9 -> 2.03e9 per frame (profiles/experiments/README.md, round 6).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-fPIC",
         "-shared", "-Wno-unused-value", "-mllvm", "-enable-ipra", "-fno-optimize-sibling-calls"]


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


def build_legacy(force=False, verbose=False):
    if not force and os.path.exists(OUT_LEGACY) and all(os.path.getmtime(os.path.join(CSRC, d)) <= os.path.getmtime(OUT_LEGACY) for d in DEPS_LEGACY):
        return OUT_LEGACY
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-DDE_LEGACY_VARIANTS"] + os.environ.get("DE_EXTRA_FLAGS", "").split() + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", OUT_LEGACY]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return OUT_LEGACY


# Geometry variants of render_kernel_v6 for tests/test_gpu_round5.py::test_ring_protocol_on_small_pools: the same sources with -D flags
# (tiny slot pools, a ring with no slack), loaded through DE_LIB_PATH by a child process; never the product library.
TEST_VARIANTS = (("v6_p64", ["-DDE_V6_P=64u", "-DDE_V6_CAP=65u"]), ("v6_p128", ["-DDE_V6_P=128u", "-DDE_V6_CAP=129u"]), ("v6_cap_p1", ["-DDE_V6_CAP=1121u"]))


def build_test_variants(force=False, verbose=False):
    root = os.path.dirname(_HERE)
    out_dir = os.path.join(root, "build", "ab")
    os.makedirs(out_dir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    newest = max(os.path.getmtime(os.path.join(CSRC, d)) for d in DEPS)
    procs = []
    for name, flags in TEST_VARIANTS:
        out = os.path.join(out_dir, name + ".so")
        if not force and os.path.exists(out) and os.path.getmtime(out) >= newest:
            continue
        cmd = [hipcc] + FLAGS + flags + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", out]
        if verbose:
            print(" ".join(cmd))
        procs.append((name, subprocess.Popen(cmd, cwd=CSRC)))      # side by side: three ~45 s compiles on 8 cores
    for name, pr in procs:
        if pr.wait() != 0:
            raise RuntimeError("building the test variant %s failed" % name)


if __name__ == "__main__":
    build(force=True, verbose=True)
