"""The hot kernel's resource budget, checked at compile time (hipcc cross-compiles without a GPU).

render_kernel_v2 is sized to exactly fill a CU at 5 waves per SIMD: 96 VGPRs, <= 8 KB of LDS per one-wave workgroup -> 20
workgroups in 160 KB, and at most a handful of spilled values (loop-invariant constants reloaded in transition code; a
build whose guards cost 36 spills ran 30 % slower).  A change that silently breaks one of these shows up
here, not in a later profile."""
import os
import re
import shutil
import subprocess

import pytest

from digital_earth_amd import build

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not available")
def test_render_kernel_v2_resource_budget(tmp_path):
    flags = [f for f in build.FLAGS if f not in ("-shared", "-fPIC")]
    cmd = [HIPCC] + flags + ["-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
                             os.path.join(build.CSRC, "de_api.hip"), "-o", str(tmp_path / "de.s")]
    out = subprocess.run(cmd, cwd=build.CSRC, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    text = out.stderr
    # the accumulate-mode instantiations: <CLAMP = false/true, MODE = 0>
    blocks = re.findall(r"Function Name: (_Z16render_kernel_v2ILb[01]ELi0EE\S*)(.*?)LDS Size \[bytes/block\]: (\d+)", text, re.S)
    assert len(blocks) == 2, "expected the two MODE 0 instantiations of render_kernel_v2"
    for name, body, lds in blocks:
        vgprs = int(re.search(r"VGPRs: (\d+)", body).group(1))
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", body).group(1))
        occupancy = int(re.search(r"Occupancy \[waves/SIMD\]: (\d+)", body).group(1))
        spills = int(re.search(r"VGPRs Spill: (\d+)", body).group(1))
        assert scratch <= 32 and spills <= 8, (name, scratch, spills)
        assert vgprs <= 96 and occupancy >= 5, (name, vgprs, occupancy)
        assert int(lds) * 20 <= 160 * 1024, (name, lds)


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not available")
def test_render_kernel_v6_budget_and_hot_loops(tmp_path):
    """render_kernel_v6 at its geometry (80 VGPRs = six waves per SIMD, 80 KB of LDS = two workgroups per CU) and, what the kernel's resource
    line does not show, no scratch access inside any stage function besides the callee-saved registers at its entry and exit: at 80 VGPRs the
    allocator is one live value away from reloading a map's base pointer inside a tracking loop (round 5: +5 % frame time)."""
    import importlib.util
    flags = [f for f in build.FLAGS if f not in ("-shared", "-fPIC")]
    asm = str(tmp_path / "de.s")
    out = subprocess.run([HIPCC] + flags + ["-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", os.path.join(build.CSRC, "de_api.hip"), "-o", asm],
                         cwd=build.CSRC, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    for inst in ("_Z9v6_kernelILb0ELb0ELb0EE", "_Z9v6_kernelILb0ELb0ELb1EE"):      # the frame's launch and its tail launches
        m = re.search(r"Function Name: (" + inst + r"\S*)(.*?)LDS Size \[bytes/block\]: (\d+)", out.stderr, re.S)
        assert m, inst + " not found"
        body, lds = m.group(2), int(m.group(3))
        assert int(re.search(r"VGPRs: (\d+)", body).group(1)) <= 80 and int(re.search(r"Occupancy \[waves/SIMD\]: (\d+)", body).group(1)) >= 6
        assert lds * 2 <= 160 * 1024
    spec = importlib.util.spec_from_file_location("v6_hot_loops", os.path.join(os.path.dirname(build.CSRC), "..", "tools", "v6_hot_loops.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for stage, r in mod.hot_loops(asm).items():
        assert r["mid_scratch"] == 0, (stage, r)
    # Round 6: the stage functions have internal linkage and the build carries -mllvm -enable-ipra -fno-optimize-sibling-calls, so LLVM's no-CSR optimisation
    # applies — no stage function saves callee-saved VGPRs through scratch any more (it was 13-24 stores + as many loads per entry, 13.9 M entries per cfg2 frame:
    # a third of the kernel's fabric traffic).  A flag or a linkage that gets lost shows here.
    text = open(asm).read()
    seen = 0
    for f in re.split(r"\n(?=_Z[\w]+:\s+; @)", text):
        m = re.match(r"(_ZL\d+v6_stage_[a-z]+ILb[01]ELb0EE\w+):", f)      # the shipped instantiations (the statistics build spills a few of its own)
        if not m:
            continue
        seen += 1
        assert len(re.findall(r"scratch_store", f)) <= 4 and len(re.findall(r"scratch_load", f)) <= 4, (m.group(1), len(re.findall(r"scratch_", f)))
    assert seen >= 18      # nine stage functions x at least the two product instantiations (REPEAT / CLAMP)
