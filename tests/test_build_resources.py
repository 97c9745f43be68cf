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
