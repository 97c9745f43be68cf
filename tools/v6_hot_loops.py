#!/usr/bin/env python3
"""Static check of render_kernel_v6's hot loops (hipcc cross-compiles: no GPU needed): for the three loop stages, the instructions of the step body
— the basic blocks between the generator's multiply and the loop's back edge — and whether any of them touches scratch.  At 80 VGPRs the register
allocator is one live value away from reloading a map's base pointer inside the tracking loop (round 5: +5 % frame time, invisible in the kernel's
resource line, which only counts the callee-saved registers at the stage functions' entries).   usage: tools/v6_hot_loops.py [asm file]"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (internal linkage since round 6 — the no-CSR optimisation of -mllvm -enable-ipra needs it —: _ZL...)
STAGES = {"sphere trace": "_ZL11v6_stage_stILb0ELb0EEvv", "gas": "_ZL12v6_stage_gasILb0ELb0EEvv", "cloud": "_ZL14v6_stage_cloudILb0ELb0EEvv"}
FLAT = {"event": "_ZL14v6_stage_eventILb0ELb0EEvv", "surface": "_ZL13v6_stage_surfILb0ELb0EEvv", "vol finish": "_ZL14v6_stage_scfinILb0ELb0EEvv", "surf finish": "_ZL16v6_stage_surffinILb0ELb0EEvv"}


def mid_scratch(lines):
    """scratch accesses of a function that are not the callee-saved registers' save at entry / restore at exit"""
    sc = [i for i, l in enumerate(lines) if "scratch_" in l]
    first_branch = next((i for i, l in enumerate(lines) if l.startswith("\ts_cbranch") or l.startswith("\ts_branch")), len(lines))
    last_label = max((i for i, l in enumerate(lines) if l.startswith(".LBB")), default=0)
    return len([i for i in sc if first_branch < i < last_label])


def assemble(out):
    sys.path.insert(0, ROOT)
    from digital_earth_amd import build
    flags = [f for f in build.FLAGS if f not in ("-shared", "-fPIC")]
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags + os.environ.get("DE_EXTRA_FLAGS", "").split() + ["-S", "--cuda-device-only", os.path.join(build.CSRC, "de_api.hip"), "-o", out]
    subprocess.run(cmd, cwd=build.CSRC, check=True, capture_output=True)


def hot_loops(path):
    txt = open(path).read()
    funcs = {}
    for f in re.split(r"\n(?=_Z[\w]+:\s+; @)", txt):
        m = re.match(r"(_Z\w+):", f)
        if m: funcs[m.group(1)] = f
    res = {}
    for stage, sym in STAGES.items():
        lines = funcs[sym].split("\n")
        # the step body: from the label before the FIRST generator multiply (the free-flight draw / the sphere trace's first vector op after the
        # loop head) to the next unconditional back edge
        # (the sphere trace draws nothing: its step is found by its map tap)
        if stage == "sphere trace":
            i0 = next(i for i, l in enumerate(lines) if "load_dword v" in l and "scratch" not in l and "global_load_dwordx" not in l)
            while i0 > 0 and "v_rsq_f32" not in lines[i0]: i0 -= 1
        else:
            i0 = next(i for i, l in enumerate(lines) if "v_mul_lo_u32" in l)
        a = i0
        while a > 0 and not lines[a].startswith(".LBB"): a -= 1
        b = i0
        while b < len(lines) - 1 and not lines[b].startswith("\ts_branch"): b += 1
        seg = lines[a:b + 1]
        res[stage] = dict(mid_scratch=mid_scratch(lines), valu=sum(l.startswith("\tv_") for l in seg), salu=sum(l.startswith("\ts_") for l in seg),
                          scratch=sum("scratch_" in l for l in seg), flat=sum("flat_load" in l for l in seg), glob=sum("global_load" in l for l in seg),
                          waits=sum("s_waitcnt" in l for l in seg), vmov=sum("v_mov_b32" in l for l in seg), nops=sum("s_nop" in l for l in seg))
    for stage, sym in FLAT.items():
        res[stage] = dict(mid_scratch=mid_scratch(funcs[sym].split("\n")))
    return res


if __name__ == "__main__":
    if len(sys.argv) > 1:
        path = sys.argv[1]
    else:
        path = os.path.join(tempfile.mkdtemp(), "de.s")
        assemble(path)
    for stage, r in hot_loops(path).items():
        if "valu" not in r:
            print("%-13s scratch accesses between the function's first branch and its last block: %d" % (stage, r["mid_scratch"]))
            continue
        print("%-13s step body (approximate cut): %3d VALU (%d moves, %d s_nop) %3d SALU, %d waits, loads: %d global %d flat; scratch accesses between the function's first branch and its last block: %d" %
              (stage, r["valu"], r["vmov"], r["nops"], r["salu"], r["waits"], r["glob"], r["flat"], r["mid_scratch"]))
