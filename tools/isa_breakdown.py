#!/usr/bin/env python3
"""Static instruction breakdown of one kernel by source line (needs an .s built with -gline-tables-only).

  hipcc ... --cuda-device-only -gline-tables-only -S de_api.hip -o de_g.s
  python tools/isa_breakdown.py de_g.s _Z16render_kernel_v2ILb0ELi0E [lo hi]

Prints VALU / SALU / LDS / VMEM counts per (file, source function) using the innermost inlined .loc, and — for lines
lo..hi of render_kernel_v2.hip — per basic block in layout order.  Static counts: every hot-loop body is laid out once, so
they approximate the per-trip dynamic counts of that body."""
import re, sys, collections, bisect
path, kern = sys.argv[1], sys.argv[2]
files = {}
src_funcs = {}   # file -> sorted list of (start_line, name)
def load_funcs(fname, full):
    out = []
    try:
        for i, l in enumerate(open(full), 1):
            m = re.match(r'\s*(?:template\s*<[^>]*>\s*)?(?:DE_DEV|__global__|static|inline)\b[^;(]*?\b([A-Za-z_][A-Za-z0-9_]*)\s*\(', l)
            if m: out.append((i, m.group(1)))
            m = re.match(r'\s*auto\s+([A-Za-z_0-9]+)\s*=\s*\[', l)
            if m: out.append((i, "lambda:" + m.group(1)))
    except OSError:
        pass
    return out
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(kern) and l.rstrip().endswith(tuple([":"])) or (l.startswith(kern) and ":" in l.split(";")[0]))
cnt = collections.Counter(); per_line = collections.Counter()
cur = (0, 0)
for l in lines[:start]:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', l)
    if m: files[int(m.group(1))] = (m.group(3), m.group(2) + "/" + m.group(3))
for fid, (fn, full) in files.items():
    src_funcs[fid] = load_funcs(fn, full)
def func_of(fid, line):
    fl = src_funcs.get(fid, [])
    k = bisect.bisect_right([s for s, _ in fl], line) - 1
    return fl[k][1] if k >= 0 else "?"
def kind(op):
    if op.startswith("v_"): return "VALU"
    if op.startswith("s_"): return "SALU"
    if op.startswith("ds_"): return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "VMEM"
    return None
total = collections.Counter()
for l in lines[start + 1:]:
    if l.startswith(".Lfunc_end"): break
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m: cur = (int(m.group(1)), int(m.group(2))); continue
    t = l.strip().split()
    if not t or t[0].startswith((".", ";")) or t[0].endswith(":"): continue
    k = kind(t[0])
    if not k: continue
    fid, line = cur
    fname = files.get(fid, ("?", ""))[0]
    cnt[(fname, func_of(fid, line), k)] += 1
    total[k] += 1
    if fname == "render_kernel_v2.hip": per_line[(line, k)] += 1
print("totals", dict(total))
agg = collections.defaultdict(collections.Counter)
for (f, fn, k), n in cnt.items(): agg[(f, fn)][k] += n
for (f, fn), c in sorted(agg.items(), key=lambda x: -x[1]["VALU"]):
    print("%-22s %-34s VALU %5d SALU %5d LDS %4d VMEM %4d" % (f, fn, c["VALU"], c["SALU"], c["LDS"], c["VMEM"]))
if len(sys.argv) > 4:
    lo, hi = int(sys.argv[3]), int(sys.argv[4])
    for ln in range(lo, hi + 1):
        v = per_line.get((ln, "VALU"), 0); s = per_line.get((ln, "SALU"), 0)
        if v or s: print("  line %4d VALU %4d SALU %4d" % (ln, v, s))
