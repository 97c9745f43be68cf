# usage: v3_run.sh <spp> : per-kernel totals of the second frame of tools/frame.py under rocprofv3 --kernel-trace (env DE_V3_* passes through)
# (the frame driver runs the kernel variant of DE_KERNEL: export DE_KERNEL=3 for the pipeline alone)
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/v3prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/v3prof -o s -- python3 $GRAFT_REPO_ROOT/tools/frame.py ${1:-64} 2>&1 | grep "^ms"
python3 - <<PY
import csv,glob,collections
for f in glob.glob("/tmp/v3prof/**/*kernel_trace.csv", recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if "v3_" in r["Kernel_Name"] or "render_kernel_v2" in r["Kernel_Name"] or "accumulate" in r["Kernel_Name"]]
    rows.sort(key=lambda r:int(r["Start_Timestamp"]))
    rows=rows[len(rows)//2:]     # second frame
    t0=int(rows[0]["Start_Timestamp"]); t1=max(int(r["End_Timestamp"]) for r in rows)
    busy=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows)
    print("frame span %.1f ms, sum of kernel durations %.1f ms, launches %d" % ((t1-t0)/1e6, busy/1e6, len(rows)))
    per=collections.defaultdict(list)
    for r in rows: per[r["Kernel_Name"].split("(")[0].replace("void ","")[:26]].append(((int(r["Start_Timestamp"])-t0)/1e6, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6))
    for k,v in per.items():
        print("%-28s total %7.1f ms | start/dur: %s" % (k, sum(d for _,d in v), " ".join("%.0f/%.1f"%x for x in v[:12])))
PY
