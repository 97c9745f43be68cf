cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/v3prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/v3prof -o s -- python3 $GRAFT_REPO_ROOT/tools/v3_frame.py ${1:-64} 3 2>&1 | grep "^ms"
python3 - <<PY
import csv,glob,collections
for f in glob.glob("/tmp/v3prof/**/*kernel_trace.csv", recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if "v3_" in r["Kernel_Name"]]
    rows.sort(key=lambda r:int(r["Start_Timestamp"]))
    rows=rows[len(rows)//2:]     # second frame
    t0=int(rows[0]["Start_Timestamp"]); t1=int(rows[-1]["End_Timestamp"])
    busy=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows)
    print("frame span %.1f ms, sum of kernel durations %.1f ms, launches %d" % ((t1-t0)/1e6, busy/1e6, len(rows)))
    per=collections.defaultdict(list)
    for r in rows: per[r["Kernel_Name"].split("(")[0].replace("void ","")[:22]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
    for k,v in per.items():
        print("%-24s total %7.1f ms | per call: %s" % (k, sum(v), " ".join("%.2f"%x for x in v[:24])))
PY
