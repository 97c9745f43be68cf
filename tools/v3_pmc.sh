cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/v3pmc
# (the frame driver runs the kernel variant of DE_KERNEL: export DE_KERNEL=3 for the pipeline alone)
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-include-regex "v3_" --output-format csv -d /tmp/v3pmc/a -o p -- python3 $GRAFT_REPO_ROOT/tools/frame.py ${1:-16} > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
rows=[]
for f in glob.glob("/tmp/v3pmc/**/*counter_collection.csv", recursive=True):
    rows+=list(csv.DictReader(open(f)))
disp=collections.defaultdict(dict)
for r in rows: disp[int(r["Dispatch_Id"])][r["Counter_Name"]]=float(r["Counter_Value"]); disp[int(r["Dispatch_Id"])]["k"]=r["Kernel_Name"].split("(")[0].replace("void ","")[:16]
ids=sorted(disp)
ids=ids[len(ids)//2:]
for i in ids[:34]:
    c=disp[i]
    print("%3d %-16s VALU %.3e util %.3f busy/wavecyc %.3f waves %5.0f wavecyc %.2e" % (i, c["k"], c["SQ_INSTS_VALU"], c["SQ_THREAD_CYCLES_VALU"]/(64*max(c["SQ_ACTIVE_INST_VALU"],1)), c["SQ_ACTIVE_INST_VALU"]/max(c["SQ_WAVE_CYCLES"],1), c["SQ_WAVES"], c["SQ_WAVE_CYCLES"]))
PY
