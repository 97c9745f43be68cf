# exclusive duration and L2 hit rate of the cloud stage kernel at several wave budgets (PMC passes serialise the launches)
cd /tmp; export TMPDIR=/tmp
for w in 8 12 16 24 32; do
  rm -rf /tmp/cl2; export DE_V3_WPC_CLOUD=$w
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --kernel-include-regex "v3_cloud" --output-format csv -d /tmp/cl2 -o p -- python3 $GRAFT_REPO_ROOT/tools/frame.py > /dev/null 2>&1
  python3 - $w <<'PY'
import csv,glob,collections,sys
rows=[]
for f in glob.glob("/tmp/cl2/**/*counter_collection.csv", recursive=True): rows+=list(csv.DictReader(open(f)))
d=collections.defaultdict(dict)
for r in rows:
    d[r["Dispatch_Id"]][r["Counter_Name"]]=float(r["Counter_Value"]); d[r["Dispatch_Id"]]["ms"]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
n=len(d); frames=n/12.0
hit=sum(v["TCC_HIT_sum"] for v in d.values()); miss=sum(v["TCC_MISS_sum"] for v in d.values()); req=sum(v["TCP_TCC_READ_REQ_sum"] for v in d.values()); ms=sum(v["ms"] for v in d.values())
print("cloud stage at %s waves per CU: %.1f ms per frame (exclusive), L2 requests %.3g, hit rate %.1f %%, misses %.3g per frame" % (sys.argv[1], ms/frames, req/frames, 100*hit/(hit+miss), miss/frames))
PY
done
