cd $GRAFT_REPO_ROOT
line() { python3 - "$1" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        j=json.loads(l); print("value %.1f Msamples/s, %.1f ms/step, single frame %.1f ms" % (j["value"], j["ms_per_step"], j.get("single_frame_ms", float("nan"))))
PY
}
for p in "config - sunset hurricane.txt" "config - florida.txt"; do
 for s in "2 2" "3 2" "3 3" "4 2" "4 3" "5 3" "6 3" "8 3"; do
  set -- $s
  echo -n "$p depths $1 pipes $2 : "; DE_AUTO_DEPTHS=$1 DE_AUTO_PIPES=$2 DE_V3_BATCHES=$2 python3 bench.py --no-cpu-baseline --steps 3 --preset "$p" > /tmp/b.json 2>/tmp/b.err; line /tmp/b.json
 done
done
