# the automatic variant's statistic (records queued for vertex round 2 / paths) and what it chose, per camera; DE_AUTO_DEEP_NS=101 = never deep
cd $GRAFT_REPO_ROOT
line() { python3 - "$1" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        j=json.loads(l); print("value %.1f Msamples/s, %.1f ms/step, single frame %.1f ms" % (j["value"], j["ms_per_step"], j.get("single_frame_ms", float("nan"))))
PY
}
export DE_AUTO_TRACE=1
for thr in ${THRS:-250 100000}; do
for p in "" "config - florida.txt" "config - sunset hurricane.txt" "config - Apollo 11.txt"; do
  echo -n "deep-ns x100 $thr | ${p:-default camera} : "
  if [ -z "$p" ]; then DE_AUTO_DEEP_NS=$thr python3 bench.py --no-cpu-baseline --steps 4 > /tmp/b.json 2>/tmp/b.err; else DE_AUTO_DEEP_NS=$thr python3 bench.py --no-cpu-baseline --steps 4 --preset "$p" > /tmp/b.json 2>/tmp/b.err; fi
  line /tmp/b.json; grep "\[de\] auto" /tmp/b.err | tail -1
done; done
for w in ${WORLDS:-2 4 8}; do for p in "" "config - sunset hurricane.txt"; do
  echo -n "simulate-world $w | ${p:-default camera} : "
  if [ -z "$p" ]; then python3 bench.py --no-cpu-baseline --steps 4 --simulate-world $w > /tmp/b.json 2>/tmp/b.err; else python3 bench.py --no-cpu-baseline --steps 4 --simulate-world $w --preset "$p" > /tmp/b.json 2>/tmp/b.err; fi
  line /tmp/b.json; grep "\[de\] auto" /tmp/b.err | tail -1
done; done
