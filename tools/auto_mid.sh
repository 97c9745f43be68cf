# calls of a tile-partition share's size under the automatic variant: what it measures and chooses (DE_AUTO_MID=0 adaptive, 1 pipeline, 2 state machine)
cd $GRAFT_REPO_ROOT
line() { python3 - "$1" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        j=json.loads(l); print("value %.1f Msamples/s, %.1f ms/step, single frame %.1f ms" % (j["value"], j["ms_per_step"], j.get("single_frame_ms", float("nan"))))
PY
}
export DE_AUTO_TRACE=1
for w in ${WORLDS:-2 4 8}; do for p in "" "config - florida.txt" "config - sunset hurricane.txt"; do for m in 0 1 2; do
  echo -n "simulate-world $w | ${p:-default camera} | DE_AUTO_MID=$m : "
  if [ -z "$p" ]; then DE_AUTO_MID=$m python3 bench.py --no-cpu-baseline --warmup 2 --steps 6 --simulate-world $w > /tmp/b.json 2>/tmp/b.err; else DE_AUTO_MID=$m python3 bench.py --no-cpu-baseline --warmup 2 --steps 6 --simulate-world $w --preset "$p" > /tmp/b.json 2>/tmp/b.err; fi
  line /tmp/b.json; grep "\[de\] auto" /tmp/b.err | tail -1
done; done; done
