# a rank's share of an N-rank tile partition on ONE GPU (bench.py --simulate-world N): MODES: 1 = the default variant's pipeline, 2 = the state machine, 5 / 6 = kernel variants 5 / 6
cd $GRAFT_REPO_ROOT
line() { python3 - "$1" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        j=json.loads(l); print("%.1f ms/step (single frame %.1f)" % (j["ms_per_step"], j.get("single_frame_ms", float("nan"))))
PY
}
for w in ${WORLDS:-2 4 8}; do for p in "" "config - florida.txt" "config - sunset hurricane.txt" "config - Apollo 11.txt"; do
  echo -n "world $w | ${p:-default camera} | pipeline / state machine / persistent : "
  for m in ${MODES:-1 2 5}; do
    if [ $m = 5 ] || [ $m = 6 ]; then export DE_KERNEL=$m; unset DE_AUTO_MID; else export DE_KERNEL=4 DE_AUTO_MID=$m; fi
    if [ -z "$p" ]; then python3 bench.py --no-cpu-baseline --warmup 3 --steps 12 --simulate-world $w > /tmp/b.json 2>/tmp/b.err; else python3 bench.py --no-cpu-baseline --warmup 3 --steps 12 --simulate-world $w --preset "$p" > /tmp/b.json 2>/tmp/b.err; fi
    echo -n "$(line /tmp/b.json) | "
  done; echo
done; done
