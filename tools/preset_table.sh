# the cameras of DESIGN §5 under the default kernel variant and under DE_KERNEL=2: Msamples/s pipelined + single-frame ms
cd $GRAFT_REPO_ROOT
line() { python3 - "$1" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        j=json.loads(l); print("value %.1f Msamples/s, %.1f ms/step, single frame %.1f ms" % (j["value"], j["ms_per_step"], j.get("single_frame_ms", float("nan"))))
PY
}
for k in ${DE_PRESET_KERNELS:-4 2}; do
  echo -n "DE_KERNEL=$k default camera : "; DE_KERNEL=$k python3 bench.py --no-cpu-baseline --steps 4 > /tmp/b.json 2>/tmp/b.err; line /tmp/b.json
  for p in "config - florida.txt" "config - sunset hurricane.txt" "config - Apollo 11.txt"; do
    echo -n "DE_KERNEL=$k $p : "; DE_KERNEL=$k python3 bench.py --no-cpu-baseline --steps 4 --preset "$p" > /tmp/b.json 2>/tmp/b.err; line /tmp/b.json
  done
  echo -n "DE_KERNEL=$k cfg4 3840x2160x128 cloud-heavy sunset : "; DE_KERNEL=$k python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 --width 3840 --height 2160 --spp 128 --cloud-heavy --preset "config - sunset hurricane.txt" > /tmp/b.json 2>/tmp/b.err; line /tmp/b.json; tail -2 /tmp/b.err
done
