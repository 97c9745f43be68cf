# knob sweep on rank 0's share of an N-rank tile partition (bench.py --simulate-world $WORLD, default 8; camera preset file in $PRESET_NAME):
# ms per step with overlapping frames, single-frame ms.   usage: [WORLD=4] [PRESET_NAME="config - florida.txt"] tools/world8_sweep.sh "ENV=1 ENV2=2" ...
cd $GRAFT_REPO_ROOT
run() {
  echo -n "$1 : "
  if [ -n "$PRESET_NAME" ]; then env $1 python3 bench.py --no-cpu-baseline --warmup 3 --steps 12 --simulate-world ${WORLD:-8} --preset "$PRESET_NAME" 2>/dev/null > /tmp/w8.json
  else env $1 python3 bench.py --no-cpu-baseline --warmup 3 --steps 12 --simulate-world ${WORLD:-8} 2>/dev/null > /tmp/w8.json; fi
  python3 -c "import sys,json; j=json.loads([l for l in open('/tmp/w8.json') if l.startswith('{')][-1]); print('%.1f ms/step, single %.1f' % (j['ms_per_step'], j['single_frame_ms']))"
}
for cfg in "$@"; do run "$cfg"; done
