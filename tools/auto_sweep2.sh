cd $GRAFT_REPO_ROOT
run() { echo -n "$* : "; env "$@" python3 tools/hybrid_try.py 2>&1 | tail -1 | sed 's/.*: \[/[/'; }
for b in 3 4 6 8; do run DE_V3_BATCHES=$b; done
for b in 3 6; do run DE_AUTO_PIPES=3 DE_V3_BATCHES=$b; done
for b in 4 8; do run DE_AUTO_PIPES=4 DE_V3_BATCHES=$b; done
run DE_AUTO_PIPES=3 DE_V3_BATCHES=6 DE_AUTO_DEPTHS=3
run DE_AUTO_PIPES=2 DE_V3_BATCHES=4 DE_AUTO_DEPTHS=3
run DE_AUTO_PIPES=2 DE_V3_BATCHES=4 DE_AUTO_DEPTHS=1
