# per-class wave budgets of the pipeline stages under the default (automatic) variant: one line per setting
cd $GRAFT_REPO_ROOT
run() { echo -n "$* : "; env "$@" python3 tools/hybrid_try.py 2>&1 | tail -1 | sed 's/.*: \[/[/'; }
run X=0
for c in 12 16 24 32; do run DE_V3_WPC_CLOUD=$c; done
for c in 12 16 24; do run DE_V3_WPC_GAS=$c; done
for c in 12 16 24; do run DE_V3_WPC_ST=$c; done
for c in 12 16 32; do run DE_V3_WPC_FLAT=$c; done
run DE_V3_WPC_CLOUD=16 DE_V3_WPC_GAS=16 DE_V3_WPC_ST=16
run DE_AUTO_WPC=6
run DE_AUTO_WPC=10
