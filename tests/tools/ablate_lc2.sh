#!/bin/bash
# k_lc2 phase ablation (timing only; results are wrong by construction): YAKAMD_DBG 16 = phase A only, 32 = no gate, 64 = no filter write-back
cd $GRAFT_REPO_ROOT
for d in 0 16 32 64; do
  echo "dbg $d: $(YAKAMD_DBG=$d YAKAMD_VERBOSE=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify --no-qv --no-pcie 2>&1 | grep 'k_lc2:' | tail -1)"
done
