#!/bin/bash
cd $GRAFT_REPO_ROOT
for f in 1024 256 64 16; do
  echo "small_f $f"
  YAKAMD_R2_SMALL_F=$f YAKAMD_VERBOSE=2 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-qv --no-pcie 2>&1 | grep -E "replay2 step|equals_reference" | tail -24 | grep -E "dsmall|drounds|equals" | sed 's/.*replay2//' | cut -c1-200 | tr '\n' ';'; echo
done
