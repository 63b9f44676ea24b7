#!/bin/bash
# HBM traffic per kernel: two separate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over one bench
# step -> gpurun_out/<name>_pmc_traffic.json (+ _pmc_hbm_bytes.csv).  Counter values are KiB;
# FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM section) prescribes for gfx950.
name=${1:-pmc}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/${name}_$c -o pmc -- \
    python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-qv > gpurun_out/${name}_$c.log 2>&1
done
python3 tests/tools/pmc_summary.py gpurun_out/${name}_FETCH_SIZE gpurun_out/${name}_WRITE_SIZE gpurun_out/${name}
