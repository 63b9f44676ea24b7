#!/bin/bash
# round-4 counter passes: HBM traffic per kernel (FETCH_SIZE, WRITE_SIZE: separate passes, --kernel-trace only) of ONE step of every configuration
# -> gpurun_out/r04pmc/r04_pmc_traffic[_<config>].json (+ _pmc_hbm_bytes.csv); SQ and LDS counters of the default step
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04pmc; mkdir -p $O
Q="--steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-qv --no-pcie --no-packed --no-nofilter"
run() {   # name, bench args...
  local name=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/raw_${name}_$c -o pmc -- python bench.py $Q "$@" > $O/${name}_$c.log 2>&1
  done
  python3 tests/tools/pmc_summary.py $O/raw_${name}_FETCH_SIZE $O/raw_${name}_WRITE_SIZE $O/$name > $O/${name}_table.txt
  rm -rf $O/raw_${name}_FETCH_SIZE $O/raw_${name}_WRITE_SIZE
  head -8 $O/${name}_table.txt
}
run r04
run r04_nofilter --config nofilter
run r04_cfg4_10x100000000 --config cfg4 --contigs 10 --contig-len 100000000
run r04_cfg4_20x100000000 --config cfg4 --contigs 20 --contig-len 100000000
run r04_cfg5 --config cfg5
run r04_cfg3shard --config cfg3shard
bash tests/tools/pmc_sq.sh r04pmc_sq > /dev/null 2>&1; cp gpurun_out/r04pmc_sq/sq_summary.txt $O/r04_sq_counters.txt
bash tests/tools/pmc_lds.sh r04pmc_lds > /dev/null 2>&1; cp gpurun_out/r04pmc_lds/*summary*.txt $O/r04_lds_counters.txt 2>/dev/null
ls $O
