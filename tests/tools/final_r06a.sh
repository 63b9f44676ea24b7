#!/bin/bash
# round-6 final measurements, part A: counter passes (FETCH_SIZE / WRITE_SIZE separately, --kernel-trace only) of ONE step of every configuration
# -> gpurun_out/r06final/r06_pmc_traffic[_<config>].json + r06_pmc_hbm_bytes[_<config>].csv (copied to profiles/ afterwards: bench.py reads roofline.traffic
# from them while the device code still has the hash they carry), then the SQ and LDS counters of the default step
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06final; mkdir -p $O
Q1="--steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-qv --no-pcie --no-packed --no-nofilter"
pmc() {   # file suffix ("" = the default line), bench args...
  local sfx=$1; shift
  local name=r06${sfx:+_$sfx}
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 1200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/raw_${name}_$c -o pmc -- python bench.py $Q1 "$@" > $O/${name}_$c.log 2>&1
  done
  python3 tests/tools/pmc_summary.py $O/raw_${name}_FETCH_SIZE $O/raw_${name}_WRITE_SIZE $O/$name > $O/${name}_table.txt
  rm -rf $O/raw_${name}_FETCH_SIZE $O/raw_${name}_WRITE_SIZE
  mv $O/${name}_pmc_traffic.json $O/r06_pmc_traffic${sfx:+_$sfx}.json; mv $O/${name}_pmc_hbm_bytes.csv $O/r06_pmc_hbm_bytes${sfx:+_$sfx}.csv
  echo "== $name"; head -7 $O/${name}_table.txt
}
pmc ""
pmc nofilter --config nofilter
pmc cfg4_10x100000000 --config cfg4 --contigs 10 --contig-len 100000000
pmc cfg4_20x100000000 --config cfg4 --contigs 20 --contig-len 100000000
pmc cfg5 --config cfg5
pmc 30m --reads 30000000
pmc cfg3shard --config cfg3shard
pmc cfg4_50x100000000_sweeps8 --config cfg4 --contigs 50
bash tests/tools/pmc_sq.sh r06final_sq > /dev/null 2>&1; cp gpurun_out/r06final_sq/sq_summary.txt $O/r06_sq_counters.txt
bash tests/tools/pmc_lds.sh r06final_lds > /dev/null 2>&1; cp gpurun_out/r06final_lds/*summary*.txt $O/r06_lds_counters_step.txt 2>/dev/null
ls $O
