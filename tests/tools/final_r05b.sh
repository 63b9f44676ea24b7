#!/bin/bash
# round 5, the two bench lines that still carried `traffic: null` (30 M reads; the 5 Gb assembly in 2 sweeps): FETCH_SIZE / WRITE_SIZE passes of ONE run of each command
# (--kernel-trace only), then the lines themselves, which read the counter bytes from profiles/r05_pmc_traffic_<config>.json
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05final_b; mkdir -p $O
Q1="--steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-qv --no-pcie --no-packed --no-nofilter"
pmc() {   # name, profiles name, bench args...
  local name=$1 pname=$2; shift; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 1200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/raw_${name}_$c -o pmc -- python bench.py $Q1 "$@" > $O/${name}_$c.log 2>&1
  done
  python3 tests/tools/pmc_summary.py $O/raw_${name}_FETCH_SIZE $O/raw_${name}_WRITE_SIZE $O/$name > $O/${name}_table.txt
  rm -rf $O/raw_${name}_FETCH_SIZE $O/raw_${name}_WRITE_SIZE
  cp $O/${name}_pmc_traffic.json profiles/r05_pmc_traffic_$pname.json; cp $O/${name}_pmc_hbm_bytes.csv profiles/r05_pmc_hbm_bytes_$pname.csv
  head -8 $O/${name}_table.txt
}
pmc r05_30m 30m --reads 30000000
pmc r05_cfg4_5gb cfg4_50x100000000_sweeps2 --config cfg4 --contigs 50
Q="--no-cpu-baseline --no-pcie --no-qv --no-packed --no-nofilter"
timeout 600 python bench.py --reads 30000000 $Q > $O/bench_30m.json 2> /dev/null
sleep 5
timeout 900 python bench.py --config cfg4 --contigs 50 --warmup 1 > $O/bench_cfg4_5gb_sweeps2.json 2> $O/bench_cfg4_5gb.err
for f in 30m cfg4_5gb_sweeps2; do python3 - $O/bench_$f.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d.get("roofline", {})
    print(sys.argv[1].split("/")[-1], "ms", round(d["ms_per_step"], 2), "value", round(d["value"] / 1e6, 1), "M/s frac", r.get("frac"), "traffic", r.get("traffic"), "hbm_util", r.get("hbm_util"), {k: x for k, x in (d.get("verify") or {}).items() if isinstance(x, bool)})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
cp profiles/r05_pmc_traffic_30m.json profiles/r05_pmc_traffic_cfg4_50x100000000_sweeps2.json profiles/r05_pmc_hbm_bytes_30m.csv profiles/r05_pmc_hbm_bytes_cfg4_50x100000000_sweeps2.csv $O/
