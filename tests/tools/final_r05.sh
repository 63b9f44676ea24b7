#!/bin/bash
# round-5 final measurements on one MI355X box, in this order:
#  1. counter passes (FETCH_SIZE / WRITE_SIZE separately, --kernel-trace only) of ONE step of every configuration -> profiles/r05_pmc_traffic[_<config>].json on the box
#     (bench.py reads roofline.traffic from them), SQ and LDS counters of the default step
#  2. the bench lines (default with all side figures; the other configurations), kernel stats of default / nofilter / cfg4 1 Gb, layout dispatch traces
#  3. the CLI end to end
# everything lands in gpurun_out/r05final; what is to be judged is copied to profiles/r05_* afterwards
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05final; mkdir -p $O
Q1="--steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-qv --no-pcie --no-packed --no-nofilter"
pmc() {   # name, bench args...
  local name=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/raw_${name}_$c -o pmc -- python bench.py $Q1 "$@" > $O/${name}_$c.log 2>&1
  done
  python3 tests/tools/pmc_summary.py $O/raw_${name}_FETCH_SIZE $O/raw_${name}_WRITE_SIZE $O/$name > $O/${name}_table.txt
  rm -rf $O/raw_${name}_FETCH_SIZE $O/raw_${name}_WRITE_SIZE
  cp $O/${name}_pmc_traffic.json profiles/ 2>/dev/null
  head -6 $O/${name}_table.txt
}
pmc r05
pmc r05_nofilter --config nofilter
pmc r05_cfg4_10x100000000 --config cfg4 --contigs 10 --contig-len 100000000
pmc r05_cfg4_20x100000000 --config cfg4 --contigs 20 --contig-len 100000000
pmc r05_cfg5 --config cfg5
pmc r05_cfg3shard --config cfg3shard
bash tests/tools/pmc_sq.sh r05final_sq > /dev/null 2>&1; cp gpurun_out/r05final_sq/sq_summary.txt $O/r05_sq_counters.txt
bash tests/tools/pmc_lds.sh r05final_lds > /dev/null 2>&1; cp gpurun_out/r05final_lds/*summary*.txt $O/r05_lds_counters.txt 2>/dev/null
# ---- bench lines
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
Q="--no-cpu-baseline --no-pcie --no-qv --no-packed --no-nofilter"
timeout 600 python bench.py --config nofilter $Q > $O/bench_nofilter.json 2> /dev/null
timeout 600 python bench.py --config cfg4 --contigs 10 --contig-len 100000000 > $O/bench_cfg4_1gb.json 2> /dev/null
timeout 600 python bench.py --config cfg4 --contigs 20 --contig-len 100000000 > $O/bench_cfg4_2gb.json 2> /dev/null
timeout 600 python bench.py --config cfg5 > $O/bench_cfg5.json 2> /dev/null
timeout 600 python bench.py --reads 30000000 $Q > $O/bench_30m.json 2> /dev/null
timeout 600 python bench.py --no-retain $Q --no-verify > $O/bench_noretain.json 2> /dev/null
YAKAMD_VERBOSE=1 timeout 900 python bench.py --config cfg3shard --warmup 1 > $O/bench_cfg3shard.json 2> $O/bench_cfg3shard.err
sleep 5
YAKAMD_VERBOSE=1 timeout 900 python bench.py --config cfg4 --contigs 50 --warmup 1 > $O/bench_cfg4_5gb_sweeps2.json 2> $O/bench_cfg4_5gb.err
grep "ranks: input\|pool after" $O/bench_cfg4_5gb.err | head -4 > $O/cfg4_5gb_stages.txt
grep "pool after\|level-2 partition\|k_lc2\|slice of the pass" $O/bench_cfg3shard.err | tail -8 > $O/cfg3shard_stages.txt
sleep 5
for f in default nofilter cfg4_1gb cfg4_2gb cfg5 30m noretain cfg3shard cfg4_5gb_sweeps2; do python3 - $O/bench_$f.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d.get("roofline", {})
    v = d.get("verify") or {}
    print(sys.argv[1].split("/")[-1], "ms", round(d["ms_per_step"], 2), "value", round(d["value"] / 1e6, 1), "M/s", {k: round(x, 4) for k, x in r.items() if ("frac" in k or k == "hbm_util") and isinstance(x, float)}, "traffic", r.get("traffic"), {k: x for k, x in v.items() if isinstance(x, bool)}, {k: d[k] for k in d if k.startswith(("first_job", "peak_hbm_bytes")) and "note" not in k})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
for cfg in "default:" "nofilter:--config nofilter" "cfg4_1gb:--config cfg4 --contigs 10 --contig-len 100000000"; do
  name=${cfg%%:*}; args=${cfg#*:}
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$name -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-pcie --no-packed --no-nofilter --no-qv $args > $O/bench_profiled_$name.json 2>/dev/null
  cp $(find $O/trace_$name -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$name.csv
  rm -rf $O/trace_$name
done
python3 - $O/kernel_stats_default.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.4: print(r["Name"][:56].ljust(56), r["Calls"].rjust(5), "%9.3f ms avg" % (float(r["AverageNs"])/1e6), "%9.2f ms tot" % (float(r["TotalDurationNs"])/1e6), r["Percentage"]+"%")
PY
R2OUT=r05final/r2_nofilter bash tests/tools/trace_r2.sh --config nofilter > /dev/null 2>&1; cp gpurun_out/r05final/r2_nofilter/r2_dispatches.txt $O/r2_dispatches_nofilter.txt
timeout 300 bash tests/tools/r04_e2e.sh gz > $O/e2e_cli.txt 2>&1
ls $O
