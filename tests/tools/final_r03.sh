#!/bin/bash
# round-3 final measurements: the default bench line (all side figures), the other configurations, profiles of the default command
# (kernel trace, the two PMC passes, SQ counters, per-dispatch trace of the layout stage) -> gpurun_out/r03final
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03final; mkdir -p $O
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
Q="--no-cpu-baseline --no-pcie --no-qv --no-packed --no-nofilter"
timeout 600 python bench.py --config nofilter $Q > $O/bench_nofilter.json 2> /dev/null
timeout 600 python bench.py --config cfg4 --contigs 10 --contig-len 100000000 > $O/bench_cfg4_1gb.json 2> /dev/null
timeout 600 python bench.py --config cfg5 > $O/bench_cfg5.json 2> /dev/null
timeout 600 python bench.py --reads 30000000 $Q > $O/bench_30m.json 2> /dev/null
timeout 600 python bench.py --no-retain $Q --no-verify > $O/bench_noretain.json 2> /dev/null
timeout 900 python bench.py --config cfg3shard > $O/bench_cfg3shard.json 2> $O/bench_cfg3shard.err
timeout 900 python bench.py --config cfg4 --contigs 50 --sweeps 8 > $O/bench_cfg4_5gb_sweeps8.json 2> /dev/null
timeout 900 python bench.py --config cfg4 --contigs 50 --sweeps 4 > $O/bench_cfg4_5gb_sweeps4.json 2> /dev/null
timeout 600 python bench.py --config cfg4 --contigs 20 --contig-len 100000000 > $O/bench_cfg4_2gb.json 2> /dev/null
for f in default nofilter cfg4_1gb cfg4_2gb cfg5 30m noretain cfg3shard cfg4_5gb_sweeps8 cfg4_5gb_sweeps4; do python3 - $O/bench_$f.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d.get("roofline", {})
    print(sys.argv[1].split("/")[-1], "ms", round(d["ms_per_step"], 2), "value", round(d["value"] / 1e6, 1), "M/s", {k: round(v, 4) for k, v in r.items() if "frac" in k and isinstance(v, float)}, d.get("verify"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
P="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-qv --no-pcie --no-packed --no-nofilter"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-pcie --no-packed --no-nofilter > $O/bench_profiled.json 2>/dev/null
cp $(find gpurun_out/r03_trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/r03_$c -o pmc -- $P > $O/pmc_$c.log 2>&1
done
python3 tests/tools/pmc_summary.py gpurun_out/r03_FETCH_SIZE gpurun_out/r03_WRITE_SIZE $O/r03 | tee $O/pmc_table.txt
python3 - $O/kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.4: print(r["Name"][:48].ljust(48), r["Calls"].rjust(5), "%9.3f ms avg" % (float(r["AverageNs"])/1e6), "%9.2f ms tot" % (float(r["TotalDurationNs"])/1e6), r["Percentage"]+"%")
PY
rm -rf gpurun_out/r03_trace gpurun_out/r03_FETCH_SIZE gpurun_out/r03_WRITE_SIZE
bash tests/tools/pmc_sq.sh r03final_sq > /dev/null 2>&1; cp gpurun_out/r03final_sq/sq_summary.txt $O/sq_counters.txt
bash tests/tools/trace_r2.sh > /dev/null 2>&1; cp gpurun_out/r2trace/r2_dispatches.txt $O/r2_dispatches.txt
timeout 300 python tests/tools/rccl_big_msg.py --gib 3 > $O/rccl_big_msg.txt 2>&1; echo "exit code $?" >> $O/rccl_big_msg.txt
