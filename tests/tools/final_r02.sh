#!/bin/bash
# round-2 final measurements: default bench line (all side figures), the other configurations, profiles of the default command
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02final; mkdir -p $O
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --config nofilter --no-cpu-baseline --no-pcie --no-qv --no-packed > $O/bench_nofilter.json 2> /dev/null
timeout 600 python bench.py --config cfg4 --contigs 10 --contig-len 100000000 > $O/bench_cfg4_1gb.json 2> /dev/null
timeout 600 python bench.py --config cfg5 > $O/bench_cfg5.json 2> /dev/null
timeout 600 python bench.py --reads 30000000 --no-cpu-baseline --no-pcie --no-qv --no-packed > $O/bench_30m_sliced.json 2> /dev/null
for f in default nofilter cfg4_1gb cfg5 30m_sliced; do python3 - $O/bench_$f.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d.get("roofline", {})
    print(sys.argv[1].split("/")[-1], "ms", round(d["ms_per_step"], 2), "value", round(d["value"] / 1e6, 1), "M/s", {k: round(v, 4) for k, v in r.items() if "frac" in k and isinstance(v, float)}, d.get("verify"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
bash tests/tools/prof_r02.sh > $O/prof.log 2>&1; tail -22 $O/prof.log
cp gpurun_out/r02_kernel_stats.csv gpurun_out/r02_pmc_hbm_bytes.csv gpurun_out/r02_pmc_traffic.json gpurun_out/r02_bench_profiled.json $O/
bash tests/tools/pmc_sq.sh r02final_sq > /dev/null 2>&1; cp gpurun_out/r02final_sq/sq_summary.txt $O/sq_counters.txt
bash tests/tools/trace_r2.sh > /dev/null 2>&1; cp gpurun_out/r2trace/r2_dispatches.txt $O/r2_dispatches.txt
