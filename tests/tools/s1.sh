#!/bin/bash
# session script (round 3): GPU tests, then A/B bench lines of the fused doubling rounds and the retained pass-2 records
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/s1; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pcie"
timeout 600 $B > $O/bench_default.json 2> $O/bench_default.err
Q="--no-verify --no-qv --no-packed"
YAKAMD_R2_FUSED=0 timeout 300 $B $Q > $O/bench_unfused.json 2>/dev/null
timeout 300 $B $Q --no-retain > $O/bench_noretain.json 2>/dev/null
timeout 300 $B $Q --config nofilter > $O/bench_nofilter.json 2>/dev/null
timeout 600 python bench.py --config cfg4 --contigs 10 --contig-len 100000000 > $O/bench_cfg4_1gb.json 2> $O/bench_cfg4_1gb.err
for f in default unfused noretain nofilter cfg4_1gb; do python3 - $O/bench_$f.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], "ms", round(d["ms_per_step"], 2), "value", round(d["value"] / 1e6, 1), "M/s", json.dumps(d.get("phase_ms_last_step")), d.get("verify"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
bash tests/tools/trace_r2.sh > $O/trace.log 2>&1; cp gpurun_out/r2trace/r2_dispatches.txt $O/ 2>/dev/null
bash tests/tools/prof_stats.sh s1prof > $O/prof_stats.log 2>&1; tail -40 $O/prof_stats.log
