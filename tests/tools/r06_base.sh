#!/bin/bash
# round-6 baseline on this round's box: default, nofilter, cfg4 1 Gb lines + the phase clocks of the layout stage
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06base; mkdir -p $O
Q="--no-cpu-baseline --no-pcie --no-qv --no-packed --no-nofilter --steps 3 --warmup 1"
timeout 600 python bench.py $Q > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --config nofilter $Q > $O/bench_nofilter.json 2> /dev/null
timeout 600 python bench.py --config cfg4 --contigs 10 --contig-len 100000000 $Q > $O/bench_cfg4_1gb.json 2> /dev/null
YAKAMD_VERBOSE=2 timeout 600 python bench.py --config cfg4 --contigs 10 --contig-len 100000000 $Q --steps 1 --warmup 1 > $O/bench_cfg4_1gb_prof.json 2> $O/cfg4_1gb_prof.err
grep "replay2\|k_r2_double" $O/cfg4_1gb_prof.err | tail -80 > $O/cfg4_1gb_r2_phases.txt
for f in default nofilter cfg4_1gb; do python3 - $O/bench_$f.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], "ms", round(d["ms_per_step"], 2), d.get("phase_ms_last_step"), (d.get("verify") or {}))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
tail -45 $O/cfg4_1gb_r2_phases.txt
