#!/bin/bash
# variants of one knob on cfg4 1 Gb (and optionally others): VAR=name VALS="a b c" [CFG="..."]
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06var}; mkdir -p $O
Q="--no-cpu-baseline --no-pcie --no-qv --no-packed --no-nofilter --steps 3 --warmup 1"
CFG=${CFG:---config cfg4 --contigs 10 --contig-len 100000000}
for v in $VALS; do
  timeout 600 python bench.py $CFG $Q --knob $VAR=$v > $O/bench_$v.json 2> $O/bench_$v.err
  python3 - $O/bench_$v.json $v <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    ph = d.get("phase_ms_last_step") or {}
    if "pass1" in ph: ph = ph["pass1"]
    v = d.get("verify") or {}
    print(sys.argv[2], "ms", round(d["ms_per_step"], 2), {k: ph[k] for k in ph if k in ("ms_sort", "ms_replay", "ms_insert", "ms_select")}, {k: v[k] for k in v if isinstance(v[k], bool)})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
