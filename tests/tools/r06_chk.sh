#!/bin/bash
# layout-stage check: the replay / insert-path GPU tests, then verified cfg4 1 Gb + nofilter + default lines
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06chk}; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "${K:-replay or insert_path or knob}" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
Q="--no-cpu-baseline --no-pcie --no-qv --no-packed --no-nofilter --steps 3 --warmup 1"
timeout 600 python bench.py --config cfg4 --contigs 10 --contig-len 100000000 $Q > $O/bench_cfg4_1gb.json 2> $O/bench_cfg4_1gb.err
timeout 600 python bench.py --config nofilter $Q > $O/bench_nofilter.json 2> /dev/null
timeout 600 python bench.py $Q > $O/bench_default.json 2> $O/bench_default.err
YAKAMD_VERBOSE=2 timeout 600 python bench.py --config cfg4 --contigs 10 --contig-len 100000000 $Q --steps 1 --warmup 1 > $O/bench_cfg4_1gb_prof.json 2> $O/cfg4_1gb_prof.err
grep "replay2\|k_r2_double" $O/cfg4_1gb_prof.err | tail -60 > $O/cfg4_1gb_r2_phases.txt
for f in cfg4_1gb nofilter default; do python3 - $O/bench_$f.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    ph = d.get("phase_ms_last_step") or {}
    if "pass1" in ph: ph = ph["pass1"]
    v = d.get("verify") or {}
    print(sys.argv[1].split("/")[-1], "ms", round(d["ms_per_step"], 2), {k: ph[k] for k in ph if k in ("ms_sort", "ms_replay", "ms_insert", "ms_select")}, {k: v[k] for k in v if isinstance(v[k], bool)})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
tail -22 $O/cfg4_1gb_r2_phases.txt
