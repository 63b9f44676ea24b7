#!/bin/bash
# round 5, tenth GPU call: the layout stage's fills and key grouping underneath the step before them (second stream)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e10; mkdir -p $O
Q="--no-cpu-baseline --no-pcie --no-qv --no-packed --no-nofilter"
line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    v = d.get("verify") or {}
    print(sys.argv[1].ljust(20), "ms", round(d["ms_per_step"], 2), {k: v[k] for k in v if isinstance(v[k], bool)}, d.get("phase_wall_ms_last_step") or d.get("phase_ms_last_step"))
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[2].replace(".json", ".err")).read()[-600:])
PY
}
run() { local name=$1; shift; timeout 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err; line $name $O/$name.json; }
run default $Q --steps 5 --warmup 2
run default_noovl $Q --steps 5 --warmup 2 --knob YAKAMD_R2_OVERLAP=0
run nofilter --config nofilter $Q --steps 5 --warmup 2
run nofilter_noovl --config nofilter $Q --steps 5 --warmup 2 --knob YAKAMD_R2_OVERLAP=0
run cfg4_1gb --config cfg4 --contigs 10 --contig-len 100000000 --steps 3 --warmup 1
run cfg4_1gb_noovl --config cfg4 --contigs 10 --contig-len 100000000 --steps 3 --warmup 1 --knob YAKAMD_R2_OVERLAP=0 --no-verify
run cfg4_2gb --config cfg4 --contigs 20 --contig-len 100000000 --steps 2 --warmup 1
(time python -m pytest tests -q -m gpu -x) > $O/gpu_test_tier.txt 2>&1; tail -6 $O/gpu_test_tier.txt
