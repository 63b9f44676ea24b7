#!/bin/bash
# round 5, eighth GPU call: level-2 scatter pipelined across the flush
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e8; mkdir -p $O
Q="--no-cpu-baseline --no-pcie --no-qv --no-packed --no-nofilter --steps 5 --warmup 2"
line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    ks = {k["kernel"].split(" (")[0][:34]: round(k["ms"], 2) for k in d["roofline"].get("all_kernels", [])} if "all_kernels" in d.get("roofline", {}) else {}
    v = d.get("verify") or {}
    print(sys.argv[1].ljust(20), "ms", round(d["ms_per_step"], 2), {k: v[k] for k in v if isinstance(v[k], bool)}, ks, d.get("phase_wall_ms_last_step"))
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[2].replace(".json", ".err")).read()[-600:])
PY
}
run() { local name=$1; shift; timeout 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err; line $name $O/$name.json; }
run default $Q
run nofilter --config nofilter $Q
run cfg4_1gb --config cfg4 --contigs 10 --contig-len 100000000 --steps 3 --warmup 1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -x -q -m gpu > $O/pytest_parity.txt 2>&1; tail -2 $O/pytest_parity.txt
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "1m_reads" > $O/pytest_cfg3.txt 2>&1; tail -2 $O/pytest_cfg3.txt
