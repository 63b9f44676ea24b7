#!/bin/bash
# round 5, eleventh GPU call: the level-2 scatters request their next records BEHIND the use of the current ones
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e11; mkdir -p $O
Q="--no-cpu-baseline --no-pcie --no-qv --no-packed --no-nofilter"
line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    v = d.get("verify") or {}
    ks = {k["kernel"].split(" (")[0][:34]: round(k["ms"], 2) for k in d["roofline"].get("all_kernels", [])} if "all_kernels" in d.get("roofline", {}) else {}
    print(sys.argv[1].ljust(16), "ms", round(d["ms_per_step"], 2), {k: v[k] for k in v if isinstance(v[k], bool)}, ks, (d.get("phase_ms_last_step") or {}).get("pass1", d.get("phase_ms_last_step")), d.get("rank_seconds"))
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[2].replace(".json", ".err")).read()[-600:])
PY
}
run() { local name=$1; shift; timeout 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err; line $name $O/$name.json; }
run default $Q --steps 5 --warmup 2
run nofilter --config nofilter $Q --steps 5 --warmup 2
run cfg4_1gb --config cfg4 --contigs 10 --contig-len 100000000 --steps 3 --warmup 1
run reads30m $Q --reads 30000000 --steps 2 --warmup 1
run cfg3shard --config cfg3shard --warmup 1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -x -q -m gpu > $O/pytest_parity.txt 2>&1; tail -2 $O/pytest_parity.txt
