#!/bin/bash
# round 5, fourteenth GPU call: k_lc2 without the per-key last-time array (4 KB less LDS): 6 / 7 / 8 workgroups per CU
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e14; mkdir -p $O
Q="--no-cpu-baseline --no-pcie --no-qv --no-packed --no-nofilter --steps 5 --warmup 2"
line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    v = d.get("verify") or {}
    ks = {k["kernel"].split(" (")[0][:34]: round(k["ms"], 2) for k in d["roofline"].get("all_kernels", [])}
    print(sys.argv[1].ljust(16), "ms", round(d["ms_per_step"], 2), {k: v[k] for k in v if isinstance(v[k], bool)}, ks)
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[2].replace(".json", ".err")).read()[-600:])
PY
}
run() { local name=$1; shift; timeout 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err; line $name $O/$name.json; }
run w6 $Q
run w7 $Q --knob YAKAMD_LC2_WV=7
run w8 $Q --knob YAKAMD_LC2_WV=8
run w6b $Q
run m30_w6 $Q --reads 30000000 --steps 2 --warmup 1
run m30_w7 $Q --reads 30000000 --steps 2 --warmup 1 --knob YAKAMD_LC2_WV=7
run noretain $Q --no-retain --steps 3 --warmup 1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_ref_cli_on_amd.py tests/test_multi_c.py -x -q -m gpu > $O/pytest_parity.txt 2>&1; tail -2 $O/pytest_parity.txt
