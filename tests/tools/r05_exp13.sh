#!/bin/bash
# round 5, thirteenth GPU call: more of a lane's records through k_lc2's table side by side
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e13; mkdir -p $O
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
run nsb3_w6 $Q
run nsb3_w5 $Q --knob YAKAMD_LC2_W6=0
run nsb4 $Q --knob YAKAMD_LC2_NSB=4
run nsb5 $Q --knob YAKAMD_LC2_NSB=5
run nsb6 $Q --knob YAKAMD_LC2_NSB=6
run nf_nsb3 --config nofilter $Q
run nf_nsb5 --config nofilter $Q --knob YAKAMD_LC2_NSB11=5
run m30_nsb3 $Q --reads 30000000 --steps 2 --warmup 1
run m30_nsb5 $Q --reads 30000000 --steps 2 --warmup 1 --knob YAKAMD_LC2_NSB=5
