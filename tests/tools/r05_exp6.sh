#!/bin/bash
# round 5, sixth GPU call: sub-buckets of 512 blocks (s2 = 9)?
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e6; mkdir -p $O
Q="--no-cpu-baseline --no-pcie --no-qv --no-packed --no-nofilter --steps 5 --warmup 2"
line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    ks = {k["kernel"].split(" (")[0][:34]: round(k["ms"], 2) for k in d["roofline"].get("all_kernels", [])} if "all_kernels" in d.get("roofline", {}) else {}
    v = d.get("verify") or {}
    print(sys.argv[1].ljust(20), "ms", round(d["ms_per_step"], 2), "verify", {k: v[k] for k in v if isinstance(v[k], bool)}, ks, d.get("phase_wall_ms_last_step"))
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[2].replace(".json", ".err")).read()[-600:])
PY
}
run() { local name=$1; shift; YAKAMD_VERBOSE=1 timeout 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err; line $name $O/$name.json; grep -h "k_lc2:" $O/$name.err | tail -1; }
run default $Q
run s2_9 $Q --knob YAKAMD_S2_BITS=9 --knob YAKAMD_LC2_LBMAX=9
run s2_9_30m $Q --reads 30000000 --steps 2 --warmup 1 --knob YAKAMD_LC2_LBMAX=9
run s2_def_30m $Q --reads 30000000 --steps 2 --warmup 1
