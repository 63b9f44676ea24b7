#!/bin/bash
# round 5, seventh GPU call: grid sizes of the persistent kernels now that a pass has 1 M sub-buckets instead of 2 M
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e7; mkdir -p $O
Q="--no-cpu-baseline --no-pcie --no-qv --no-packed --no-nofilter --no-verify --steps 5 --warmup 2"
line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    ks = {k["kernel"].split(" (")[0][:34]: round(k["ms"], 2) for k in d["roofline"].get("all_kernels", [])} if "all_kernels" in d.get("roofline", {}) else {}
    print(sys.argv[1].ljust(20), "ms", round(d["ms_per_step"], 2), ks, d.get("phase_wall_ms_last_step"))
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[2].replace(".json", ".err")).read()[-600:])
PY
}
run() { local name=$1; shift; timeout 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err; line $name $O/$name.json; }
run lc2_24k $Q --knob YAKAMD_LC2_WGS=24576
run lc2_49k $Q --knob YAKAMD_LC2_WGS=49152
run lc2_98k $Q --knob YAKAMD_LC2_WGS=98304
run lc2_196k $Q --knob YAKAMD_LC2_WGS=196608
run lc2_1m $Q --knob YAKAMD_LC2_WGS=1048576
run cnt2_64k $Q --knob YAKAMD_CNT2_WGS=65536
run cnt2_1m $Q --knob YAKAMD_CNT2_WGS=1048576
