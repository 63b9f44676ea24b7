#!/bin/bash
# round 5, first GPU call: does the new code still give the reference's bytes, and which of the new variants is fastest
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e1; mkdir -p $O
timeout 120 tests/tools/mb/mb_malloc 16 4 > $O/mb_malloc.txt 2>&1
Q="--no-cpu-baseline --no-pcie --no-qv --no-packed --no-nofilter --steps 5 --warmup 2"
line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    ks = {k["kernel"].split(" (")[0][:34]: round(k["ms"], 2) for k in d["roofline"].get("all_kernels", [])} if "all_kernels" in d.get("roofline", {}) else {}
    print(sys.argv[1].ljust(28), "ms", round(d["ms_per_step"], 2), "verify", (d.get("verify") or {}).get("equals_reference"), ks)
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[2].replace(".json", ".err")).read()[-600:])
PY
}
run() { local name=$1; shift; YAKAMD_VERBOSE=1 timeout 300 python bench.py $Q "$@" > $O/$name.json 2> $O/$name.err; line $name $O/$name.json; grep -h "k_lc2:" $O/$name.err | tail -1; }
run base_nostage0_cap8 --knob YAKAMD_LC2_NOSTAGE=0 --knob YAKAMD_P2_CAP7=0
run nostage_s2_11      --knob YAKAMD_S2_BITS=11
run nostage_s2_10_cap8 --knob YAKAMD_P2_CAP7=0
run nostage_s2_10      
run nostage_s2_10_w6   --knob YAKAMD_LC2_W6=1
Q2="--no-cpu-baseline --no-pcie --no-qv --no-packed --no-nofilter --steps 3 --warmup 1"
run2() { local name=$1; shift; YAKAMD_VERBOSE=1 timeout 400 python bench.py $Q2 "$@" > $O/$name.json 2> $O/$name.err; line $name $O/$name.json; }
run2 nofilter_cap10 --config nofilter --knob YAKAMD_LC2_CAPB=10 --knob YAKAMD_KC_INPLACE=0
run2 nofilter_cap11 --config nofilter
run2 cfg4_1gb --config cfg4 --contigs 10 --contig-len 100000000
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -x -q -m gpu > $O/pytest_parity.txt 2>&1; tail -3 $O/pytest_parity.txt
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "1m_reads or rank_share_equals_oracle" > $O/pytest_cfg3.txt 2>&1; tail -5 $O/pytest_cfg3.txt
