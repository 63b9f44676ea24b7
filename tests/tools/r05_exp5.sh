#!/bin/bash
# round 5, fifth GPU call: k_lc2 with its cold arguments out of the scalar registers; superblocks of 16 GB (first-job times of the large configurations)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e5; mkdir -p $O
Q="--no-cpu-baseline --no-pcie --no-qv --no-packed --no-nofilter --steps 5 --warmup 2"
line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    ks = {k["kernel"].split(" (")[0][:34]: round(k["ms"], 2) for k in d["roofline"].get("all_kernels", [])} if "all_kernels" in d.get("roofline", {}) else {}
    v = d.get("verify") or {}
    print(sys.argv[1].ljust(20), "ms", round(d["ms_per_step"], 2), "verify", {k: v[k] for k in v if isinstance(v[k], bool)}, ks, d.get("phase_wall_ms_last_step"), {k: d[k] for k in d if k.startswith(("first_job", "peak_hbm_bytes", "rank_seconds")) and "note" not in k})
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[2].replace(".json", ".err")).read()[-600:])
PY
}
run() { local name=$1; shift; YAKAMD_VERBOSE=1 timeout 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err; line $name $O/$name.json; }
run default $Q
run nofilter --config nofilter $Q
run cfg4_5gb --config cfg4 --contigs 50 --warmup 1
grep -h "ranks: input\|pool after" $O/cfg4_5gb.err | head -4
run cfg4_5gb_q0 --config cfg4 --contigs 50 --knob YAKAMD_POOL_QUANTUM_MB=0 --no-verify
grep -h "ranks: input\|pool after" $O/cfg4_5gb_q0.err | head -2
run cfg3shard --config cfg3shard --warmup 1
grep -h "pool after" $O/cfg3shard.err | head -3
run cfg4_1gb --config cfg4 --contigs 10 --contig-len 100000000 --steps 3 --warmup 1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -x -q -m gpu > $O/pytest_parity.txt 2>&1; tail -2 $O/pytest_parity.txt
