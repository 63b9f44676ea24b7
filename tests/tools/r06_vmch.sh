#!/bin/bash
# the pool's large-buffer tier with smaller chunks and a lower threshold: how much device memory the 5 Gb job (8 sweeps, the library's own rule) obtains and what its first job costs
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06vmch}; mkdir -p $O
run() {  # name, knobs...
  local name=$1; shift
  sleep 10
  YAKAMD_VERBOSE=1 timeout 400 python bench.py --config cfg4 --contigs 50 --warmup 1 --no-verify "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python3 - $O/bench_$name.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], "ms", round(d["ms_per_step"], 2), d["config"].get("sweeps_of_every_job"), {k: d[k] for k in d if k.startswith(("first_job", "peak_hbm_bytes")) and "note" not in k})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
  grep "pool after" $O/bench_$name.err | tail -2 | cut -c1-420
}
run base
run ch64_min256 --knob YAKAMD_POOL_VM_CH=67108864 --knob YAKAMD_POOL_VM_MIN=268435456
run ch64_min64 --knob YAKAMD_POOL_VM_CH=67108864 --knob YAKAMD_POOL_VM_MIN=67108864
run ch32_min32 --knob YAKAMD_POOL_VM_CH=33554432 --knob YAKAMD_POOL_VM_MIN=33554432
