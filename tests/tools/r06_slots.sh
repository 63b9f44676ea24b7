#!/bin/bash
# the N = 2 line with a slot per rank on one device (176 GB in use): why 576 -> 1431 ms after the pool changes?
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06slots}; mkdir -p $O
SL="--knob YAKAMD_MGPU_SLOT_PER_RANK=1 --knob YAKAMD_MGPU_LOOPBACK=1"
run() { local name=$1; shift
  sleep 5
  YAKAMD_VERBOSE=1 timeout 900 python bench.py --gpus 2 --reads 37500000 --steps 2 --warmup 1 --no-verify --no-cpu-baseline --no-weak-base $SL "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python3 - $O/bench_$name.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], "ms", round(d["ms_per_step"], 1), "first", d.get("first_job_ms"), d.get("peak_hbm_bytes_per_device"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
  grep "pool after" $O/bench_$name.err | tail -2 | cut -c1-560; }
run default
run vmmin1g --knob YAKAMD_POOL_VM_MIN=1073741824
run roomy0 --knob YAKAMD_POOL_VM_ROOMY=0
run roomy0_vmmin1g --knob YAKAMD_POOL_VM_ROOMY=0 --knob YAKAMD_POOL_VM_MIN=1073741824
