#!/bin/bash
# the pool's large-buffer tier without taking ranges apart while the device is roomy: the lines that lost 5-10 % to remapping (default 50.7 -> 56.2 ms)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06roomy}; mkdir -p $O
Q="--no-cpu-baseline --no-pcie --no-qv --no-packed --no-nofilter --steps 3 --warmup 1"
run() { local name=$1; shift; YAKAMD_VERBOSE=1 timeout 600 python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python3 - $O/bench_$name.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    v = d.get("verify") or {}
    print(sys.argv[1].split("/")[-1], "ms", round(d["ms_per_step"], 2), {k: x for k, x in v.items() if isinstance(x, bool)}, {k: d[k] for k in d if k.startswith(("first_job_ms", "peak_hbm_bytes")) and "note" not in k})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
  grep "pool after" $O/bench_$name.err | tail -2 | cut -c1-520; }
run default $Q
run default_vm0 $Q --knob YAKAMD_POOL_VM=0
run nofilter --config nofilter $Q
run 30m --reads 30000000 $Q
run cfg4_1gb --config cfg4 --contigs 10 --contig-len 100000000 $Q
run cfg4_2gb --config cfg4 --contigs 20 --contig-len 100000000 $Q
run cfg3shard --config cfg3shard --warmup 1
sleep 5
run cfg4_5gb --config cfg4 --contigs 50 --warmup 1
timeout 700 python -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
