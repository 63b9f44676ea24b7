#!/bin/bash
# cfg3shard with the pool's roomy threshold at 75 / 85 per cent of the device (default 60)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06roomy2}; mkdir -p $O
for r in 75 85; do
  sleep 5
  YAKAMD_VERBOSE=1 timeout 600 python bench.py --config cfg3shard --warmup 1 --knob YAKAMD_POOL_VM_ROOMY=$r > $O/bench_cfg3shard_$r.json 2> $O/bench_cfg3shard_$r.err
  python3 - $O/bench_cfg3shard_$r.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], "ms", round(d["ms_per_step"], 2), d.get("rank_seconds"), d.get("first_job_rank_seconds"), {k: x for k, x in (d.get("verify") or {}).items() if isinstance(x, bool)})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
  grep "pool after" $O/bench_cfg3shard_$r.err | tail -2 | cut -c1-520
done
