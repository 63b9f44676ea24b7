#!/bin/bash
# the 5 Gb assembly (BASELINE configs[3]) in 4 and 8 sweeps instead of 2: what a one-job process pays (first_job_ms) against the warm step
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06sweeps}; mkdir -p $O
for s in 4 8; do
  YAKAMD_VERBOSE=1 timeout 400 python bench.py --config cfg4 --contigs 50 --sweeps $s --warmup 1 > $O/bench_cfg4_5gb_sweeps$s.json 2> $O/bench_cfg4_5gb_sweeps$s.err
  python3 - $O/bench_cfg4_5gb_sweeps$s.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], "ms", round(d["ms_per_step"], 2), {k: x for k, x in (d.get("verify") or {}).items() if isinstance(x, bool)}, {k: d[k] for k in d if k.startswith(("first_job", "peak_hbm_bytes")) and "note" not in k})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
  grep "pool after" $O/bench_cfg4_5gb_sweeps$s.err | tail -2 | cut -c1-600
  sleep 5
done
