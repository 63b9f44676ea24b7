#!/bin/bash
# a cfg3 rank share with larger sub-buckets in TWO sweeps (2^4 groups, then 2^(s-4) sub-buckets each): s = 14, 13, 12
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06s2c}; mkdir -p $O
for s in 14 13 12; do
  YAKAMD_VERBOSE=1 timeout 400 python bench.py --config cfg3shard --knob YAKAMD_S2_BITS=$s --knob YAKAMD_P3_MIN=$((s-1)) > $O/bench_cfg3shard_s$s.json 2> $O/bench_cfg3shard_s$s.err
  python3 - $O/bench_cfg3shard_s$s.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], "ms", round(d["ms_per_step"], 2), d.get("rank_seconds"), {k: x for k, x in (d.get("verify") or {}).items() if isinstance(x, bool)})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
  grep -h "k_lc2:\|two sweeps\|passed on\|level-2" $O/bench_cfg3shard_s$s.err | tail -6
done
