#!/bin/bash
# a cfg3 rank share with the level-2 partition forced into ONE sweep of 2^11 / 2^12 / 2^13 sub-buckets per sub-table (k_lc2 then takes sub-buckets of 34 K / 17 K / 8 K records in rounds of 768)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06s2b}; mkdir -p $O
for s in 13 12 11; do
  YAKAMD_VERBOSE=1 timeout 400 python bench.py --config cfg3shard --knob YAKAMD_S2_BITS=$s > $O/bench_cfg3shard_s$s.json 2> $O/bench_cfg3shard_s$s.err
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
