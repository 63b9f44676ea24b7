#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c3; mkdir -p $O
YAKAMD_VERBOSE=${VERB:-2} timeout 1500 python bench.py --config cfg3shard "$@" > $O/cfg3shard.json 2> $O/cfg3shard.err
grep -v "^\[M::" $O/cfg3shard.err | tail -90
python3 - $O/cfg3shard.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("rank ms", round(d["ms_per_step"], 1), d["rank_seconds"], "peak GB", d["peak_hbm_bytes"] / 1e9, d["verify"], d["prediction"]["job_distinct_kmers_per_s"])
PY
