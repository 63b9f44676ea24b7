#!/bin/bash
# the three lines of final_r04.sh that changed afterwards (cfg5: traffic of the step's own kernels; 5 Gb: md5 in a job of its own; cfg3shard once more)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04final; mkdir -p $O
timeout 600 python bench.py --config cfg5 > $O/bench_cfg5.json 2> /dev/null
YAKAMD_VERBOSE=1 timeout 900 python bench.py --config cfg3shard --warmup 1 > $O/bench_cfg3shard.json 2> $O/bench_cfg3shard.err
YAKAMD_VERBOSE=1 timeout 900 python bench.py --config cfg4 --contigs 50 --warmup 1 > $O/bench_cfg4_5gb_sweeps2.json 2> $O/bench_cfg4_5gb.err
grep "ranks: input\|pool after" $O/bench_cfg4_5gb.err | head -6 > $O/cfg4_5gb_stages.txt
for f in cfg5 cfg3shard cfg4_5gb_sweeps2; do python3 - $O/bench_$f.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = d.get("roofline", {})
print(sys.argv[1].split("/")[-1], "ms", round(d["ms_per_step"], 2), d.get("seconds_second_chunking"), d.get("rank_seconds"), {k: round(v, 4) for k, v in r.items() if ("frac" in k or k == "hbm_util") and isinstance(v, float)}, d.get("verify"))
PY
done
cat $O/cfg4_5gb_stages.txt
