#!/bin/bash
# whole -m gpu tier, then one line of every configuration
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06all}; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > $O/gpu_test_tier.txt 2>&1; tail -6 $O/gpu_test_tier.txt
Q="--no-cpu-baseline --no-pcie --no-qv --no-packed --no-nofilter --steps 3 --warmup 1"
timeout 600 python bench.py $Q > $O/bench_default.json 2> /dev/null
timeout 600 python bench.py --config nofilter $Q > $O/bench_nofilter.json 2> /dev/null
timeout 600 python bench.py --config cfg4 --contigs 10 --contig-len 100000000 $Q > $O/bench_cfg4_1gb.json 2> /dev/null
timeout 600 python bench.py --config cfg4 --contigs 20 --contig-len 100000000 $Q > $O/bench_cfg4_2gb.json 2> /dev/null
timeout 600 python bench.py --reads 30000000 $Q > $O/bench_30m.json 2> /dev/null
YAKAMD_VERBOSE=1 timeout 900 python bench.py --config cfg3shard --warmup 1 > $O/bench_cfg3shard.json 2> $O/bench_cfg3shard.err
for f in default nofilter cfg4_1gb cfg4_2gb 30m cfg3shard; do python3 - $O/bench_$f.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    v = d.get("verify") or {}
    print(sys.argv[1].split("/")[-1], "ms", round(d["ms_per_step"], 2), "value", round(d["value"] / 1e6, 1), "M/s", {k: x for k, x in v.items() if isinstance(x, bool)}, {k: d[k] for k in d if k.startswith(("first_job", "rank_seconds"))})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
