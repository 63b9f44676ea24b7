#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/s5; mkdir -p $O
timeout 900 python -m pytest tests/test_multi_c.py tests/test_gpu_api.py tests/test_gpu_multirank.py -m gpu -x -q > $O/pytest.log 2>&1; tail -6 $O/pytest.log
timeout 300 python bench.py --config cfg3shard --reads 2000000 --batch-reads 1000000 > $O/cfg3shard_small.json 2> $O/cfg3shard_small.err; tail -3 $O/cfg3shard_small.err
timeout 900 python bench.py --config cfg3shard > $O/cfg3shard.json 2> $O/cfg3shard.err; tail -3 $O/cfg3shard.err
timeout 600 python bench.py --config cfg4 --contigs 50 --sweeps 4 > $O/cfg4_5gb_s4.json 2> $O/cfg4_5gb_s4.err; tail -5 $O/cfg4_5gb_s4.err
timeout 600 python bench.py --config cfg4 --contigs 50 --sweeps 2 > $O/cfg4_5gb_s2.json 2> $O/cfg4_5gb_s2.err; tail -5 $O/cfg4_5gb_s2.err
for f in cfg3shard_small cfg3shard cfg4_5gb_s4 cfg4_5gb_s2; do python3 - $O/$f.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], "ms", round(d["ms_per_step"], 2), "value", round(d["value"] / 1e6, 1), "M/s", {k: d[k] for k in ("rank_seconds", "peak_hbm_bytes", "prediction", "seconds_second_chunking", "verify") if k in d})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
