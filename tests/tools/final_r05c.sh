#!/bin/bash
# round 5, last change: k_r2_place / k_r2_headfill dealt over the sub-tables that place (a shard's contiguous range used to land on one or four of the eight XCDs).
# Parity of the sharded paths, then the bench lines of the configurations it changes (each verifies its bytes: oracle share md5 / golden md5)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05final_c; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_multi_c.py -q -x -k "streaming_replay or large_subtables or shard or multi or sweep" > $O/tests.txt 2>&1; tail -2 $O/tests.txt
line() { python3 - $1 <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], "ms", round(d["ms_per_step"], 2), "value", round(d["value"] / 1e6, 1), "M/s", {k: x for k, x in (d.get("verify") or {}).items() if isinstance(x, bool)}, {k: d[k] for k in d if k.startswith("first_job")})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
YAKAMD_VERBOSE=1 timeout 600 python bench.py --config cfg3shard --warmup 1 > $O/bench_cfg3shard.json 2> $O/bench_cfg3shard.err; line $O/bench_cfg3shard.json
timeout 600 python bench.py --config cfg3shard --rank 5 --warmup 1 > $O/bench_cfg3shard_rank5.json 2> /dev/null; line $O/bench_cfg3shard_rank5.json
sleep 3
YAKAMD_VERBOSE=1 timeout 900 python bench.py --config cfg4 --contigs 50 --warmup 1 > $O/bench_cfg4_5gb_sweeps2.json 2> $O/bench_cfg4_5gb.err; line $O/bench_cfg4_5gb_sweeps2.json
timeout 300 python bench.py --config cfg4 --contigs 10 --contig-len 100000000 --sweeps 4 > $O/bench_cfg4_1gb_sweeps4.json 2> /dev/null; line $O/bench_cfg4_1gb_sweeps4.json
grep "ranks: input\|pool after" $O/bench_cfg4_5gb.err | head -4 > $O/cfg4_5gb_stages.txt
grep "pool after\|level-2 partition\|k_lc2\|slice of the pass" $O/bench_cfg3shard.err | tail -8 > $O/cfg3shard_stages.txt
