#!/bin/bash
# round-6 final measurements, part D: the lines the pool's 1 GiB threshold (back from 2 GiB after part C) touches, on the final host code
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06final_d; mkdir -p $O
Q="--no-cpu-baseline --no-pcie --no-qv --no-packed --no-nofilter"
timeout 600 python bench.py $Q > $O/bench_default_short.json 2> /dev/null
timeout 600 python bench.py --config cfg4 --contigs 10 --contig-len 100000000 > $O/bench_cfg4_1gb.json 2> /dev/null
YAKAMD_VERBOSE=1 timeout 900 python bench.py --config cfg3shard --warmup 1 > $O/bench_cfg3shard.json 2> $O/bench_cfg3shard.err
sleep 5
YAKAMD_VERBOSE=1 timeout 900 python bench.py --config cfg4 --contigs 50 --warmup 1 > $O/bench_cfg4_5gb_sweeps8.json 2> $O/bench_cfg4_5gb.err
sleep 5
timeout 900 python bench.py --gpus 2 --reads 37500000 --steps 2 --warmup 1 > $O/bench_gpus2.json 2> $O/bench_gpus2.err
timeout 900 python bench.py --gpus 2 --reads 37500000 --steps 2 --warmup 1 --no-verify --no-cpu-baseline --no-weak-base --knob YAKAMD_MGPU_SLOT_PER_RANK=1 --knob YAKAMD_MGPU_LOOPBACK=1 > $O/bench_gpus2_slots.json 2> $O/bench_gpus2_slots.err
grep "ranks: input\|pool after" $O/bench_cfg4_5gb.err | head -4 > $O/cfg4_5gb_stages.txt
grep "pool after\|level-2 partition\|k_lc2\|slice of the pass" $O/bench_cfg3shard.err | tail -8 > $O/cfg3shard_stages.txt
for f in default_short cfg4_1gb cfg3shard cfg4_5gb_sweeps8 gpus2 gpus2_slots; do python3 - $O/bench_$f.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d.get("roofline", {})
    v = d.get("verify") or {}
    print(sys.argv[1].split("/")[-1], "ms", round(d["ms_per_step"], 2), "value", round(d["value"] / 1e6, 1), "M/s", {k: round(x, 4) for k, x in r.items() if ("frac" in k or k == "hbm_util") and isinstance(x, float)}, {k: x for k, x in v.items() if isinstance(x, bool)}, {k: d[k] for k in d if k.startswith(("first_job", "peak_hbm_bytes")) and "note" not in k})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
timeout 600 python -m pytest tests/test_multi_c.py -m gpu -x -q -k "named_pipes or sweeps" 2>&1 | tail -3
