#!/bin/bash
# round-6 final measurements, part C = part B's bench lines, GPU test tier and CLI end to end again on the final host code (the kernels, their stats and their counter
# passes are those of parts A and B: the pool change between B and C touches no device code)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06final_c; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gpu_test_tier.txt 2>&1; tail -4 $O/gpu_test_tier.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default_steps20_warmup5.json 2> /dev/null
Q="--no-cpu-baseline --no-pcie --no-qv --no-packed --no-nofilter"
timeout 600 python bench.py --config nofilter $Q > $O/bench_nofilter.json 2> /dev/null
timeout 600 python bench.py --config cfg4 --contigs 10 --contig-len 100000000 > $O/bench_cfg4_1gb.json 2> /dev/null
timeout 600 python bench.py --config cfg4 --contigs 20 --contig-len 100000000 > $O/bench_cfg4_2gb.json 2> /dev/null
timeout 600 python bench.py --config cfg5 > $O/bench_cfg5.json 2> /dev/null
timeout 600 python bench.py --reads 30000000 $Q > $O/bench_30m.json 2> /dev/null
timeout 600 python bench.py --no-retain $Q --no-verify > $O/bench_noretain.json 2> /dev/null
YAKAMD_VERBOSE=1 timeout 900 python bench.py --config cfg3shard --warmup 1 > $O/bench_cfg3shard.json 2> $O/bench_cfg3shard.err
sleep 5
YAKAMD_VERBOSE=1 timeout 900 python bench.py --config cfg4 --contigs 50 --warmup 1 > $O/bench_cfg4_5gb_sweeps8.json 2> $O/bench_cfg4_5gb.err
sleep 5
timeout 900 python bench.py --config cfg4 --contigs 50 --sweeps 2 --warmup 1 --no-verify > $O/bench_cfg4_5gb_sweeps2.json 2> /dev/null
grep "ranks: input\|pool after" $O/bench_cfg4_5gb.err | head -4 > $O/cfg4_5gb_stages.txt
grep "pool after\|level-2 partition\|k_lc2\|slice of the pass" $O/bench_cfg3shard.err | tail -8 > $O/cfg3shard_stages.txt
sleep 5
timeout 900 python bench.py --gpus 2 --reads 37500000 --steps 2 --warmup 1 > $O/bench_gpus2.json 2> $O/bench_gpus2.err
timeout 900 python bench.py --gpus 2 --reads 37500000 --steps 2 --warmup 1 --no-verify --no-cpu-baseline --no-weak-base --knob YAKAMD_MGPU_SLOT_PER_RANK=1 --knob YAKAMD_MGPU_LOOPBACK=1 > $O/bench_gpus2_slots.json 2> $O/bench_gpus2_slots.err
for f in default default_steps20_warmup5 nofilter cfg4_1gb cfg4_2gb cfg5 30m noretain cfg3shard cfg4_5gb_sweeps8 cfg4_5gb_sweeps2 gpus2 gpus2_slots; do python3 - $O/bench_$f.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d.get("roofline", {})
    v = d.get("verify") or {}
    print(sys.argv[1].split("/")[-1], "ms", round(d["ms_per_step"], 2), "value", round(d["value"] / 1e6, 1), "M/s", {k: round(x, 4) for k, x in r.items() if ("frac" in k or k == "hbm_util") and isinstance(x, float)}, "traffic", r.get("traffic"), {k: x for k, x in v.items() if isinstance(x, bool)}, {k: d[k] for k in d if k.startswith(("first_job", "peak_hbm_bytes")) and "note" not in k})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
timeout 300 bash tests/tools/r04_e2e.sh gz > $O/e2e_cli.txt 2>&1
ls $O
