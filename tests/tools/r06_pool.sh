#!/bin/bash
# the large-buffer tier of the device pool (physical chunks behind virtual ranges): API / parity tests, then first job vs warm of the large configurations
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06pool}; mkdir -p $O
timeout 700 python -m pytest tests/test_gpu_api.py tests/test_qv.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
Q="--no-cpu-baseline --no-pcie --no-qv --no-packed --no-nofilter --steps 3 --warmup 1"
YAKAMD_VERBOSE=1 timeout 200 python bench.py $Q > $O/bench_default.json 2> $O/bench_default.err
YAKAMD_VERBOSE=1 timeout 200 python bench.py --config cfg4 --contigs 10 --contig-len 100000000 $Q > $O/bench_cfg4_1gb.json 2> $O/bench_cfg4_1gb.err
YAKAMD_VERBOSE=1 timeout 200 python bench.py --config cfg4 --contigs 20 --contig-len 100000000 $Q > $O/bench_cfg4_2gb.json 2> $O/bench_cfg4_2gb.err
sleep 5
YAKAMD_VERBOSE=1 timeout 300 python bench.py --config cfg4 --contigs 50 --warmup 1 > $O/bench_cfg4_5gb_sweeps2.json 2> $O/bench_cfg4_5gb.err
sleep 5
YAKAMD_VERBOSE=1 timeout 300 python bench.py --config cfg3shard --warmup 1 > $O/bench_cfg3shard.json 2> $O/bench_cfg3shard.err
for f in default cfg4_1gb cfg4_2gb cfg4_5gb_sweeps2 cfg3shard; do python3 - $O/bench_$f.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    v = d.get("verify") or {}
    print(sys.argv[1].split("/")[-1], "ms", round(d["ms_per_step"], 2), {k: x for k, x in v.items() if isinstance(x, bool)}, {k: d[k] for k in d if k.startswith(("first_job", "peak_hbm_bytes")) and "note" not in k})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
for f in default cfg4_1gb cfg4_2gb cfg4_5gb cfg3shard; do echo "== $f"; grep "pool after" $O/bench_$f.err | tail -2 | cut -c1-700; done
