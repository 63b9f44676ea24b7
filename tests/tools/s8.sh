#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/s8; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pcie --no-qv --no-packed --no-nofilter"
timeout 300 $B > $O/bench_default.json 2> $O/bench_default.err
YAKAMD_VERBOSE=1 timeout 300 python bench.py --reads 30000000 --steps 2 --warmup 1 --no-cpu-baseline --no-pcie --no-qv --no-packed --no-nofilter > $O/bench_30m.json 2> $O/bench_30m.err
grep "k_lc2" $O/bench_30m.err | tail -2
timeout 300 $B --no-verify --config nofilter > $O/bench_nofilter.json 2>/dev/null
timeout 600 python bench.py --config cfg4 --contigs 10 --contig-len 100000000 > $O/bench_cfg4_1gb.json 2> $O/bench_cfg4_1gb.err
for f in default 30m nofilter cfg4_1gb; do python3 - $O/bench_$f.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], "ms", round(d["ms_per_step"], 2), "value", round(d["value"] / 1e6, 1), "M/s", json.dumps(d.get("phase_ms_last_step")), d.get("verify"), d["roofline"].get("step_frac"), d["roofline"].get("pass1_frac"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
YAKAMD_VERBOSE=2 timeout 600 python bench.py --config cfg4 --contigs 10 --contig-len 100000000 --steps 1 --warmup 1 --no-verify > $O/cfg4_verbose.json 2> $O/cfg4_verbose.err
grep -E "replay2 step|k_r2_double" $O/cfg4_verbose.err | tail -60 | cut -c1-260
timeout 300 python tests/tools/rccl_big_msg.py --gib 3 > $O/rccl_big_msg.log 2>&1; echo "rccl_big_msg rc=$?" >> $O/rccl_big_msg.log; tail -5 $O/rccl_big_msg.log
