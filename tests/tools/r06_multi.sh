#!/bin/bash
# the multi-GPU code on the one-GPU box: the tests that force a slot per rank (exchange between slots of device 0, in-process rig for the grouped calls),
# then the N = 2 bench line (weak_base, cpu_baseline) as it is and with slots per rank, rounds overlapped and stage by stage
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06multi}; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_multi_c.py tests/test_gpu_multirank.py -m gpu -x -q ${K:+-k "$K"} ) > $O/pytest.log 2>&1; tail -8 $O/pytest.log
R=${READS:-37500000}
timeout 900 python bench.py --gpus 2 --reads $R --steps 2 --warmup 1 --no-verify > $O/bench_gpus2.json 2> $O/bench_gpus2.err
SL="--knob YAKAMD_MGPU_SLOT_PER_RANK=1 --knob YAKAMD_MGPU_LOOPBACK=1"
timeout 900 python bench.py --gpus 2 --reads $R --steps 2 --warmup 1 --no-verify --no-cpu-baseline --no-weak-base $SL > $O/bench_gpus2_slots.json 2> $O/bench_gpus2_slots.err
timeout 900 python bench.py --gpus 2 --reads $R --steps 2 --warmup 1 --no-verify --no-cpu-baseline --no-weak-base $SL --knob YAKAMD_MGPU_NO_OVERLAP=1 > $O/bench_gpus2_slots_nooverlap.json 2> $O/bench_gpus2_slots_nooverlap.err
for f in gpus2 gpus2_slots gpus2_slots_nooverlap; do python3 - $O/bench_$f.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    wb = d.get("weak_base") or {}
    print(sys.argv[1].split("/")[-1], "ms", round(d["ms_per_step"], 1), "first", d.get("first_job_ms"), d["config"]["exchange"][:40], "| base", wb.get("ms_per_step"), wb.get("base_ms_over_job_ms"), "| cpu", (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
grep -v "^$\|amdgpu.ids\|socket.cpp\|OMP_NUM\|\*\*\*\*" $O/bench_gpus2.err | tail -12
