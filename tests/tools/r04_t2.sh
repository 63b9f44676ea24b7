#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04t2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -x -q 2>&1 | tail -8 > $O/pytest.txt
cat $O/pytest.txt
Q="--steps 3 --warmup 1 --no-cpu-baseline --no-qv --no-pcie --no-packed --no-nofilter"
timeout 300 python bench.py $Q > $O/default.json 2> $O/default.err
python3 - $O/default.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("default", round(d["ms_per_step"], 2), d["phase_ms_last_step"]["pass1"], d["verify"])
PY
