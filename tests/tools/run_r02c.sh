#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c; mkdir -p $O
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-qv"
echo "== bench lc2"; YAKAMD_VERBOSE=1 timeout 300 $B 2>$O/b.err | grep '^{' > $O/b.json; grep -E "k_lc2|lds tier" $O/b.err | tail -2; tail -3 $O/b.err | grep -v yak_amd
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r02c/b.json")); print("step", d["ms_per_step"], d["phase_ms_last_step"]["pass1"], d["verify"])
except Exception as e: print("bench failed", e)
PY
for D in 16 32; do YAKAMD_VERBOSE=1 YAKAMD_DBG=$D timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify --no-qv 2>&1 >/dev/null | grep -E "k_lc2" | tail -1; done
for W in 512 1024 2560; do echo "wgs $W"; YAKAMD_LC2_WGS=$W YAKAMD_VERBOSE=1 timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify --no-qv 2>&1 >/dev/null | grep -E "k_lc2" | tail -1; done
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
