#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02d; mkdir -p $O
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-qv"
echo "== bench"; YAKAMD_VERBOSE=1 timeout 300 $B 2>$O/b.err | grep '^{' > $O/b.json; grep -E "k_lc2|lds tier|key-owning" $O/b.err | tail -3; grep -v yak_amd $O/b.err | tail -3
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r02d/b.json")); print("step", d["ms_per_step"], d["phase_ms_last_step"], d["phase_wall_ms_last_step"], d["verify"])
except Exception as e: print("bench failed", e)
PY
for D in 16 32; do YAKAMD_VERBOSE=1 YAKAMD_DBG=$D timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify --no-qv 2>&1 >/dev/null | grep -E "k_lc2" | tail -1; done
echo "== own lds 156000 (1 WG/CU, fewer ranges)"; YAKAMD_OWN_LDS=156000 YAKAMD_VERBOSE=1 timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify --no-qv 2>$O/c.err | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['phase_ms_last_step']['pass2'])"; grep key-owning $O/c.err | tail -1
echo "== own lds 40000"; YAKAMD_OWN_LDS=40000 YAKAMD_VERBOSE=1 timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify --no-qv 2>$O/c.err | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['phase_ms_last_step']['pass2'])"; grep key-owning $O/c.err | tail -1
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
