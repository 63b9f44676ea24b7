#!/bin/bash
# the round's standard check on the GPU box: whole -m gpu suite + a short verified bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-chk}; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -25
YAKAMD_VERBOSE=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>$O/b.err | grep '^{' > $O/b.json
grep -E "key-owning|k_lc2" $O/b.err | tail -2
python - $O <<'PY'
import json,sys
d = json.load(open(sys.argv[1] + "/b.json")); p=d["phase_ms_last_step"]
print("step", round(d["ms_per_step"],2), p, d["phase_wall_ms_last_step"], d["verify"], d.get("qv_lookup_probe",{}).get("ms"))
PY
