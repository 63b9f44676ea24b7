#!/bin/bash
# a short verified bench line of the default workload -> gpurun_out/$1/b.json
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-chk}; mkdir -p $O; shift
YAKAMD_VERBOSE=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>$O/b.err | grep '^{' > $O/b.json
grep -E "key-owning|k_lc2" $O/b.err | tail -2
python - $O <<'PY'
import json,sys
d = json.load(open(sys.argv[1] + "/b.json")); p=d["phase_ms_last_step"]
print("step", round(d["ms_per_step"],2), p, d["phase_wall_ms_last_step"], d["verify"], d.get("qv_lookup_probe",{}).get("ms"))
r = d["roofline"]; print({k: r[k] for k in r if "frac" in k})
PY
