#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02e; mkdir -p $O
one() { # label, env...
  local label=$1; shift
  env "$@" YAKAMD_VERBOSE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-qv 2>$O/e.err | grep '^{' > $O/$label.json
  grep -E "key-owning" $O/e.err | tail -1
  python - $label <<'PY'
import json,sys
try:
    d = json.load(open(f"gpurun_out/r02e/{sys.argv[1]}.json")); p=d["phase_ms_last_step"]
    print(sys.argv[1], "step", round(d["ms_per_step"],2), "lds", p["pass1"]["ms_insert"], "replay", p["pass1"]["ms_replay"], "p2count", p["pass2"]["ms_insert"], "shrink", d["phase_wall_ms_last_step"]["shrink"], "ok", d["verify"]["equals_reference"])
except Exception as e: print(sys.argv[1], "failed", e)
PY
}
one base A=1
one own156 YAKAMD_OWN_LDS=156000
one own120 YAKAMD_OWN_LDS=120000
one rp16k YAKAMD_REPLAY_LDS=16384
one rp8k YAKAMD_REPLAY_LDS=8192
one rp16k512 YAKAMD_REPLAY_LDS=16384 YAKAMD_REPLAY_THREADS=512
