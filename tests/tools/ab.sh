#!/bin/bash
# A/B of env settings on the default bench: each argument is "VAR=val VAR=val" (or "-" for none); prints the phase times
cd $GRAFT_REPO_ROOT
for e in "$@"; do
  [ "$e" = "-" ] && e=""
  echo "[$e] $(env $e timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-qv --no-pcie 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); p=d["phase_ms_last_step"]; print(round(d["ms_per_step"],2), {k: round(v,2) for k,v in p["pass1"].items() if v}, {k: round(v,2) for k,v in p["pass2"].items() if v}, d["phase_wall_ms_last_step"]["shrink"])')"
done
