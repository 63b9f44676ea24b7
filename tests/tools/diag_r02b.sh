#!/bin/bash
# replay occupancy experiments: fewer threads / less LDS per workgroup -> more workgroups per CU
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02b; mkdir -p $O
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-qv"
for cfg in "1024 32768" "512 32768" "512 16384" "512 8192" "256 8192" "256 16384"; do
  set -- $cfg
  YAKAMD_REPLAY_THREADS=$1 YAKAMD_REPLAY_LDS=$2 timeout 300 $B 2>$O/e.err | grep '^{' > $O/r_$1_$2.json || tail -3 $O/e.err
  python - "$1" "$2" <<'PY'
import json,sys
d = json.load(open(f"gpurun_out/r02b/r_{sys.argv[1]}_{sys.argv[2]}.json"))
p = d["phase_ms_last_step"]
print("threads", sys.argv[1], "lds_words", sys.argv[2], "step", round(d["ms_per_step"],2), "replay p1", p["pass1"]["ms_replay"], "shrink", d["phase_wall_ms_last_step"]["shrink"], "verify", d["verify"] and d["verify"].get("equals_reference"))
PY
done
