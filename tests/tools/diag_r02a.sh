#!/bin/bash
# round-2 diagnostics: where the time goes inside k_replay and k_lds_count, scatter microbenchmark
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02a; mkdir -p $O
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify --no-qv"
echo "== mb_scatter" ; (cd tests/tools/mb && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 mb_scatter.hip -o mb_scatter 2>/dev/null; timeout 120 ./mb_scatter) | tee $O/mb_scatter.txt
echo "== replay profile (block 0)"; YAKAMD_DBG=32 YAKAMD_VERBOSE=1 timeout 300 $B 2>$O/replay_prof.err | grep '^{' > $O/replay_prof.json; grep -E "replay block|lds tier" $O/replay_prof.err
echo "== lds ablations"; bash tests/tools/ablate_lds.sh 2>&1 | tee $O/ablate.txt
echo "== no filter"; YAKAMD_DBG=32 YAKAMD_VERBOSE=1 timeout 300 $B --bf-shift 0 2>$O/nofilter.err | grep '^{' > $O/nofilter.json; grep -E "replay block" $O/nofilter.err
python - <<'PY'
import json
for n in ("replay_prof", "nofilter"):
    d = json.load(open(f"gpurun_out/r02a/{n}.json"))
    print(n, d["ms_per_step"], d["phase_ms_last_step"], d["phase_wall_ms_last_step"])
PY
