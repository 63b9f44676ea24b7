#!/bin/bash
# timing ablations of k_lds_count (results are wrong with YAKAMD_DBG != 0; timing only)
for D in 0 16 32; do
  YAKAMD_VERBOSE=1 YAKAMD_DBG=$D timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify 2>/tmp/abl.err | grep "^{" > /tmp/abl.json
  grep "lds tier" /tmp/abl.err | tail -1
  python - <<PY
import json
d = json.load(open("/tmp/abl.json"))
print("dbg", $D, "k_lds_count ms", d["phase_ms_last_step"]["pass1"]["ms_insert"])
PY
done
