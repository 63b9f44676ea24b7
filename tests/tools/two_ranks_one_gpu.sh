#!/bin/bash
# the N > 1 bench path with two ranks sharing one GPU (gloo instead of RCCL): its result must equal
# the single-rank run on the same logical input (2 x reads)
set -u
cd "$(dirname "$0")/../.."
R=${1:-1000000}
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
  bench.py --gpus 2 --backend gloo --reads $R --bf-shift 34 --steps 1 --warmup 0 --no-cpu-baseline --no-verify 2>/tmp/two.err | grep "^{" > /tmp/two.json || { tail -20 /tmp/two.err; exit 1; }
python bench.py --reads $((2*R)) --bf-shift 34 --steps 1 --warmup 0 --no-cpu-baseline --no-verify 2>/dev/null | grep "^{" > /tmp/one.json
python - <<PY
import json
a, b = json.load(open("/tmp/two.json")), json.load(open("/tmp/one.json"))
print("2 ranks: final_distinct", a["final_distinct"], "ms", round(a["ms_per_step"], 1))
print("1 rank : final_distinct", b["final_distinct"], "ms", round(b["ms_per_step"], 1))
print("MATCH" if a["final_distinct"] == b["final_distinct"] else "MISMATCH")
PY
