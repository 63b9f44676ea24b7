#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04t5; mkdir -p $O
Q="--steps 3 --warmup 1 --no-cpu-baseline --no-qv --no-pcie --no-packed --no-nofilter"
for cfg in "" "--config nofilter" "--config cfg4 --contigs 10 --contig-len 100000000"; do
timeout 300 python bench.py $Q $cfg > $O/b.json 2> $O/b.err
python3 - $O/b.json "$cfg" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
p = d["phase_ms_last_step"]
v = d["verify"]
print(sys.argv[2] or "default", round(d["ms_per_step"], 2), p["pass1"] if "pass1" in p else p, v.get("equals_reference"))
PY
done
