#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/s9; mkdir -p $O
C4="python bench.py --config cfg4 --contigs 10 --contig-len 100000000 --steps 2 --warmup 1 --no-verify"
run() { name=$1; shift; env "$@" timeout 300 $C4 > $O/$name.json 2>/dev/null; python3 - $O/$name.json $name <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[2], "ms", round(d["ms_per_step"], 2), json.dumps(d.get("phase_ms_last_step")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run base A=1
run seg13_t512 YAKAMD_R2_SEG_LOG=13 YAKAMD_R2_PLACE_THREADS=512
run seg13_t1024 YAKAMD_R2_SEG_LOG=13
run seg14_t512 YAKAMD_R2_PLACE_THREADS=512
run seg12_t256 YAKAMD_R2_SEG_LOG=12 YAKAMD_R2_PLACE_THREADS=256
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pcie --no-qv --no-packed --no-nofilter --no-verify"
for e in "A=1" "YAKAMD_R2_SEG_LOG=13 YAKAMD_R2_PLACE_THREADS=512" "YAKAMD_R2_SEG_LOG=12 YAKAMD_R2_PLACE_THREADS=256"; do
  env $e timeout 300 $B > $O/d.json 2>/dev/null; echo "default [$e]: $(grep -o '"ms_per_step": [0-9.]*' $O/d.json) $(grep -o '"ms_replay": [0-9.]*' $O/d.json | head -1)"
done
