#!/bin/bash
# which buffer of the layout stage is read before it is written?  One bench run per buffer, that buffer filled with 0xA5 first (YAKAMD_R2_POISON bit i), classic pool
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06poison}; mkdir -p $O
Q="--no-cpu-baseline --no-pcie --no-qv --no-packed --no-nofilter --steps 1 --warmup 0 --config cfg4 --contigs 10 --contig-len 100000000 --knob YAKAMD_POOL_VM=0"
for i in ${BITS:-none 0 1 2 3 4 5 6 7 8 9 10 12}; do
  K=""; [ $i != none ] && K="--knob YAKAMD_R2_POISON=$((1 << i))"
  YAKAMD_VERBOSE=1 timeout 120 python bench.py $Q $K > $O/b_$i.json 2> $O/b_$i.err; rc=$?
  python3 - $O/b_$i.json $i $rc <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    v = d.get("verify") or {}
    print("buffer", sys.argv[2], "rc", sys.argv[3], "ms", round(d["ms_per_step"], 1), v.get("yak_md5"), {k: x for k, x in v.items() if isinstance(x, bool)})
except Exception as e:
    print("buffer", sys.argv[2], "rc", sys.argv[3], "FAILED", e)
PY
  grep "refused\|fault" $O/b_$i.err | head -3
done
