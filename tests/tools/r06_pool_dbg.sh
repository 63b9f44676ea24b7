#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06pooldbg}; mkdir -p $O
Q="--no-cpu-baseline --no-pcie --no-qv --no-packed --no-nofilter --steps 3 --warmup 1 --config cfg4 --contigs 10 --contig-len 100000000"
run() { name=$1; shift; YAKAMD_VERBOSE=1 timeout 100 python bench.py $Q "$@" > $O/b_$name.json 2> $O/b_$name.err; rc=$?
  python3 - $O/b_$name.json $name $rc <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    v = d.get("verify") or {}
    print(sys.argv[2], "rc", sys.argv[3], "ms", round(d["ms_per_step"], 1), v.get("yak_md5"), {k: x for k, x in v.items() if isinstance(x, bool)})
except Exception as e:
    print(sys.argv[2], "rc", sys.argv[3], "FAILED", e)
PY
  grep "refused\|fault" $O/b_$name.err | sort | uniq -c | head -4; }
run vm_keepva --knob YAKAMD_POOL_VM_KEEPVA=1
