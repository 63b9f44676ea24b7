#!/bin/bash
# correctness of the tagged records at size (repeated) + kernel trace of the default bench
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5
timeout 600 python tests/tools/diag_rec8b.py 2>&1 | grep -v amdgpu.ids > $O/diag.log; tail -15 $O/diag.log
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-pcie --no-qv"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B > $O/bench_profiled.json 2>/dev/null
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; rm -rf $O/trace
python3 - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/r02o/kernel_stats.csv")):
    if float(r["Percentage"]) > 0.4: print(r["Name"][:48].ljust(48), r["Calls"].rjust(5), "%9.3f ms avg" % (float(r["AverageNs"])/1e6), "%9.2f ms tot" % (float(r["TotalDurationNs"])/1e6), r["Percentage"]+"%")
PY
