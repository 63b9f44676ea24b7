#!/bin/bash
# kernel stats of one config: r04_prof.sh <name> <bench args...>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
N=$1; shift
O=gpurun_out/r04prof; mkdir -p $O
Q="--steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-qv --no-pcie --no-packed --no-nofilter"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${N}_trace -- python bench.py $Q "$@" > $O/${N}.json 2>/dev/null
cp $(find $O/${N}_trace -name "*kernel_stats.csv" | head -1) $O/${N}_kernel_stats.csv
rm -rf $O/${N}_trace
python3 - $O/${N}_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.25: print(r["Name"][:70].ljust(70), r["Calls"].rjust(5), "%9.3f ms avg" % (float(r["AverageNs"])/1e6), "%9.2f ms tot" % (float(r["TotalDurationNs"])/1e6), r["Percentage"]+"%")
PY
