#!/bin/bash
# round-4 probe 1: what the bloom write-back costs k_lc2 (YAKAMD_DBG=64 skips it), per-kernel stats + layout dispatch trace of the unfiltered protocol
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04p1; mkdir -p $O
Q="--steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-qv --no-pcie --no-packed --no-nofilter"
YAKAMD_VERBOSE=1 timeout 300 python bench.py $Q > $O/default.json 2> $O/default.err
YAKAMD_VERBOSE=1 YAKAMD_DBG=64 timeout 300 python bench.py $Q > $O/dbg64.json 2> $O/dbg64.err
grep -h "k_lc2" $O/default.err | tail -2; grep -h "k_lc2" $O/dbg64.err | tail -2
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/nf_trace -- python bench.py --config nofilter $Q > $O/nofilter_prof.json 2>/dev/null
cp $(find $O/nf_trace -name "*kernel_stats.csv" | head -1) $O/nofilter_kernel_stats.csv
rm -rf $O/nf_trace
R2OUT=r04p1/nf_r2 bash tests/tools/trace_r2.sh --config nofilter > $O/nofilter_r2.txt 2>&1
python3 - $O/nofilter_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.3: print(r["Name"][:60].ljust(60), r["Calls"].rjust(5), "%9.3f ms avg" % (float(r["AverageNs"])/1e6), "%9.2f ms tot" % (float(r["TotalDurationNs"])/1e6), r["Percentage"]+"%")
PY
head -40 $O/nofilter_r2.txt
