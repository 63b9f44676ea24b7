#!/bin/bash
# kernel-trace summary of any bench configuration: prof_stats_cfg.sh <name> <bench args...> -> gpurun_out/<name>_kernel_stats.csv (+ printed table, ms per step)
name=${1:-prof}; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$name -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-pcie --no-packed --no-nofilter --no-qv "$@" > gpurun_out/$name.json 2>/dev/null
f=$(find gpurun_out/$name -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${name}_kernel_stats.csv
rm -rf gpurun_out/$name
python3 - gpurun_out/${name}_kernel_stats.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r['Percentage']) > 0.4: print(r['Name'][:56].ljust(56), r['Calls'].rjust(4), '%8.3f ms avg' % (float(r['AverageNs'])/1e6), '%8.2f ms/step' % (float(r['TotalDurationNs'])/1e6/4), r['Percentage']+'%')
PY
grep -o '"ms_per_step": [0-9.]*' gpurun_out/$name.json
