#!/bin/bash
# kernel-trace summary of the default bench run -> gpurun_out/<name>_kernel_stats.csv (+ printed table)
name=${1:-prof}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$name -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify > gpurun_out/$name.json 2>/dev/null
f=$(find gpurun_out/$name -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${name}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r['Percentage']) > 0.2: print(r['Name'][:60].ljust(60), r['Calls'].rjust(4), '%8.3f ms avg' % (float(r['AverageNs'])/1e6), r['Percentage']+'%')
PY
grep -o '"ms_per_step": [0-9.]*' gpurun_out/$name.json
