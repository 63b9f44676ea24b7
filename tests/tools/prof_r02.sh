#!/bin/bash
# round-2 profiles of the default bench command: kernel trace + the two PMC passes -> gpurun_out/r02_*
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-pcie --no-qv"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-pcie > gpurun_out/r02_bench_profiled.json 2>/dev/null
cp $(find gpurun_out/r02_trace -name "*kernel_stats.csv" | head -1) gpurun_out/r02_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/r02_$c -o pmc -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-qv --no-pcie > gpurun_out/r02_$c.log 2>&1
done
python3 tests/tools/pmc_summary.py gpurun_out/r02_FETCH_SIZE gpurun_out/r02_WRITE_SIZE gpurun_out/r02
python3 - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/r02_kernel_stats.csv")):
    if float(r["Percentage"]) > 0.4: print(r["Name"][:48].ljust(48), r["Calls"].rjust(5), "%9.3f ms avg" % (float(r["AverageNs"])/1e6), "%9.2f ms tot" % (float(r["TotalDurationNs"])/1e6), r["Percentage"]+"%")
PY
rm -rf gpurun_out/r02_trace gpurun_out/r02_FETCH_SIZE gpurun_out/r02_WRITE_SIZE
