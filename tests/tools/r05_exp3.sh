#!/bin/bash
# round 5, third GPU call: hash once (the histogram sweep leaves the hash stream for the scatter sweep), pass 2's small table for fuller sub-buckets
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e3; mkdir -p $O
Q="--no-cpu-baseline --no-pcie --no-qv --no-packed --no-nofilter --steps 5 --warmup 2"
line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    ks = {k["kernel"].split(" (")[0][:34]: round(k["ms"], 2) for k in d["roofline"].get("all_kernels", [])} if "all_kernels" in d.get("roofline", {}) else {}
    print(sys.argv[1].ljust(28), "ms", round(d["ms_per_step"], 2), "verify", (d.get("verify") or {}).get("equals_reference"), ks, d.get("phase_wall_ms_last_step"))
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[2].replace(".json", ".err")).read()[-600:])
PY
}
run() { local name=$1; shift; YAKAMD_VERBOSE=1 timeout 300 python bench.py $Q "$@" > $O/$name.json 2> $O/$name.err; line $name $O/$name.json; }
run default
run no_hash_once --knob YAKAMD_HASH_ONCE=0
run cnt2_small48 --knob YAKAMD_CNT2_SMALL=48
Q2="--no-cpu-baseline --no-pcie --no-qv --no-packed --no-nofilter --steps 3 --warmup 1"
run2() { local name=$1; shift; YAKAMD_VERBOSE=1 timeout 400 python bench.py $Q2 "$@" > $O/$name.json 2> $O/$name.err; line $name $O/$name.json; }
run2 nofilter --config nofilter
run2 cfg4_1gb --config cfg4 --contigs 10 --contig-len 100000000
run2 noretain --no-retain
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-pcie --no-packed --no-nofilter --no-qv > $O/bench_profiled.json 2>/dev/null
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats_default.csv; rm -rf $O/trace
python3 - $O/kernel_stats_default.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.8: print(r["Name"][:56].ljust(56), r["Calls"].rjust(5), "%9.3f ms avg" % (float(r["AverageNs"])/1e6), "%9.2f ms tot" % (float(r["TotalDurationNs"])/1e6), r["Percentage"]+"%")
PY
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_multi_c.py tests/test_gpu_multirank.py -x -q -m gpu > $O/pytest_parity.txt 2>&1; tail -3 $O/pytest_parity.txt
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "1m_reads" > $O/pytest_cfg3.txt 2>&1; tail -3 $O/pytest_cfg3.txt
