#!/bin/bash
# how busy is the LDS in each kernel of one default bench step: SQ LDS counters (cycles are summed over the CUs)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-lds}; mkdir -p $O
timeout 900 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_ATOMIC_RETURN --kernel-trace --output-format csv -d $O/raw -o pmc -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-qv --no-pcie --no-packed --no-nofilter > $O/run.log 2>&1
python3 - $O <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
O = sys.argv[1]
fs = glob.glob(os.path.join(O, "raw", "**", "*counter_collection.csv"), recursive=True)
if not fs:
    print("no counter file"); print(open(os.path.join(O, "run.log")).read()[-1500:]); sys.exit(0)
v = defaultdict(lambda: defaultdict(float)); ns = defaultdict(float); seen = set()
for r in csv.DictReader(open(fs[0])):
    m = re.match(r"(?:void )?([A-Za-z_0-9]+(?:<[^>(]*>)?)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"]
    v[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"],)
    if key not in seen:
        seen.add(key); ns[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
with open(os.path.join(O, "lds_summary.txt"), "w") as out:
    for k in sorted(ns, key=lambda k: -ns[k])[:14]:
        c = v[k]
        line = "%-26s %8.2f ms  busy_cu_cycles %.3g  wave_cycles %.3g  lds insts %.3g  idx_active %.3g  bank_conflict %.3g  addr_conflict %.3g  wait_inst_lds %.3g  atomic_return %.3g" % (
            k, ns[k] / 1e6, c["SQ_BUSY_CU_CYCLES"], c["SQ_WAVE_CYCLES"], c["SQ_INSTS_LDS"], c["SQ_LDS_IDX_ACTIVE"], c["SQ_LDS_BANK_CONFLICT"], c["SQ_LDS_ADDR_CONFLICT"], c["SQ_WAIT_INST_LDS"], c["SQ_LDS_ATOMIC_RETURN"])
        print(line); out.write(line + "\n")
PY
rm -rf $O/raw
