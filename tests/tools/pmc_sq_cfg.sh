#!/bin/bash
# SQ counters per kernel for any bench configuration: pmc_sq_cfg.sh <outdir> <bench args...>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-sq}; mkdir -p $O; shift
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/raw -o pmc -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-qv --no-pcie --no-packed --no-nofilter "$@" > $O/run.log 2>&1
timeout 900 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/raw2 -o pmc -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-qv --no-pcie --no-packed --no-nofilter "$@" > $O/run2.log 2>&1
python3 - $O <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
O = sys.argv[1]
v = defaultdict(lambda: defaultdict(float)); ns = defaultdict(float); seen = set(); calls = defaultdict(int)
for raw in ("raw", "raw2"):
    fs = glob.glob(os.path.join(O, raw, "**", "*counter_collection.csv"), recursive=True)
    if not fs: continue
    for r in csv.DictReader(open(fs[0])):
        m = re.match(r"(?:void )?([A-Za-z_0-9]+(?:<[^>(]*>)?)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"]
        v[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (raw, r["Dispatch_Id"])
        if raw == "raw" and key not in seen:
            seen.add(key); ns[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); calls[k] += 1
with open(os.path.join(O, "sq_summary.txt"), "w") as out:
    for k in sorted(ns, key=lambda k: -ns[k])[:16]:
        c = v[k]; wc = c["SQ_WAVE_CYCLES"] or 1
        line = "%-34s %3d x %8.2f ms  wait_any %4.1f%%  wait_inst %4.1f%%  active %4.1f%% (valu %4.1f%% lds %4.1f%%)  insts valu %.3g salu %.3g lds %.3g vmem_rd %.3g vmem_wr %.3g smem %.3g  busy_cycles %.3g wave_cycles %.3g" % (
            k, calls[k], ns[k] / 1e6, 100 * c["SQ_WAIT_ANY"] / wc, 100 * c["SQ_WAIT_INST_ANY"] / wc, 100 * c["SQ_ACTIVE_INST_ANY"] / wc, 100 * c["SQ_ACTIVE_INST_VALU"] / wc, 100 * c["SQ_ACTIVE_INST_LDS"] / wc,
            c["SQ_INSTS_VALU"], c["SQ_INSTS_SALU"], c["SQ_INSTS_LDS"], c["SQ_INSTS_VMEM_RD"], c["SQ_INSTS_VMEM_WR"], c["SQ_INSTS_SMEM"], c["SQ_BUSY_CYCLES"], wc)
        print(line); out.write(line + "\n")
PY
rm -rf $O/raw $O/raw2
