#!/bin/bash
# per-dispatch durations of the selection / sort / layout kernels of the last bench step, in launch order; arguments go to bench.py (e.g. --config nofilter)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${R2OUT:-r2trace}; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/raw -o t -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify --no-qv --no-pcie --no-packed --no-nofilter "$@" > $O/bench.json 2> $O/bench.err
python3 - $O <<'PY'
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "raw", "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
out = []
prev_end = None
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if n.startswith(("k_r2_", "k_replay", "k_seg_sort", "k_shrink", "k_lc_", "k_cnt2", "k_nsel", "k_ts_", "k_part2<0, true", "k_part2<1, true", "k_part2_wc<true", "k_part2_scan", "k_kt_", "__amd_rocclr_fill")):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end else 0
        out.append("%-18s %8.1f us  gap %6.1f us  grid %s" % (n[:18], (e - s) / 1e3, gap, r.get("Grid_Size_X", "") + "x" + r.get("Grid_Size_Y", "")))
    prev_end = int(r["End_Timestamp"])
last = max([i for i, l in enumerate(out) if l.startswith("k_lc_sum")] or [0])       # the last pass-1 of the run
out = out[last:]
open(os.path.join(sys.argv[1], "r2_dispatches.txt"), "w").write("\n".join(out) + "\n")
from collections import defaultdict
agg = defaultdict(lambda: [0, 0.0])
for l in out:
    agg[l[:18].strip()][0] += 1; agg[l[:18].strip()][1] += float(l[18:].split()[0])
print("\n".join("%-20s x%-4d %9.1f us" % (k, v[0], v[1]) for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])))
print("\n".join(out[:120]))
PY
rm -rf $O/raw
