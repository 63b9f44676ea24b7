#!/bin/bash
# the LDS conflict counters of uniformly random accesses (tests/tools/mb/mb_lds.hip): the floor a hash table in LDS cannot get under, whatever its layout
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06_mb_lds}; mkdir -p $O
hipcc --offload-arch=gfx950 -O2 tests/tools/mb/mb_lds.hip -o /tmp/mb_lds 2> /dev/null
/tmp/mb_lds > $O/mb_lds.txt 2>&1
timeout 600 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN --kernel-trace --output-format csv -d $O/raw -o pmc -- /tmp/mb_lds > $O/run.log 2>&1
python3 - $O <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
O = sys.argv[1]
fs = glob.glob(os.path.join(O, "raw", "**", "*counter_collection.csv"), recursive=True)
if not fs:
    print("no counter file"); print(open(os.path.join(O, "run.log")).read()[-1500:]); sys.exit(0)
names = {0: "ds_read_b32", 1: "ds_read_b64", 2: "ds_add_rtn_u32", 3: "ds_add_u32", 4: "ds_min_rtn_u32", 5: "ds_min_u32", 6: "ds_cmpst_rtn_b64", 7: "ds_min_rtn_u64", 8: "ds_write_b32", 9: "add_rtn + min_rtn"}
v = defaultdict(lambda: defaultdict(float)); order = []
for r in csv.DictReader(open(fs[0])):
    m = re.search(r"k<(\d+), (\d+)>", r["Kernel_Name"])
    if not m: continue
    k = (int(m.group(1)), int(m.group(2)))
    if k not in order: order.append(k)
    v[k][r["Counter_Name"]] += float(r["Counter_Value"])      # (the short warm-up launch of each variant is in the sums: 64 of 4160 iterations)
with open(os.path.join(O, "mb_lds_counters.txt"), "w") as out:
    for k in order:
        c = v[k]
        ia = c["SQ_LDS_IDX_ACTIVE"] or 1
        line = "%-20s lanes per address %2d: lds insts %.3g  idx_active %.3g  bank_conflict %.3g (%.3f of idx_active)  addr_conflict %.3g (%.3f)  idx_active per lds inst %.2f" % (
            names.get(k[0], str(k[0])), 1 << k[1], c["SQ_INSTS_LDS"], c["SQ_LDS_IDX_ACTIVE"], c["SQ_LDS_BANK_CONFLICT"], c["SQ_LDS_BANK_CONFLICT"] / ia, c["SQ_LDS_ADDR_CONFLICT"], c["SQ_LDS_ADDR_CONFLICT"] / ia, ia / (c["SQ_INSTS_LDS"] or 1))
        print(line); out.write(line + "\n")
PY
rm -rf $O/raw
cat $O/mb_lds.txt
