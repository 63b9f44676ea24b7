#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes -> per-kernel HBM bytes per launch (JSON + CSV)."""
import csv, glob, json, os, re, sys
from collections import defaultdict

def short(name):
    m = re.match(r"(?:void )?([A-Za-z_0-9]+(?:<[^>(]*>)?)", name)
    return m.group(1) if m else name

def load(d, counter):
    tot, calls, ns = defaultdict(float), defaultdict(int), defaultdict(float)
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"]) * 1024.0
        calls[k] += 1
        ns[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return tot, calls, ns

fd, wd, out = sys.argv[1:4]
ft, fc, fns = load(fd, "FETCH_SIZE")
wt, wc, _ = load(wd, "WRITE_SIZE")
res, rows = {}, []
for k in sorted(ft, key=lambda k: -fns[k]):
    n = fc[k]
    res[k] = {"launches": n, "fetch_bytes_per_launch_x2_corrected": 2.0 * ft[k] / n, "write_bytes_per_launch": wt.get(k, 0.0) / max(1, wc.get(k, n))}
    rows.append((k, n, fns[k] / 1e6, ft[k] / 1e9, 2 * ft[k] / 1e9, wt.get(k, 0.0) / 1e9))
# the device code these bytes were measured on (bench.py reports them only while the tree it runs from still has this value)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
try:
    import yak_amd
    res["_measured_on"] = {"kernels_sha16": yak_amd.kernels_sha16()}
except Exception as e:
    res["_measured_on"] = {"kernels_sha16": None, "error": str(e)}
json.dump(res, open(out + "_pmc_traffic.json", "w"), indent=1)
with open(out + "_pmc_hbm_bytes.csv", "w") as f:
    f.write("kernel,calls,total_ms,FETCH_SIZE_raw_GB,FETCH_SIZE_x2_GB,WRITE_SIZE_GB\n")
    for r in rows:
        f.write("%s,%d,%.2f,%.2f,%.2f,%.2f\n" % r)
for r in rows[:14]:
    print("%-28s calls %3d  %8.2f ms  fetch x2 %7.2f GB  write %7.2f GB" % (r[0], r[1], r[2], r[4], r[5]))
