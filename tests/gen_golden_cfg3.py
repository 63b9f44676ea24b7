#!/usr/bin/env python3
"""Golden for BASELINE configs[2] (`yak count -k31` on 600 M x 150 bp synthetic reads, prefix-sharded over 8 GPUs) -- build container only,
~10 min on 8 cores, 10 GB of scratch.  The stream (reads 0 .. 599 999 999 of tools/yaksynth.c: G = 3.0 Gb, seed 42, e = 0.1 %, N = 0.05 %, each read
followed by '\\n') is 90 GB of bases and a full table ~41 GB of keys, so what is pinned is what ONE RANK of the 8-GPU job owns: the bytes
({capacity, size, keys in slot order} per sub-table, htab.c:385-389) of sub-tables [0, 128) = rank 0 and [640, 768) = rank 5, counted by the oracle
over the whole stream (oracle/yko_synth.c: generated chunk by chunk, never stored; the oracle's own yko_extract / yko_ch_insert_list, prefixes in
parallel as the reference's kt_for does).  A sub-table is a function of its own put-calls in stream order (count.c:129-143), so these are exactly
the bytes the full job's .yak holds at those sub-tables.  The procedure is checked here first at 1 M reads against `yko count -R` on the same
reads written as FASTQ.  -> tests/golden/cfg3_full.json; tests/test_gpu_fullsize.py::test_cfg3_rank_share_equals_oracle compares a rank's
share measured on one GPU (bench.py --config cfg3shard --rank R) with it, and bench.py --gpus 8 the ranks of the real job."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
YKS = os.path.join(ROOT, "oracle", "yko_synth")
YKO = os.path.join(ROOT, "oracle", "yko")
SYN = os.path.join(ROOT, "tools", "yaksynth")
READS, L, G, SEED, ERR, NR, K = 600_000_000, 150, 3_000_000_000, 42, 0.001, 0.0005, 31
RANGES = [(0, 128), (640, 768)]


def md5_file(fn, skip=0):
    h, n = hashlib.md5(), 0
    with open(fn, "rb") as f:
        f.seek(skip)
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk); n += len(blk)
    return h.hexdigest(), n


def run(tmp, reads, genome, tag):
    cmd = [YKS, "-n", str(reads), "-l", str(L), "-g", str(genome), "-s", str(SEED), "-e", str(ERR), "-N", str(NR), "-k", str(K), "-t", str(os.cpu_count() or 8),
           "-c", "4000000", "-o", os.path.join(tmp, tag)]
    for lo, hi in RANGES:
        cmd += ["-R", f"{lo}:{hi}"]
    out = subprocess.run(cmd, check=True, stdout=subprocess.PIPE).stdout.decode()
    res = {}
    for line in out.splitlines():
        f = line.split()
        if f[0] == "RANGE":
            fn = os.path.join(tmp, f"{tag}.{f[1]}-{f[2]}.part")
            md5, size = md5_file(fn)
            res[f"{f[1]}:{f[2]}"] = {"md5": md5, "size": size, "distinct": int(f[3]), "instances": int(f[4])}
            os.remove(fn)
    return res


def main():
    """gen_golden_cfg3.py [tmpdir] [--ranges lo:hi,lo:hi]: two ranges per run fit the 62 GB of the build container (a rank's 128 tables of 8 Mi slots are
    8.6 GB); the ranges of a run are merged into what tests/golden/cfg3_full.json already holds.  Round 6 added the six ranks that rounds 5 left out."""
    global RANGES
    tmp = next((a for a in sys.argv[1:] if not a.startswith("--") and ":" not in a), "/tmp/cfg3")
    if "--ranges" in sys.argv:
        RANGES = [tuple(int(x) for x in r.split(":")) for r in sys.argv[sys.argv.index("--ranges") + 1].split(",")]
    os.makedirs(tmp, exist_ok=True)
    # the procedure against `yko count -R` (the oracle's prefix-range mode, which reproduces the reference's md5 at 2 Gb: cfg45_full.json) on 1 M reads
    small = run(tmp, 1_000_000, 5_000_000, "small")
    fq = os.path.join(tmp, "small.fq")
    subprocess.check_call([SYN, "-n", "1000000", "-l", str(L), "-g", "5000000", "-s", str(SEED), "-e", str(ERR), "-N", str(NR), "-t", "8", "-o", fq])
    for lo, hi in RANGES:
        part = os.path.join(tmp, "small.ref.part")
        subprocess.run([YKO, "count", f"-k{K}", "-R", f"{lo}:{hi}", "-o", part, fq], check=True, stderr=subprocess.DEVNULL)
        md5, size = md5_file(part, 16 if lo == 0 else 0)             # yko writes the .yak header in front of range 0
        if (md5, size) != (small[f"{lo}:{hi}"]["md5"], small[f"{lo}:{hi}"]["size"]):
            raise SystemExit(f"yko_synth does not reproduce yko count -R {lo}:{hi}")
        os.remove(part)
    os.remove(fq)
    res = {"workload": f"yak count -k{K} (no filter) on reads 0 .. {READS - 1} of yaksynth (l = {L}, G = {G}, seed {SEED}, e = {ERR}, N = {NR}), in read order",
           "reads": READS, "read_len": L, "genome": G, "seed": SEED, "err": ERR, "nrate": NR, "k": K, "pre": 10,
           "what": "md5 / size of the bytes {u32 capacity, u32 size, keys in slot order} of the sub-tables [lo, hi): a rank's share of the .yak file, no header",
           "produced_by": "oracle/yko_synth (the oracle over the generated stream; checked against `yko count -R` at 1 M reads by this script)",
           "procedure_check_1M_reads": small,
           "ranges": run(tmp, READS, G, "cfg3")}
    out_fn = os.path.join(ROOT, "tests", "golden", "cfg3_full.json")
    try:
        old = json.load(open(out_fn))
        if all(old.get(k_) == res[k_] for k_ in ("reads", "read_len", "genome", "seed", "err", "nrate", "k", "pre")):
            for name, sec in (("ranges", old.get("ranges", {})), ("procedure_check_1M_reads", old.get("procedure_check_1M_reads", {}))):
                merged = dict(sec); merged.update(res[name]); res[name] = dict(sorted(merged.items(), key=lambda kv: int(kv[0].split(":")[0])))
    except (OSError, ValueError):
        pass
    json.dump(res, open(out_fn, "w"), indent=1)
    print(json.dumps(res["ranges"], indent=1))


if __name__ == "__main__":
    main()
