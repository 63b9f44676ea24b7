#!/usr/bin/env python3
"""Full-size golden for BASELINE configs[1] FROM THE REFERENCE ITSELF (build container only, ~3 min,
3 GB of scratch): the bench workload written as FASTQ by tools/yaksynth, counted by the reference
binary compiled from /root/reference (oracle/_ref/yak), md5 + size of its .yak stored in
tests/golden/cfg2_full.json.  bench.py compares the device result of the same workload with it."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "yak")
SYN = os.path.join(ROOT, "tools", "yaksynth")
N, L, G, SEED, K, BF = 10_000_000, 150, 50_000_000, 42, 31, 37


def main():
    tmp = next((a for a in sys.argv[1:] if not a.startswith("--")), "/tmp/cfg2")
    os.makedirs(tmp, exist_ok=True)
    fq, out = os.path.join(tmp, "r.fq"), os.path.join(tmp, "ref.yak")
    if not os.path.exists(fq):
        subprocess.check_call([SYN, "-n", str(N), "-l", str(L), "-g", str(G), "-s", str(SEED), "-t", "8", "-o", fq])
    res = {}
    if "--with-30m" in sys.argv:                              # 9.4 GB of FASTQ, ~10 min: the sliced-pass case
        fq3, out3 = os.path.join(tmp, "r30.fq"), os.path.join(tmp, "ref30.yak")
        if not os.path.exists(fq3):
            subprocess.check_call([SYN, "-n", str(3 * N), "-l", str(L), "-g", str(3 * G), "-s", str(SEED), "-t", "8", "-o", fq3])
        subprocess.run([REF, "count", f"-k{K}", f"-b{BF}", "-t8", "-o", out3, fq3], check=True, stderr=subprocess.DEVNULL)
        h = hashlib.md5()
        with open(out3, "rb") as f:
            for blk in iter(lambda: f.read(1 << 24), b""):
                h.update(blk)
        res["b37_30M_sliced"] = {"workload": f"yak count -k{K} -b{BF} on yaksynth -n {3 * N} -l {L} -g {3 * G} -s {SEED}", "reads": 3 * N, "read_len": L,
                                 "genome": 3 * G, "seed": SEED, "k": K, "bf_shift": BF, "md5": h.hexdigest(), "size": os.path.getsize(out3),
                                 "produced_by": "oracle/_ref/yak (the reference, compiled from /root/reference)"}
    for name, bf in (("b37", BF), ("no_filter", 0)):
        subprocess.run([REF, "count", f"-k{K}"] + ([f"-b{bf}"] if bf else []) + ["-t8", "-o", out, fq], check=True, stderr=subprocess.DEVNULL)
        h = hashlib.md5()
        with open(out, "rb") as f:
            for blk in iter(lambda: f.read(1 << 24), b""):
                h.update(blk)
        res[name] = {"workload": f"yak count -k{K}" + (f" -b{bf}" if bf else "") + f" on yaksynth -n {N} -l {L} -g {G} -s {SEED} (e=0.5%, N=0.05%)",
                     "reads": N, "read_len": L, "genome": G, "seed": SEED, "k": K, "bf_shift": bf,
                     "md5": h.hexdigest(), "size": os.path.getsize(out), "produced_by": "oracle/_ref/yak (the reference, compiled from /root/reference)"}
        print(res[name])
    json.dump(res, open(os.path.join(ROOT, "tests", "golden", "cfg2_full.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
