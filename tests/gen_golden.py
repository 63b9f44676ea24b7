#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ FROM THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference): `make -C oracle ref` compiles the reference
sources where they lie into oracle/_ref/ (binary `yak`, plus a shim exposing its static-inline hash
functions); this script then
  * records known-answer values of yak_hash64 / yak_hash64_64 / yak_hash_long / yak_hash64_inv /
    __kh_h2b / yak_bf_insert            -> tests/golden/kat.json
  * runs `yak count` on small deterministic inputs (tools/yaksynth seeds, or the literal files in
    tests/golden/inputs/) and stores the .yak bytes (small cases) or their md5 (larger cases)
                                        -> tests/golden/*.yak, tests/golden/manifest.json
Only data is stored: inputs/outputs, never reference source.
"""
import ctypes as C
import hashlib
import json
import os
import random
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
REF = os.path.join(ROOT, "oracle", "_ref")
SYNTH = os.path.join(ROOT, "tools", "yaksynth")

# name -> (synth args or literal input file(s), yak count args, store bytes?)
CASES = {
    "nb_k31":      (dict(n=600, l=150, g=5000, s=1),              ["-k31"], True),
    "nb_k21":      (dict(n=600, l=150, g=5000, s=2),              ["-k21"], True),
    "nb_k15_fa":   (dict(n=20, l=4000, g=30000, s=3, a=1, N=0.002), ["-k15"], True),
    "nb_k27_p12":  (dict(n=600, l=150, g=5000, s=4),              ["-k27", "-p12"], True),
    "b20_k31":     (dict(n=600, l=150, g=2500, s=5),              ["-k31", "-b20"], True),
    "b24_k31":     (dict(n=600, l=150, g=2500, s=5),              ["-k31", "-b24"], True),
    "b19_k31":     (dict(n=600, l=150, g=2500, s=5),              ["-k31", "-b19"], True),       # 1 block per sub-table: heavy FP load
    "b19_H7":      (dict(n=600, l=150, g=2500, s=6),              ["-k31", "-b19", "-H7"], True),
    "b22_H40":     (dict(n=600, l=150, g=2500, s=6),              ["-k31", "-b22", "-H40"], True),
    "b15_nobf":    (dict(n=600, l=150, g=2500, s=6),              ["-k31", "-b15"], True),   # pre < b < pre+9: no filter, still two passes
    "nb_k32":      (dict(n=600, l=150, g=5000, s=8),              ["-k32"], True),            # long k-mer path (count.c:45-60)
    "nb_k41":      (dict(n=600, l=150, g=5000, s=8),              ["-k41"], True),
    "b24_k63_fa":  (dict(n=20, l=4000, g=30000, s=9, a=1, N=0.002), ["-k63", "-b24"], True),
    "edge_fx":     ("inputs/edge.fx",                              ["-k5"], True),
    "one_read":    ("inputs/one3000.fa",                           ["-k31"], True),
    "one_read_x2": ("inputs/one3000x2.fa",                         ["-k31"], True),           # grow-on-existing-key
    "polyA":       ("inputs/polya.fa",                             ["-k31"], True),           # saturation at 1023
    "mid_nb":      (dict(n=12000, l=150, g=60000, s=7),           ["-k31"], False),
    "mid_b24":     (dict(n=12000, l=150, g=60000, s=7),           ["-k31", "-b24"], False),
    "mid_b20":     (dict(n=12000, l=150, g=60000, s=7),           ["-k31", "-b20"], False),      # saturated filter
    "mid_b30":     (dict(n=12000, l=150, g=60000, s=7),           ["-k31", "-b30"], False),
    "cfg1":        (dict(n=100000, l=150, g=500000, s=42),        ["-k31", "-K64m", "-t1"], False),
}


# `yak qv` (lookup-only path): table = a stored .yak above, query = synthetic contigs of the same genome
QV_CASES = {
    "qv_b24_k31":       ("b24_k31", dict(n=30, l=1000, g=2500, s=5, a=1, e=0.01, N=0.001), ["-p", "-E"]),
    "qv_b24_k31_filt":  ("b24_k31", dict(n=30, l=1000, g=2500, s=5, a=1, e=0.05, N=0.001), ["-p", "-f", "0.25", "-l", "1000"]),
    "qv_nb_k21":        ("nb_k21",  dict(n=12, l=3000, g=5000, s=2, a=1, e=0.002),          ["-p", "-K", "5k"]),
}


def synth_args(d):
    a = ["-n", str(d["n"]), "-l", str(d["l"]), "-g", str(d["g"]), "-s", str(d["s"])]
    if d.get("a"):
        a.append("-a")
    if "N" in d:
        a += ["-N", str(d["N"])]
    if "e" in d:
        a += ["-e", str(d["e"])]
    return a


def write_inputs():
    os.makedirs(os.path.join(G, "inputs"), exist_ok=True)
    with open(os.path.join(G, "inputs", "edge.fx"), "wb") as f:
        f.write(b">a\nACGTNNACGTACGTTTGACCA\r\n>b desc\nAC\n\nGT\n@c\nACGTAGGCATTACGGACTA\n+\nIIIIIIIIIIIIIIIIIII\n"
                b">short\nACG\n@d\nACGTAGGCATTACGGACTAGG\n+\nIIII\n")
    rnd = random.Random(3000)
    one = "".join(rnd.choice("ACGT") for _ in range(3000))
    with open(os.path.join(G, "inputs", "one3000.fa"), "w") as f:
        f.write(">r\n" + one + "\n")
    with open(os.path.join(G, "inputs", "one3000x2.fa"), "w") as f:
        f.write(">r\n" + one + "\n>r2\n" + one + "\n")
    with open(os.path.join(G, "inputs", "polya.fa"), "w") as f:
        f.write(">a\n" + "A" * 1500 + "\n>t\n" + "T" * 700 + "\n>m\n" + "ACGTTGCA" * 40 + "\n")


def kat():
    S = C.CDLL(os.path.join(REF, "libyakshim.so"))
    R = C.CDLL(os.path.join(REF, "libyakref.so"))
    u64 = C.c_uint64
    S.shim_hash64.restype = u64; S.shim_hash64.argtypes = [u64, u64]
    S.shim_hash64_64.restype = u64; S.shim_hash64_64.argtypes = [u64]
    S.shim_hash64_inv.restype = u64; S.shim_hash64_inv.argtypes = [u64, u64]
    S.shim_hash_long.restype = u64; S.shim_hash_long.argtypes = [C.POINTER(u64)]
    S.shim_h2b.restype = C.c_uint32; S.shim_h2b.argtypes = [C.c_uint32, C.c_uint32]
    R.yak_bf_init.restype = C.c_void_p; R.yak_bf_init.argtypes = [C.c_int, C.c_int]
    R.yak_bf_insert.restype = C.c_int; R.yak_bf_insert.argtypes = [C.c_void_p, u64]
    rnd = random.Random(20260927)
    out = {"hash64": [], "hash64_64": [], "hash_long": [], "h2b": [], "bf": []}
    for k in (31, 21, 15, 5, 27):
        m = (1 << 2 * k) - 1
        vals = [0, 1, 2, m, 0x2aaaaaaaaaaaaaaa & m, 0x0123456789abcdef & m] + [rnd.getrandbits(2 * k) for _ in range(20)]
        for v in vals:
            h = S.shim_hash64(v, m)
            assert S.shim_hash64_inv(h, m) == v
            out["hash64"].append([v, m, h])
    for v in [0, 1, 2**64 - 1] + [rnd.getrandbits(64) for _ in range(20)]:
        out["hash64_64"].append([v, S.shim_hash64_64(v)])
    for _ in range(20):
        x = [rnd.getrandbits(40) for _ in range(4)]
        out["hash_long"].append([x, S.shim_hash_long((u64 * 4)(*x))])
    for h, b in [(1, 10), (0x12345678, 12), (0xffffffff, 2)] + [(rnd.getrandbits(32), rnd.randint(2, 24)) for _ in range(20)]:
        out["h2b"].append([h, b, S.shim_h2b(h, b)])
    for (ns, nh) in [(27, 4), (9, 4), (12, 7), (10, 40)]:
        bf = R.yak_bf_init(ns, nh)
        seq = [0x123456789abc, 0x123456789abc] + [rnd.getrandbits(50) for _ in range(300)]
        out["bf"].append({"n_shift": ns, "n_hash": nh, "hashes": seq, "ret": [R.yak_bf_insert(bf, v) for v in seq]})
    out["bf_init_null"] = [[8, 4, R.yak_bf_init(8, 4) is None], [56, 4, R.yak_bf_init(56, 4) is None]]
    json.dump(out, open(os.path.join(G, "kat.json"), "w"))


def main():
    if not os.path.exists(os.path.join(REF, "yak")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tools")])
    write_inputs()
    kat()
    man = {}
    tmp = os.path.join("/tmp", "ykgold")
    os.makedirs(tmp, exist_ok=True)
    for name, (inp, args, store) in CASES.items():
        if isinstance(inp, dict):
            fn = os.path.join(tmp, name + (".fa" if inp.get("a") else ".fq"))
            subprocess.check_call([SYNTH] + synth_args(inp) + ["-o", fn])
            desc = {"synth": inp}
        else:
            fn = os.path.join(G, inp)
            desc = {"file": inp}
        out = os.path.join(tmp, name + ".yak")
        subprocess.run([os.path.join(REF, "yak"), "count"] + args + ["-o", out, fn], check=True, stderr=subprocess.DEVNULL)
        data = open(out, "rb").read()
        desc.update(args=args, md5=hashlib.md5(data).hexdigest(), size=len(data), stored=store)
        if store:
            open(os.path.join(G, name + ".yak"), "wb").write(data)
        man[name] = desc
        print(name, len(data), desc["md5"])
    json.dump(man, open(os.path.join(G, "manifest.json"), "w"), indent=1)
    sys.path.insert(0, ROOT)
    from oracle.pyoracle import parse_qv_output
    qv = {}
    for name, (table, inp, args) in QV_CASES.items():
        fn = os.path.join(tmp, name + ".fa")
        subprocess.check_call([SYNTH] + synth_args(inp) + ["-o", fn])
        txt = subprocess.run([os.path.join(REF, "yak"), "qv"] + args + [os.path.join(G, table + ".yak"), fn], check=True,
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
        ct, sq, ek = parse_qv_output(txt)
        qv[name] = {"table": table, "synth": inp, "args": args, "cnt": {str(c): v[1] for c, v in ct.items() if v[1]},
                    "sq": sq, "n_ek": len(ek), "ek_md5": hashlib.md5("\n".join(ek).encode()).hexdigest()}
        print(name, sum(v[1] for v in ct.values()), len(sq), len(ek))
    json.dump(qv, open(os.path.join(G, "qv.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
