"""nature of the difference between a right and a wrong full step (diagnostic, GPU box)"""
import hashlib, os, sys, struct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch, yak_amd, bench
reads, bf = 10_000_000, 37
h = bench.make_reads(reads, 5 * reads, 42, 0, torch, 32)
d = h.to("cuda:0"); nb = d.numel()
def run(env, stop=3):
    for k, v in env.items(): os.environ[k] = v
    t = yak_amd.Table(31, 10, 4, bf)
    t.count_pass(1, [(d.data_ptr(), nb, 0)])
    if stop > 0: t.destroy_bf(); t.clear()
    if stop > 1: t.count_pass(0, [(d.data_ptr(), nb, 0)])
    if stop > 2: t.shrink(2, 1023)
    b = t.dump_bytes(); t.close()
    for k in env: del os.environ[k]
    return b
def split(b):
    off, out = 16, []
    for p in range(1024):
        cap, n = struct.unpack_from("<II", b, off)
        out.append((cap, np.frombuffer(b, dtype=np.uint64, count=n, offset=off + 8)))
        off += 8 + 8 * n
    return out
for stop in (3, 2, 1):
    good = run({"YAKAMD_REC8": "0"}, stop)
    gm = hashlib.md5(good).hexdigest()
    for it in range(4):
        bad = run({}, stop)
        bm = hashlib.md5(bad).hexdigest()
        print("stop", stop, "try", it, gm[:8], bm[:8], flush=True)
        if bm == gm: continue
        A, B = split(good), split(bad)
        nbad = 0
        for p in range(1024):
            (ca, ka), (cb, kb) = A[p], B[p]
            if ca != cb or len(ka) != len(kb) or not np.array_equal(ka, kb):
                nbad += 1
                if nbad <= 5:
                    same_set = len(ka) == len(kb) and np.array_equal(np.sort(ka >> 10), np.sort(kb >> 10))
                    same_order = len(ka) == len(kb) and np.array_equal(ka >> 10, kb >> 10)
                    nd = int((ka != kb).sum()) if len(ka) == len(kb) else -1
                    print("  sub-table", p, "cap", ca, cb, "n", len(ka), len(kb), "same key set", same_set, "same order", same_order, "entries differing", nd)
                    if same_order:
                        i = np.nonzero(ka != kb)[0][:8]
                        print("    idx", i, "good counts", (ka[i] & 1023), "bad counts", (kb[i] & 1023))
        print("  sub-tables differing:", nbad)
        break
