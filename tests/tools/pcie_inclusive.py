"""PCIe-inclusive rate of the benchmark workload: the same protocol as bench.py, but both passes take the
base image from HOST memory (yakamd_feed_bases_host: pageable source -> staged copy -> device), as
yak_count() does after parsing.  Never `value`; quoted in DESIGN.md section 5."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch          # first: two HIP runtimes in the other order leave torch without devices
import bench
import yak_amd

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
h = bench.make_reads(n_reads, 5 * n_reads, 42, 0, torch, 32)
nb = h.numel()
L = yak_amd.lib()
for pinned in (True, False):
    src = h if pinned else h.clone().pin_memory() if False else torch.empty(nb, dtype=torch.uint8).copy_(h)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        t = yak_amd.Table(31, 10, 4, 37)
        for create_new in (1, 0):
            if L.yakamd_pass_begin(t.h, create_new) != 0 or L.yakamd_feed_bases_host(t.h, src.data_ptr(), nb, 0) != 0:
                raise SystemExit(yak_amd._err())
            n_ins = L.yakamd_pass_end(t.h)
            t.h.contents.tot += n_ins
            if create_new:
                t.destroy_bf(); t.clear()
        t.shrink(2, 1023)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tot = t.tot
        t.close()
        best = min(best, dt)
    print(f"{'pinned' if pinned else 'pageable'} host image: {best * 1e3:.1f} ms per protocol run, {tot / best / 1e6:.1f} M distinct k-mers/s (tot {tot})")
