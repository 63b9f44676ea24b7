"""how does the per-rank cost grow when a rank owns fewer, larger sub-tables (multi-GPU weak scaling)?"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
import yak_amd, bench
L = yak_amd.lib()
dev = torch.device("cuda", 0)
for n_reads, frac in ((10_000_000, 1), (20_000_000, 2), (40_000_000, 4))[int(os.environ.get("CASE0", "0")):int(os.environ.get("NCASE", "2"))]:
    h = bench.make_reads(n_reads, 5 * n_reads, 42, 0, torch, 32)
    d = h.to(dev); torch.cuda.synchronize(); nb = d.numel()
    for rep in range(2):
        t = yak_amd.Table(31, 10, 4, 37)
        L.yakamd_set_shard(t.h, 0, 1024 // frac)
        t0 = time.perf_counter()
        t.count_pass(1, [(d.data_ptr(), nb, 0)])
        torch.cuda.synchronize(); t1 = time.perf_counter()
        s1 = t.stats()
        t.destroy_bf(); t.clear()
        t.count_pass(0, [(d.data_ptr(), nb, 0)])
        torch.cuda.synchronize(); t2 = time.perf_counter()
        s2 = t.stats()
        t.shrink(2, 1023); torch.cuda.synchronize(); t3 = time.perf_counter()
        s3 = t.stats()
        tot = t.tot; t.close()
    print(f"reads={n_reads} shard=1/{frac}: pass1 {1e3*(t1-t0):.1f} (xpart+part2 {s1['ms_extract']:.1f} lds {s1['ms_insert']:.1f} sort {s1['ms_sort']:.1f} replay {s1['ms_replay']:.1f}) pass2 {1e3*(t2-t1):.1f} (count {s2['ms_insert']:.1f}) shrink {1e3*(t3-t2):.1f} tot={tot}", flush=True)
    del d, h
