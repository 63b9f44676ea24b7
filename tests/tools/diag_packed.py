import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, yak_amd, bench
L = yak_amd.lib()
reads = 10_000_000
h = bench.make_reads(reads, 5 * reads, 42, 0, torch, 32)
d = h.to("cuda:0"); n = d.numel(); nw = (n + 31) // 32
dc = torch.empty(2 * nw + 4, dtype=torch.int32, device="cuda:0"); dv = torch.empty(nw + 4, dtype=torch.int32, device="cuda:0")
assert L.yakamd_pack_bases_dev(d.data_ptr(), n, dc.data_ptr(), dv.data_ptr(), None) == 0
torch.cuda.synchronize()
for mode in ("ascii", "packed", "packed", "ascii"):
    t = yak_amd.Table(31, 10, 4, 37)
    t0 = time.perf_counter()
    if mode == "ascii": t.count_pass(1, [(d.data_ptr(), n, 0)])
    else: t.count_pass_packed(1, [(dc.data_ptr(), dv.data_ptr(), n, 0)])
    torch.cuda.synchronize(); t1 = time.perf_counter()
    s = t.stats()
    print(mode, round((t1 - t0) * 1e3, 1), {k: round(v, 2) for k, v in s.items() if k.startswith("ms_") and v}, flush=True)
    t.close()
