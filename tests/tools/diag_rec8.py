"""which of the record formats / batch sizes changes the bytes at size (diagnostic, GPU box)"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, yak_amd, bench
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
bf = int(sys.argv[2]) if len(sys.argv) > 2 else 37
h = bench.make_reads(reads, 5 * reads, 42, 0, torch, 32)
d = h.to("cuda:0"); nb = d.numel()
full = len(sys.argv) > 3
ck0 = int(d[:nb // 8 * 8].view(torch.int64).sum().item())
def run(env):
    for k, v in env.items(): os.environ[k] = v
    t = yak_amd.Table(31, 10, 4, bf)
    t.count_pass(1, [(d.data_ptr(), nb, 0)])
    s = t.stats()
    torch.cuda.synchronize()
    print("   bases checksum after pass 1:", int(d[:nb // 8 * 8].view(torch.int64).sum().item()), "before:", ck0)
    if full:
        m1 = t.dump_md5()[0][:8]
        t.destroy_bf(); t.clear(); mc = t.dump_md5()[0][:8]; t.count_pass(0, [(d.data_ptr(), nb, 0)]); m2 = t.dump_md5()[0][:8]
        t.clear(); t.count_pass(0, [(d.data_ptr(), nb, 0)]); m2b = t.dump_md5()[0][:8]
        t.shrink(2, 1023)
        print("   pass1", m1, "cleared", mc, "pass2", m2, "pass2 again", m2b)
    md5 = t.dump_md5()[0]; tot = t.tot
    t.close()
    for k in env: del os.environ[k]
    return md5, tot, round(s["ms_extract"], 2), round(s.get("ms_part2", 0), 2), round(s["ms_insert"], 2), round(s["ms_sort"], 2), round(s["ms_replay"], 2)
for env in ({}, {}, {"YAKAMD_BATCH": str(1 << 28)}, {"YAKAMD_BATCH": str(1 << 25)}): print(env, run(env), flush=True)
for env in () and ({"YAKAMD_REC8": "0"}, {}, {"YAKAMD_REC8_OUT": "0"}, {"YAKAMD_BATCH": str(1 << 25)}, {"YAKAMD_BATCH": str(1 << 25), "YAKAMD_REC8": "0"},
            {"YAKAMD_BATCH": str(1 << 25), "YAKAMD_REC8_OUT": "0"}, {"YAKAMD_BATCH": str(1 << 28)}, {}, {"YAKAMD_BATCH": str(1 << 25)}, {}, {"YAKAMD_BATCH": str(1 << 25)}):
    print(env, run(env), flush=True)
