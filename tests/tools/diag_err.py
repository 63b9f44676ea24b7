"""which API phase leaves a HIP error behind (diagnostic, GPU box)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, yak_amd, bench
hip = C.CDLL("libamdhip64.so")
hip.hipGetErrorString.restype = C.c_char_p
def chk(what):
    torch.cuda.synchronize()
    e = hip.hipGetLastError()
    print(what, "->", e, hip.hipGetErrorString(e).decode() if e else "ok", flush=True)
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
for env in ({}, {"YAKAMD_REC8": "0"}):
    os.environ.update(env)
    print("env", env)
    h = bench.make_reads(reads, 5 * reads, 42, 0, torch, 32)
    d = h.to("cuda:0"); nb = d.numel()
    chk("upload")
    t = yak_amd.Table(31, 10, 4, 37); chk("init")
    t.count_pass(1, [(d.data_ptr(), nb, 0)]); chk("pass 1")
    t.destroy_bf(); t.clear(); chk("destroy_bf + clear")
    t.count_pass(0, [(d.data_ptr(), nb, 0)]); chk("pass 2")
    t.shrink(2, 1023); chk("shrink")
    x = torch.empty(nb, dtype=torch.int16, device="cuda:0"); chk("torch.empty")
    yak_amd.lib().yakamd_lookup_dev(t.h, d.data_ptr(), nb, x.data_ptr()); chk("lookup")
    t.close(); chk("close")
    for k in env: del os.environ[k]
import __graft_entry__ as ge
ge.smoke(); chk("smoke")
