"""N > 1 path without GPUs: two gloo processes run the sharding logic of yak_amd/shard.py
(ownership, exchange, stream-order reconstruction) with the ORACLE standing in for the device
extraction and insertion; the concatenated per-rank sub-tables must equal the single-process
reference result byte for byte -- no bloom and the full two-pass bloom protocol."""
import ctypes as C
import os
import subprocess
import sys

import pytest

from conftest import ROOT

WORKER = r'''
import ctypes as C, os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from oracle import pyoracle
from yak_amd import shard
import __graft_entry__ as ge

K, PRE, P = 31, 10, 1024
bf_shift = int(sys.argv[1]); out = sys.argv[2]
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
L = pyoracle.lib()
n_reads = 1500
img = ge._synth(n_reads, 150, 12000, 5, first=rank * n_reads)          # this rank's slice of the reads
slice_bytes = len(img)
h = np.empty(len(img), dtype=np.uint64); t = np.empty(len(img), dtype=np.uint32)
m = L.yko_extract_pos(K, img, len(img), h.ctypes.data, t.ctypes.data)
h, t = h[:m], t[:m]
lo, hi = shard.owner_range(rank, world, P)
tab = L.yko_ch_init(K, PRE, 4, bf_shift)

def one_pass(create_new):
    dest = (h & np.uint64(P - 1)).astype(np.int64) // (P // world)
    order = np.argsort(dest, kind="stable")
    counts = [int((dest == d).sum()) for d in range(world)]
    sh = torch.from_numpy(h[order].astype(np.int64)); st = torch.from_numpy(t[order].astype(np.int32))
    rh, rt, rc = shard.exchange(sh, st, counts)
    rh = rh.numpy().astype(np.uint64); rt = rt.numpy().astype(np.uint32).astype(np.uint64)
    tg = np.empty(len(rh), dtype=np.uint64)
    for src, off, n, t0 in shard.segments(rc, slice_bytes):
        tg[off:off + n] = rt[off:off + n] + np.uint64(t0)
    # the owner replays its k-mers in stream order, prefix by prefix (what the device layout replay encodes)
    o = np.lexsort((tg, rh & np.uint64(P - 1)))
    rh = rh[o]; pref = (rh & np.uint64(P - 1)).astype(np.int64)
    n_ins = 0
    for p in range(lo, hi):
        a = np.ascontiguousarray(rh[pref == p])
        if len(a):
            n_ins += L.yko_ch_insert_list(tab, create_new, len(a), a.ctypes.data_as(C.POINTER(C.c_uint64)))
    tab.contents.tot += n_ins

one_pass(1)
if bf_shift > 0:
    L.yko_ch_destroy_bf(tab); L.yko_ch_clear(tab); one_pass(0); L.yko_ch_shrink(tab, 2, 1023)
data = pyoracle.dump_bytes(tab)
# keep only the owned sub-tables' bytes
import struct
off, parts = 16, []
for p in range(P):
    cap, n = struct.unpack_from("<II", data, off)
    if lo <= p < hi: parts.append(data[off:off + 8 + 8 * n])
    off += 8 + 8 * n
blob = b"".join(parts)
gathered = [None] * world
dist.all_gather_object(gathered, (blob, int(tab.contents.tot)))
if rank == 0:
    open(out, "wb").write(data[:16] + b"".join(g[0] for g in gathered))
    open(out + ".tot", "w").write(str(sum(g[1] for g in gathered)))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("bf_shift", [0, 22])
def test_two_rank_sharding_equals_single_process(bf_shift, tmp_path, oracle):
    import __graft_entry__ as ge
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    out = str(tmp_path / "sharded.yak")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                    "--master-addr", "127.0.0.1", "--master-port", str(29500 + bf_shift), str(script), str(bf_shift), out],
                   check=True, env=env, timeout=600, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    whole = ge._synth(1500, 150, 12000, 5, first=0) + ge._synth(1500, 150, 12000, 5, first=1500)
    want, wtot = oracle.count_protocol_mem(whole, k=31, bf_shift=bf_shift)
    assert open(out, "rb").read() == want
    assert int(open(out + ".tot").read()) == wtot


def test_owner_ranges():
    from yak_amd import shard
    for world in (1, 2, 4, 8):
        r = [shard.owner_range(i, world, 1024) for i in range(world)]
        assert r[0][0] == 0 and r[-1][1] == 1024 and all(a[1] == b[0] for a, b in zip(r, r[1:]))
        assert all(shard.owner_of(p, world, 1024) == i for i, (lo, hi) in enumerate(r) for p in (lo, hi - 1))
    with pytest.raises(ValueError):
        shard.owner_range(0, 3, 1024)


WORKER2 = r'''
import sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from yak_amd import shard

P = 1024
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
lo, hi = shard.owner_range(rank, world, P)

def records_of(r):
    """what rank r partitions: {{hash, position}} sorted by prefix = low 10 bits of the hash (ragged groups, some empty)"""
    g = np.random.default_rng(100 + r)
    n = 5000 + 700 * r
    h = g.integers(0, 1 << 62, size=n, dtype=np.int64)
    h[(h & (P - 1)) % 7 == 3] |= 5                     # leave some prefixes empty
    o = np.argsort(h & (P - 1), kind="stable")
    h = h[o]; t = g.integers(0, 1 << 31, size=n, dtype=np.int64)
    bst = np.searchsorted(h & (P - 1), np.arange(P + 1))
    return np.stack([h, t], axis=1), bst

rec, bst = records_of(rank)
for width in (2, 1):
    send = torch.from_numpy(rec if width == 2 else np.ascontiguousarray(rec[:, 0]))
    got = shard.exchange_partitioned(send, [int(x) for x in bst], P)
    assert len(got) == world
    for src, (sl, offs) in enumerate(got):
        r2, b2 = records_of(src)
        want = r2[b2[lo]:b2[hi]] if width == 2 else r2[b2[lo]:b2[hi], 0]
        assert sl.shape[0] == b2[hi] - b2[lo] and np.array_equal(sl.numpy(), want), (width, src)
        assert len(offs) == P + 1 and offs[0] == 0 and offs[-1] == sl.shape[0]
        assert all(offs[p + 1] - offs[p] == (b2[p + 1] - b2[p] if lo <= p < hi else 0) for p in range(P))
dist.destroy_process_group()
'''


def test_partitioned_exchange_formats_two_ranks(tmp_path):
    """the production exchange (per-source slices + per-prefix offsets), 16-byte records of a counting
    pass and the 8-byte hashes of a count-existing pass, ragged and empty prefix groups"""
    script = tmp_path / "worker2.py"
    script.write_text(WORKER2.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                    "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                   check=True, env=env, timeout=600, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
