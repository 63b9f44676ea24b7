"""Prefix sharding of the 1<<pre sub-tables over the GPUs of one node.

Sub-table p is touched only by k-mers whose hash has prefix p (reference count.c:19, htab.c:56),
so rank r of `world` owns the contiguous prefixes [r*P/world, (r+1)*P/world) and the job needs
exactly one exchange per pass: every hashed k-mer travels to the owner of its prefix together with
its position in the logical input stream.  The logical stream of the whole job is rank 0's reads,
then rank 1's, ...; a k-mer at local position t of rank s therefore has stream time
s * slice_bytes + t, which is all the owner needs to reproduce the reference's insertion order.

This module is pure plumbing (torch.distributed tensors in, tensors out): the same code drives
RCCL on GPUs (bench.py) and gloo on CPU (tests/test_shard_gloo.py).
"""
import torch
import torch.distributed as dist


def owner_range(rank, world, n_prefix):
    """prefixes [lo, hi) owned by `rank`"""
    if n_prefix % world:
        raise ValueError("the number of ranks must divide the number of sub-tables")
    return rank * n_prefix // world, (rank + 1) * n_prefix // world


def owner_of(prefix, world, n_prefix):
    return prefix // (n_prefix // world)


def _a2a(recv, send, recv_counts, send_counts):
    """all_to_all_single on backends that have it (nccl == RCCL); send/recv pairs on gloo"""
    if dist.get_backend() != "gloo":
        dist.all_to_all_single(recv, send, recv_counts, send_counts)
        return
    rank, world = dist.get_rank(), dist.get_world_size()
    so = [0]
    for c in send_counts:
        so.append(so[-1] + c)
    ro = [0]
    for c in recv_counts:
        ro.append(ro[-1] + c)
    recv[ro[rank]:ro[rank + 1]] = send[so[rank]:so[rank + 1]]
    reqs, landing = [], []
    for peer in range(world):
        if peer == rank:
            continue
        if send_counts[peer]:
            reqs.append(dist.isend(send[so[peer]:so[peer + 1]].cpu().contiguous(), peer))   # gloo moves host memory
        if recv_counts[peer]:
            buf = torch.empty(recv_counts[peer], dtype=recv.dtype)
            reqs.append(dist.irecv(buf, peer))
            landing.append((peer, buf))
    for r in reqs:
        r.wait()
    for peer, buf in landing:
        recv[ro[peer]:ro[peer + 1]] = buf.to(recv.device)


MAX_MSG_ELEMS = 1 << 26      # per-peer message size cap (elements): RCCL/torch mis-handle messages of several GB


def _a2a_rounds(recv, send, recv_counts, send_counts, async_op=False):
    """all-to-all of variable-size segments in rounds of at most MAX_MSG_ELEMS elements per peer;
    every round writes straight into its final place (views), so `recv` ends up grouped by source.
    async_op (RCCL only): the rounds are queued on the collective stream and a list of work handles is
    returned for the caller to wait on -- the exchange then overlaps whatever the caller runs next."""
    world = dist.get_world_size()
    so, ro = [0], [0]
    for c in send_counts:
        so.append(so[-1] + c)
    for c in recv_counts:
        ro.append(ro[-1] + c)
    biggest = torch.tensor([max(list(send_counts) + list(recv_counts) + [0])], dtype=torch.int64, device=send.device)
    if world > 1:
        dist.all_reduce(biggest, op=dist.ReduceOp.MAX)           # every rank must run the same number of rounds
    rounds = (int(biggest.item()) + MAX_MSG_ELEMS - 1) // MAX_MSG_ELEMS
    works = []
    for r in range(rounds):
        a, b = r * MAX_MSG_ELEMS, (r + 1) * MAX_MSG_ELEMS
        sc = [max(0, min(c, b) - a) if c > a else 0 for c in send_counts]
        rc = [max(0, min(c, b) - a) if c > a else 0 for c in recv_counts]
        if dist.get_backend() == "gloo":
            ins = torch.cat([send[so[d] + a:so[d] + a + sc[d]] for d in range(world)]) if sum(sc) else send[:0]
            out = torch.empty(sum(rc), dtype=recv.dtype, device=recv.device)
            _a2a(out, ins, rc, sc)
            o = 0
            for s_ in range(world):
                recv[ro[s_] + a:ro[s_] + a + rc[s_]] = out[o:o + rc[s_]]
                o += rc[s_]
        else:
            w = dist.all_to_all([recv[ro[s_] + a:ro[s_] + a + rc[s_]] for s_ in range(world)],
                                [send[so[d] + a:so[d] + a + sc[d]] for d in range(world)], async_op=async_op)
            if async_op:
                works.append(w)
    return works


def exchange(send_hash, send_t, send_counts):
    """send_hash (int64) / send_t (int32, may be None) are grouped by destination rank with
    `send_counts[d]` entries for rank d.  Returns (recv_hash, recv_t, recv_counts) grouped by
    SOURCE rank -- i.e. in the stream order of the whole job."""
    world = dist.get_world_size()
    dev = send_hash.device
    sc = torch.tensor(send_counts, dtype=torch.int64, device=dev)
    rc = torch.empty_like(sc)
    _a2a(rc, sc, [1] * world, [1] * world)
    recv_counts = [int(x) for x in rc.tolist()]
    n_recv = sum(recv_counts)
    recv_hash = torch.empty(n_recv, dtype=send_hash.dtype, device=dev)
    _a2a_rounds(recv_hash, send_hash, recv_counts, list(send_counts))
    recv_t = None
    if send_t is not None:
        recv_t = torch.empty(n_recv, dtype=send_t.dtype, device=dev)
        _a2a_rounds(recv_t, send_t, recv_counts, list(send_counts))
    return recv_hash, recv_t, recv_counts


def segments(recv_counts, slice_bytes):
    """(source rank, offset, n, t0) of each received segment, in stream order: t0 is the stream
    time of the first byte of that source rank's slice"""
    off = 0
    for src, n in enumerate(recv_counts):
        if n:
            yield src, off, n, src * slice_bytes
        off += n


def exchange_hashes(send_rec, bstart, n_prefix):
    """pass 2 (count existing only): ship just the 8-byte hash of every record to its owner; order and
    grouping no longer matter.  Returns one int64 tensor of received hashes."""
    world = dist.get_world_size()
    per = n_prefix // world
    send_counts = [int(bstart[(d + 1) * per] - bstart[d * per]) for d in range(world)]
    h = send_rec[:, 0].contiguous()
    recv_hash, _, _ = exchange(h, None, send_counts)
    return recv_hash


def exchange_partitioned(send_rec, bstart, n_prefix, async_op=False):
    """send_rec: int64 tensor of records grouped by sub-table prefix (ascending) -- [n, 2] {hash,
    position} for a counting pass (yakamd_partition_dev), or [n] bare hashes for a pass that only
    counts existing keys (yakamd_partition_hashes_dev); `bstart` the n_prefix + 1 group offsets.
    Owners are contiguous prefix ranges, so the per-destination send buffers are slices.  Returns a
    list, in source-rank order, of (recv_slice, offsets[n_prefix + 1] of that slice) ready for
    yakamd_feed_partitioned_dev / yakamd_count_partitioned_dev.
    async_op: the payload all-to-all is only queued; the return value is a function that waits for
    it and yields that list (send_rec must stay untouched until then)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = send_rec.device
    width = send_rec.shape[1] if send_rec.dim() == 2 else 1
    per = n_prefix // world
    send_counts = [int(bstart[(d + 1) * per] - bstart[d * per]) for d in range(world)]
    # relative offsets of the owned prefixes inside each destination's slice
    rel = torch.tensor([[int(bstart[d * per + j] - bstart[d * per]) for j in range(per + 1)] for d in range(world)],
                       dtype=torch.int64, device=dev).reshape(-1)
    rel_in = torch.empty_like(rel)
    _a2a(rel_in, rel, [per + 1] * world, [per + 1] * world)
    rel_in = rel_in.reshape(world, per + 1).tolist()
    recv_counts = [r[-1] for r in rel_in]
    flat = send_rec.reshape(-1)
    recv = torch.empty(width * sum(recv_counts), dtype=torch.int64, device=dev)
    works = _a2a_rounds(recv, flat, [width * c for c in recv_counts], [width * c for c in send_counts], async_op=async_op)

    def finish():
        for w in works:
            w.wait()
        got = recv.reshape(-1, width) if width > 1 else recv
        lo = rank * per
        out, off = [], 0
        for src in range(world):
            m = recv_counts[src]
            offs = [0] * lo + rel_in[src] + [m] * (n_prefix - lo - per)
            out.append((got[off:off + m], offs))
            off += m
        return out
    return finish if async_op else finish()
