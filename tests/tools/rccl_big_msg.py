#!/usr/bin/env python3
"""Reproducer for the reason yak_amd/shard.py cuts its all-to-all into rounds of <= 2^26 elements (512 MB of int64) per peer.

Round 1 of this project saw a multi-GB all-to-all message arrive corrupted with the torch / RCCL build of this image; the only rig it ever
had was ONE GPU, i.e. a single rank sending to itself (`bench.py --force-exchange`, commit "shard.exchange: all-to-all in rounds of <= 64M
elements per peer").  No box with several GPUs has been available to any round, so whether messages between DIFFERENT devices are affected
is not known.  This script checks both situations with position- and rank-tagged payloads:

    python tests/tools/rccl_big_msg.py [--gib 3]                                  # one rank, message to itself (runs on a 1-GPU box)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tests/tools/rccl_big_msg.py

Every rank sends one message of --gib GiB per peer and checks every element it receives; then the same payload in rounds of 2^26
elements.  Exit code 0 = both ways clean (the cap in shard.py can go), 1 = the single big message is corrupted but the rounds are clean
(keep the cap), 2 = something else is wrong.  The versions are printed for the record.
"""
import argparse
import os
import sys

import torch
import torch.distributed as dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=3.0, help="GiB per peer message")
    ap.add_argument("--round-elems", type=int, default=1 << 26)
    a = ap.parse_args()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    n = int(a.gib * (1 << 30)) // 8
    if rank == 0:
        print(f"torch {torch.__version__}, hip {torch.version.hip}, nccl/rccl {torch.cuda.nccl.version()}, world {world}, "
              f"devices {torch.cuda.device_count()}, {a.gib} GiB per peer", flush=True)

    def payload(src, dst):
        return (torch.arange(n, dtype=torch.int64, device=dev) << 8) | (src << 4) | dst

    def check(recv, label):
        bad = 0
        for src in range(world):
            want = payload(src, rank)
            bad += int((recv[src] != want).sum().item())
            del want
        print(f"rank {rank}: {label}: {bad} wrong elements of {world * n}", flush=True)
        return bad

    send = [payload(rank, d) for d in range(world)]
    recv = [torch.zeros(n, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_to_all(recv, send)
    torch.cuda.synchronize()
    bad_big = check(recv, "one message per peer")
    for r in recv:
        r.zero_()
    for lo in range(0, n, a.round_elems):
        hi = min(n, lo + a.round_elems)
        dist.all_to_all([r[lo:hi] for r in recv], [s[lo:hi] for s in send])
    torch.cuda.synchronize()
    bad_rounds = check(recv, f"rounds of {a.round_elems} elements")
    t = torch.tensor([bad_big, bad_rounds], dtype=torch.int64, device=dev)
    dist.all_reduce(t)
    dist.destroy_process_group()
    big, rounds = int(t[0].item()), int(t[1].item())
    sys.exit(0 if big == 0 and rounds == 0 else 1 if rounds == 0 else 2)


if __name__ == "__main__":
    main()
