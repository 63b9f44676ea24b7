import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
for n in (100_000_000, 600_000_000, 1_200_000_000):
    a = torch.arange(n, dtype=torch.int64, device=dev) * 7 + 3
    b = torch.empty_like(a)
    dist.all_to_all_single(b, a, [n], [n])
    torch.cuda.synchronize()
    bad = int((a != b).sum())
    print("all_to_all_single", n, "mismatches", bad, flush=True)
    b.zero_()
    dist.all_to_all([b], [a])
    torch.cuda.synchronize()
    print("all_to_all(list)  ", n, "mismatches", int((a != b).sum()), flush=True)
    del a, b
dist.destroy_process_group()
