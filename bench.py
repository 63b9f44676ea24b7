#!/usr/bin/env python3
"""bench.py -- `yak count -k31 -b37` on synthetic 150-bp reads, device path, N GPUs of one node.

One "step" = the whole counting job of reference main.c:53-61 on one batch of synthetic reads that
is already resident in HBM: yak_ch_init (bloom filters included) -> pass 1 (extract, bloom gate,
insert; exact khashl layout) -> destroy_bf -> clear -> pass 2 (count existing) -> shrink(2,1023).
N = 1 runs BASELINE.json configs[1] (10 M x 150 bp, G = 50 Mb, e = 0.5 %).  N > 1 is weak scaling:
every rank brings its own 10 M reads of a genome N times as long, the 1024 sub-tables are sharded
by hash prefix over the ranks and the hashed k-mers travel to their owner with one RCCL
all-to-all per pass.

`--gpus N` without a torch.distributed environment re-launches itself as N ranks (torch.distributed.run
on 127.0.0.1).  `--config nofilter|cfg4|cfg5` run the other BASELINE.json configurations (no filter;
the long-contig k = 21 assembly; the lookup-only path of `yak qv`) with the same output contract.

Prints ONE JSON line (rank 0).  `value` = distinct k-mers in the final table (h->tot, the number
the reference logs) per second, whole job over all ranks; `kmer_instances_per_s` = k-mer windows
consumed per second (both passes).  `roofline` is the whole protocol step (SURVEY 8(d)'s algorithmic bytes of every
instance over the step's wall-clock time; `roofline.traffic` the rocprofv3 counter bytes of the same step);
`roofline.dominant_kernel` is the single kernel with the largest time, timed with HIP events on the engine's own
stream inside the library (to be compared with profiles/*_kernel_stats.csv); `cpu_baseline` is the reference
(oracle/_ref) or the oracle port timed on this host on a bounded sample.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K, PRE, N_HASH = 31, 10, 4
READ_LEN = 150
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
B_INSERT = 24.0                # algorithmic B / instance of k_acc_insert
B_LOOKUP = 16.0                # k_img_count: 8 read + 8 slot read (+ 8 * f_hit, added at run time)


def synth_lib():
    L = C.CDLL(os.path.join(ROOT, "tools", "libyaksynth.so"))
    L.yaksynth_reads.restype = C.c_int64
    L.yaksynth_reads.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_uint64,
                                 C.c_double, C.c_double, C.c_int64, C.c_int]
    return L


class HostBuf:
    """page-locked host bytes from the library's own runtime (yakamd_host_alloc): the single-GPU mode needs no PyTorch"""
    def __init__(self, L, nbytes):
        self.L, self.n = L, nbytes
        self.p = L.yakamd_host_alloc(max(1, nbytes))
        if not self.p:
            raise MemoryError("yakamd_host_alloc")

    def data_ptr(self):
        return self.p

    def numel(self):
        return self.n

    def __del__(self):
        if getattr(self, "p", None):
            self.L.yakamd_host_free(self.p); self.p = None


class DevBuf:
    """device bytes (yakamd_dev_alloc)"""
    def __init__(self, L, nbytes):
        self.L, self.n = L, nbytes
        self.p = L.yakamd_dev_alloc(max(16, nbytes))
        if not self.p:
            raise MemoryError("yakamd_dev_alloc")

    def data_ptr(self):
        return self.p

    def numel(self):
        return self.n

    def to_numpy(self, dtype):
        import numpy as np
        out = np.empty(self.n // np.dtype(dtype).itemsize, dtype=dtype)
        if self.L.yakamd_memcpy_d2h(out.ctypes.data, self.p, out.nbytes) != 0:
            raise RuntimeError("d2h")
        return out

    def __del__(self):
        if getattr(self, "p", None):
            self.L.yakamd_dev_free(self.p); self.p = None


def make_reads(n_reads, genome, seed, first, torch, threads, L=None):
    """reads of this rank in page-locked host memory (memory image: 150 bases + '\\n' per read): a pinned uint8 tensor for the torch driver, else a HostBuf"""
    if torch is not None:
        t = torch.empty(n_reads * (READ_LEN + 1), dtype=torch.uint8, pin_memory=True)
    else:
        t = HostBuf(L, n_reads * (READ_LEN + 1))
    synth_lib().yaksynth_reads(t.data_ptr(), n_reads, READ_LEN, genome, seed, 0.005, 0.0005, first, threads)
    return t


def _ref_count(fq, bits, threads, k=K):
    from oracle import pyoracle
    t0 = time.time()
    r = subprocess.run([pyoracle.REF_BIN, "count", f"-k{k}"] + ([f"-b{bits}"] if bits else []) + [f"-t{threads}", "-o", "/dev/null", fq],
                       stderr=subprocess.PIPE, check=True)
    dt = time.time() - t0
    m = re.findall(rb"(\d+) distinct k-mers", r.stderr)
    return dt, (int(m[-1]) if m else 0)


def cpu_baseline(n_reads, genome, bf_shift, threads, t1_reads=500_000, err=None):
    """The reference itself (oracle/_ref/yak, compiled in the build container, carried as a prebuilt file) on
    the SAME reads as the device run (tools/yaksynth writes them as FASTQ: same seed, same order), all host
    cores; plus a -t1 figure on a stated fraction.  Without the prebuilt reference: the oracle port (1 core)
    on a bounded sample."""
    from oracle import pyoracle
    tmp = tempfile.mkdtemp(prefix="ykb", dir=os.environ.get("YAKAMD_TMP", None))
    try:
        if pyoracle.have_ref():
            fq = os.path.join(tmp, "s.fq")
            subprocess.check_call([os.path.join(ROOT, "tools", "yaksynth"), "-n", str(n_reads), "-l", str(READ_LEN),
                                   "-g", str(genome), "-s", "42", "-t", str(min(threads, 32)), "-o", fq] + (["-e", str(err)] if err is not None else []))
            dt, tot = _ref_count(fq, bf_shift, threads)
            inst = n_reads * (READ_LEN - K + 1) * (2 if bf_shift else 1)
            out = {"value": tot / dt, "unit": "distinct k-mers/s", "cores": threads, "kind": "reference",
                   "kmer_instances_per_s": inst / dt, "seconds": round(dt, 2),
                   "sample": f"yak count -k{K}" + (f" -b{bf_shift}" if bf_shift else "") + f" -t{threads} on the SAME {n_reads} x {READ_LEN} bp reads as the device run "
                             f"(tools/yaksynth -g {genome} -s 42 written as FASTQ, file in page cache), " + ("both passes" if bf_shift else "one pass")}
            if t1_reads <= 0:
                return out
            # one core: a fraction of the reads (same genome), the filter scaled with the instances
            f1 = os.path.join(tmp, "s1.fq")
            subprocess.check_call(["head", "-n", str(4 * t1_reads), fq], stdout=open(f1, "wb"))
            bits1 = bf_shift
            while bits1 > 20 and (1 << bits1) > (1 << bf_shift) * t1_reads * 2 // n_reads:
                bits1 -= 1
            dt1, tot1 = _ref_count(f1, bits1 if bf_shift else 0, 1)
            out["t1"] = {"value": tot1 / dt1, "cores": 1, "seconds": round(dt1, 2), "kmer_instances_per_s": t1_reads * (READ_LEN - K + 1) * (2 if bf_shift else 1) / dt1,
                         "sample": f"first {t1_reads} of those reads, -t1" + (f" -b{bits1}" if bf_shift else "")}
            return out
        import __graft_entry__ as ge
        sample = min(n_reads, 200_000)
        g = max(sample * READ_LEN // 30, 1000)
        bits = bf_shift
        while bits > 20 and (1 << bits) > 114 * sample * 120 * 2:
            bits -= 1
        reads = ge._synth(sample, READ_LEN, g, 4242)
        t0 = time.time()
        _, tot = pyoracle.count_protocol_mem(reads, k=K, bf_shift=bits if bf_shift else 0)
        dt = time.time() - t0
        return {"value": tot / dt, "unit": "distinct k-mers/s", "cores": 1, "kind": "port", "seconds": round(dt, 2),
                "kmer_instances_per_s": sample * (READ_LEN - K + 1) * (2 if bf_shift else 1) / dt,
                "sample": f"oracle restatement on {sample} x {READ_LEN} bp synthetic reads (G={g}), -b{bits}"}
    finally:
        subprocess.call(["rm", "-rf", tmp])


def maybe_spawn(a):
    """`python bench.py --gpus N` run bare: become N ranks (one process per GPU, RCCL over xGMI)"""
    if a.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
                              "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])


def run_multi_c(a):
    """N > 1, the product's own multi-GPU driver (yakamd_count_multi_dev, the device-resident twin of yak_count() under YAKAMD_GPUS): ONE process
    -- rank 0 -- holds every GPU of the node, the library partitions the chunks on their devices, exchanges the records (RCCL grouped send / recv
    over xGMI, or peer copies) and counts every rank's prefix range; the other ranks torch.distributed.run started only take part in the
    barriers (gloo: they never touch a GPU).  Default workload: weak scaling of BASELINE configs[2] -- 75 M x 150 bp reads per GPU (8 GPUs = the
    600 M reads), G = 5 x reads, e = 0.1 %, no filter, one pass; --bf-shift B runs the filtered two-pass protocol instead."""
    import torch
    import torch.distributed as dist
    import yak_amd
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if world > 1:
        dist.init_process_group("gloo")
    P = 1 << PRE
    if P % a.gpus:
        raise SystemExit("the number of GPUs must divide 1024 sub-tables")
    bf = a.bf_shift if a.bf_shift_given else 0
    per_gpu = a.reads if a.reads_given else 75_000_000
    if a.scaling == "strong":
        if a.total_reads <= 0:
            raise SystemExit("--scaling strong needs --total-reads")
        per_gpu = a.total_reads // a.gpus
    err = 0.001 if bf == 0 else 0.005

    def barrier():
        if world > 1:
            dist.barrier()

    if rank != 0:                                           # the job is rank 0's process; the others keep the launcher's contract
        for _ in range(3):
            barrier()
        dist.destroy_process_group()
        return
    L = yak_amd.lib()
    n_dev = torch.cuda.device_count()
    if L.yakamd_device_count() < 1 or n_dev < 1:
        raise SystemExit("no gfx950 device: refusing to run (no CPU fallback)")
    N = a.gpus
    devs = [r % n_dev for r in range(N)]                     # fewer devices than ranks: ranks share them (a one-GPU box posing as N)
    own_slots = any(kv.partition("=")[0] == "YAKAMD_MGPU_SLOT_PER_RANK" and int(kv.partition("=")[2]) for kv in a.knob)
    sdev = []                                                # a slot per distinct device -- or, under the library's test switch, per rank (ranks that share a device then exchange as if they did not)
    for d in devs:
        if own_slots or d not in sdev:
            sdev.append(d)
    S = len(sdev)
    n_distinct = len(set(sdev))
    genome = 5 * per_gpu * N
    rec_len = READ_LEN + 1
    # the job's reads, dealt in rounds of one chunk per device (chunk c = round * S + device: the job's own read numbering, as yak_count() deals a file)
    per_dev = per_gpu * N // S
    chunk_reads = min(per_dev, max(1, a.batch_reads if a.batch_reads_given else (1 << 29) // rec_len))
    n_rounds = -(-per_dev // chunk_reads)
    threads = min(os.cpu_count() or 8, 64)
    syn = synth_lib()
    h_buf = torch.empty(chunk_reads * rec_len, dtype=torch.uint8, pin_memory=True)
    bufs, ptrs, sizes = [], [], []
    tg = time.perf_counter()
    for b in range(n_rounds):
        for s in range(S):
            n_reads = max(0, min(chunk_reads, per_dev - b * chunk_reads))
            first = b * S * chunk_reads + s * n_reads           # the job's stream = reads 0, 1, 2, ... in order (every round before this one was a full one): the stream tests/golden/cfg3_full.json pins
            t = torch.empty(max(16, n_reads * rec_len), dtype=torch.uint8, device=f"cuda:{sdev[s]}")
            if n_reads:
                syn.yaksynth_reads(h_buf.data_ptr(), n_reads, READ_LEN, genome, 42, err, 0.0005, first, threads)
                t[:n_reads * rec_len].copy_(h_buf[:n_reads * rec_len])
                torch.cuda.synchronize(sdev[s])
            bufs.append(t); ptrs.append(t.data_ptr()); sizes.append(n_reads * rec_len)
    gen_s = time.perf_counter() - tg
    d_chunk = (C.c_void_p * len(ptrs))(*ptrs)
    n_bytes = (C.c_int64 * len(sizes))(*sizes)
    dev_arr = (C.c_int * N)(*devs)
    opt = yak_amd.CoptT(); L.yak_copt_init(C.byref(opt)); opt.k, opt.pre, opt.bf_shift, opt.bf_n_hash = K, PRE, bf, N_HASH
    exch = C.c_int(-1)

    def sync_all():
        for d in sdev:
            torch.cuda.synchronize(d)

    def step(keep=False):
        h = L.yakamd_count_multi_dev(C.byref(opt), None, N, dev_arr, n_rounds, d_chunk, n_bytes, C.byref(exch))
        if not h:
            raise RuntimeError("yakamd_count_multi_dev: " + yak_amd._err())
        if bf > 0:
            L.yak_ch_destroy_bf(h); L.yak_ch_clear(h, 1)
            if not L.yakamd_count_multi_dev(C.byref(opt), h, N, dev_arr, n_rounds, d_chunk, n_bytes, C.byref(exch)):
                raise RuntimeError("yakamd_count_multi_dev (count pass): " + yak_amd._err())
            L.yak_ch_shrink(h, 2, 1023, 1)
        sync_all()
        tot = h.contents.tot
        if keep:
            return h, tot
        L.yak_ch_destroy(h)
        return None, tot

    t_first = time.perf_counter()
    first_ms = None
    for i_ in range(a.warmup):
        step()
        if i_ == 0:
            sync_all(); first_ms = (time.perf_counter() - t_first) * 1e3     # the first job of the process: the driver hands out (and clears) the memory for the first time
    for d in sdev:
        L.yakamd_peak_bytes(d, 1)
    sync_all(); barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        _, tot = step()
    sync_all(); barrier()
    dt = time.perf_counter() - t0
    exch_timed = exch.value                                  # how the timed steps exchanged (the verification runs behind them are jobs of their own)
    # what a device needs at its peak: the library's buffers in use (tables, records, exchange) + the input chunks this harness keeps resident on it
    in_bytes = {d: sum(bufs[i_].numel() for i_ in range(n_rounds * S) if sdev[i_ % S] == d) for d in sdev}
    peak = {d: int(L.yakamd_peak_bytes(d, 0)) + in_bytes[d] for d in sdev}
    hbm_total = {d: torch.cuda.mem_get_info(d)[1] for d in sdev}
    if any(peak[d] > 0.9 * hbm_total[d] for d in sdev):
        raise SystemExit(f"FAILED: the job needs {max(peak.values()) / 1e9:.1f} GB on one device, more than 0.9 of its {hbm_total[sdev[0]] / 1e9:.0f} GB")
    inst = sum(sizes) // rec_len * (READ_LEN - K + 1) * (2 if bf else 1)
    verify = None
    gfn = os.path.join(ROOT, "tests", "golden", "cfg3_full.json")
    if bf == 0 and not a.no_verify and os.path.exists(gfn) and P % N == 0:
        # BASELINE configs[2] itself (600 M reads, any N that divides the pinned ranges): the oracle counted the whole stream for all eight ranges of 128
        # sub-tables (tests/gen_golden_cfg3.py: the whole table, 41 GB of .yak bytes); the job's bytes at every one of them must be the oracle's
        g = json.load(open(gfn))
        if (per_gpu * N, genome, PRE, K) == (g["reads"], g["genome"], g["pre"], g["k"]):
            h, _ = step(keep=True)
            tm = yak_amd.Table(K, PRE, N_HASH, bf, ptr=h)
            verify = {"oracle": f"tests/golden/cfg3_full.json ({g['produced_by']})"}
            for rng, ref in g["ranges"].items():
                lo_, hi_ = (int(x) for x in rng.split(":"))
                md5_, size_ = tm.range_md5(lo_, hi_)
                verify[f"subtables_{lo_}_{hi_}_equal_oracle"] = (md5_, size_) == (ref["md5"], ref["size"])
            tm.close()
            if not all(v for k_, v in verify.items() if k_.startswith("subtables_")):
                raise SystemExit(f"FAILED: the job's sub-tables differ from the oracle's: {verify}")
    if a.job_md5 or per_gpu * N <= 4_000_000:
        # the sharded table's .yak bytes against ONE unsharded table fed the same stream, chunk after chunk
        h, _ = step(keep=True)
        tm = yak_amd.Table(K, PRE, N_HASH, bf, ptr=h)
        md5_n = tm.dump_md5()[0]
        tm.close()
        torch.cuda.set_device(sdev[0])
        t1 = yak_amd.Table(K, PRE, N_HASH, bf)
        feeds, off = [], 0
        for i, (p_, n_) in enumerate(zip(ptrs, sizes)):
            if n_:
                if S > 1:                                    # bring the chunk to device 0
                    c0 = bufs[i].to(f"cuda:{sdev[0]}"); bufs.append(c0); p_ = c0.data_ptr()
                feeds.append((p_, n_, off))
            off += n_
        t1.count_pass(1, feeds)
        if bf > 0:
            t1.destroy_bf(); t1.clear(); t1.count_pass(0, feeds); t1.shrink(2, 1023)
        md5_1 = t1.dump_md5()[0]
        t1.close()
        verify = dict(verify or {}, job_yak_md5=md5_n, one_table_yak_md5=md5_1, equals_one_table=md5_n == md5_1)
        if md5_n != md5_1:
            raise SystemExit("FAILED: the sharded job's .yak differs from the single table's")
    ms = dt / a.steps * 1e3
    # ---- the same per-GPU workload on ONE device in this process: per_gpu reads (G = 5 x reads: 30x as the job), all 1024 sub-tables, the same
    # driver with one rank -- the N = 1 point of THIS curve (the `--gpus 1` line of the driver's sweep is configs[1], a different job)
    weak_base = None
    if not a.no_weak_base:
        del d_chunk
        bufs.clear()                                         # the job's chunks go before the base's are made (a one-GPU box holds N ranks' input)
        torch.cuda.empty_cache()
        g1 = 5 * per_gpu
        nr1 = -(-per_gpu // chunk_reads)
        b1, p1, s1 = [], [], []
        for b in range(nr1):
            n_reads = min(chunk_reads, per_gpu - b * chunk_reads)
            t = torch.empty(max(16, n_reads * rec_len), dtype=torch.uint8, device=f"cuda:{sdev[0]}")
            syn.yaksynth_reads(h_buf.data_ptr(), n_reads, READ_LEN, g1, 42, err, 0.0005, b * chunk_reads, threads)
            t[:n_reads * rec_len].copy_(h_buf[:n_reads * rec_len])
            torch.cuda.synchronize(sdev[0])
            b1.append(t); p1.append(t.data_ptr()); s1.append(n_reads * rec_len)
        dc1, nb1, dv1 = (C.c_void_p * nr1)(*p1), (C.c_int64 * nr1)(*s1), (C.c_int * 1)(sdev[0])

        def step1():
            h = L.yakamd_count_multi_dev(C.byref(opt), None, 1, dv1, nr1, dc1, nb1, None)
            if not h:
                raise RuntimeError("yakamd_count_multi_dev (weak base): " + yak_amd._err())
            if bf > 0:
                L.yak_ch_destroy_bf(h); L.yak_ch_clear(h, 1)
                if not L.yakamd_count_multi_dev(C.byref(opt), h, 1, dv1, nr1, dc1, nb1, None):
                    raise RuntimeError("yakamd_count_multi_dev (weak base, count pass): " + yak_amd._err())
                L.yak_ch_shrink(h, 2, 1023, 1)
            torch.cuda.synchronize(sdev[0])
            tot1 = h.contents.tot
            L.yak_ch_destroy(h)
            return tot1
        step1()
        tb = time.perf_counter()
        for _ in range(a.steps):
            tot1 = step1()
        ms1 = (time.perf_counter() - tb) / a.steps * 1e3
        weak_base = {"ms_per_step": ms1, "value": tot1 / (ms1 / 1e3), "unit": "distinct k-mers/s", "n_gpus": 1, "device": sdev[0], "steps": a.steps, "reads": per_gpu, "final_distinct": tot1,
                     "kmer_instances_per_s": per_gpu * (READ_LEN - K + 1) * (2 if bf else 1) / (ms1 / 1e3),
                     "workload": f"the per-GPU share of the job as a job of its own: {per_gpu} x {READ_LEN} bp reads (G={g1}, e={err * 100:g}%), all {P} sub-tables on device {sdev[0]}, "
                                 "same driver (yakamd_count_multi_dev, one rank), timed in this process right behind the job",
                     "base_ms_over_job_ms": ms1 / ms,
                     "note": "weak scaling: with N devices the job does N x this work; base_ms_over_job_ms = 1 is linear" +
                             (f" (here the {N} ranks share {n_distinct} device(s): the job does {N // n_distinct} x the base's work per device, so {n_distinct / N:.3g} is this box's ceiling)" if n_distinct < N else "")}
        del b1
    barrier()
    cpu_base = None
    if not a.no_cpu_baseline:
        # the reference itself on a bounded sample of the per-GPU workload (same generator, same error rate, 30x), all host cores of this box
        n_cpu = min(per_gpu, 4_000_000)
        cpu_base = cpu_baseline(n_cpu, 5 * n_cpu, bf, min(os.cpu_count() or 8, 32), t1_reads=0, err=err)
        cpu_base["sample_note"] = f"bounded sample: {n_cpu} of the {per_gpu} reads a GPU takes, genome scaled with it (G = 5 x reads) so that the coverage, and with it the share of repeated k-mers, is the job's"
    by = (32.0 if bf == 0 else 16.0 + 128.0 + 16.0 + 16.0 + 8.0 + 8.0) * (inst / (2 if bf else 1))
    import bench_configs
    traffic, traffic_src = bench_configs.pmc_step_traffic(f"multi_{N}x{per_gpu}" + (f"_b{bf}" if bf else ""))
    if traffic is None and (N, per_gpu, bf) == (8, 75_000_000, 0):
        # BASELINE configs[2] itself: no box holds eight ranks' jobs on one device, so the counter passes of ONE rank's share (--config cfg3shard: its own
        # partition, every feed, its pass; the stand-ins for the peers' partitions scaled out) stand for each of the eight
        t1, src1 = bench_configs.pmc_step_traffic("cfg3shard", scale={"k_xpart": 1.0 / 8, "k_part_": 1.0 / 8})
        if t1:
            traffic, traffic_src = 8 * t1, "8 x one rank's share, " + src1
    out = {"metric": "distinct k-mers counted/sec (k=31), prefix-sharded over the GPUs of one node, .yak bit-exact",
           "value": tot / (dt / a.steps), "unit": "distinct k-mers/s", "n_gpus": N, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms,
           "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": "u64", "data": "synthetic",
           "config": {"workload": f"yak count -k{K}" + (f" -b{bf}, both passes + shrink" if bf else ", no filter, one pass") +
                                  f" on {per_gpu * N} x {READ_LEN} bp synthetic reads ({per_gpu} per GPU; G={genome}, e={err * 100:g}%, N=0.05%), 30x"
                                  + ("; 8 GPUs = BASELINE configs[2]" if (bf == 0 and per_gpu == 75_000_000) else ""),
                      "reads_per_gpu": per_gpu, "k": K, "pre": PRE, "bf_shift": bf,
                      "sharding": f"{N} ranks own {P // N} sub-tables each; per round one chunk of {chunk_reads} reads per device, k-mers grouped by prefix on the device that holds the chunk, "
                                  "8-byte tagged records to the owner of their prefix",
                      "driver": "C: yakamd_count_multi_dev (libyak_amd.so), one process holding all devices; torch only allocates the input buffers",
                      "exchange": {0: "none (all ranks on one device: slices are fed where the partition left them)", 1: "RCCL grouped ncclSend/ncclRecv (one round per chunk set)",
                                   2: "hipMemcpyPeerAsync peer copies (RCCL unavailable, switched off, or a round it failed)",
                                   3: "grouped ncclSend/ncclRecv call pattern served by the library's in-process test rig (slots on one device)"}.get(exch_timed, "?"),
                      "devices": devs, "rounds": n_rounds},
           "kmer_instances_per_s": inst / (dt / a.steps), "final_distinct": tot,
           "input_generation_s_not_timed": round(gen_s, 2), "first_job_ms": first_ms,
           "peak_hbm_bytes_per_device": {str(d): peak[d] for d in sdev}, "peak_hbm_note": "library buffers in use at their high-water mark (yakamd_peak_bytes: tables, records, exchange buffers; the pool's idle ranges are not in it) + the input chunks resident on the device; the job refuses to report above 0.9 of the device's memory",
           "roofline": {"bound": "hbm", "kernel": "whole job step (partition + exchange + per-rank count + exact layout), all GPUs", "achieved": by / (dt / a.steps) / 1e9,
                        "peak": HBM_PEAK_GBS * n_distinct, "unit": "GB/s", "frac": by / (dt / a.steps) / 1e9 / (HBM_PEAK_GBS * n_distinct), "traffic": traffic,
                        "traffic_per_device": (traffic / n_distinct) if traffic else None,
                        "hbm_util": (traffic / (dt / a.steps) / 1e9 / (HBM_PEAK_GBS * n_distinct)) if traffic else None,
                        "traffic_source": (traffic_src + "; the counter passes ran this command with all ranks on ONE device (the only box there is): the bytes of every rank's kernels, "
                                           "summed; records that cross xGMI are not in them") if traffic else (traffic_src or "no counter pass of this command (N, reads per GPU) under profiles/"),
                        "kernels_sha16": yak_amd.kernels_sha16(),
                        "algorithmic_bytes_per_instance": by / max(1, inst / (2 if bf else 1))},
           "weak_base": weak_base, "cpu_baseline": cpu_base,
           "scaling_note": f"weak: {per_gpu} reads per GPU at every N > 1 (8 GPUs = BASELINE configs[2]); `weak_base` is the N = 1 point of this curve, measured in this process "
                           "(the `--gpus 1` line of a sweep is configs[1], 10 M reads through the filtered protocol: a different job); ONE rank's share of the 8-GPU job on one GPU is "
                           "`--config cfg3shard`",
           "verify": verify}
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def _device_or_retry():
    """A process that starts within seconds of the exit of another one that used the device (seen in round 5 right behind 1 Gb-assembly runs, and once right
    behind a 15 s pytest run) can find NO device: `No HIP GPUs are available` from torch, an empty device list from the HIP runtime -- and the runtime keeps
    that answer for the life of the process.  So the question is asked in a CHILD process first (torch, then libyak_amd.so: the import order INTEGRATION.md
    asks for), again every 5 s for up to a minute while the answer is no or the child fails; if the device stays away, the configuration's own check fails
    loudly as before."""
    probe = ("import sys; sys.path.insert(0, %r)\n"
             "import torch\n"
             "ok = torch.cuda.is_available() and torch.cuda.device_count() >= 1\n"
             "if ok: torch.cuda.init()\n"
             "import yak_amd\n"
             "sys.exit(0 if ok and yak_amd.lib().yakamd_device_count() >= 1 else 3)\n" % ROOT)
    for attempt in range(12):
        try:
            rc = subprocess.run([sys.executable, "-c", probe], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=180).returncode
        except Exception:
            return
        if rc == 0:
            return
        print(f"[bench] no gfx950 device visible yet (probe exit code {rc}): asking again in 5 s (attempt {attempt + 1} of 12)", file=sys.stderr)
        time.sleep(5)


def main():
    if int(os.environ.get("RANK", "0")) == 0 or "--driver" in sys.argv:   # (under torch.distributed.run only rank 0 drives the devices, the others wait at its barrier -- unless `--driver torch` gives every rank its own)
        _device_or_retry()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU")
    ap.add_argument("--batch-reads", type=int, default=10_000_000, help="N > 1: reads per exchange round and rank (the job's input is dealt to the ranks in chunks of this size)")
    ap.add_argument("--bf-shift", type=int, default=37)
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "nofilter", "cfg3shard", "cfg4", "cfg5"],
                    help="BASELINE.json configuration: cfg2 = configs[1] (default, the metric's workload); nofilter = same reads, no bloom filter, one pass; "
                         "cfg4 = configs[3], yak count -k21 on a synthetic assembly (long contigs, singletons kept); cfg5 = configs[4], lookup-only path of yak qv")
    ap.add_argument("--of", type=int, default=8, help="cfg3shard: GPUs of the job whose rank 0 is measured on this one GPU (BASELINE configs[2]: 8)")
    ap.add_argument("--rank", type=int, default=0, help="cfg3shard: which of the --of ranks (it owns sub-tables [rank * 1024 / of, (rank + 1) * 1024 / of); all eight ranks of the 600 M-read job are pinned on the oracle: tests/golden/cfg3_full.json)")
    ap.add_argument("--contigs", type=int, default=50, help="cfg4: number of contigs")
    ap.add_argument("--contig-len", type=int, default=100_000_000, help="cfg4: bases per contig")
    ap.add_argument("--sweeps", type=int, default=1, help="cfg4: > 1 = count through yak_count() in that many sweeps over prefix ranges (sizes beyond one pass: --contigs 50 --sweeps 8 = 5 Gb)")
    ap.add_argument("--qv-reads", type=int, default=75_000, help="cfg5: number of 20 kb query reads")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"], help="N > 1: reads per GPU fixed (weak) or --total-reads fixed (strong)")
    ap.add_argument("--total-reads", type=int, default=0, help="strong scaling: reads of the whole job (BASELINE configs[2]: 600000000)")
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive and CLI end-to-end side measurements")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-weak-base", action="store_true", help="N > 1: skip the one-device run of the per-GPU workload (`weak_base`)")
    ap.add_argument("--knob", action="append", default=[], metavar="NAME=VALUE", help="a test switch of the library (yakamd_test_set); tests force code paths with it")
    ap.add_argument("--force-exchange", action="store_true", help="run the sharded (all-to-all) data path even on 1 GPU")
    ap.add_argument("--exchange16", action="store_true", help="N > 1: exchange 16-byte {hash, position} records instead of 8-byte tagged ones")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: do not queue the pass-2 exchange behind pass 1's computation")
    ap.add_argument("--job-md5", action="store_true", help="also report the md5 of the whole job's .yak bytes (sub-tables gathered from all ranks)")
    ap.add_argument("--no-qv", action="store_true", help="skip the lookup-kernel side measurement")
    ap.add_argument("--no-packed", action="store_true", help="skip the packed-image (0.375 B/base) side measurement")
    ap.add_argument("--no-nofilter", action="store_true", help="skip the unfiltered-protocol side measurement (roofline.no_bloom_step_frac)")
    ap.add_argument("--no-retain", action="store_true", help="pass 2 extracts and hashes the input again instead of counting the records pass 1 retained")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend: nccl (= RCCL, default); gloo lets several ranks share one GPU for testing")
    ap.add_argument("--driver", default="c", choices=["c", "torch"], help="N > 1: c = the library's own multi-GPU driver (yakamd_count_multi_dev: one process, RCCL inside the library; default); "
                                                                          "torch = one process per GPU, yak_amd/shard.py + torch.distributed all-to-all")
    a = ap.parse_args()
    a.reads_given = any(x == "--reads" or x.startswith("--reads=") for x in sys.argv[1:])
    a.warmup_given = any(x == "--warmup" or x.startswith("--warmup=") for x in sys.argv[1:])
    a.batch_reads_given = any(x == "--batch-reads" or x.startswith("--batch-reads=") for x in sys.argv[1:])
    a.bf_shift_given = any(x == "--bf-shift" or x.startswith("--bf-shift=") for x in sys.argv[1:])
    maybe_spawn(a)
    if a.knob:                                                    # the library is loaded once per process: one place serves every mode below
        if a.config in ("cfg3shard", "cfg4", "cfg5") or a.gpus > 1:
            import torch                                          # (these modes hold their inputs in torch tensors: torch finds no device if the library's HIP runtime is up first)
        import yak_amd
        for kv in a.knob:
            name, _, val = kv.partition("=")
            yak_amd.lib().yakamd_test_set(name.encode(), int(val))
    if a.gpus > 1 and a.driver == "c" and a.config == "cfg2" and not a.force_exchange:
        return run_multi_c(a)
    if a.config == "nofilter":
        a.bf_shift = 0
    if a.config in ("cfg3shard", "cfg4", "cfg5"):
        import bench_configs
        try:
            return bench_configs.run(a)
        finally:
            if os.environ.get("YAKAMD_VERBOSE"):
                import yak_amd as _y
                _y.lib().yakamd_pool_report(b"the bench")
    if a.scaling == "strong":
        if a.total_reads <= 0:
            raise SystemExit("--scaling strong needs --total-reads")
        a.reads = a.total_reads // max(1, a.gpus)

    import yak_amd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    sharded = world > 1 or a.force_exchange
    torch = dist = dev = None                       # PyTorch only serves the torch driver of the N > 1 mode (device buffers of the all-to-all, torch.distributed)
    if sharded:
        import torch
        import torch.distributed as dist
        local = local % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    if sharded:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)
    L = yak_amd.lib()
    if L.yakamd_device_count() < 1:
        raise SystemExit("no gfx950 device: refusing to run (no CPU fallback)")

    def dev_sync():
        if torch is not None:
            torch.cuda.synchronize()
        elif L.yakamd_device_sync() != 0:
            raise RuntimeError(yak_amd._err())

    def dev_empty(nbytes):
        return torch.empty(nbytes, dtype=torch.uint8, device=dev) if torch is not None else DevBuf(L, nbytes)

    P = 1 << PRE
    if P % world:
        raise SystemExit("the number of GPUs must divide 1024 sub-tables")
    lo, hi = rank * P // world, (rank + 1) * P // world
    threads = max(1, (os.cpu_count() or 8) // max(1, world))
    genome = 5 * a.reads * world                    # 30x coverage of the whole job
    # The logical input of the whole job is cut into chunks of `batch_reads` reads, dealt round-robin to the
    # ranks: chunk c = b * world + r is batch b of rank r.  A single rank therefore sees the same stream as N
    # ranks do (N-GPU bytes == 1-GPU bytes), and every rank is busy in every round however long the input.
    n_batches = 1 if world == 1 else -(-a.reads // max(1, min(a.reads, a.batch_reads)))
    while a.reads % n_batches:                      # equal chunks, so that the chunks of all ranks tile the job's reads without gaps
        n_batches += 1
    batch_reads = a.reads // n_batches
    rec_len = READ_LEN + 1
    B = batch_reads * rec_len                       # bytes (= stream positions) of a full chunk
    d_reads = dev_empty(a.reads * rec_len)
    h_reads = None
    for b in range(n_batches):
        nb_reads = min(batch_reads, a.reads - b * batch_reads)
        h_reads = make_reads(nb_reads, genome, 42, (b * world + rank) * batch_reads, torch, min(threads, 64), L)
        if torch is not None:
            d_reads[b * B:b * B + nb_reads * rec_len].copy_(h_reads, non_blocking=False)
        elif L.yakamd_memcpy_h2d(d_reads.data_ptr() + b * B, h_reads.data_ptr(), nb_reads * rec_len) != 0:
            raise RuntimeError("h2d")
    dev_sync()
    n_bytes = d_reads.numel()
    batch_span = [(b * B, min(B, n_bytes - b * B)) for b in range(n_batches)]

    # exchange buffers (sharded path only): records grouped by prefix -- 8-byte tagged records where k / pre allow them
    # (yakamd_partition_tagged_dev: the stream order of a prefix's records is implied, no position travels), else 16-byte {hash, position}
    tagged = sharded and L.yakamd_tagged_ok(K, PRE) != 0 and not a.exchange16
    if sharded:
        s_rec = torch.empty(min(B, n_bytes) if tagged else (min(B, n_bytes), 2), dtype=torch.int64, device=dev)
        s_hash = torch.empty(min(B, n_bytes), dtype=torch.int64, device=dev)
        h_bstart = (C.c_uint64 * (P + 1))()

    def exchange(b, create_new, async_op=False):
        """partition this rank's k-mers of batch b by sub-table prefix once, then one all-to-all moves every
        record (pass 2: only its hash) to the owner of its prefix (RCCL over xGMI).  async_op: the
        all-to-all is queued and a function that waits for it is returned."""
        from yak_amd import shard
        off, nb = batch_span[b]
        if create_new:
            n = (L.yakamd_partition_tagged_dev if tagged else L.yakamd_partition_dev)(K, PRE, d_reads.data_ptr() + off, nb, s_rec.data_ptr(), h_bstart)
            send = s_rec[:n]
        else:                                               # counting existing keys only needs the hashes: 8-byte records
            n = L.yakamd_partition_hashes_dev(K, PRE, d_reads.data_ptr() + off, nb, s_hash.data_ptr(), h_bstart)
            send = s_hash[:n]
        if n < 0:
            raise RuntimeError("partition failed")
        out = shard.exchange_partitioned(send, list(h_bstart), P, async_op=async_op)
        if not async_op:
            torch.cuda.synchronize()
        return out

    pending = [None]      # the pass-2 exchange of the step in flight (queued while pass 1 computes)

    packed = [None]       # (codes, valid): the side measurement with the image at 0.375 B per base (yakamd_feed_packed_dev)

    def one_pass(t, create_new):
        if not sharded:
            if packed[0] is not None:
                t.count_pass_packed(create_new, [(packed[0][0].data_ptr(), packed[0][1].data_ptr(), n_bytes, 0)], same_input=not a.no_retain)
            else:
                t.count_pass(create_new, [(d_reads.data_ptr(), n_bytes, 0)], same_input=not a.no_retain)
            return
        if L.yakamd_pass_begin(t.h, create_new) != 0:
            raise RuntimeError("pass_begin")
        held = []
        for b in range(n_batches):
            if create_new:
                segs = exchange(b, 1)
                # both passes read the same input (main.c:53-57), so the second pass's k-mers can travel while
                # the first pass is still counting: queue that all-to-all now, wait for it when pass 2 starts
                if n_batches == 1 and a.bf_shift > 0 and not a.no_overlap:
                    pending[0] = exchange(0, 0, async_op=True)
            else:
                segs = pending[0]() if pending[0] is not None else exchange(b, 0)
                pending[0] = None
                torch.cuda.synchronize()
            held.append(segs)                                   # lent buffers must outlive pass_end
            for src, (rec, offs) in enumerate(segs):            # by source rank = stream order of the job
                if rec.shape[0]:
                    ob = (C.c_uint64 * (P + 1))(*offs)
                    if create_new:
                        if (L.yakamd_feed_partitioned_tagged_dev(t.h, rec.data_ptr(), rec.shape[0], ob, (b * world + src) * B, B, 1) if tagged else
                                L.yakamd_feed_partitioned_lent_dev(t.h, rec.data_ptr(), rec.shape[0], ob, (b * world + src) * B, B)) != 0:
                            raise RuntimeError("feed_partitioned: " + yak_amd._err())
                    elif L.yakamd_count_partitioned_dev(t.h, rec.data_ptr(), rec.shape[0], ob) != 0:
                        raise RuntimeError("count_partitioned: " + yak_amd._err())
        n_ins = L.yakamd_pass_end(t.h)
        if n_ins < 0:
            raise RuntimeError("pass_end: " + yak_amd._err())
        t.h.contents.tot += n_ins
        del held

    wall = {}
    last_stats = [None]

    def step(keep=False):
        def tick(name, t0):
            dev_sync()
            t1 = time.perf_counter()
            wall[name] = (t1 - t0) * 1e3
            return t1
        tp = time.perf_counter()
        t = yak_amd.Table(K, PRE, N_HASH, a.bf_shift)
        if sharded:
            L.yakamd_set_shard(t.h, lo, hi)
        elif a.bf_shift > 0 and not a.no_retain:
            L.yakamd_retain_input(t.h, 1)           # pass 2 reads the same input (main.c:57): its hashed k-mers stay on the device
        tp = tick("init", tp)
        one_pass(t, 1)
        tp = tick("pass1", tp)
        s1 = t.stats()
        if a.bf_shift > 0:
            t.destroy_bf(); t.clear()
            tp = tick("destroy_bf_clear", tp)
            one_pass(t, 0)
            tp = tick("pass2", tp)
            s2 = t.stats()
            t.shrink(2, 1023)
            tp = tick("shrink", tp)
            last_stats[0] = t.stats()
        else:
            s2 = None
        tot = t.tot
        if keep:
            return t, tot, s1, s2
        t.close()
        tick("close", tp)
        return None, tot, s1, s2

    def barrier():
        if sharded:
            dist.barrier()
        dev_sync()

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        _, tot, s1, s2 = step()
    barrier()
    dt = time.perf_counter() - t0
    wall_timed = dict(wall)
    if sharded:
        v = torch.tensor([dt, float(tot), float(s1["n_instances"] + (s2["n_instances"] if s2 else 0))],
                         dtype=torch.float64, device=dev)
        vmax = v.clone(); dist.all_reduce(vmax, op=dist.ReduceOp.MAX)
        vsum = v.clone(); dist.all_reduce(vsum, op=dist.ReduceOp.SUM)
        dt, tot_all, inst_all = vmax[0].item(), vsum[1].item(), vsum[2].item()
    else:
        tot_all, inst_all = float(tot), float(s1["n_instances"] + (s2["n_instances"] if s2 else 0))
    ms_step = dt / a.steps * 1e3

    job_md5 = None
    if a.job_md5:
        # bytes of the whole job's .yak: every rank dumps the sub-tables it owns, rank 0 concatenates them in
        # prefix order behind the 16-byte header (htab.c:373-394) -- must equal the single-rank file
        import struct
        t_m, _, _, _ = step(keep=True)
        data = t_m.dump_bytes(); t_m.close()
        plo_, phi_ = (lo, hi) if sharded else (0, P)
        off, parts = 16, []
        for p_ in range(P):
            n_ = struct.unpack_from("<I", data, off + 4)[0]
            if plo_ <= p_ < phi_:
                parts.append(data[off:off + 8 + 8 * n_])
            off += 8 + 8 * n_
        blob = b"".join(parts)
        if sharded and world > 1:
            gathered = [None] * world
            dist.all_gather_object(gathered, blob)
            blob = b"".join(gathered)
        job_md5 = hashlib.md5(data[:16] + blob).hexdigest()

    verify = None
    if not a.no_verify and not sharded:
        # full-size property: the .yak bytes do not depend on how the stream is cut into device
        # batches (the reference's independence of -K / -t)
        t_a, _, _, _ = step(keep=True)
        md5_a = hashlib.md5(t_a.dump_bytes()).hexdigest(); t_a.close()
        os.environ["YAKAMD_BATCH"] = str(1 << 25)
        t_b, _, _, _ = step(keep=True)
        md5_b = hashlib.md5(t_b.dump_bytes()).hexdigest(); t_b.close()
        del os.environ["YAKAMD_BATCH"]
        verify = {"yak_md5": md5_a, "batch_independent": md5_a == md5_b}
        if md5_a != md5_b:
            raise SystemExit("FAILED: .yak bytes depend on the device batch size")
        # full-size golden: the .yak the REFERENCE wrote for this very workload (tests/gen_golden_full.py)
        try:
            gold = json.load(open(os.path.join(ROOT, "tests", "golden", "cfg2_full.json")))
        except Exception:
            gold = None
        gold = next((g for g in (gold or {}).values() if (g["reads"], g["genome"], g["k"], g["bf_shift"], g["seed"]) == (a.reads, genome, K, a.bf_shift, 42)), None)
        if gold and world == 1:
            verify["reference_md5"] = gold["md5"]
            verify["equals_reference"] = md5_a == gold["md5"]
            if md5_a != gold["md5"]:
                raise SystemExit("FAILED: .yak differs from the reference's for the benchmark workload")
        # small-size gate against the oracle, byte for byte
        import __graft_entry__ as ge
        ge.smoke()

    # the same protocol from the PACKED image (2-bit codes + validity bits, 0.375 B per base; north_star: "packed reads"): a side
    # measurement with its own md5 check -- the default line reads the ASCII image the reference's parser hands over
    packed_probe = None
    if not a.no_packed and not sharded:
        nw = (n_bytes + 31) // 32
        d_codes = dev_empty(4 * (2 * nw + 4))
        d_valid = dev_empty(4 * (nw + 4))
        dev_sync()
        tpk = time.perf_counter()
        if L.yakamd_pack_bases_dev(d_reads.data_ptr(), n_bytes, d_codes.data_ptr(), d_valid.data_ptr(), None) != 0:
            raise RuntimeError("pack: " + yak_amd._err())
        dev_sync()
        pack_ms = (time.perf_counter() - tpk) * 1e3
        packed[0] = (d_codes, d_valid)
        step()
        barrier()
        tpk = time.perf_counter()
        for _ in range(2):
            step()
        barrier()
        pk_ms = (time.perf_counter() - tpk) / 2 * 1e3
        t_p, _, _, _ = step(keep=True)
        md5_p = t_p.dump_md5()[0]; t_p.close()
        packed[0] = None
        packed_probe = {"ms_per_step": pk_ms, "bytes_per_base": 0.375, "pack_kernel_ms": pack_ms, "yak_md5": md5_p,
                        "equals_ascii_run": (md5_p == verify["yak_md5"]) if verify else None}
        if verify and md5_p != verify["yak_md5"]:
            raise SystemExit("FAILED: the packed image gives another .yak than the ASCII image")
        del d_codes, d_valid

    # second kernel family (BASELINE configs[4]): the lookup-only kernel of `yak qv`
    # on the table just built, over the same resident reads -- a side measurement, never `value`
    qv_probe = None
    if not a.no_qv and not sharded:
        t_q, _, _, _ = step(keep=True)
        d_t16 = dev_empty(2 * n_bytes)
        L.yakamd_lookup_dev(t_q.h, d_reads.data_ptr(), n_bytes, d_t16.data_ptr())       # warm-up
        dev_sync()
        tq = time.perf_counter()
        for _ in range(3):
            if L.yakamd_lookup_dev(t_q.h, d_reads.data_ptr(), n_bytes, d_t16.data_ptr()) != 0:
                raise RuntimeError("lookup")
        dev_sync()
        ms_q = (time.perf_counter() - tq) / 3 * 1e3
        import numpy as np
        h_t16 = d_t16.to_numpy(np.uint16)
        n_q = int((h_t16 != 0xffff).sum())
        present = int(((h_t16 != 0xffff) & (h_t16 > 0)).sum())
        del h_t16
        # lookup = 8 algorithmic bytes per k-mer instance (+ the input, reported separately)
        qv_probe = {"kernel": "k_lookup", "kmers_looked_up": n_q, "present": present, "ms": ms_q, "lookups_per_s": n_q / (ms_q * 1e-3),
                    "achieved_GBs": 8.0 * n_q / (ms_q * 1e-3) / 1e9, "frac_of_hbm_peak": 8.0 * n_q / (ms_q * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "algorithmic_bytes_per_instance": 8.0,
                    "note": "bound by random 64-byte-granule HBM reads: the 1 GB table image exceeds the 256 MB Infinity Cache, ~50 G such reads/s measured"}
        t_q.close()
        del d_t16

    # the unfiltered protocol on the same resident reads (main.c:53, one pass, singletons kept): its roofline fraction at SURVEY 8(d)'s
    # 32 B per instance carries no bloom-block credit -- printed beside the default line, never `value`
    nb_probe = None
    if not a.no_nofilter and not sharded and a.bf_shift > 0:
        def nofilter_step():
            t_ = yak_amd.Table(K, PRE, N_HASH, 0)
            t_.count_pass(1, [(d_reads.data_ptr(), n_bytes, 0)])
            st_ = t_.stats()
            tot_ = t_.tot
            t_.close()
            return tot_, st_
        nofilter_step()
        barrier()
        tq = time.perf_counter()
        for _ in range(2):
            tot_nb, st_nb = nofilter_step()
        barrier()
        ms_nb = (time.perf_counter() - tq) / 2 * 1e3
        nb_probe = {"ms_per_step": ms_nb, "distinct": tot_nb, "instances": st_nb["n_instances"], "bytes_per_instance": 32.0,
                    "step_frac": 32.0 * st_nb["n_instances"] / (ms_nb * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "phase_ms": {k: round(v, 3) for k, v in st_nb.items() if k.startswith("ms_") and k != "ms_bloom"}}

    # the rate with the base image handed over from HOST memory in both passes (packed by the host: yakamd_feed_packed_host, what
    # yak_count() does after parsing) and the whole `yak-amd count` command on the same reads as a FASTQ file
    # (process start, parsing, PCIe, counting, writing the .yak) -- side figures, never `value`
    pcie_ms = e2e = host_pack = None
    if not a.no_pcie and not sharded:
        # the parser's share, outside the timed region like the parsing itself: the image packed to 0.375 B per base (yakamd_pack_bases_host,
        # what yak_count()'s parser threads do with what they parsed) in page-locked memory
        h_packed = HostBuf(L, L.yakamd_packed_bytes(n_bytes))
        tpk = time.perf_counter()
        L.yakamd_pack_bases_host(h_reads.data_ptr(), n_bytes, h_packed.data_ptr())
        host_pack = {"bytes_over_the_bus": L.yakamd_packed_bytes(n_bytes), "ascii_bytes": n_bytes, "one_thread_gb_per_s": n_bytes / (time.perf_counter() - tpk) / 1e9}

        def host_protocol():
            t = yak_amd.Table(K, PRE, N_HASH, a.bf_shift)
            if a.bf_shift > 0 and not a.no_retain:
                L.yakamd_retain_input(t.h, 1)                       # what yak_count() does for a filtered count of one file: pass 2 crosses the bus only if nothing was kept
            for create_new in ((1, 0) if a.bf_shift > 0 else (1,)):
                if L.yakamd_pass_begin(t.h, create_new) != 0:
                    raise RuntimeError(yak_amd._err())
                kept = L.yakamd_count_retained(t.h) if (not create_new and not a.no_retain) else 1
                if kept < 0 or (kept and L.yakamd_feed_packed_host(t.h, h_packed.data_ptr(), n_bytes, 0) != 0):
                    raise RuntimeError(yak_amd._err())
                n_ins = L.yakamd_pass_end(t.h)
                t.h.contents.tot += n_ins
                if create_new and a.bf_shift > 0:
                    t.destroy_bf(); t.clear()
            if a.bf_shift > 0:
                t.shrink(2, 1023)
            dev_sync()
            tot_ = t.tot
            t.close()
            return tot_
        host_protocol()
        tq = time.perf_counter()
        for _ in range(2):
            tot_h = host_protocol()
        pcie_ms = (time.perf_counter() - tq) / 2 * 1e3
        if tot_h != tot:
            raise SystemExit("FAILED: host-fed run counted a different table")
        cli = os.path.join(ROOT, "yak_amd", "yak-amd")
        if os.path.exists(cli) and a.reads <= 12_000_000:
            tmp = tempfile.mkdtemp(prefix="yke", dir=os.environ.get("YAKAMD_TMP", None))
            try:
                fq = os.path.join(tmp, "r.fq")
                subprocess.check_call([os.path.join(ROOT, "tools", "yaksynth"), "-n", str(a.reads), "-l", str(READ_LEN), "-g", str(genome), "-s", "42",
                                       "-t", str(min(threads, 32)), "-o", fq])
                cmd = [cli, "count", f"-k{K}", f"-t{min(threads, 32)}", "-o", os.path.join(tmp, "o.yak")] + ([f"-b{a.bf_shift}"] if a.bf_shift else []) + [fq]
                # A process that starts right after another one released tens of GB of HBM waits for the driver to hand those pages
                # out again (measured: +1.5 s before its first batch).  So this process gives its cached device memory back first and
                # the device gets a few idle seconds before each run; both runs are reported, `ms` is the better one
                L.yakamd_trim()
                runs = []
                for _ in range(2):
                    time.sleep(8.0)
                    tq = time.perf_counter()
                    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
                    runs.append((time.perf_counter() - tq) * 1e3)
                e2e = {"ms": min(runs), "ms_runs": runs, "command": " ".join(os.path.basename(x) if os.sep in x else x for x in cmd),
                       "fastq_bytes": os.path.getsize(fq), "yak_md5": hashlib.md5(open(os.path.join(tmp, "o.yak"), "rb").read()).hexdigest()}
                if verify and e2e["yak_md5"] != verify["yak_md5"]:
                    raise SystemExit("FAILED: the CLI's .yak differs from the device-resident run's")
            finally:
                subprocess.call(["rm", "-rf", tmp])

    if rank != 0:
        dist.destroy_process_group()
        return

    dbgc = (C.c_uint32 * 4)()
    L.yakamd_debug_counters(dbgc)
    # per-kernel view of the last step (HIP events on the engine's stream, inside the library); the
    # dominant kernel is the one with the largest accumulated time.  Algorithmic bytes per unit
    # (DESIGN.md section 4): partition sweeps 8 B/instance, insert/count 24 B/instance, pass-2 lookup
    # 16 + 8 f_hit, layout replay 16 B per key placed.
    n1, n2 = s1["n_instances"], (s2["n_instances"] if s2 else 0)
    # no-bloom count = 32 B/instance, of which the insert kernel reads the bucket back and reads +
    # writes a slot (24); bloom mode pass 1 = 16 + 128 (the 64-byte bloom block read + written per instance,
    # bbf.c:25-42) + 16 f_ins, of which the insert kernel's share is 8 + 128 + 16 f_ins (f_ins measured)
    f_ins = s1["n_new_keys"] / max(1, n1)
    b_insert = (8.0 + 128.0 + 16.0 * f_ins) if (a.bf_shift > PRE and s1["ms_part2"] > 0) else B_INSERT
    f_hit = (qv_probe["present"] / max(1, qv_probe["kmers_looked_up"])) if qv_probe else 0.9
    dev_batch = int(os.environ.get("YAKAMD_BATCH", 1 << 31))
    n_batches = -(-n_bytes // dev_batch)
    p2_extracted = bool(s2) and s2["ms_extract"] > 0          # False: pass 2 counted the records pass 1 retained (no second extraction)
    kern = [
        {"kernel": "k_xpart (extract + level-1 partition" + (", both passes)" if p2_extracted else ", pass 1; pass 2 reads its records)"),
         "ms": s1["ms_extract"] - s1["ms_part2"] + (s2["ms_extract"] if s2 else 0),
         "launches": 2 * n_batches * (2 if p2_extracted else 1), "bytes": 8.0 * (n1 + (n2 if p2_extracted else 0))},   # histogram + scatter per device batch
        {"kernel": "k_part2 (level-2 partition)", "ms": s1["ms_part2"], "launches": 2, "bytes": 8.0 * n1},
        {"kernel": "k_lc2 (insert + bloom gate)" if s1["ms_part2"] > 0 else "k_acc_insert", "ms": s1["ms_insert"],
         "launches": max(1, s1["n_dominant_launches"]), "bytes": b_insert * n1, "bytes_no_bloom_model": B_INSERT * n1},
    ]
    if s2:
        st_after = last_stats[0]
        kern.append({"kernel": "k_cnt2 + k_img_count_own (pass 2: k_cnt2 and k_cnt2_apply on the records pass 1 retained by sub-bucket, else k_img_count_own)", "ms": s2["ms_insert"], "launches": max(1, s2["n_dominant_launches"]),
                     "bytes": (B_LOOKUP + 8.0 * f_hit) * n2})
        kern.append({"kernel": "k_r2_* + k_replay (exact khashl layout: pass 1 + shrink)", "ms": s1["ms_replay"] + st_after["ms_shrink"],
                     "launches": 2, "bytes": 16.0 * (s1["n_new_keys"] + tot_all / max(1, world))})
    else:
        kern.append({"kernel": "k_r2_* + k_replay (exact khashl layout)", "ms": s1["ms_replay"], "launches": 1, "bytes": 16.0 * s1["n_new_keys"]})
    for k_ in kern:
        k_["avg_launch_ms"] = k_["ms"] / k_["launches"]
        k_["achieved_GBs"] = k_["bytes"] / (k_["ms"] * 1e-3) / 1e9 if k_["ms"] > 0 else 0.0
        k_["frac"] = k_["achieved_GBs"] / HBM_PEAK_GBS
    # HBM traffic per launch from the committed rocprofv3 PMC passes of this same command
    # (profiles/r01k_pmc_traffic.json: FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes, + WRITE_SIZE)
    pmc, pmc_src = {}, None
    default_cfg = world == 1 and not a.no_retain and ((a.reads == 10_000_000 and a.bf_shift in (37, 0)) or (a.reads == 30_000_000 and a.bf_shift == 37))     # (the counter passes were taken on these commands, no other)
    for cand_ in (("r06_pmc_traffic_30m.json", "r05_pmc_traffic_30m.json") if a.reads == 30_000_000 else ("r06_pmc_traffic_nofilter.json", "r05_pmc_traffic_nofilter.json", "r04_pmc_traffic_nofilter.json") if a.bf_shift == 0 else
                  ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json")):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", cand_)))
            pmc_src = "profiles/" + cand_
            break
        except Exception:
            pass
    # the counter bytes belong to the device code they were measured on: another tree -> no traffic figure (a stale constant would say nothing about this build)
    k_now = yak_amd.kernels_sha16()
    k_on = (pmc.get("_measured_on") or {}).get("kernels_sha16") if pmc else None
    traffic_note = None
    if pmc and k_on != k_now:
        traffic_note = f"{pmc_src} was measured on kernels {k_on}, this tree builds {k_now}: traffic is null until the counter passes are taken again (tests/tools/final_r06.sh)"
        pmc = {}
    for k_ in kern:
        keys_ = [x for x in k_["kernel"].split(" (")[0].replace("*", "").split(" + ")]
        cand = [v for n_, v in pmc.items() if any(n_.startswith(key) for key in keys_) and isinstance(v, dict) and "launches" in v] if default_cfg else []
        steps_pmc = 1 if a.reads == 30_000_000 else max(1, sum(v["launches"] for n_, v in pmc.items() if n_.startswith("k_lc2") and isinstance(v, dict) and "launches" in v))     # k_lc2 (any template variant) runs once per step: the steps of the profiled command
        k_["traffic_bytes"] = sum(v["launches"] / steps_pmc * (v["fetch_bytes_per_launch_x2_corrected"] + v["write_bytes_per_launch"]) for v in cand) if cand else None
        # how busy the HBM really is while this kernel (group) runs: counter bytes / its time / peak
        k_["hbm_util"] = (k_["traffic_bytes"] / (k_["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if (k_["traffic_bytes"] and k_["ms"] > 0) else None
    step_traffic = None
    if default_cfg and pmc:                                       # every kernel of one protocol step, whatever its name
        steps_pmc = 1 if a.reads == 30_000_000 else max(1, sum(v["launches"] for n_, v in pmc.items() if n_.startswith("k_lc2") and isinstance(v, dict) and "launches" in v))
        step_traffic = sum(v["launches"] / steps_pmc * (v["fetch_bytes_per_launch_x2_corrected"] + v["write_bytes_per_launch"])
                           for v in pmc.values() if isinstance(v, dict) and "launches" in v)
    # the pass and the step as a whole against the same roof: SURVEY 8(d)'s algorithmic bytes of every instance
    # the pass consumed / its wall-clock time (pass 1 with and without the exact-layout tail: sort + replay)
    bloom_on = a.bf_shift > PRE and s2 is not None
    b_pass1 = (16.0 + 128.0 + 16.0 * f_ins) * n1 if bloom_on else 32.0 * n1
    b_pass2 = (16.0 + 8.0 + 8.0 * f_hit) * n2
    w = wall_timed
    ms_p1 = w.get("pass1", 0.0)
    frac = lambda by, ms: by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else None
    b_step = b_pass1 + b_pass2
    b_step_nc = ((16.0 + 16.0 * f_ins) * n1 if bloom_on else 32.0 * n1) + b_pass2       # the same step without the 128 B per instance of bloom-block credit
    step_gbs = b_step / (ms_step * 1e-3) / 1e9 if world == 1 else None
    # the dominant KERNEL: the single kernel with the largest time (the extraction and replay rows are groups of several kernels)
    dom = max([k_ for k_ in kern if not k_["kernel"].startswith(("k_xpart", "k_r2_"))], key=lambda x: x["ms"])
    name, avg_ms, launches, ach = dom["kernel"], dom["avg_launch_ms"], dom["launches"], dom["achieved_GBs"]
    out = {
        "metric": (f"distinct k-mers counted/sec (k=31), yak count -b{a.bf_shift} two-pass protocol, .yak bit-exact" if bloom_on or a.bf_shift > 0 else
                   "distinct k-mers counted/sec (k=31), yak count without a filter (one pass, singletons kept), .yak bit-exact"),
        "value": tot_all / (dt / a.steps), "unit": "distinct k-mers/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step,
        "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": f"yak count -k{K}" + (f" -b{a.bf_shift}" if a.bf_shift > 0 else "") + f" on {a.reads} x {READ_LEN} bp synthetic reads per GPU "
                               f"(G={genome}, e=0.5%, N=0.05%), 30x, " + ("bloom prefilter on, both passes + shrink" if a.bf_shift > 0 else "no filter: one pass, every k-mer kept"),
                   "reads_per_gpu": a.reads, "k": K, "pre": PRE, "bf_shift": a.bf_shift,
                   "sharding": "prefix-sharded sub-tables, RCCL all-to-all of hashed k-mers" if sharded else "1 GPU"},
        "kmer_instances_per_s": inst_all / (dt / a.steps),
        "final_distinct": tot_all,
        "phase_ms_last_step": {"pass1": {k: round(v, 3) for k, v in s1.items() if k.startswith("ms_") and k != "ms_bloom"},
                               "pass2": {k: round(v, 3) for k, v in s2.items() if k.startswith("ms_") and k != "ms_bloom"} if s2 else None},
        "phase_wall_ms_last_step": {k: round(v, 2) for k, v in wall_timed.items()},
        "pass1_distinct_seen": s1["n_distinct_seen"], "pass1_table_keys": s1["n_new_keys"],
        "bloom_exact_resolutions": s1["n_bloom_candidates"],
        # the roofline line is the WHOLE protocol step: SURVEY 8(d)'s algorithmic bytes of every instance both passes consumed / the step's
        # wall-clock time (every kernel and every host gap in the denominator).  The dominant kernel keeps its own entry for the rocprof cross-check
        "roofline": {"bound": "hbm", "kernel": "whole protocol step (init, pass 1 incl. exact layout, clear, pass 2, shrink)",
                     "achieved": step_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": (step_gbs / HBM_PEAK_GBS) if step_gbs else None,
                     "traffic": step_traffic,
                     "hbm_util": (step_traffic / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS) if step_traffic else None,
                     "traffic_source": (pmc_src + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, summed over the kernels of one step; FETCH_SIZE x2 as MI355X_MICROARCH.md prescribes)") if step_traffic else traffic_note,
                     "kernels_sha16": k_now, "traffic_measured_on_kernels_sha16": k_on,
                     "algorithmic_bytes_per_step": b_step,
                     "model": "SURVEY 8(d): bloom mode pass 1 = 16 + 128 + 16 f_ins bytes per instance (the reference touches a 64-byte bloom block per instance; "
                              "this design stages every bloom range once in LDS instead, so the 128 is a credit, not traffic), pass 2 = 16 + 8 + 8 f_hit; "
                              "no_bloom_step_frac prices the one-pass unfiltered protocol on the same reads at 32 B per instance",
                     "f_ins": f_ins, "f_hit": f_hit,
                     "pass1_frac": frac(b_pass1, ms_p1),
                     "pass2_frac": frac(b_pass2, w.get("pass2", 0.0)) if s2 else None,
                     "step_frac": frac(b_step, ms_step if world == 1 else 0.0),
                     "no_credit_step_frac": frac(b_step_nc, ms_step if world == 1 else 0.0),
                     "no_credit_note": "the step priced without the model's 128 B per instance for the reference's bloom-block read-modify-write (this design stages each bloom range once in LDS): "
                                       "pass 1 = 16 + 16 f_ins, pass 2 unchanged.  frac (credit in), hbm_util (counter bytes) and no_bloom_step_frac (the unfiltered protocol at 32 B) stand beside it",
                     "no_bloom_step_frac": nb_probe["step_frac"] if nb_probe else None,
                     "no_bloom_step": nb_probe,
                     "pass_bytes": {"pass1": b_pass1, "pass2": b_pass2, "per_instance_pass1": b_pass1 / max(1, n1), "per_instance_pass2": (b_pass2 / n2) if n2 else None},
                     "pass_ms": {"pass1": ms_p1, "pass2": w.get("pass2"), "step": ms_step},
                     "pass2_frac_note": ("pass 2 ran on the level-2 records pass 1 retained: of the 16 B per instance the model counts for writing a prefix bucket and reading it back, "
                                         "it only reads the 8 (the write was pass 1's); the counter bytes of k_cnt2 + k_cnt2_apply are in all_kernels") if (s2 and not p2_extracted) else None,
                     "pass2_input": "records retained by pass 1 (same input: main.c:57; sub-bucket records + key lists when pass 1 was one slice, else prefix-grouped records)" if (s2 and not p2_extracted) else "extracted again",
                     "dominant_kernel": {"kernel": name, "avg_launch_ms": avg_ms, "launches": launches,
                                         "algorithmic_bytes_per_launch": dom["bytes"] / launches, "algorithmic_bytes_per_instance": dom["bytes"] / max(1, n1),
                                         "achieved": ach, "frac": ach / HBM_PEAK_GBS,
                                         "achieved_no_bloom_model": (dom["bytes_no_bloom_model"] / (dom["ms"] * 1e-3) / 1e9) if dom.get("bytes_no_bloom_model") else None,
                                         "traffic": (dom["traffic_bytes"] / launches) if dom.get("traffic_bytes") else None,
                                         "hbm_util": dom.get("hbm_util"),
                                         "note": ("frac > 1 is the model's credit, not bandwidth: SURVEY 8(d) prices a 64-byte bloom block read + written per instance (128 B), "
                                                  "this kernel stages each bloom range once in LDS, so those bytes never reach the HBM; hbm_util (counter bytes / time / peak) "
                                                  "and achieved_no_bloom_model (32 B per instance) are the physical figures") if ach / HBM_PEAK_GBS > 1.0 else None},
                     "all_kernels": [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in kk.items()} for kk in kern]},
        "verify": verify,
        "job_yak_md5": job_md5,
        "qv_lookup_probe": qv_probe,
        "packed_input": packed_probe, "pcie_inclusive_ms": pcie_ms, "pcie_inclusive_host_pack": host_pack, "pcie_inclusive_value": (tot_all / (pcie_ms * 1e-3)) if pcie_ms else None,
        "e2e_cli": e2e,
        "replay_doublings_parallel_vs_serial_fallback": list(dbgc),
    }
    if not a.no_cpu_baseline and world == 1:                      # rank 0 at N = 1 only
        out["cpu_baseline"] = cpu_baseline(a.reads, genome, a.bf_shift, min(os.cpu_count() or 8, 32))
    print(json.dumps(out))
    if os.environ.get("YAKAMD_VERBOSE"):
        yak_amd.lib().yakamd_pool_report(b"the bench")
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
