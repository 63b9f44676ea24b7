"""bench.py --config cfg4 | cfg5: the two BASELINE.json configurations that are not the read-counting protocol.

cfg4 = configs[3]: `yak count -k21` on a synthetic assembly FASTA (long contigs, no filter, singletons
kept; reference count.c:28-43,120-125 + khashl.h:152-221 on multi-million-slot sub-tables).  A step is
one whole pass (init -> count -> exact layout) over a base image resident in HBM.
cfg5 = configs[4]: the lookup-only path of `yak qv -p` (reference qv.c:34-135): 20 kb reads with 0.2 %
errors looked up in the GPU-resident table of the cfg2 reads.  A step is one pass of k_lookup +
k_qv_reduce over the resident reads.
Same one-line JSON contract as bench.py (rank 0, N = 1: these paths do not shard).
"""
import ctypes as C
import hashlib
import json
import os
import struct
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0
PRE = 10


def _synth():
    L = C.CDLL(os.path.join(ROOT, "tools", "libyaksynth.so"))
    L.yaksynth_reads.restype = C.c_int64
    L.yaksynth_reads.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_uint64, C.c_double, C.c_double, C.c_int64, C.c_int]
    L.yaksynth_tiles.restype = C.c_int64
    L.yaksynth_tiles.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_uint64, C.c_int]
    return L


def _gold(name):
    try:
        return json.load(open(os.path.join(ROOT, "tests", "golden", "cfg45_full.json"))).get(name)
    except Exception:
        return None


def _timed(torch, steps, warmup, fn):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, r


def run_cfg4_sweeps(a, yak_amd):
    """cfg4 beyond what one pass holds in HBM (5 Gb: 68 GB of table, twice that while a doubling is replayed, 80 GB of selected
    keys): `yak_count()` itself with YAKAMD_GPUS = N ranks on ONE device -- the pass in N sweeps over prefix ranges (every rank owns
    P / N sub-tables; ranks that share a device finish their passes one after the other).  File-inclusive: the assembly is written
    as FASTA and parsed by the library's reader.  Properties only (no reference golden exists at this size: the reference needs
    more host memory than the build container has)."""
    K = 21
    L = yak_amd.lib()
    threads = min(os.cpu_count() or 8, 32)
    tmp = tempfile.mkdtemp(prefix="ykc4", dir=os.environ.get("YAKAMD_TMP", None))
    fa = os.path.join(tmp, "asm.fa")
    try:
        subprocess.check_call([os.path.join(ROOT, "tools", "yaksynth"), "-T", "-n", str(a.contigs), "-l", str(a.contig_len), "-s", "42", "-w", "60", "-t", str(threads), "-o", fa])
        inst = a.contigs * (a.contig_len - K + 1)
        os.environ["YAKAMD_GPUS"] = str(a.sweeps); os.environ["YAKAMD_GPU_LIST"] = ",".join(["0"] * a.sweeps)
        res = []
        for chunk in ((1 << 28), (1 << 27)):                    # two chunkings of the stream: the result must not depend on it
            os.environ["YAKAMD_MGPU_CHUNK"] = str(chunk)
            o = yak_amd.CoptT(); L.yak_copt_init(C.byref(o)); o.k, o.n_thread = K, threads
            t0 = time.perf_counter()
            h = L.yak_count(fa.encode(), C.byref(o), None)
            dt = time.perf_counter() - t0
            if not h:
                raise SystemExit("FAILED: yak_count: " + yak_amd._err())
            hist = (C.c_int64 * 1024)()
            L.yak_ch_hist(h, hist, 1)
            tot = h.contents.tot
            res.append((dt, tot, list(hist)))
            L.yak_ch_destroy(h)
            L.yakamd_trim()
        for k_ in ("YAKAMD_GPUS", "YAKAMD_GPU_LIST", "YAKAMD_MGPU_CHUNK"):
            del os.environ[k_]
    finally:
        subprocess.call(["rm", "-rf", tmp])
    dt, tot, hist = res[0]
    verify = {"count_mass_equals_instances": sum(c * hist[c] for c in range(1024)) == inst and hist[1023] == 0, "sum_hist_equals_tot": sum(hist) == tot,
              "chunking_independent": res[0][1:] == res[1][1:], "yak_size_bytes": 16 + 8 * (1 << PRE) + 8 * tot, "distinct": tot}
    if not all(v for v in verify.values() if isinstance(v, bool)):
        raise SystemExit(f"FAILED: {verify}")
    by = 32.0 * inst
    return {"metric": "distinct k-mers counted/sec (k=21), yak count on an assembly (no filter, singletons kept), file-inclusive, pass in sweeps over prefix ranges",
            "value": tot / dt, "unit": "distinct k-mers/s", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": dt * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"yak count -k{K} -t{threads} on a synthetic assembly FASTA: {a.contigs} contigs x {a.contig_len} bp (tools/yaksynth -T, seed 42), no filter, "
                                   f"yak_count() in {a.sweeps} sweeps over prefix ranges on one device (YAKAMD_GPUS={a.sweeps}, YAKAMD_GPU_LIST=0,...)", "k": K, "pre": PRE, "bf_shift": 0},
            "kmer_instances_per_s": inst / dt, "final_distinct": tot, "seconds_second_chunking": res[1][0],
            "roofline": {"bound": "hbm", "kernel": "whole yak_count() call (parse + sweeps)", "achieved": by / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": by / dt / 1e9 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_instance": 32.0},
            "verify": verify}


def run_cfg4(a, torch, yak_amd):
    if a.sweeps == 1 and a.contigs * a.contig_len > 2_500_000_000:
        a.sweeps = 8                                           # beyond one pass's memory (the default 50 x 100 Mb = BASELINE configs[3])
    if a.sweeps > 1:
        return run_cfg4_sweeps(a, yak_amd)
    K = 21
    L = yak_amd.lib()
    threads = min(os.cpu_count() or 8, 64)
    n_img = a.contigs * (a.contig_len + 1)
    h = torch.empty(n_img, dtype=torch.uint8)
    _synth().yaksynth_tiles(h.data_ptr(), a.contigs, a.contig_len, 42, threads)
    d = h.to("cuda:0")
    del h
    torch.cuda.synchronize()
    inst = a.contigs * (a.contig_len - K + 1)
    stats = {}

    def step(keep=False):
        t = yak_amd.Table(K, PRE, 0, 0)
        t.count_pass(1, [(d.data_ptr(), n_img, 0)])
        stats.update(t.stats())
        tot = t.tot
        if keep:
            return t
        t.close()
        return tot

    dt, tot = _timed(torch, a.steps, a.warmup, step)
    # size-independent properties: .yak size = 16 + 8 P + 8 D; every count adds up to the instances
    # consumed (yak_ch_hist, no saturation at this coverage); khashl's load rule per sub-table
    verify = {}
    if not a.no_verify:
        t = step(keep=True)
        hist = (C.c_int64 * 1024)()
        L.yak_ch_hist(t.h, hist, 1)
        caps = [t.subtable(p) for p in range(1 << PRE)]
        verify = {"sum_sizes_equals_tot": sum(s for _, s in caps) == t.tot,
                  "count_mass_equals_instances": sum(c * hist[c] for c in range(1024)) == inst and hist[1023] == 0,
                  "load_rule": all((cap == 0 and s == 0) or (cap >= 4 and cap & (cap - 1) == 0 and s <= (cap >> 1) + (cap >> 2)) for cap, s in caps),
                  "yak_size_bytes": 16 + 8 * (1 << PRE) + 8 * t.tot, "distinct": t.tot}
        g = _gold(f"cfg4_{a.contigs}x{a.contig_len}")
        if g:                                                  # the reference's own .yak for this very input (tests/gen_golden_full.py --cfg4)
            md5, nbytes = t.dump_md5()
            verify.update(yak_md5=md5, reference_md5=g["md5"], equals_reference=md5 == g["md5"] and nbytes == g["size"])
            if md5 != g["md5"]:
                raise SystemExit("FAILED: .yak differs from the reference's")
        t.close()
        if not all(v for k, v in verify.items() if isinstance(v, bool)):
            raise SystemExit(f"FAILED: {verify}")
    by = 32.0 * inst                                           # no-filter count = 32 B per instance
    out = {"metric": "distinct k-mers counted/sec (k=21), yak count on an assembly (no filter, singletons kept), .yak layout exact",
           "value": tot / dt, "unit": "distinct k-mers/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
           "config": {"workload": f"yak count -k{K} on a synthetic assembly: {a.contigs} contigs x {a.contig_len} bp tiling a random genome (tools/yaksynth -T, seed 42), "
                                  "no filter, one pass, base image resident in HBM", "k": K, "pre": PRE, "bf_shift": 0},
           "kmer_instances_per_s": inst / dt, "final_distinct": tot,
           "phase_ms_last_step": {k: round(v, 3) for k, v in stats.items() if k.startswith("ms_")},
           "roofline": {"bound": "hbm", "kernel": "whole pass (extract + partition + insert + exact layout)", "achieved": by / dt / 1e9, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": by / dt / 1e9 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_instance": 32.0,
                        "extract_insert_frac": by / ((stats["ms_extract"] + stats["ms_insert"]) * 1e-3) / 1e9 / HBM_PEAK_GBS},
           "verify": verify}
    return out


def run_cfg5(a, torch, yak_amd):
    import bench
    K, N_HASH, BF = 31, 4, 37
    L = yak_amd.lib()
    threads = min(os.cpu_count() or 8, 64)
    genome = 5 * a.reads
    h = bench.make_reads(a.reads, genome, 42, 0, torch, threads)
    d = h.to("cuda:0")
    t = yak_amd.Table(K, PRE, N_HASH, BF)
    t.count_pass(1, [(d.data_ptr(), d.numel(), 0)])
    t.destroy_bf(); t.clear()
    t.count_pass(0, [(d.data_ptr(), d.numel(), 0)])
    t.shrink(2, 1023)
    del d, h
    QL = 20000
    nq = a.qv_reads
    hq = torch.empty(nq * (QL + 1), dtype=torch.uint8)
    _synth().yaksynth_reads(hq.data_ptr(), nq, QL, genome, 42, 0.002, 0.0, 0, threads)
    dq = hq.to("cuda:0")
    nb = dq.numel()
    d_t = torch.empty(nb, dtype=torch.int16, device="cuda:0")
    d_off = (torch.arange(nq, dtype=torch.int64, device="cuda:0") * (QL + 1)).contiguous()
    d_len = torch.full((nq,), QL, dtype=torch.int32, device="cuda:0")
    d_tot = torch.empty(nq, dtype=torch.int32, device="cuda:0")
    d_non0 = torch.empty(nq, dtype=torch.int32, device="cuda:0")
    d_hist = torch.zeros(1024, dtype=torch.int64, device="cuda:0")

    def step():
        d_hist.zero_()
        if L.yakamd_lookup_dev(t.h, dq.data_ptr(), nb, d_t.data_ptr()) != 0:
            raise RuntimeError(yak_amd._err())
        if L.yakamd_qv_reduce_dev(t.h, d_t.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), nq, 0, 0.5, d_tot.data_ptr(), d_non0.data_ptr(), d_hist.data_ptr()) != 0:
            raise RuntimeError(yak_amd._err())

    dt, _ = _timed(torch, a.steps, a.warmup, step)
    hist = d_hist.cpu().tolist()
    n_k = nq * (QL - K + 1)
    ct_md5 = hashlib.md5("\n".join(f"{c}\t{v}" for c, v in enumerate(hist) if v).encode()).hexdigest()
    verify = {"kmers": sum(hist), "kmers_expected": n_k, "ct_md5": ct_md5}
    g = _gold(f"cfg5_{a.reads}_{nq}")
    if g:                                                      # `yak qv` of the reference on the same table and reads: its CT lines
        verify.update(reference_ct_md5=g["ct_md5"], equals_reference=ct_md5 == g["ct_md5"])
        if ct_md5 != g["ct_md5"]:
            raise SystemExit("FAILED: CT histogram differs from the reference's")
    if sum(hist) != n_k:
        raise SystemExit("FAILED: k-mers looked up != k-mers in the reads")
    t.close()
    by = 8.0 * n_k                                             # lookup = 8 B per instance
    return {"metric": "k-mers looked up/sec, lookup-only path of yak qv -p (k=31) against the GPU-resident table of the cfg2 reads",
            "value": n_k / dt, "unit": "k-mer lookups/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"yak qv -p -K3.2g: {nq} x {QL} bp reads (e = 0.2 %) of the genome the table's {a.reads} x 150 bp reads come from (yak count -k31 -b37), "
                                   "reads resident in HBM, per-position lookup + per-read reduction + 1024-bin histogram", "k": K, "pre": PRE},
            "roofline": {"bound": "hbm", "kernel": "k_lookup + k_qv_reduce", "achieved": by / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": by / dt / 1e9 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_instance": 8.0},
            "verify": verify}


def run(a):
    import torch
    import yak_amd
    if a.gpus != 1:
        raise SystemExit("--config cfg4 / cfg5 are single-GPU configurations")
    if yak_amd.lib().yakamd_device_count() < 1:
        raise SystemExit("no gfx950 device: refusing to run (no CPU fallback)")
    torch.cuda.set_device(0)
    out = run_cfg4(a, torch, yak_amd) if a.config == "cfg4" else run_cfg5(a, torch, yak_amd)
    print(json.dumps(out))
