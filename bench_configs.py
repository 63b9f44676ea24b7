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


def pmc_step_traffic(name, scale=None, only=None):
    """HBM bytes of ONE step of a configuration from its committed rocprofv3 counter passes (profiles/r04_pmc_traffic_<name>.json: FETCH_SIZE x 2 as
    MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE, of a `--steps 1 --warmup 0 --no-verify` run of that very command): (bytes, source) or (None, None)"""
    d = None
    for rnd in ("r06", "r05", "r04"):                          # the latest round that profiled this configuration
        fn = os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic_{name}.json")
        try:
            d = json.load(open(fn))
            break
        except Exception:
            pass
    if d is None:
        return None, None
    import yak_amd
    on, now = (d.get("_measured_on") or {}).get("kernels_sha16"), yak_amd.kernels_sha16()
    if on != now:                                             # counter bytes of other kernels than the ones that run: not this build's traffic
        return None, f"profiles/{rnd}_pmc_traffic_{name}.json was measured on kernels {on}, this tree builds {now}: no traffic figure until the counter passes are taken again"
    def w(k_):                                                # cfg3shard: the stand-ins for the peers' GPUs partition 7 of 8 chunks on this device
        return next((f for pre, f in (scale or {}).items() if k_.startswith(pre)), 1.0)
    by = sum(w(k_) * v["launches"] * (v["fetch_bytes_per_launch_x2_corrected"] + v["write_bytes_per_launch"]) for k_, v in d.items()
             if isinstance(v, dict) and "launches" in v and (only is None or k_.startswith(only)))
    return by, f"profiles/{rnd}_pmc_traffic_{name}.json (every kernel of one step; FETCH_SIZE x2 + WRITE_SIZE; measured on kernels {on})"


def _roof_traffic(roof, name, seconds, scale=None, only=None):
    by, src = pmc_step_traffic(name, scale, only)
    roof["traffic"] = by
    roof["hbm_util"] = (by / seconds / 1e9 / HBM_PEAK_GBS) if by else None
    roof["traffic_source"] = src
    import yak_amd
    roof["kernels_sha16"] = yak_amd.kernels_sha16()
    return roof


FIRST = {}                                                     # "ms": wall-clock time of the first step of the process (the driver hands out its memory for the first time)


def _timed(torch, steps, warmup, fn):
    for i_ in range(warmup):
        t0 = time.perf_counter()
        fn()
        if i_ == 0:
            torch.cuda.synchronize(); FIRST["ms"] = (time.perf_counter() - t0) * 1e3
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
        own_rule = a.sweeps == 0                               # no --sweeps: yak_count() picks the number itself (yak_multi.cpp auto_sweeps), as it does for a user's `yak count`
        if not own_rule:
            os.environ["YAKAMD_GPUS"] = str(a.sweeps); os.environ["YAKAMD_GPU_LIST"] = ",".join(["0"] * a.sweeps)
        sweeps_seen = []
        res = []
        g = _gold(f"cfg4_{a.contigs}x{a.contig_len}")
        # two chunkings of the stream (the result must not depend on it: the first one is the timed run), then -- where a golden exists for this
        # input -- a third job whose 40 GB dump is hashed (after the timed runs: the dump's host memory pushes the FASTA out of the page cache)
        n_warm = max(0, a.warmup) if a.warmup_given else 0    # --warmup W: W whole jobs first (a device whose memory has never been handed out pays the driver ~27 ms per GB on first use)
        warm_s = []
        for chunk in ((1 << 28),) * n_warm + ((1 << 28), (1 << 27)) + (((1 << 28),) if (g and not a.no_verify) else ()):
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
            sweeps_seen.append(int(L.yakamd_last_sweeps()))
            FIRST.setdefault("ms", dt * 1e3)
            FIRST["peak"] = max(FIRST.get("peak", 0), int(L.yakamd_peak_bytes(0, 0)))
            if len(warm_s) < n_warm:
                warm_s.append(dt); L.yak_ch_destroy(h); continue
            md5 = None
            if len(res) == 2:                                  # the .yak bytes against the golden of this very input
                tm = yak_amd.Table(K, PRE, 0, 0, ptr=h)
                md5, nbytes = tm.dump_md5()
                tm.h = None
                if (md5, nbytes) != (g["md5"], g["size"]):
                    raise SystemExit(f"FAILED: .yak differs from the golden ({md5} / {nbytes} bytes against {g['md5']} / {g['size']})")
            res.append((dt, tot, list(hist), md5))
            L.yak_ch_destroy(h)
        for k_ in ("YAKAMD_GPUS", "YAKAMD_GPU_LIST", "YAKAMD_MGPU_CHUNK"):
            os.environ.pop(k_, None)
        if own_rule:
            a.sweeps = sweeps_seen[n_warm]                     # of the timed job
    finally:
        subprocess.call(["rm", "-rf", tmp])
    dt, tot, hist, _ = res[0]
    md5 = res[2][3] if len(res) > 2 else None
    if len(res) > 2 and res[2][1:3] != res[0][1:3]:
        raise SystemExit("FAILED: the job whose dump was hashed counted something else than the timed one")
    verify = {"count_mass_equals_instances": sum(c * hist[c] for c in range(1024)) == inst and hist[1023] == 0, "sum_hist_equals_tot": sum(hist) == tot,
              "chunking_independent": res[0][1:3] == res[1][1:3], "yak_size_bytes": 16 + 8 * (1 << PRE) + 8 * tot, "distinct": tot}
    if md5:
        verify.update(yak_md5=md5, golden_md5=g["md5"], equals_golden=md5 == g["md5"], golden_produced_by=g.get("produced_by"))
    if not all(v for v in verify.values() if isinstance(v, bool)):
        raise SystemExit(f"FAILED: {verify}")
    by = 32.0 * inst
    # counter bytes of ONE job: the profiled command (`--no-verify`: the two chunkings, no warm-up) runs two
    roof_t = _roof_traffic({}, f"cfg4_{a.contigs}x{a.contig_len}_sweeps{a.sweeps}", dt, scale={"": 0.5})
    return {"metric": "distinct k-mers counted/sec (k=21), yak count on an assembly (no filter, singletons kept), file-inclusive, pass in sweeps over prefix ranges",
            "value": tot / dt, "unit": "distinct k-mers/s", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": dt * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"yak count -k{K} -t{threads} on a synthetic assembly FASTA: {a.contigs} contigs x {a.contig_len} bp (tools/yaksynth -T, seed 42), no filter, "
                                   f"yak_count() in {a.sweeps} sweeps over prefix ranges on one device " + ("(the library's own rule for a process that does not own the device memory of fewer sweeps yet: YAKAMD_COLD_GB)" if own_rule else f"(YAKAMD_GPUS={a.sweeps}, YAKAMD_GPU_LIST=0,...)"), "k": K, "pre": PRE, "bf_shift": 0, "sweeps_of_every_job": sweeps_seen},
            "kmer_instances_per_s": inst / dt, "final_distinct": tot, "seconds_second_chunking": res[1][0], "seconds_jobs_before_the_timed_one": [round(x, 3) for x in warm_s],
            "first_job_ms": FIRST.get("ms"), "first_job_note": "the first yak_count() of the process: beyond the ~112-160 GB the driver hands out at once, device memory costs ~30 ms per GB the first time (tests/tools/mb/mb_malloc.hip, mb_vmm5.hip; profiles/r05_mb_malloc.txt, r06_mb_vmm5.txt)",
            "peak_hbm_bytes": FIRST.get("peak"),
            "roofline": {"bound": "hbm", "kernel": "whole yak_count() call (parse + sweeps)", "achieved": by / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": by / dt / 1e9 / HBM_PEAK_GBS, "traffic": roof_t["traffic"], "hbm_util": roof_t["hbm_util"],
                         "traffic_source": (roof_t["traffic_source"] + "; half of the profiled command's two jobs") if roof_t["traffic_source"] else None, "algorithmic_bytes_per_instance": 32.0},
            "verify": verify}


def run_cfg4(a, torch, yak_amd):
    if a.sweeps == 1 and a.contigs * a.contig_len > 2_500_000_000:
        a.sweeps = 0                                           # beyond one pass's memory: yak_count() on the FASTA, the library's own rule picks the sweeps (yak_multi.cpp auto_sweeps:
                                                               # the default 50 x 100 Mb = BASELINE configs[3] runs in 8 in a process that does not own the memory yet, in 2 in one that does)
    if a.sweeps != 1:
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
           "kmer_instances_per_s": inst / dt, "final_distinct": tot, "first_job_ms": FIRST.get("ms"), "peak_hbm_bytes": int(L.yakamd_peak_bytes(0, 0)),
           "phase_ms_last_step": {k: round(v, 3) for k, v in stats.items() if k.startswith("ms_") and k != "ms_bloom"},
           "roofline": {"bound": "hbm", "kernel": "whole pass (extract + partition + insert + exact layout)", "achieved": by / dt / 1e9, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": by / dt / 1e9 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_instance": 32.0,
                        "extract_insert_frac": by / ((stats["ms_extract"] + stats["ms_insert"]) * 1e-3) / 1e9 / HBM_PEAK_GBS},
           "verify": verify}
    _roof_traffic(out["roofline"], f"cfg4_{a.contigs}x{a.contig_len}", dt)
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
            "roofline": _roof_traffic({"bound": "hbm", "kernel": "k_lookup + k_qv_reduce", "achieved": by / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": by / dt / 1e9 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_instance": 8.0}, "cfg5", dt,
                                      only=("k_lookup", "k_qv_reduce")),           # (the profiled command also builds the table: only the step's two kernels count)
            "verify": verify}


def run_cfg3shard(a, torch, yak_amd):
    """ONE rank's share of BASELINE configs[2] (600 M x 150 bp over `of` = 8 GPUs, prefix-sharded) on one GPU: the rank owns P / of sub-tables and
    receives, round after round, the records of its prefixes from `of` sources.  Every source's chunk of reads is generated on the host (the job's
    own read numbering: chunk c = round * of + source), partitioned on THIS device as its source would do it, and the owned slice is fed in source
    order -- exactly what the exchange delivers (yakamd_feed_partitioned_tagged_dev).  Timed per rank: the partition of the rank's OWN chunks
    (source 0), every feed, and the end of the pass; the other sources' partitions stand in for the peers' GPUs and are not this rank's time.
    The 8-GPU rate printed is a PREDICTION from these per-rank times and the exchange volume, not a measurement."""
    import bench
    K, N_HASH, P = 31, 4, 1 << PRE
    L = yak_amd.lib()
    of = a.of
    bf = a.bf_shift if a.bf_shift_given else 0                   # configs[2] is written without a filter; --bf-shift 37 runs the two-pass protocol
    per_src = a.reads if a.reads_given else 75_000_000
    batch = min(a.batch_reads, per_src)
    n_rounds = -(-per_src // batch)
    genome = 5 * per_src * of
    ERR = 0.001                                                # e = 0.1 % for configs[2] (at 0.5 % an unfiltered count of 600 M reads holds ~15 G distinct k-mers: ~200 GB of slots)
    threads = min(os.cpu_count() or 8, 64)
    rk = a.rank
    if not 0 <= rk < of or P % of:
        raise SystemExit("--rank must name one of the --of ranks, and --of divide the 1024 sub-tables")
    lo, hi = rk * P // of, (rk + 1) * P // of
    rec_len = bench.READ_LEN + 1
    B = batch * rec_len
    h_buf = torch.empty(B, dtype=torch.uint8, pin_memory=True)
    d_reads = torch.empty(B, dtype=torch.uint8, device="cuda:0")
    d_rec = torch.empty(B, dtype=torch.int64, device="cuda:0")
    h_bst = (C.c_uint64 * (P + 1))()
    syn = bench.synth_lib()
    if L.yakamd_tagged_ok(K, PRE) == 0:
        raise SystemExit("tagged records not available")
    tot_mem = torch.cuda.mem_get_info()[1]
    min_free = [torch.cuda.mem_get_info()[0]]
    T = {"own_partition": 0.0, "feed": 0.0, "finish": 0.0, "peer_partitions_not_counted": 0.0, "host_generation_not_counted": 0.0}
    fed = [0]

    def chunk(b, s, hashes):
        """source s's chunk of round b on the device, partitioned; returns (records, offsets of the owned slice, stream offset)"""
        n_reads = min(batch, per_src - b * batch)
        first = b * of * batch + s * n_reads                     # the job's stream = reads 0, 1, 2, ... in order, dealt round by round, source by source (every round before this one was a full one)
        tg = time.perf_counter()
        syn.yaksynth_reads(h_buf.data_ptr(), n_reads, bench.READ_LEN, genome, 42, ERR, 0.0005, first, threads)
        d_reads[:n_reads * rec_len].copy_(h_buf[:n_reads * rec_len])
        torch.cuda.synchronize()
        T["host_generation_not_counted"] += time.perf_counter() - tg
        tp = time.perf_counter()
        n = (L.yakamd_partition_hashes_dev if hashes else L.yakamd_partition_tagged_dev)(K, PRE, d_reads.data_ptr(), n_reads * rec_len, d_rec.data_ptr(), h_bst)
        torch.cuda.synchronize()
        T["own_partition" if s == rk else "peer_partitions_not_counted"] += time.perf_counter() - tp
        if n < 0:
            raise RuntimeError("partition: " + yak_amd._err())
        bst = list(h_bst)
        cnt = bst[hi] - bst[lo]
        ob = (C.c_uint64 * (P + 1))(*[bst[min(max(p_, lo), hi)] - bst[lo] for p_ in range(P + 1)])
        return d_rec.data_ptr() + 8 * bst[lo], cnt, ob, first * rec_len, n_reads * rec_len

    def one_pass(t, create_new):
        if L.yakamd_pass_begin(t.h, create_new) != 0:
            raise RuntimeError(yak_amd._err())
        reuse = 1
        if not create_new:
            tf = time.perf_counter()
            reuse = L.yakamd_count_retained(t.h)                 # 0: the records of pass 1 were kept (YAKAMD_RETAIN_GB) and are counted now
            torch.cuda.synchronize()
            T["feed"] += time.perf_counter() - tf
            if reuse < 0:
                raise RuntimeError(yak_amd._err())
        for b in range(n_rounds if reuse else 0):
            for s in range(of):
                ptr, cnt, ob, t0, span = chunk(b, s, hashes=not create_new)
                tf = time.perf_counter()
                rc = (L.yakamd_feed_partitioned_tagged_dev(t.h, ptr, cnt, ob, t0, span, 0) if create_new else L.yakamd_count_partitioned_dev(t.h, ptr, cnt, ob)) if cnt else 0
                torch.cuda.synchronize()
                T["feed"] += time.perf_counter() - tf
                if rc != 0:
                    raise RuntimeError("feed: " + yak_amd._err())
                fed[0] += cnt
            min_free[0] = min(min_free[0], torch.cuda.mem_get_info()[0])
        tf = time.perf_counter()
        n_ins = L.yakamd_pass_end(t.h)
        torch.cuda.synchronize()
        T["finish"] += time.perf_counter() - tf
        min_free[0] = min(min_free[0], torch.cuda.mem_get_info()[0])
        if n_ins < 0:
            raise RuntimeError("pass_end: " + yak_amd._err())
        t.h.contents.tot += n_ins
        return reuse == 0

    def job(last_job=True):
        for k_ in T:
            T[k_] = 0.0
        fed[0] = 0
        t = yak_amd.Table(K, PRE, N_HASH, bf)
        L.yakamd_set_shard(t.h, lo, hi)
        if bf > 0:
            L.yakamd_retain_input(t.h, 1)
        one_pass(t, 1)
        inst1 = fed[0]
        pass1 = dict(T)
        reused = None
        if bf > 0:
            t.destroy_bf(); t.clear()
            reused = one_pass(t, 0)
            tf = time.perf_counter()
            t.shrink(2, 1023)
            torch.cuda.synchronize()
            T["finish"] += time.perf_counter() - tf
        hist = (C.c_int64 * 1024)()
        L.yak_ch_hist(t.h, hist, 1)
        tot = t.tot
        caps = [t.subtable(p_) for p_ in range(lo, hi)]
        share = t.range_md5(lo, hi) if last_job and not a.no_verify else None    # the bytes this rank contributes to the job's .yak file
        t.close()
        return inst1, pass1, reused, hist, tot, caps, share

    # --warmup W: W whole jobs before the measured one (the device memory pool is then warm: a fresh process pays the driver ~27 ms per GB it
    # allocates for the first time, which a job of this size -- 200 GB of buffers -- feels; the first job's time is reported beside the last's)
    first_job = None
    n_jobs = max(0, a.warmup if a.warmup_given else 0) + 1
    for it_ in range(n_jobs):
        L.yakamd_peak_bytes(0, 1)
        inst1, pass1, reused, hist, tot, caps, share = job(it_ == n_jobs - 1)
        if first_job is None:
            first_job = {k_: round(v, 3) for k_, v in T.items()}
    rank_s = T["own_partition"] + T["feed"] + T["finish"]
    # what the rank receives over xGMI per pass: (of - 1) / of of its records, 8 bytes each, over of - 1 point-to-point links (MI355X_MICROARCH.md: ~153 GB/s each way per link)
    exch_s = inst1 * 8.0 * (of - 1) / of / ((of - 1) * 153e9 * 0.8) * (2 if (bf > 0 and not reused) else 1)
    # what the rank's device must hold in the REAL of-GPU job (bench.py --gpus of): the library's buffers at their high-water mark (measured here:
    # tables, the slice's records, the counting stage; the pool's idle ranges are not in it) + its share of the input resident in HBM + the two sets of
    # send / receive buffers of the exchange (one 2^29-byte chunk of reads per device and round: 8 bytes per k-mer instance each way)
    lib_peak = int(L.yakamd_peak_bytes(0, 0))
    chunk_reads_real = min(per_src, (1 << 29) // rec_len)
    exch_buf = 2 * 2 * chunk_reads_real * (bench.READ_LEN - K + 1) * 8
    real_peak = lib_peak + per_src * rec_len + exch_buf
    if real_peak > 0.9 * tot_mem:
        raise SystemExit(f"FAILED: a rank of the {of}-GPU job would need {real_peak / 1e9:.1f} GB of its device's {tot_mem / 1e9:.0f} GB")
    verify = {"count_mass_equals_instances": (sum(c * hist[c] for c in range(1024)) == inst1 and hist[1023] == 0) if bf == 0 else None,
              "sum_hist_equals_tot": sum(hist) == tot, "largest_subtable_slots": max(c_ for c_, _ in caps), "distinct": tot}
    if share is not None:
        verify["share_md5"], verify["share_bytes"] = share
        # the oracle counted the whole 600 M-read stream for the sub-tables of every rank of eight (tests/gen_golden_cfg3.py -> tests/golden/cfg3_full.json;
        # its own check at 1 M reads is on record too): a rank's share must be those bytes
        gfn = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "cfg3_full.json")
        if bf == 0 and os.path.exists(gfn):
            g = json.load(open(gfn))
            ref = None
            if (per_src * of, genome, PRE) == (g["reads"], g["genome"], g["pre"]):
                ref = g["ranges"].get(f"{lo}:{hi}")
            elif (per_src * of, genome) == (1_000_000, 5_000_000):
                ref = g["procedure_check_1M_reads"].get(f"{lo}:{hi}")
            if ref is not None:
                verify["share_equals_oracle"] = (ref["md5"], ref["size"], ref["distinct"]) == (share[0], share[1], tot)
                verify["oracle"] = f"tests/golden/cfg3_full.json {lo}:{hi} ({g['produced_by']})"
                if not verify["share_equals_oracle"]:
                    raise SystemExit(f"FAILED: the rank's share differs from the oracle's: {verify} vs {ref}")
    if verify["count_mass_equals_instances"] is False or not verify["sum_hist_equals_tot"]:
        raise SystemExit(f"FAILED: {verify}")
    by = (32.0 if bf == 0 else (16.0 + 128.0 + 16.0 + 16.0 + 8.0 + 8.0)) * inst1
    return {"metric": f"distinct k-mers counted/sec (k=31): ONE rank's share of a {of}-GPU prefix-sharded job, measured on one GPU", "value": tot / rank_s, "unit": "distinct k-mers/s (this rank)",
            "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": rank_s * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"rank {rk} of {of}: owns sub-tables [{lo}, {hi}); receives the records of its prefixes from {of} sources x {per_src} x {bench.READ_LEN} bp reads "
                                   f"(G = {genome}, e = 0.1 %) in {n_rounds} rounds of {batch} reads per source; yak count -k{K}" + (f" -b{bf}, both passes + shrink" if bf else ", no filter"),
                       "k": K, "pre": PRE, "bf_shift": bf, "of": of, "reads_per_source": per_src, "batch_reads": batch},
            "rank_seconds": {k_: round(v, 3) for k_, v in T.items()}, "first_job_rank_seconds": first_job, "jobs_before_the_measured_one": max(0, a.warmup if a.warmup_given else 0), "pass1_seconds": {k_: round(v, 3) for k_, v in pass1.items()},
            "instances_received_per_pass": inst1, "final_distinct_this_rank": tot,
            "peak_hbm_bytes": lib_peak, "peak_hbm_bytes_incl_pool_cache_and_harness": tot_mem - min_free[0],
            "peak_hbm_bytes_in_the_real_job": real_peak, "peak_hbm_note": f"library buffers in use at their high-water mark {lib_peak / 1e9:.1f} GB + this rank's {per_src * rec_len / 1e9:.1f} GB of the input resident in HBM "
                                                                        f"+ {exch_buf / 1e9:.1f} GB of exchange buffers (two sets of send + receive, one 2^29-byte chunk of reads each); the line fails above 0.9 of the device's {tot_mem / 1e9:.0f} GB",
            "pass2_counted_retained_records": reused,
            "prediction": {"label": "PREDICTED, not measured: all ranks take this rank's time; the exchange (8-byte records over of-1 xGMI links at 80 % of 153 GB/s) is added in full, not overlapped",
                           "exchange_seconds": exch_s, "job_seconds": rank_s + exch_s, "job_distinct_kmers_per_s": tot * of / (rank_s + exch_s),
                           "job_reads": per_src * of},
            "roofline": _roof_traffic({"bound": "hbm", "kernel": "this rank's passes (own partition + feeds + finish)", "achieved": by / rank_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": by / rank_s / 1e9 / HBM_PEAK_GBS, "traffic": None}, "cfg3shard", rank_s,
                                      scale={"k_xpart": 1.0 / of, "k_part_": 1.0 / of}),   # (the profiled command ran one job: --warmup 0)
            "verify": verify}


def run(a):
    import torch
    import yak_amd
    if a.gpus != 1:
        raise SystemExit("--config cfg3shard / cfg4 / cfg5 are single-GPU configurations")
    if yak_amd.lib().yakamd_device_count() < 1:
        raise SystemExit("no gfx950 device: refusing to run (no CPU fallback)")
    torch.cuda.set_device(0)
    out = run_cfg4(a, torch, yak_amd) if a.config == "cfg4" else run_cfg3shard(a, torch, yak_amd) if a.config == "cfg3shard" else run_cfg5(a, torch, yak_amd)
    print(json.dumps(out))
