"""yak_amd -- MI355X-native k-mer counting engine behind lh3/yak's C API.

This package is only the Python-side loader used by tests and bench.py.  The product is the
C-ABI shared library ``yak_amd/libyak_amd.so`` (hand-written gfx950 HIP kernels + the drop-in
``yak.h`` surface declared in ``include/yak.h`` / ``include/yak_amd.h``).  There is no Python or
CPU implementation of the counting path here: if the library is missing, importing :func:`lib`
raises; if no MI355X is visible, ``yak_ch_init`` / ``yak_count`` return NULL and the wrappers raise.

Note for processes that also use PyTorch-ROCm: import torch BEFORE calling :func:`lib` (torch ships
its own copy of the HIP runtime; whichever is loaded first serves both).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libyak_amd.so")

# every symbol include/yak.h and include/yak_amd.h declare (checked by tests/test_abi.py)
YAK_H_SYMBOLS = [
    "yak_copt_init", "yak_bf_init", "yak_bf_destroy", "yak_bf_insert",
    "yak_ch_init", "yak_ch_destroy", "yak_ch_destroy_bf", "yak_ch_insert_list", "yak_ch_get",
    "yak_ch_inc", "yak_ch_getseq", "yak_ch_clear", "yak_ch_hist", "yak_ch_shrink", "yak_ch_dump",
    "yak_ch_restore", "yak_count", "yak_verbose", "seq_nt4_table", "yak_qopt_init", "yak_qv", "yak_recount", "yak_ch_setcnt",
    "yak_ch_tighten", "yak_ch_merge", "yak_ch_subtract", "yak_ch_isec", "yak_ch_restore_core", "yak_qv_solve",
]
YAK_AMD_H_SYMBOLS = [
    "yakamd_device_count", "yakamd_last_error", "yakamd_ctx_of", "yakamd_set_shard",
    "yakamd_pass_begin", "yakamd_feed_bases_dev", "yakamd_feed_bases_host", "yakamd_feed_packed_dev", "yakamd_pack_bases_dev", "yakamd_packed_bytes", "yakamd_pack_bases_host", "yakamd_feed_packed_host", "yakamd_feed_packed_pieces_host", "yakamd_feed_hashed_dev",
    "yakamd_pass_end", "yakamd_extract_dev", "yakamd_sync_host", "yakamd_dump_mem", "yakamd_dump_range_mem", "yakamd_subtable",
    "yakamd_get_stats", "yakamd_trim", "yakamd_peak_bytes", "yakamd_dev_alloc", "yakamd_dev_free", "yakamd_memcpy_h2d",
    "yakamd_memcpy_d2h", "yakamd_partition_dev", "yakamd_feed_partitioned_dev", "yakamd_debug_counters", "yakamd_count_hashes_dev",
    "yakamd_partition_hashes_dev", "yakamd_count_partitioned_dev", "yakamd_feed_partitioned_lent_dev",
    "yakamd_tagged_ok", "yakamd_pass_fast", "yakamd_partition_tagged_dev", "yakamd_feed_partitioned_tagged_dev",
    "yakamd_lookup_dev", "yakamd_qv_reduce_dev", "yakamd_host_image", "yakamd_host_image_packed", "yakamd_gz_tune", "yakamd_gz_inflate", "yakamd_test_set", "yakamd_test_reset",
    "yakamd_retain_input", "yakamd_count_retained", "yakamd_retained_instances", "yakamd_count_multi_dev",
    "yakamd_host_alloc", "yakamd_host_free", "yakamd_device_sync", "yakamd_mem_info", "yakamd_last_sweeps", "yakamd_pool_report",
]


class CoptT(C.Structure):                      # yak_copt_t, include/yak.h (reference yak.h:25-31)
    _fields_ = [("bf_shift", C.c_int32), ("bf_n_hash", C.c_int32), ("k", C.c_int32),
                ("pre", C.c_int32), ("n_thread", C.c_int32), ("chunk_size", C.c_int64)]


class ChT(C.Structure):                        # yak_ch_t (reference yak.h:61-65)
    _fields_ = [("k", C.c_int), ("pre", C.c_int), ("n_hash", C.c_int), ("n_shift", C.c_int),
                ("tot", C.c_uint64), ("h", C.c_void_p)]


class QoptT(C.Structure):                      # yak_qopt_t (reference yak.h:33-40)
    _fields_ = [("print_each", C.c_int32), ("print_err_kmer", C.c_int32), ("min_len", C.c_int32),
                ("n_threads", C.c_int32), ("min_frac", C.c_double), ("fpr", C.c_double), ("chunk_size", C.c_int64)]


class QstatT(C.Structure):                     # yak_qstat_t (reference yak.h:42-47)
    _fields_ = [("tot", C.c_int64), ("qv_raw", C.c_double), ("qv", C.c_double), ("cov", C.c_double), ("err", C.c_double),
                ("fpr_lower", C.c_double), ("fpr_upper", C.c_double), ("adj_cnt", C.c_double * 1024)]


class StatsT(C.Structure):                     # yakamd_stats_t
    _fields_ = [("ms_extract", C.c_double), ("ms_insert", C.c_double), ("ms_bloom", C.c_double),
                ("ms_select", C.c_double), ("ms_sort", C.c_double), ("ms_replay", C.c_double),
                ("ms_total", C.c_double), ("ms_dominant_kernel", C.c_double),
                ("n_dominant_launches", C.c_int64), ("n_instances", C.c_int64),
                ("n_distinct_seen", C.c_int64), ("n_new_keys", C.c_int64),
                ("n_bloom_candidates", C.c_int64), ("ms_part2", C.c_double), ("ms_shrink", C.c_double)]


_lib = None


def lib():
    """Load libyak_amd.so (once).  Raises if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `make lib` "
                           "(python __graft_entry__.py build); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    P = C.POINTER
    L.yak_copt_init.argtypes = [P(CoptT)]
    L.yak_ch_init.restype = P(ChT); L.yak_ch_init.argtypes = [C.c_int] * 4
    L.yak_ch_destroy.argtypes = [P(ChT)]
    L.yak_ch_destroy_bf.argtypes = [P(ChT)]
    L.yak_ch_insert_list.restype = C.c_int
    L.yak_ch_insert_list.argtypes = [P(ChT), C.c_int, C.c_int, P(C.c_uint64)]
    L.yak_ch_get.restype = C.c_int; L.yak_ch_get.argtypes = [P(ChT), C.c_uint64]
    L.yak_ch_inc.restype = C.c_int; L.yak_ch_inc.argtypes = [P(ChT), C.c_uint64]
    L.yak_ch_clear.argtypes = [P(ChT), C.c_int]
    L.yak_ch_shrink.argtypes = [P(ChT), C.c_int, C.c_int, C.c_int]
    L.yak_ch_hist.argtypes = [P(ChT), P(C.c_int64), C.c_int]
    L.yak_ch_dump.restype = C.c_int; L.yak_ch_dump.argtypes = [P(ChT), C.c_char_p]
    L.yak_ch_restore.restype = P(ChT); L.yak_ch_restore.argtypes = [C.c_char_p]
    L.yak_count.restype = P(ChT); L.yak_count.argtypes = [C.c_char_p, P(CoptT), P(ChT)]
    L.yak_bf_init.restype = C.c_void_p; L.yak_bf_init.argtypes = [C.c_int, C.c_int]
    L.yak_bf_insert.restype = C.c_int; L.yak_bf_insert.argtypes = [C.c_void_p, C.c_uint64]
    L.yak_bf_destroy.argtypes = [C.c_void_p]
    L.yakamd_device_count.restype = C.c_int
    L.yakamd_last_error.restype = C.c_char_p
    L.yakamd_set_shard.restype = C.c_int; L.yakamd_set_shard.argtypes = [P(ChT), C.c_int, C.c_int]
    L.yakamd_pass_begin.restype = C.c_int; L.yakamd_pass_begin.argtypes = [P(ChT), C.c_int]
    L.yakamd_feed_bases_dev.restype = C.c_int
    L.yakamd_feed_bases_dev.argtypes = [P(ChT), C.c_void_p, C.c_int64, C.c_uint64]
    L.yakamd_packed_bytes.restype = C.c_int64
    L.yakamd_packed_bytes.argtypes = [C.c_int64]
    L.yakamd_pack_bases_host.restype = None
    L.yakamd_pack_bases_host.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    L.yakamd_feed_packed_host.restype = C.c_int
    L.yakamd_feed_packed_host.argtypes = [P(ChT), C.c_void_p, C.c_int64, C.c_uint64]
    L.yakamd_feed_packed_pieces_host.restype = C.c_int
    L.yakamd_feed_packed_pieces_host.argtypes = [P(ChT), C.c_int, P(C.c_void_p), P(C.c_void_p), P(C.c_int64), C.c_uint64]
    L.yakamd_feed_packed_dev.restype = C.c_int
    L.yakamd_feed_packed_dev.argtypes = [P(ChT), C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64]
    L.yakamd_pack_bases_dev.restype = C.c_int
    L.yakamd_pack_bases_dev.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.yakamd_feed_bases_host.restype = C.c_int
    L.yakamd_feed_bases_host.argtypes = [P(ChT), C.c_void_p, C.c_int64, C.c_uint64]
    L.yakamd_feed_hashed_dev.restype = C.c_int
    L.yakamd_feed_hashed_dev.argtypes = [P(ChT), C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64]
    L.yakamd_pass_end.restype = C.c_int64; L.yakamd_pass_end.argtypes = [P(ChT)]
    L.yakamd_extract_dev.restype = C.c_int64
    L.yakamd_extract_dev.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.yakamd_sync_host.restype = C.c_int; L.yakamd_sync_host.argtypes = [P(ChT)]
    L.yakamd_dump_mem.restype = C.c_int64
    L.yakamd_dump_mem.argtypes = [P(ChT), P(P(C.c_uint8))]
    L.yakamd_dump_range_mem.restype = C.c_int64
    L.yakamd_dump_range_mem.argtypes = [P(ChT), C.c_int, C.c_int, P(P(C.c_uint8))]
    L.yakamd_peak_bytes.restype = C.c_int64; L.yakamd_peak_bytes.argtypes = [C.c_int, C.c_int]
    L.yakamd_last_sweeps.restype = C.c_int; L.yakamd_last_sweeps.argtypes = []
    L.yakamd_pool_report.restype = None; L.yakamd_pool_report.argtypes = [C.c_char_p]
    L.yakamd_subtable.restype = C.c_int
    L.yakamd_subtable.argtypes = [P(ChT), C.c_int, P(C.c_uint32), P(C.c_uint32)]
    L.yakamd_get_stats.restype = C.c_int; L.yakamd_get_stats.argtypes = [P(ChT), P(StatsT)]
    L.yakamd_partition_dev.restype = C.c_int64
    L.yakamd_partition_dev.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, P(C.c_uint64)]
    L.yakamd_feed_partitioned_dev.restype = C.c_int
    L.yakamd_feed_partitioned_dev.argtypes = [P(ChT), C.c_void_p, C.c_int64, P(C.c_uint64), C.c_uint64, C.c_uint64]
    L.yakamd_feed_partitioned_lent_dev.restype = C.c_int
    L.yakamd_feed_partitioned_lent_dev.argtypes = L.yakamd_feed_partitioned_dev.argtypes
    L.yakamd_tagged_ok.restype = C.c_int; L.yakamd_tagged_ok.argtypes = [C.c_int, C.c_int]
    L.yakamd_partition_tagged_dev.restype = C.c_int64
    L.yakamd_partition_tagged_dev.argtypes = L.yakamd_partition_dev.argtypes
    L.yakamd_feed_partitioned_tagged_dev.restype = C.c_int
    L.yakamd_feed_partitioned_tagged_dev.argtypes = list(L.yakamd_feed_partitioned_dev.argtypes) + [C.c_int]
    L.yakamd_partition_hashes_dev.restype = C.c_int64
    L.yakamd_partition_hashes_dev.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, P(C.c_uint64)]
    L.yakamd_count_partitioned_dev.restype = C.c_int
    L.yakamd_count_partitioned_dev.argtypes = [P(ChT), C.c_void_p, C.c_int64, P(C.c_uint64)]
    L.yakamd_count_hashes_dev.restype = C.c_int
    L.yakamd_count_hashes_dev.argtypes = [P(ChT), C.c_void_p, C.c_int64]
    L.yakamd_host_image.restype = C.c_int64
    L.yakamd_host_image.argtypes = [C.c_char_p, C.c_int, C.c_int, P(C.c_void_p)]
    L.yakamd_host_image_packed.restype = C.c_int64
    L.yakamd_host_image_packed.argtypes = [C.c_char_p, C.c_int, P(C.c_void_p)]
    L.yakamd_test_set.restype = None
    L.yakamd_test_set.argtypes = [C.c_char_p, C.c_int64]
    L.yakamd_test_reset.restype = None
    L.yakamd_test_reset.argtypes = []
    L.yakamd_gz_tune.restype = None
    L.yakamd_gz_tune.argtypes = [C.c_int64, C.c_int64, C.c_int64]
    L.yakamd_gz_inflate.restype = C.c_int64
    L.yakamd_gz_inflate.argtypes = [C.c_char_p, C.c_int, P(C.c_void_p)]
    L.yak_ch_setcnt.argtypes = [P(ChT), C.c_int, C.c_int]
    L.yak_ch_restore_core.restype = P(ChT)
    L.yak_qv_solve.restype = C.c_int
    L.yak_qv_solve.argtypes = [P(C.c_int64), P(C.c_int64), C.c_int, C.c_double, P(QstatT)]
    L.yak_ch_tighten.argtypes = [P(ChT)]
    L.yak_ch_merge.argtypes = [P(ChT), P(ChT), C.c_int, C.c_int, C.c_int, C.c_int]
    L.yak_ch_subtract.argtypes = [P(ChT), P(ChT), C.c_int]
    L.yak_ch_isec.argtypes = [P(ChT), P(ChT), C.c_int]
    L.yak_recount.restype = None; L.yak_recount.argtypes = [C.c_char_p, P(ChT)]
    L.yak_qopt_init.argtypes = [P(QoptT)]
    L.yak_qv.restype = None; L.yak_qv.argtypes = [P(QoptT), C.c_char_p, P(ChT), P(C.c_int64)]
    L.yakamd_lookup_dev.restype = C.c_int; L.yakamd_lookup_dev.argtypes = [P(ChT), C.c_void_p, C.c_int64, C.c_void_p]
    L.yakamd_retain_input.restype = C.c_int; L.yakamd_retain_input.argtypes = [P(ChT), C.c_int]
    L.yakamd_count_retained.restype = C.c_int; L.yakamd_count_retained.argtypes = [P(ChT)]
    L.yakamd_retained_instances.restype = C.c_int64; L.yakamd_retained_instances.argtypes = [P(ChT)]
    L.yakamd_count_multi_dev.restype = P(ChT)
    L.yakamd_count_multi_dev.argtypes = [P(CoptT), P(ChT), C.c_int, P(C.c_int), C.c_int, P(C.c_void_p), P(C.c_int64), P(C.c_int)]
    L.yakamd_qv_reduce_dev.restype = C.c_int
    L.yakamd_qv_reduce_dev.argtypes = [P(ChT), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_double,
                                       C.c_void_p, C.c_void_p, C.c_void_p]
    L.yakamd_host_alloc.restype = C.c_void_p; L.yakamd_host_alloc.argtypes = [C.c_size_t]
    L.yakamd_host_free.argtypes = [C.c_void_p]
    L.yakamd_device_sync.restype = C.c_int
    L.yakamd_mem_info.restype = C.c_int; L.yakamd_mem_info.argtypes = [P(C.c_size_t), P(C.c_size_t)]
    L.yakamd_dev_alloc.restype = C.c_void_p; L.yakamd_dev_alloc.argtypes = [C.c_size_t]
    L.yakamd_dev_free.argtypes = [C.c_void_p]
    L.yakamd_memcpy_h2d.restype = C.c_int; L.yakamd_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.yakamd_memcpy_d2h.restype = C.c_int; L.yakamd_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    _lib = L
    return L


def _err():
    return (lib().yakamd_last_error() or b"").decode()


class Table:
    """Thin owner of a ``yak_ch_t *`` created through the C ABI."""

    def __init__(self, k=31, pre=10, n_hash=4, bf_shift=0, ptr=None):
        self.L = lib()
        self.h = ptr if ptr is not None else self.L.yak_ch_init(k, pre, n_hash, bf_shift)
        if not self.h:
            raise RuntimeError("yak_ch_init failed: " + _err())

    # -- reference protocol pieces ---------------------------------------------------------
    def count_pass(self, create_new, feeds, same_input=False):
        """one pass: feeds = iterable of (device_ptr, n_bytes, t0).  same_input (create_new = 0 only): the feeds are the ones of
        the create_new pass before -- the records that pass retained (yakamd_retain_input) are counted instead, if there are any"""
        if self.L.yakamd_pass_begin(self.h, create_new) != 0:
            raise RuntimeError(_err())
        r = self.L.yakamd_count_retained(self.h) if (same_input and not create_new) else 1
        if r < 0:
            raise RuntimeError(_err())
        for ptr, n, t0 in (feeds if r else ()):
            if self.L.yakamd_feed_bases_dev(self.h, ptr, n, t0) != 0:
                raise RuntimeError(_err())
        n_ins = self.L.yakamd_pass_end(self.h)
        if n_ins < 0:
            raise RuntimeError(_err())
        self.h.contents.tot += n_ins
        return n_ins

    def count_pass_packed(self, create_new, feeds, same_input=False):
        """one pass over packed images: feeds = iterable of (codes_ptr, valid_ptr, n_bases, t0) (yakamd_feed_packed_dev)"""
        if self.L.yakamd_pass_begin(self.h, create_new) != 0:
            raise RuntimeError(_err())
        r = self.L.yakamd_count_retained(self.h) if (same_input and not create_new) else 1
        if r < 0:
            raise RuntimeError(_err())
        for codes, valid, n, t0 in (feeds if r else ()):
            if self.L.yakamd_feed_packed_dev(self.h, codes, valid, n, t0) != 0:
                raise RuntimeError(_err())
        n_ins = self.L.yakamd_pass_end(self.h)
        if n_ins < 0:
            raise RuntimeError(_err())
        self.h.contents.tot += n_ins
        return n_ins

    def count_pass_packed_host(self, create_new, pieces, as_one=False):
        """one pass over pieces of the stream, each packed on the host (pack_bases_host) and fed from host memory: pieces = iterable of (ascii bytes, t0);
        as_one: all of them in ONE feed (yakamd_feed_packed_pieces_host), every piece taken to end at a multiple of 32 positions"""
        if self.L.yakamd_pass_begin(self.h, create_new) != 0:
            raise RuntimeError(_err())
        if as_one:
            pieces = list(pieces)
            pk = [pack_bases_host(buf) for buf, _ in pieces]
            nw = [(len(buf) + 31) // 32 for buf, _ in pieces]
            keep = [C.create_string_buffer(x, len(x)) for x in pk]
            codes = (C.c_void_p * len(pk))(*[C.addressof(k) for k in keep])
            valid = (C.c_void_p * len(pk))(*[C.addressof(k) + ((8 * w + 15) & ~15) for k, w in zip(keep, nw)])
            if self.L.yakamd_feed_packed_pieces_host(self.h, len(pk), codes, valid, (C.c_int64 * len(pk))(*nw), pieces[0][1] if pieces else 0) != 0:
                raise RuntimeError(_err())
            pieces = ()
        for buf, t0 in pieces:
            pk = pack_bases_host(buf)
            if self.L.yakamd_feed_packed_host(self.h, pk, len(buf), t0) != 0:
                raise RuntimeError(_err())
        n_ins = self.L.yakamd_pass_end(self.h)
        if n_ins < 0:
            raise RuntimeError(_err())
        self.h.contents.tot += n_ins
        return n_ins

    def count_pass_host(self, create_new, buf, t0=0):
        if self.L.yakamd_pass_begin(self.h, create_new) != 0:
            raise RuntimeError(_err())
        cbuf = (C.c_char * len(buf)).from_buffer_copy(buf)
        if self.L.yakamd_feed_bases_host(self.h, cbuf, len(buf), t0) != 0:
            raise RuntimeError(_err())
        n_ins = self.L.yakamd_pass_end(self.h)
        if n_ins < 0:
            raise RuntimeError(_err())
        self.h.contents.tot += n_ins
        return n_ins

    def destroy_bf(self):
        self.L.yak_ch_destroy_bf(self.h)

    def clear(self):
        self.L.yak_ch_clear(self.h, 1)

    def shrink(self, lo, hi):
        self.L.yak_ch_shrink(self.h, lo, hi, 1)

    @property
    def tot(self):
        return self.h.contents.tot

    def stats(self):
        st = StatsT()
        self.L.yakamd_get_stats(self.h, C.byref(st))
        return {f: getattr(st, f) for f, _ in StatsT._fields_}

    def dump_bytes(self):
        out = C.POINTER(C.c_uint8)()
        n = self.L.yakamd_dump_mem(self.h, C.byref(out))
        if n < 0:
            raise RuntimeError(_err())
        addr = C.cast(out, C.c_void_p).value
        data = bytes((C.c_char * n).from_address(addr))          # C.string_at takes a C int: .yak files pass 2 GB
        C.CDLL(None).free(out)
        return data

    def dump_md5(self):
        """md5 of the .yak bytes without a second copy of them (multi-GB tables)"""
        import hashlib
        out = C.POINTER(C.c_uint8)()
        n = self.L.yakamd_dump_mem(self.h, C.byref(out))
        if n < 0:
            raise RuntimeError(_err())
        addr = C.cast(out, C.c_void_p).value
        h = hashlib.md5()
        step = 1 << 28
        for o in range(0, n, step):
            h.update((C.c_char * min(step, n - o)).from_address(addr + o))
        C.CDLL(None).free(out)
        return h.hexdigest(), n

    def range_md5(self, lo, hi):
        """md5 and size of the bytes of sub-tables [lo, hi) ({capacity, size, keys in slot order} each, no header): one rank's share of the .yak file"""
        import hashlib
        out = C.POINTER(C.c_uint8)()
        n = self.L.yakamd_dump_range_mem(self.h, lo, hi, C.byref(out))
        if n < 0:
            raise RuntimeError(_err())
        addr = C.cast(out, C.c_void_p).value
        h = hashlib.md5()
        step = 1 << 28
        for o in range(0, n, step):
            h.update((C.c_char * min(step, n - o)).from_address(addr + o))
        C.CDLL(None).free(out)
        return h.hexdigest(), n

    def subtable(self, i):
        cap, size = C.c_uint32(), C.c_uint32()
        self.L.yakamd_subtable(self.h, i, C.byref(cap), C.byref(size))
        return cap.value, size.value

    def close(self):
        if self.h:
            self.L.yak_ch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def count_protocol_host(buf1, k=31, pre=10, n_hash=4, bf_shift=0, buf2=None):
    """`yak count` protocol of reference main.c:53-60 on in-memory base images -> .yak bytes"""
    t = Table(k, pre, n_hash, bf_shift)
    try:
        t.count_pass_host(1, buf1)
        if bf_shift > 0:
            t.destroy_bf()
            t.clear()
            t.count_pass_host(0, buf2 if buf2 is not None else buf1)
            t.shrink(2, 1023)
        return t.dump_bytes(), t.tot
    finally:
        t.close()


def qv_counts(table_fn, seq_fn, min_len=0, min_frac=0.5, chunk=1000000000):
    """`yak qv` counting step through the C ABI (yak_ch_restore + yak_qv): the 1024-bin histogram of
    table counts over the k-mers of the accepted sequences"""
    L = lib()
    h = L.yak_ch_restore(table_fn.encode())
    if not h:
        raise RuntimeError("yak_ch_restore failed: " + _err())
    o = QoptT()
    L.yak_qopt_init(C.byref(o))
    o.min_len, o.min_frac, o.chunk_size = min_len, min_frac, chunk
    cnt = (C.c_int64 * 1024)()
    L.yak_qv(C.byref(o), seq_fn.encode(), h, cnt)
    L.yak_ch_destroy(h)
    return list(cnt)


def pack_bases_host(buf):
    """the packed image of an ASCII base image (host only): code words, then -- 16-byte aligned -- validity words"""
    L = lib()
    n = len(buf)
    out = C.create_string_buffer(max(16, L.yakamd_packed_bytes(n)))
    src = (C.c_char * max(1, n)).from_buffer_copy(buf if n else b"\0")
    L.yakamd_pack_bases_host(src, n, out)
    return out.raw[:L.yakamd_packed_bytes(n)]


def gz_tune(chunk_bytes=0, min_file_bytes=-1, front_bytes=-1):
    """test hook: the gzip reader's bytes per thread and batch, the smallest file it takes, the room in front of a batch"""
    lib().yakamd_gz_tune(chunk_bytes, min_file_bytes, front_bytes)


def gz_inflate(fn, threads=4):
    """the inflated stream of gzip file `fn` as the parallel reader delivers it (host only); None if it does not take the file"""
    L = lib()
    out = C.c_void_p()
    n = L.yakamd_gz_inflate(fn.encode(), threads, C.byref(out))
    if n == -1:
        return None
    if n < 0:
        raise OSError("invalid gzip stream in " + fn + ": " + L.yakamd_last_error().decode())
    data = C.string_at(out, n)
    C.CDLL(None).free(out)
    return data


def host_image_packed(fn, min_len=0):
    """the stream yak_count() feeds for `fn` when its parser threads pack, unpacked again (host only); None if the parallel parser does not take the file"""
    L = lib()
    out = C.c_void_p()
    n = L.yakamd_host_image_packed(fn.encode(), min_len, C.byref(out))
    if n < 0:
        return None
    data = C.string_at(out, n)
    C.CDLL(None).free(out)
    return data


def host_image(fn, min_len=0, fast=True):
    """the base image yak_count() would feed for file `fn` (host only)"""
    L = lib()
    out = C.c_void_p()
    n = L.yakamd_host_image(fn.encode(), min_len, 1 if fast else 0, C.byref(out))
    if n < 0:
        raise OSError("cannot read " + fn)
    data = C.string_at(out, n)
    C.CDLL(None).free(out)
    return data


def kernels_sha16():
    """sha256 (first 16 hex digits) over the device code the library is built from (csrc/kernels.hip, kern_*.inc, yk_device.h): the counter passes under
    profiles/ carry the value they were measured on, and bench.py reports `roofline.traffic` only while it equals the value of the tree it runs from."""
    import glob, hashlib
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    h = hashlib.sha256()
    for fn in sorted(glob.glob(os.path.join(d, "kern_*.inc")) + [os.path.join(d, "kernels.hip"), os.path.join(d, "yk_device.h")]):
        h.update(os.path.basename(fn).encode() + b"\0")
        h.update(open(fn, "rb").read())
    return h.hexdigest()[:16]
