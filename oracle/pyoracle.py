"""ORACLE loader (test infrastructure only -- never imported by the yak_amd package).

ctypes access to oracle/liboracle.so (the CPU restatement, oracle/yko*.c) and to the prebuilt
reference in oracle/_ref/ when present.  Used by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py only.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")
REF_BIN = os.path.join(HERE, "_ref", "yak")
REF_LIB = os.path.join(HERE, "_ref", "libyakref.so")
REF_SHIM = os.path.join(HERE, "_ref", "libyakshim.so")
YKO_BIN = os.path.join(HERE, "yko")


class Copt(C.Structure):
    _fields_ = [("bf_shift", C.c_int32), ("bf_n_hash", C.c_int32), ("k", C.c_int32),
                ("pre", C.c_int32), ("n_thread", C.c_int32), ("chunk_size", C.c_int64)]


class Qopt(C.Structure):                       # yko_qopt_t == yak_qopt_t (yak.h:33-40)
    _fields_ = [("print_each", C.c_int32), ("print_err_kmer", C.c_int32), ("min_len", C.c_int32),
                ("n_threads", C.c_int32), ("min_frac", C.c_double), ("fpr", C.c_double), ("chunk_size", C.c_int64)]


class Ch(C.Structure):
    _fields_ = [("k", C.c_int), ("pre", C.c_int), ("n_hash", C.c_int), ("n_shift", C.c_int),
                ("tot", C.c_uint64), ("h", C.c_void_p)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, "all", "ref"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        P = C.POINTER
        L.yko_hash64.restype = C.c_uint64; L.yko_hash64.argtypes = [C.c_uint64, C.c_uint64]
        L.yko_hash64_64.restype = C.c_uint64; L.yko_hash64_64.argtypes = [C.c_uint64]
        L.yko_hash64_inv.restype = C.c_uint64; L.yko_hash64_inv.argtypes = [C.c_uint64, C.c_uint64]
        L.yko_hash_long.restype = C.c_uint64; L.yko_hash_long.argtypes = [P(C.c_uint64)]
        L.yko_h2b.restype = C.c_uint32; L.yko_h2b.argtypes = [C.c_uint32, C.c_uint32]
        L.yko_bf_init.restype = C.c_void_p; L.yko_bf_init.argtypes = [C.c_int, C.c_int]
        L.yko_bf_insert.restype = C.c_int; L.yko_bf_insert.argtypes = [C.c_void_p, C.c_uint64]
        L.yko_bf_destroy.argtypes = [C.c_void_p]
        L.yko_copt_init.argtypes = [P(Copt)]
        L.yko_ch_init.restype = P(Ch); L.yko_ch_init.argtypes = [C.c_int] * 4
        L.yko_ch_destroy.argtypes = [P(Ch)]
        L.yko_ch_insert_list.restype = C.c_int
        L.yko_ch_insert_list.argtypes = [P(Ch), C.c_int, C.c_int, P(C.c_uint64)]
        L.yko_ch_get.restype = C.c_int; L.yko_ch_get.argtypes = [P(Ch), C.c_uint64]
        L.yko_ch_clear.argtypes = [P(Ch)]
        L.yko_ch_shrink.argtypes = [P(Ch), C.c_int, C.c_int]
        L.yko_ch_destroy_bf.argtypes = [P(Ch)]
        L.yko_ch_dump.restype = C.c_int; L.yko_ch_dump.argtypes = [P(Ch), C.c_char_p]
        L.yko_ch_restore.restype = P(Ch); L.yko_ch_restore.argtypes = [C.c_char_p]
        L.yko_ch_dump_mem.restype = C.c_size_t; L.yko_ch_dump_mem.argtypes = [P(Ch), P(P(C.c_uint8))]
        L.yko_ch_subtable.argtypes = [P(Ch), C.c_int, P(C.c_uint32), P(C.c_uint32)]
        L.yko_extract_pos.restype = C.c_int64
        L.yko_extract_pos.argtypes = [C.c_int, C.c_char_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.yko_count_mem.restype = P(Ch)
        L.yko_count_mem.argtypes = [C.c_char_p, C.c_int64, P(Copt), P(Ch)]
        L.yko_count_protocol_mem.restype = P(Ch)
        L.yko_count_protocol_mem.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, C.c_int64, P(Copt)]
        L.yko_read_image.restype = C.c_int64; L.yko_read_image.argtypes = [C.c_char_p, C.c_int, P(C.c_void_p)]
        L.yko_ch_hist.argtypes = [P(Ch), P(C.c_int64)]
        L.yko_ch_setcnt.argtypes = [P(Ch), C.c_int]
        L.yko_qopt_init.argtypes = [P(Qopt)]
        L.yko_qv.restype = C.c_int; L.yko_qv.argtypes = [P(Qopt), C.c_char_p, P(Ch), P(C.c_int64), C.c_void_p]
        L.yko_count_file.restype = P(Ch); L.yko_count_file.argtypes = [C.c_char_p, P(Copt), P(Ch)]
        L.yko_count_protocol_file.restype = P(Ch)
        L.yko_count_protocol_file.argtypes = [C.c_char_p, C.c_char_p, P(Copt)]
        _lib = L
    return _lib


def copt(k=31, pre=10, n_hash=4, bf_shift=0, chunk=10000000):
    o = Copt()
    lib().yko_copt_init(C.byref(o))
    o.k, o.pre, o.bf_n_hash, o.bf_shift, o.chunk_size = k, pre, n_hash, bf_shift, chunk
    return o


def dump_bytes(h):
    L = lib()
    out = C.POINTER(C.c_uint8)()
    n = L.yko_ch_dump_mem(h, C.byref(out))
    data = C.string_at(out, n)
    C.CDLL(None).free(out)
    return data


def count_protocol_mem(buf1, k=31, pre=10, n_hash=4, bf_shift=0, buf2=None, chunk=10000000):
    """`yak count` protocol (reference main.c:53-60) on memory images -> (.yak bytes, tot)"""
    L = lib()
    o = copt(k, pre, n_hash, bf_shift, chunk)
    h = L.yko_count_protocol_mem(buf1, len(buf1), buf2, len(buf2) if buf2 is not None else 0, C.byref(o))
    try:
        return dump_bytes(h), h.contents.tot
    finally:
        L.yko_ch_destroy(h)


def have_ref():
    return os.path.exists(REF_BIN)


def ref_count_cli(args, timeout=3600):
    """run the prebuilt reference binary: `yak count <args>`; returns wall seconds"""
    import time
    t = time.time()
    subprocess.run([REF_BIN, "count"] + list(args), check=True, stderr=subprocess.DEVNULL, timeout=timeout)
    return time.time() - t


def qv_counts(table_fn, seq_fn, min_len=0, min_frac=0.5):
    """counting step of `yak qv` (qv.c:34-135) -> the 1024-bin histogram"""
    L = lib()
    h = L.yko_ch_restore(table_fn.encode())
    o = Qopt()
    L.yko_qopt_init(C.byref(o))
    o.min_len, o.min_frac = min_len, min_frac
    cnt = (C.c_int64 * 1024)()
    assert L.yko_qv(C.byref(o), seq_fn.encode(), h, cnt, None) == 0
    L.yko_ch_destroy(h)
    return list(cnt)


def parse_qv_output(text):
    """stdout of `yak qv` / `yko qv` / `yak-amd qv` -> ({count: (table k-mers, input k-mers)}, sorted SQ lines, sorted EK lines)"""
    ct, sq, ek = {}, [], []
    for l in text.splitlines():
        f = l.split("\t")
        if f[0] == "CT":
            ct[int(f[1])] = (int(f[2]), int(f[3]))
        elif f[0] == "SQ":
            sq.append(l)
        elif f[0] == "EK":
            ek.append(l)
    return ct, sorted(sq), sorted(ek)


def read_image(fn, min_len=0):
    L = lib()
    out = C.c_void_p()
    n = L.yko_read_image(fn.encode(), min_len, C.byref(out))
    if n < 0:
        raise OSError("cannot read " + fn)
    data = C.string_at(out, n)
    C.CDLL(None).free(out)
    return data
