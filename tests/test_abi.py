"""The C-ABI library loads on a machine without a GPU and exports every symbol that
include/yak.h and include/yak_amd.h declare (no compute calls here)."""
import ctypes as C
import os
import re

from conftest import ROOT


def declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(yak(?:amd)?_[a-z0-9_]+)\s*\(", src))
    names |= set(re.findall(r"extern\s+[a-z ]+\b(yak_verbose|seq_nt4_table)\b", src))
    return names


def test_library_exports_every_declared_symbol():
    import yak_amd
    L = yak_amd.lib()
    want = declared("yak.h") | declared("yak_amd.h")
    assert {"yak_count", "yak_ch_init", "yak_ch_insert_list", "yak_ch_dump", "yak_bf_insert",
            "yakamd_feed_bases_dev", "yakamd_pass_end"} <= want
    missing = [n for n in sorted(want) if not hasattr(L, n)]
    assert not missing, missing
    assert set(yak_amd.YAK_H_SYMBOLS) | set(yak_amd.YAK_AMD_H_SYMBOLS) <= want | {"yak_verbose", "seq_nt4_table"}


def test_struct_layouts_match_reference_abi():
    import yak_amd
    assert C.sizeof(yak_amd.CoptT) == 32           # 5 x int32 + pad + int64 (yak.h:25-31)
    assert yak_amd.CoptT.chunk_size.offset == 24
    assert C.sizeof(yak_amd.ChT) == 32 and yak_amd.ChT.tot.offset == 16 and yak_amd.ChT.h.offset == 24


def test_defaults_and_pure_host_entry_points():
    import yak_amd
    L = yak_amd.lib()
    o = yak_amd.CoptT()
    L.yak_copt_init(C.byref(o))
    assert (o.k, o.pre, o.bf_shift, o.bf_n_hash, o.n_thread, o.chunk_size) == (31, 10, 0, 4, 4, 10000000)
    t = (C.c_ubyte * 256).in_dll(L, "seq_nt4_table")
    assert [t[ord(c)] for c in "ACGTNacgtu"] == [0, 1, 2, 3, 4, 0, 1, 2, 3, 3] and t[0] == 0 and t[3] == 3
    assert L.yak_bf_init(8, 4) is None               # bbf.c:9


def test_no_cpu_fallback_without_gpu():
    """without a gfx950 device the table constructor must fail (NULL), not count on the CPU"""
    import yak_amd
    L = yak_amd.lib()
    if L.yakamd_device_count() > 0:
        return
    assert not L.yak_ch_init(31, 10, 4, 0)
    o = yak_amd.CoptT(); L.yak_copt_init(C.byref(o))
    assert not L.yak_count(os.path.join(ROOT, "tests", "golden", "inputs", "edge.fx").encode(), C.byref(o), None)
    assert b"no gfx950" in L.yakamd_last_error()
