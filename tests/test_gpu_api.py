"""yak_ch_* entry points on the device state vs the same call sequence on the oracle."""
import ctypes as C
import os
import random

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ya():
    import yak_amd
    assert yak_amd.lib().yakamd_device_count() >= 1
    return yak_amd


def _lists(oracle, seed, n_lists, n, pre=10, dup=0.5):
    """hashed k-mer lists sharing one prefix each, with repeats inside and across lists"""
    rnd = random.Random(seed)
    m = (1 << 62) - 1
    pool = {}
    out = []
    for _ in range(n_lists):
        p = rnd.randrange(8)
        cur = []
        while len(cur) < n:
            if pool.get(p) and rnd.random() < dup:
                cur.append(rnd.choice(pool[p]))
            else:
                h = (oracle.lib().yko_hash64(rnd.getrandbits(62), m) >> pre << pre) | p
                pool.setdefault(p, []).append(h); cur.append(h)
        out.append(cur)
    return out


@pytest.mark.parametrize("bf", [0, 21])
def test_insert_list_sequences(bf, ya, oracle):
    """htab.c:51-78 called list by list, as count.c:82 does; create_new = 1 then 0; table bytes and
    return values must match after every call (covers growth on an existing table image, the bloom
    carried across calls, the foreign-prefix skip and count saturation)"""
    L, O = ya.lib(), oracle.lib()
    t = ya.Table(31, 10, 4, bf)
    o = O.yko_ch_init(31, 10, 4, bf)
    lists = _lists(oracle, 5, 40, 300)
    lists[3] = lists[3][:100] + [lists[3][0] ^ 1] + lists[3][100:]       # a foreign prefix in the middle
    lists.append([lists[0][0]] * 1500)                                    # saturate one key
    for i, a in enumerate(lists):
        arr = (C.c_uint64 * len(a))(*a)
        create = 1 if i % 5 != 4 else 0
        assert L.yak_ch_insert_list(t.h, create, len(a), arr) == O.yko_ch_insert_list(o, create, len(a), arr), i
        if i % 7 == 0 or i == len(lists) - 1:
            assert t.dump_bytes() == oracle.dump_bytes(o), i
    for a in lists[:5]:
        for x in a[:20]:
            assert L.yak_ch_get(t.h, x) == O.yko_ch_get(o, x)
    assert L.yak_ch_get(t.h, 12345 << 10) == -1
    t.close(); O.yko_ch_destroy(o)


def test_inc_and_concurrent_insert_lists(ya, oracle):
    """yak_ch_inc (htab.c:80-91) as a single-slot device update that keeps a valid host mirror coherent, and
    yak_ch_insert_list called from several threads at once (the reference's kt_for workers call it
    concurrently, one sub-table each, count.c:129-143): calls take turns, nothing is dropped"""
    import threading
    L, O = ya.lib(), oracle.lib()
    t = ya.Table(31, 10, 0, 0)
    o = O.yko_ch_init(31, 10, 0, 0)
    lists = _lists(oracle, 5, 24, 200)
    res = [None] * len(lists)

    def work(j0):
        for i in range(j0, len(lists), 4):
            arr = (C.c_uint64 * len(lists[i]))(*lists[i])
            res[i] = L.yak_ch_insert_list(t.h, 1, len(lists[i]), arr)
    th = [threading.Thread(target=work, args=(j,)) for j in range(4)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert all(r is not None and r >= 0 for r in res)
    # the final multiset of (key, count) does not depend on the order in which the lists were taken
    for a in lists:
        O.yko_ch_insert_list(o, 1, len(a), (C.c_uint64 * len(a))(*a))
    keys = sorted({x for a in lists for x in a})
    assert [L.yak_ch_get(t.h, x) for x in keys] == [O.yko_ch_get(o, x) for x in keys]
    assert sum(res) == len(keys)
    O.yko_ch_inc.restype = C.c_int; O.yko_ch_inc.argtypes = [C.POINTER(oracle.Ch), C.c_uint64]
    for x in keys[:50] + [keys[0]] * 5 + [(12345 << 10) | 3]:
        assert L.yak_ch_inc(t.h, x) == O.yko_ch_inc(o, x)
        assert L.yak_ch_get(t.h, x) == O.yko_ch_get(o, x)      # the mirror was patched, not refreshed
    h1 = (C.c_int64 * 1024)(); h2 = (C.c_int64 * 1024)()
    L.yak_ch_hist(t.h, h1, 1); O.yko_ch_hist.argtypes = [C.POINTER(oracle.Ch), C.POINTER(C.c_int64)]; O.yko_ch_hist(o, h2)
    assert list(h1) == list(h2)                                # ... and the device image holds the increments
    t.close(); O.yko_ch_destroy(o)


def test_clear_shrink_restore_hist(ya, oracle, synth, tmp_path):
    L, O = ya.lib(), oracle.lib()
    img = synth(3000, g=15000, s=4)
    t = ya.Table(31, 10, 4, 0)
    t.count_pass_host(1, img)
    oc = oracle.copt()
    o = O.yko_count_mem(img, len(img), C.byref(oc), None)
    assert t.tot == o.contents.tot
    h1 = (C.c_int64 * 1024)(); h2 = (C.c_int64 * 1024)()
    L.yak_ch_hist(t.h, h1, 1); O.yko_ch_hist.argtypes = [C.POINTER(oracle.Ch), C.POINTER(C.c_int64)]; O.yko_ch_hist(o, h2)
    assert list(h1) == list(h2)
    for lo, hi in ((2, 1023), (3, 7), (1, 2000), (5, 4)):
        t.shrink(lo, hi); O.yko_ch_shrink(o, lo, hi)
        assert t.dump_bytes() == oracle.dump_bytes(o) and t.tot == o.contents.tot
    fn = str(tmp_path / "t.yak").encode()
    assert L.yak_ch_dump(t.h, fn) == 0
    r = ya.Table(ptr=L.yak_ch_restore(fn))
    ro = O.yko_ch_restore(fn)
    assert r.dump_bytes() == oracle.dump_bytes(ro)       # restore re-puts in file order; empties get capacity 4
    t.clear(); O.yko_ch_clear(o)
    assert t.dump_bytes() == oracle.dump_bytes(o)
    t.count_pass_host(0, img)                            # recount into the cleared table
    O.yko_count_mem(img, len(img), C.byref(oc), o)
    assert t.dump_bytes() == oracle.dump_bytes(o)
    assert L.yak_ch_dump(t.h, b"/nonexistent_dir/x.yak") == -1
    assert not L.yak_ch_restore(b"/nonexistent.yak")
    t.close(); r.close(); O.yko_ch_destroy(o); O.yko_ch_destroy(ro)


def test_yak_count_file_api(ya, oracle, tmp_path):
    """yak_count() itself: NULL on an unreadable file, h0 semantics, chunking by opt->chunk_size"""
    import subprocess
    from conftest import ROOT
    L = ya.lib()
    fq = str(tmp_path / "r.fq")
    subprocess.check_call([os.path.join(ROOT, "tools", "yaksynth"), "-n", "2000", "-g", "10000", "-s", "3", "-o", fq])
    o = ya.CoptT(); L.yak_copt_init(C.byref(o)); o.chunk_size = 30000
    assert not L.yak_count(b"/no/such/file.fq", C.byref(o), None)
    h = L.yak_count(fq.encode(), C.byref(o), None)
    t = ya.Table(ptr=h)
    oc = oracle.copt(chunk=30000)
    want = oracle.lib().yko_count_protocol_file(fq.encode(), None, C.byref(oc))
    assert t.dump_bytes() == oracle.dump_bytes(want) and t.tot == want.contents.tot
    assert L.yak_count(fq.encode(), C.byref(o), h)       # second call returns h0 itself
    t.close(); oracle.lib().yko_ch_destroy(want)


@pytest.mark.gpu
def test_recount_equals_clear_plus_count_existing(ya, oracle, tmp_path):
    """yak_recount (count.c:168-193): counts of the table's k-mers in ANOTHER file; layout untouched"""
    import subprocess
    from conftest import ROOT
    L, O = ya.lib(), oracle.lib()
    syn = os.path.join(ROOT, "tools", "yaksynth")
    f1, f2, tab = str(tmp_path / "a.fq"), str(tmp_path / "b.fa"), str(tmp_path / "t.yak")
    subprocess.check_call([syn, "-n", "8000", "-l", "150", "-g", "40000", "-s", "21", "-o", f1])
    subprocess.check_call([syn, "-a", "-n", "30", "-l", "5000", "-g", "40000", "-s", "21", "-e", "0.01", "-N", "0.001", "-o", f2])
    subprocess.run([os.path.join(ROOT, "oracle", "yko"), "count", "-k27", "-o", tab, f1], check=True, stderr=subprocess.DEVNULL)
    h = L.yak_ch_restore(tab.encode())
    assert h

    def dump():
        out = C.POINTER(C.c_uint8)()
        n = L.yakamd_dump_mem(h, C.byref(out))
        data = C.string_at(out, n)
        C.CDLL(None).free(out)
        return data
    L.yak_recount(f2.encode(), h)
    got = dump()
    ho = O.yko_ch_restore(tab.encode())
    O.yko_ch_clear(ho)
    o = oracle.copt(k=27)
    assert O.yko_count_file(f2.encode(), C.byref(o), ho)
    want = oracle.dump_bytes(ho)
    O.yko_ch_destroy(ho)
    assert got == want
    L.yak_recount(str(tmp_path / "missing.fa").encode(), h)          # unreadable file: table untouched
    assert dump() == want
    L.yak_ch_destroy(h)


@pytest.mark.gpu
def test_hist_and_setcnt_on_the_device(ya, oracle, synth):
    """yak_ch_hist (htab.c:145-169) and yak_ch_setcnt (htab.c:219-235) run on the device image"""
    L, O = ya.lib(), oracle.lib()
    img = synth(6000, g=30000, s=8)
    o = oracle.copt(k=31, bf_shift=24)
    ho = O.yko_count_protocol_mem(img, len(img), None, 0, C.byref(o))
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".yak") as f:
        assert O.yko_ch_dump(ho, f.name.encode()) == 0
        O.yko_ch_destroy(ho)
        h = L.yak_ch_restore(f.name.encode())
        ho = O.yko_ch_restore(f.name.encode())              # restore re-inserts in file order: compare restored with restored
    assert h
    want = (C.c_int64 * 1024)(); have = (C.c_int64 * 1024)()
    O.yko_ch_hist(ho, want)
    L.yak_ch_hist(h, have, 4)
    n_keys = sum(want)
    assert list(have) == list(want) and n_keys > 10000

    def dump():
        out = C.POINTER(C.c_uint8)()
        n = L.yakamd_dump_mem(h, C.byref(out))
        data = C.string_at(out, n)
        C.CDLL(None).free(out)
        return data
    for cnt in (1, 1023, 0):
        O.yko_ch_setcnt(ho, cnt)
        L.yak_ch_setcnt(h, cnt, 4)
        assert dump() == oracle.dump_bytes(ho)
        L.yak_ch_hist(h, have, 4)
        assert have[cnt] == n_keys and sum(have) == n_keys
    L.yak_ch_destroy(h); O.yko_ch_destroy(ho)


def _two_tables(ya, oracle, synth, tmp_path, k=25):
    """a and b: tables of two read sets over the same genome (b with more errors and fewer reads)"""
    import subprocess
    L, O = ya.lib(), oracle.lib()
    imgs = [synth(9000, g=50000, s=17, e=0.004), synth(4000, g=50000, s=17, e=0.02, first=9000)]
    out = []
    for j, img in enumerate(imgs):
        o = oracle.copt(k=k)
        ho = O.yko_count_protocol_mem(img, len(img), None, 0, C.byref(o))
        fn = str(tmp_path / f"t{j}.yak")
        assert O.yko_ch_dump(ho, fn.encode()) == 0
        O.yko_ch_destroy(ho)
        out.append(fn)
    return out


def _dump(L, h):
    out = C.POINTER(C.c_uint8)()
    n = L.yakamd_dump_mem(h, C.byref(out))
    data = C.string_at(out, n)
    C.CDLL(None).free(out)
    return data


@pytest.mark.gpu
@pytest.mark.parametrize("op", ["subtract", "isec", "tighten", "shrink_tighten", "merge", "merge_presize", "merge_range", "setcnt_merge"])
def test_set_operations_on_the_device(op, ya, oracle, synth, tmp_path):
    """yak_ch_subtract / isec / tighten / merge (htab.c:102-110, 246-347) against the same calls on the
    oracle, starting from restored tables (what `yak cntasm / subtract / isec` do, main.c:90-284)"""
    L, O = ya.lib(), oracle.lib()
    O.yko_ch_merge.argtypes = [C.POINTER(oracle.Ch), C.POINTER(oracle.Ch), C.c_int, C.c_int, C.c_int]
    O.yko_ch_subtract.argtypes = [C.POINTER(oracle.Ch)] * 2
    O.yko_ch_isec.argtypes = [C.POINTER(oracle.Ch)] * 2
    O.yko_ch_tighten.argtypes = [C.POINTER(oracle.Ch)]
    O.yko_ch_shrink.argtypes = [C.POINTER(oracle.Ch), C.c_int, C.c_int]
    fa, fb = _two_tables(ya, oracle, synth, tmp_path)
    h0, h1 = L.yak_ch_restore(fa.encode()), L.yak_ch_restore(fb.encode())
    o0, o1 = O.yko_ch_restore(fa.encode()), O.yko_ch_restore(fb.encode())
    assert h0 and h1
    h1_alive = True
    if op == "subtract":
        L.yak_ch_subtract(h0, h1, 4); O.yko_ch_subtract(o0, o1)
    elif op == "isec":
        L.yak_ch_isec(h0, h1, 4); O.yko_ch_isec(o0, o1)
    elif op == "tighten":
        L.yak_ch_tighten(h0); O.yko_ch_tighten(o0)
    elif op == "shrink_tighten":
        L.yak_ch_shrink(h0, 3, 1023, 4); O.yko_ch_shrink(o0, 3, 1023)
        assert _dump(L, h0) == oracle.dump_bytes(o0)
        L.yak_ch_tighten(h0); O.yko_ch_tighten(o0)
    elif op in ("merge", "merge_presize", "merge_range", "setcnt_merge"):
        if op == "setcnt_merge":                              # main.c:144-150: presence counting across samples
            L.yak_ch_setcnt(h0, 1, 4); O.yko_ch_setcnt(o0, 1)
        lo, hi = (2, 5) if op == "merge_range" else (0, 1023)
        L.yak_ch_merge(h0, h1, lo, hi, 4, 1 if op == "merge_presize" else 0)
        O.yko_ch_merge(o0, o1, lo, hi, 1 if op == "merge_presize" else 0)
        h1_alive = False
    assert _dump(L, h0) == oracle.dump_bytes(o0)
    assert h0.contents.tot == o0.contents.tot
    L.yak_ch_destroy(h0); O.yko_ch_destroy(o0)
    if h1_alive:
        assert _dump(L, h1) == oracle.dump_bytes(o1)          # the second operand is left alone
        L.yak_ch_destroy(h1); O.yko_ch_destroy(o1)


@pytest.mark.gpu
@pytest.mark.parametrize("sweeps", [0, 1], ids=["one_table", "tables_sharded_by_sweeps"])
@pytest.mark.parametrize("pre_resize", [0, 1], ids=["plain", "resize_before_merge"])
def test_cntasm_protocol_equals_reference_cli(pre_resize, sweeps, ya, oracle, tmp_path, monkeypatch):
    """`yak cntasm` (main.c:90-161) -- count each assembly, keep its unique k-mers, merge sample after
    sample, shrink, tighten, dump -- as the same call sequence on the library; the file must equal the
    one the reference binary writes (oracle/_ref/yak, where it travelled) and the oracle's.  With sweeps
    every yak_count() returns a table sharded over prefix ranges (what an unfiltered count of a large
    plain file does by itself): merge, shrink, setcnt, tighten and dump must take it shard by shard"""
    import subprocess
    from conftest import ROOT
    if sweeps:
        monkeypatch.setenv("YAKAMD_AUTO_SWEEP_GB", "0.000001")
        monkeypatch.delenv("YAKAMD_GPUS", raising=False)
    L, O = ya.lib(), oracle.lib()
    syn, ref = os.path.join(ROOT, "tools", "yaksynth"), os.path.join(ROOT, "oracle", "_ref", "yak")
    fas = []
    for j, (seed, e) in enumerate(((31, 0.0), (31, 0.004), (31, 0.008))):           # three "assemblies" of one genome
        fa = str(tmp_path / f"asm{j}.fa")
        subprocess.check_call([syn, "-a", "-n", "12", "-l", "20000", "-g", "150000", "-s", str(seed), "-e", str(e), "-N", "0.0002", "-o", fa])
        if j:                                                                        # different contigs per sample
            txt = open(fa).read().split(">")[1:]
            open(fa, "w").write("".join(">" + r for r in txt[j:] + txt[:j]))
        fas.append(fa)
    K, min_cnt, max_cnt, max_out, check_n = 21, 1, 1, 0, 10
    o = ya.CoptT(); L.yak_copt_init(C.byref(o)); o.k = K; o.chunk_size = 1900000000
    h = None
    oo = oracle.copt(k=K, chunk=1900000000); ho = None
    O.yko_ch_merge.argtypes = [C.POINTER(oracle.Ch), C.POINTER(oracle.Ch), C.c_int, C.c_int, C.c_int]
    O.yko_ch_shrink.argtypes = [C.POINTER(oracle.Ch), C.c_int, C.c_int]
    O.yko_ch_tighten.argtypes = [C.POINTER(oracle.Ch)]
    for i, fa in enumerate(fas):
        h1 = L.yak_count(fa.encode(), C.byref(o), None)
        g1 = O.yko_count_file(fa.encode(), C.byref(oo), None)
        assert h1 and g1
        if h is None:
            h, ho = h1, g1
            L.yak_ch_shrink(h, min_cnt, max_cnt, 4); L.yak_ch_setcnt(h, 1, 4)
            O.yko_ch_shrink(ho, min_cnt, max_cnt); O.yko_ch_setcnt(ho, 1)
        else:
            L.yak_ch_merge(h, h1, min_cnt, max_cnt, 4, pre_resize)
            O.yko_ch_merge(ho, g1, min_cnt, max_cnt, pre_resize)
        if i == len(fas) - 1 or (i + 1 > max_out and (i + 1) % check_n == 0):
            L.yak_ch_shrink(h, i + 1 - max_out, 1023, 4)
            O.yko_ch_shrink(ho, i + 1 - max_out, 1023)
    L.yak_ch_tighten(h); O.yko_ch_tighten(ho)
    got, want = _dump(L, h), oracle.dump_bytes(ho)
    assert got == want and h.contents.tot == ho.contents.tot and h.contents.tot > 1000
    if os.path.exists(ref):
        out = str(tmp_path / "ref.yak")
        subprocess.run([ref, "cntasm", f"-k{K}"] + (["-r"] if pre_resize else []) + ["-o", out] + fas, check=True, stderr=subprocess.DEVNULL)
        assert open(out, "rb").read() == got
    L.yak_ch_destroy(h); O.yko_ch_destroy(ho)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["first_sharded", "second_sharded", "both_sharded"])
@pytest.mark.parametrize("op", ["subtract", "isec", "merge_presize"])
def test_two_table_operations_on_sharded_tables(op, which, ya, oracle, tmp_path, monkeypatch):
    """yak_ch_subtract / isec / merge when yak_count() handed out tables sharded over prefix ranges (sweeps): the result
    must be the one the unsharded tables give (the oracle's)"""
    import subprocess
    from conftest import ROOT
    L, O = ya.lib(), oracle.lib()
    O.yko_ch_merge.argtypes = [C.POINTER(oracle.Ch), C.POINTER(oracle.Ch), C.c_int, C.c_int, C.c_int]
    O.yko_ch_subtract.argtypes = [C.POINTER(oracle.Ch)] * 2
    O.yko_ch_isec.argtypes = [C.POINTER(oracle.Ch)] * 2
    syn = os.path.join(ROOT, "tools", "yaksynth")
    fas = []
    for j, e in enumerate((0.001, 0.01)):
        fa = str(tmp_path / f"s{j}.fa")
        subprocess.check_call([syn, "-a", "-n", "10", "-l", "15000", "-g", "120000", "-s", "41", "-e", str(e), "-N", "0.0002", "-o", fa])
        fas.append(fa)
    o = ya.CoptT(); L.yak_copt_init(C.byref(o)); o.k = 23
    oo = oracle.copt(k=23)
    hs = []
    for j, fa in enumerate(fas):
        sharded = which == "both_sharded" or (which == "first_sharded") == (j == 0)
        if sharded:
            monkeypatch.setenv("YAKAMD_AUTO_SWEEP_GB", "0.000001")
        else:
            monkeypatch.setenv("YAKAMD_AUTO_SWEEP_GB", "0")
        hs.append(L.yak_count(fa.encode(), C.byref(o), None))
        assert hs[-1]
    gs = [O.yko_count_file(fa.encode(), C.byref(oo), None) for fa in fas]
    if op == "subtract":
        L.yak_ch_subtract(hs[0], hs[1], 4); O.yko_ch_subtract(gs[0], gs[1])
    elif op == "isec":
        L.yak_ch_isec(hs[0], hs[1], 4); O.yko_ch_isec(gs[0], gs[1])
    else:
        L.yak_ch_merge(hs[0], hs[1], 0, 1023, 4, 1); O.yko_ch_merge(gs[0], gs[1], 0, 1023, 1)
    assert _dump(L, hs[0]) == oracle.dump_bytes(gs[0])
    assert hs[0].contents.tot == gs[0].contents.tot and hs[0].contents.tot > 1000
    L.yak_ch_destroy(hs[0]); O.yko_ch_destroy(gs[0])
    if op != "merge_presize":
        assert _dump(L, hs[1]) == oracle.dump_bytes(gs[1])
        L.yak_ch_destroy(hs[1]); O.yko_ch_destroy(gs[1])


@pytest.mark.gpu
@pytest.mark.parametrize("cmd", ["subtract", "isec"])
def test_subtract_and_isec_protocols_equal_reference_cli(cmd, ya, oracle, synth, tmp_path):
    """`yak subtract` / `yak isec` (main.c:217-284): restore, set operation, tighten, dump"""
    import subprocess
    from conftest import ROOT
    L = ya.lib()
    ref = os.path.join(ROOT, "oracle", "_ref", "yak")
    fa, fb = _two_tables(ya, oracle, synth, tmp_path)
    h0, h1 = L.yak_ch_restore(fa.encode()), L.yak_ch_restore(fb.encode())
    (L.yak_ch_subtract if cmd == "subtract" else L.yak_ch_isec)(h0, h1, 8)
    L.yak_ch_destroy(h1)
    L.yak_ch_tighten(h0)
    out = str(tmp_path / "dev.yak")
    assert L.yak_ch_dump(h0, out.encode()) == 0
    L.yak_ch_destroy(h0)
    want = str(tmp_path / "want.yak")
    tool = ref if os.path.exists(ref) else None
    if tool:
        subprocess.run([tool, cmd, "-o", want, fa, fb], check=True, stderr=subprocess.DEVNULL)
    else:                                                     # the oracle's own call sequence
        O = oracle.lib()
        O.yko_ch_subtract.argtypes = [C.POINTER(oracle.Ch)] * 2; O.yko_ch_isec.argtypes = [C.POINTER(oracle.Ch)] * 2
        O.yko_ch_tighten.argtypes = [C.POINTER(oracle.Ch)]
        o0, o1 = O.yko_ch_restore(fa.encode()), O.yko_ch_restore(fb.encode())
        (O.yko_ch_subtract if cmd == "subtract" else O.yko_ch_isec)(o0, o1)
        O.yko_ch_tighten(o0)
        assert O.yko_ch_dump(o0, want.encode()) == 0
    a, b = open(out, "rb").read(), open(want, "rb").read()
    assert a == b and len(a) > 16 + 8 * 1024


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["triobin", "sexchr", "triobin_sliced", "sexchr_sliced", "load_all_into_existing", "load_all_into_existing_sliced"])
def test_restore_core_flag_modes_on_the_device(family, ya, oracle, synth, tmp_path, monkeypatch, knob):
    """yak_ch_restore_core modes 2-6 (htab.c:396-476): flag sets of several .yak files ORed into one
    table -- the loads of `yak triobin` (main.c) and `yak sexchr`; bytes against the oracle (itself pinned
    on the reference's library, tests/test_oracle_vs_ref.py)"""
    L, O = ya.lib(), oracle.lib()
    O.yko_ch_restore_core.restype = C.POINTER(oracle.Ch)
    O.yko_ch_restore_core.argtypes = [C.POINTER(oracle.Ch), C.c_char_p, C.c_int, C.c_int, C.c_int]
    fa, fb = _two_tables(ya, oracle, synth, tmp_path)
    img = synth(5000, g=50000, s=18, e=0.006)
    o = oracle.copt(k=25)
    hc = O.yko_count_protocol_mem(img, len(img), None, 0, C.byref(o))
    fc = str(tmp_path / "t2.yak")
    assert O.yko_ch_dump(hc, fc.encode()) == 0
    O.yko_ch_destroy(hc)
    if family.endswith("_sliced"):                           # several passes per file, as files with >= 2^28 (2^22) selected k-mers need
        knob("YAKAMD_LOAD_SLICE", "777")
    # YAK_LOAD_ALL into a table that already holds keys (htab.c:436-448): present keys stay as they are, new ones keep their saved count
    steps = [(2, fa), (3, fb)] if family.startswith("triobin") else [(4, fa), (5, fb), (6, fc)] if family.startswith("sexchr") else [(1, fa), (1, fb), (1, fc), (1, fa)]
    h, ho = None, None
    for mode, fn in steps:
        h = L.yak_ch_restore_core(h, fn.encode(), C.c_int(mode), C.c_int(2), C.c_int(5))
        ho = O.yko_ch_restore_core(ho, fn.encode(), mode, 2, 5)
        assert h and ho
        assert _dump(L, h) == oracle.dump_bytes(ho), (family, mode)
    assert not L.yak_ch_restore_core(None, fa.encode(), C.c_int(3), C.c_int(2), C.c_int(5))     # htab.c:413: needs a table
    assert L.yak_ch_get(h, 0) in (-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 15)                       # lookups work on the flag table
    L.yak_ch_destroy(h); O.yko_ch_destroy(ho)


@pytest.mark.gpu
def test_concurrent_get_after_restore(ya, oracle, synth, tmp_path):
    """yak_ch_get from several threads at once, the first calls racing on the host-mirror refresh
    (the reference's kt_for workers do exactly that: qv.c:59, triobin.c)"""
    import threading
    import numpy as np
    L, O = ya.lib(), oracle.lib()
    fa, _ = _two_tables(ya, oracle, synth, tmp_path)
    h, ho = L.yak_ch_restore(fa.encode()), O.yko_ch_restore(fa.encode())
    img = synth(300, g=50000, s=17, e=0.01)
    hh = np.empty(len(img), dtype=np.uint64); tt = np.empty(len(img), dtype=np.uint32)
    m = O.yko_extract_pos(25, img, len(img), hh.ctypes.data, tt.ctypes.data)
    want = [O.yko_ch_get(ho, int(x)) for x in hh[:m]]
    got = [None] * 8

    def work(j):
        got[j] = [L.yak_ch_get(h, int(x)) for x in hh[:m]]
    th = [threading.Thread(target=work, args=(j,)) for j in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert all(g == want for g in got) and any(v > 0 for v in want) and any(v < 0 for v in want)
    L.yak_ch_destroy(h); O.yko_ch_destroy(ho)


@pytest.mark.parametrize("k,bf", [(31, 0), (31, 22), (21, 0), (41, 20)])
def test_packed_image_feed_equals_ascii_feed(k, bf, ya, oracle, synth):
    """yakamd_feed_packed_dev (2-bit codes + validity bits, 0.375 B per base) must give what the ASCII image gives --
    count.c:28-43 keeps nothing else of a base -- for both passes of the protocol, Ns and read ends included, and with
    the image cut where a tile does not end (n not a multiple of 16 / 32 / 4096)"""
    L, O = ya.lib(), oracle.lib()
    img = synth(4001, g=20000, s=9, N=0.004)
    img = img[:len(img) - 7]
    n = len(img)
    d_a = L.yakamd_dev_alloc(n + 64); assert d_a
    nw = (n + 31) // 32
    d_c = L.yakamd_dev_alloc(8 * nw + 64); d_v = L.yakamd_dev_alloc(4 * nw + 64); assert d_c and d_v
    buf = (C.c_char * n).from_buffer_copy(img)
    assert L.yakamd_memcpy_h2d(d_a, buf, n) == 0
    assert L.yakamd_pack_bases_dev(d_a, n, d_c, d_v, None) == 0
    # the packer itself against a host restatement
    codes = (C.c_uint32 * (2 * nw))(); valid = (C.c_uint32 * nw)()
    assert L.yakamd_memcpy_d2h(codes, d_c, 8 * nw) == 0 and L.yakamd_memcpy_d2h(valid, d_v, 4 * nw) == 0
    nt4 = {65: 0, 97: 0, 67: 1, 99: 1, 71: 2, 103: 2, 84: 3, 116: 3}
    for j in list(range(0, 200)) + list(range(n - 70, n)):
        c = nt4.get(img[j])
        assert (valid[j >> 5] >> (j & 31) & 1) == (c is not None)
        if c is not None:
            assert (codes[j >> 4] >> (2 * (j & 15)) & 3) == c
    tp = ya.Table(k, 10, 4, bf); ta = ya.Table(k, 10, 4, bf)
    tp.count_pass_packed(1, [(d_c, d_v, n, 0)]); ta.count_pass(1, [(d_a, n, 0)])
    assert tp.dump_bytes() == ta.dump_bytes() and tp.tot == ta.tot
    oc = oracle.copt(k=k, bf_shift=bf)
    o = O.yko_count_mem(img, n, C.byref(oc), None)
    assert tp.dump_bytes() == oracle.dump_bytes(o)
    if bf:
        tp.destroy_bf(); tp.clear(); tp.count_pass_packed(0, [(d_c, d_v, n, 0)])
        O.yko_ch_destroy_bf(o); O.yko_ch_clear(o); O.yko_count_mem(img, n, C.byref(oc), o)
        assert tp.dump_bytes() == oracle.dump_bytes(o)
    # the packer of the host side (yak_count()'s parser threads) writes the same words, and its image fed from host memory -- in three pieces, as
    # parsed segments arrive, each packed on its own -- counts to the same bytes
    pk = ya.pack_bases_host(img)
    cb = (8 * nw + 15) & ~15
    assert pk[:8 * nw] == bytes(codes) and pk[cb:cb + 4 * nw] == bytes(valid)
    th = ya.Table(k, 10, 4, bf)
    cuts = [0, img.index(b"\n", n // 3) + 1, img.index(b"\n", 2 * n // 3) + 1, n]
    th.count_pass_packed_host(1, [(img[a:b], a) for a, b in zip(cuts, cuts[1:])])
    assert th.dump_bytes() == ta.dump_bytes() and th.tot == ta.tot
    # ... and the way yak_count() hands a window over: the pieces (an empty one among them) in one feed, each ending at a multiple of 32 positions
    tw = ya.Table(k, 10, 4, bf)
    tw.count_pass_packed_host(1, [(img[a:b], 0) for a, b in zip(cuts, cuts[1:])][:1] + [(b"", 0)] + [(img[a:b], 0) for a, b in zip(cuts, cuts[1:])][1:], as_one=True)
    assert tw.dump_bytes() == ta.dump_bytes() and tw.tot == ta.tot
    tp.close(); ta.close(); th.close(); tw.close(); O.yko_ch_destroy(o)
    for d in (d_a, d_c, d_v):
        L.yakamd_dev_free(d)


@pytest.mark.parametrize("env", [dict(), dict(YAKAMD_BF_DEFER="0")], ids=["filter_rebuilt_from_retained_records", "filter_written_by_the_pass"])
def test_filter_survives_a_pass_that_kept_it_in_lds(env, ya, oracle, synth, monkeypatch, knob):
    """a filtered pass that retains its records does not write its 2^bf_shift filter bits back (yak_ch_destroy_bf usually follows, main.c:55); a later
    create_new call on the same table -- yak_ch_insert_list here (htab.c:51-78 consults the filter, htab.c:63-65) -- must still meet exactly the bits
    every instance of the pass set (bbf.c:34-40): the filter is rebuilt from the retained records before they are dropped"""
    for k_, v in env.items():
        knob(k_, v)
    L, O = ya.lib(), oracle.lib()
    img = synth(6000, g=30000, s=77)
    t = ya.Table(31, 10, 4, 22)
    assert L.yakamd_retain_input(t.h, 1) == 0
    d = L.yakamd_dev_alloc(len(img) + 64)
    assert L.yakamd_memcpy_h2d(d, img, len(img)) == 0
    t.count_pass(1, [(d, len(img), 0)])
    assert L.yakamd_retained_instances(t.h) > 0
    o = O.yko_count_mem(img, len(img), C.byref(oracle.copt(31, 10, 4, 22, 10000000)), None)
    lists = _lists(oracle, 9, 30, 400)
    for i, a in enumerate(lists):
        arr = (C.c_uint64 * len(a))(*a)
        assert L.yak_ch_insert_list(t.h, 1, len(a), arr) == O.yko_ch_insert_list(o, 1, len(a), arr), i
    assert t.dump_bytes() == oracle.dump_bytes(o)
    t.close(); O.yko_ch_destroy(o); L.yakamd_dev_free(d)
