"""The lookup-only path (`yak qv`, reference qv.c:34-135).
CPU: the oracle's restatement against the reference-produced golden vectors (tests/golden/qv.json)
and, where the prebuilt reference binary exists, against `yak qv` itself.
GPU: the library (yak_ch_restore + yak_qv through the C ABI, the `yak-amd qv` caller, and the two
device entry points on their own) against the oracle and the golden vectors."""
import ctypes as C
import hashlib
import json
import os
import subprocess

import pytest

from conftest import GOLD, ROOT

REF = os.path.join(ROOT, "oracle", "_ref", "yak")
YKO = os.path.join(ROOT, "oracle", "yko")
YAM = os.path.join(ROOT, "yak_amd", "yak-amd")
SYN = os.path.join(ROOT, "tools", "yaksynth")
QV = json.load(open(os.path.join(GOLD, "qv.json")))


def query_file(desc, tmp_path):
    d = desc["synth"]
    fn = str(tmp_path / "q.fa")
    a = [SYN, "-a", "-n", str(d["n"]), "-l", str(d["l"]), "-g", str(d["g"]), "-s", str(d["s"]), "-e", str(d["e"]), "-o", fn]
    if "N" in d:
        a += ["-N", str(d["N"])]
    subprocess.check_call(a)
    return fn


def check_against_golden(out_text, desc, oracle):
    ct, sq, ek = oracle.parse_qv_output(out_text)
    assert {str(c): v[1] for c, v in ct.items() if v[1]} == desc["cnt"]
    assert sq == desc["sq"]
    assert (len(ek), hashlib.md5("\n".join(ek).encode()).hexdigest()) == (desc["n_ek"], desc["ek_md5"])


@pytest.mark.parametrize("name", sorted(QV))
def test_oracle_qv_matches_reference_golden(name, oracle, tmp_path):
    desc = QV[name]
    out = subprocess.run([YKO, "qv"] + desc["args"] + [os.path.join(GOLD, desc["table"] + ".yak"), query_file(desc, tmp_path)],
                         check=True, stdout=subprocess.PIPE).stdout.decode()
    check_against_golden(out, desc, oracle)


@pytest.mark.skipif(not os.path.exists(REF), reason="prebuilt reference binary not present")
@pytest.mark.parametrize("args", [["-p", "-E"], ["-p", "-l", "2500"], ["-f", "0.88", "-p"], ["-K", "10k", "-p"]], ids=lambda a: "".join(a))
def test_oracle_qv_cli_equals_reference_cli(args, oracle, tmp_path):
    fq, fa, tab = str(tmp_path / "r.fq"), str(tmp_path / "a.fa"), str(tmp_path / "t.yak")
    subprocess.check_call([SYN, "-n", "5000", "-l", "150", "-g", "30000", "-s", "11", "-o", fq])
    subprocess.check_call([SYN, "-a", "-n", "25", "-l", "3000", "-g", "30000", "-s", "11", "-e", "0.004", "-N", "0.0005", "-o", fa])
    subprocess.run([REF, "count", "-k27", "-b24", "-o", tab, fq], check=True, stderr=subprocess.DEVNULL)
    a = oracle.parse_qv_output(subprocess.run([REF, "qv"] + args + [tab, fa], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode())
    b = oracle.parse_qv_output(subprocess.run([YKO, "qv"] + args + [tab, fa], check=True, stdout=subprocess.PIPE).stdout.decode())
    assert a == b and sum(v[1] for v in a[0].values()) > 0


# ------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def ya():
    import yak_amd
    if yak_amd.lib().yakamd_device_count() < 1:
        pytest.skip("no MI355X visible")
    return yak_amd


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(QV))
def test_device_qv_matches_reference_golden(name, ya, oracle, tmp_path):
    desc = QV[name]
    out = subprocess.run([YAM, "qv"] + desc["args"] + [os.path.join(GOLD, desc["table"] + ".yak"), query_file(desc, tmp_path)],
                         check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
    check_against_golden(out, desc, oracle)


@pytest.mark.gpu
@pytest.mark.parametrize("opt", [dict(), dict(min_len=2500), dict(min_frac=0.88), dict(min_frac=0.0, min_len=100), dict(chunk=10000)],
                         ids=["default", "min_len", "min_frac", "all_pass", "small_batches"])
@pytest.mark.parametrize("k,bf", [(27, 24), (21, 0), (15, 0)])
def test_yak_qv_through_the_c_abi(opt, k, bf, ya, oracle, tmp_path):
    fq, fa, tab = str(tmp_path / "r.fq"), str(tmp_path / "a.fa"), str(tmp_path / "t.yak")
    subprocess.check_call([SYN, "-n", "20000", "-l", "150", "-g", "100000", "-s", "13", "-o", fq])
    subprocess.check_call([SYN, "-a", "-n", "60", "-l", "4000", "-g", "100000", "-s", "13", "-e", "0.004", "-N", "0.0005", "-o", fa])
    subprocess.run([YKO, "count", f"-k{k}", f"-b{bf}", "-o", tab, fq], check=True, stderr=subprocess.DEVNULL)
    o2 = {k_: v for k_, v in opt.items() if k_ != "chunk"}
    want = oracle.qv_counts(tab, fa, **o2)
    assert ya.qv_counts(tab, fa, **opt) == want and sum(want) > 0


@pytest.mark.gpu
def test_yak_qv_reads_large_files_through_the_parallel_reader(ya, oracle, tmp_path, monkeypatch):
    """without -p / -E yak_qv needs nothing of a record but its bases: a file of more than 1 MB -- plain, ordinary gzip, wrapped FASTA with short
    and empty records among the long ones -- goes through the parallel reader (threads x windows), and the histogram is the one-thread reader's
    and the oracle's"""
    import gzip
    import random
    fq, fa, tab = str(tmp_path / "r.fq"), str(tmp_path / "a.fa"), str(tmp_path / "t.yak")
    subprocess.check_call([SYN, "-n", "20000", "-l", "150", "-g", "100000", "-s", "13", "-o", fq])
    subprocess.check_call([SYN, "-a", "-n", "700", "-l", "4000", "-g", "100000", "-s", "13", "-e", "0.004", "-N", "0.0005", "-o", fa])
    rnd = random.Random(4)
    recs = open(fa).read().split(">")[1:]
    wrapped = []
    for i, r in enumerate(recs):                                # wrap some records, put empty and short ones in between
        name, seq = r.split("\n", 1)
        seq = seq.replace("\n", "")
        if i % 3 == 0:
            seq = "\n".join(seq[j:j + 70] for j in range(0, len(seq), 70))
        wrapped.append(">" + name + "\n" + seq + "\n")
        if i % 50 == 7:
            wrapped.append(">empty%d\n\n>short%d\nACGTACGT\n" % (i, i))
    fa2 = str(tmp_path / "b.fa")
    open(fa2, "w").write("".join(wrapped))
    gz = str(tmp_path / "b.fa.gz")
    with gzip.open(gz, "wb") as f:
        f.write(open(fa2, "rb").read())
    assert os.path.getsize(fa2) > (2 << 20)
    subprocess.run([YKO, "count", "-k27", "-b24", "-o", tab, fq], check=True, stderr=subprocess.DEVNULL)
    ya.gz_tune(100000, 0, -1)
    try:
        for opt in (dict(), dict(min_len=2500), dict(chunk=300000)):
            o2 = {k_: v for k_, v in opt.items() if k_ != "chunk"}
            want = oracle.qv_counts(tab, fa2, **o2)
            assert sum(want) > 0
            for thr, win in (("1", None), ("4", None), ("7", "300000")):
                monkeypatch.setenv("YAKAMD_PARSE_THREADS", thr)
                if win:
                    monkeypatch.setenv("YAKAMD_PARSE_WINDOW", win)
                else:
                    monkeypatch.delenv("YAKAMD_PARSE_WINDOW", raising=False)
                assert ya.qv_counts(tab, fa2, **opt) == want, (opt, thr, win)
                assert ya.qv_counts(tab, gz, **opt) == want, (opt, thr, win, "gz")
    finally:
        ya.gz_tune(1 << 20, 4 << 20, 64 << 20)


@pytest.mark.gpu
def test_lookup_and_reduce_entry_points(ya, oracle, synth, tmp_path):
    """yakamd_lookup_dev / yakamd_qv_reduce_dev on their own: per-position counts against
    yko_ch_get on the oracle's table; ragged sequences incl. empty, shorter than k and runs of N"""
    import numpy as np
    L, O = ya.lib(), oracle.lib()
    K = 25
    reads = synth(3000, g=20000, s=3)
    o = oracle.copt(k=K)
    ho = O.yko_count_protocol_mem(reads, len(reads), None, 0, C.byref(o))
    tab = str(tmp_path / "t.yak")
    assert O.yko_ch_dump(ho, tab.encode()) == 0
    h = L.yak_ch_restore(tab.encode())
    assert h
    other = synth(40, g=20000, s=3, e=0.02)                     # same genome, more errors: absent k-mers
    seqs = [b"", b"ACGT", b"N" * 40] + [other[i * 151:i * 151 + 150] for i in range(40)] + [reads[:1500].replace(b"\n", b"A")]
    img = b"".join(x + b"\n" for x in seqs)
    off = np.cumsum([0] + [len(x) + 1 for x in seqs[:-1]]).astype(np.uint64)
    ln = np.array([len(x) for x in seqs], dtype=np.uint32)
    # expected per-position values from the oracle
    hh = np.empty(len(img), dtype=np.uint64); tt = np.empty(len(img), dtype=np.uint32)
    m = O.yko_extract_pos(K, img, len(img), hh.ctypes.data, tt.ctypes.data)
    want = np.full(len(img), 0xffff, dtype=np.uint16)
    for j in range(m):
        want[tt[j]] = max(0, O.yko_ch_get(ho, int(hh[j])))
    pad = (len(img) + 15) // 16 * 16
    d_b, d_t = L.yakamd_dev_alloc(pad), L.yakamd_dev_alloc(2 * pad)
    assert L.yakamd_memcpy_h2d(d_b, img + b"\n" * (pad - len(img)), pad) == 0
    assert L.yakamd_lookup_dev(h, d_b, len(img), d_t) == 0
    got = np.empty(len(img), dtype=np.uint16)
    assert L.yakamd_memcpy_d2h(got.ctypes.data, d_t, 2 * len(img)) == 0
    assert np.array_equal(got, want) and (want == 0).sum() > 0 and ((want > 0) & (want < 0xffff)).sum() > 0
    ns = len(seqs)
    d_off, d_len, d_tot, d_non0, d_hist = (L.yakamd_dev_alloc(8 * ns), L.yakamd_dev_alloc(4 * ns), L.yakamd_dev_alloc(4 * ns),
                                           L.yakamd_dev_alloc(4 * ns), L.yakamd_dev_alloc(8 * 1024))
    assert L.yakamd_memcpy_h2d(d_off, off.tobytes(), 8 * ns) == 0 and L.yakamd_memcpy_h2d(d_len, ln.tobytes(), 4 * ns) == 0
    for min_len, min_frac in ((0, 0.5), (100, 0.9), (0, 0.0)):
        assert L.yakamd_memcpy_h2d(d_hist, bytes(8 * 1024), 8 * 1024) == 0
        assert L.yakamd_qv_reduce_dev(h, d_t, d_off, d_len, ns, min_len, min_frac, d_tot, d_non0, d_hist) == 0
        tot, non0, hist = np.empty(ns, np.uint32), np.empty(ns, np.uint32), np.empty(1024, np.uint64)
        L.yakamd_memcpy_d2h(tot.ctypes.data, d_tot, 4 * ns); L.yakamd_memcpy_d2h(non0.ctypes.data, d_non0, 4 * ns)
        L.yakamd_memcpy_d2h(hist.ctypes.data, d_hist, 8 * 1024)
        eh = np.zeros(1024, np.uint64)
        for j in range(ns):
            v = want[int(off[j]):int(off[j]) + int(ln[j])]
            v = v[v != 0xffff]
            if ln[j] < min_len:
                assert tot[j] == 0xffffffff
                continue
            assert (tot[j], non0[j]) == (len(v), int((v > 0).sum()))
            if not (int((v > 0).sum()) < len(v) * min_frac):
                eh += np.bincount(v, minlength=1024).astype(np.uint64)
        assert np.array_equal(hist, eh)
    for p_ in (d_b, d_t, d_off, d_len, d_tot, d_non0, d_hist):
        L.yakamd_dev_free(p_)
    L.yak_ch_destroy(h); O.yko_ch_destroy(ho)


# ------------------------------------------------------------------------------------------ host statistics
@pytest.mark.skipif(not os.path.exists(REF), reason="prebuilt reference binary not present")
@pytest.mark.parametrize("cov,e_asm", [(40, 0.002), (25, 0.0005), (60, 0.01), (3, 0.002)], ids=["40x", "25x_clean", "60x_noisy", "3x_low"])
def test_qv_solve_prints_what_the_reference_prints(cov, e_asm, tmp_path):
    """yak_qv_solve (qv.c:146-244) is host arithmetic: fed with the very histograms the reference
    printed (CT columns 3 and 4), it must reproduce the reference's CT / FR / ER / CV / QV lines"""
    import yak_amd
    L = yak_amd.lib()
    G = 200000
    fq, fa, tab = str(tmp_path / "r.fq"), str(tmp_path / "a.fa"), str(tmp_path / "t.yak")
    subprocess.check_call([SYN, "-n", str(G * cov // 150), "-l", "150", "-g", str(G), "-s", "23", "-o", fq])
    subprocess.check_call([SYN, "-a", "-n", "40", "-l", "10000", "-g", str(G), "-s", "23", "-e", str(e_asm), "-N", "0.0001", "-o", fa])
    subprocess.run([REF, "count", "-k21", "-b28", "-o", tab, fq], check=True, stderr=subprocess.DEVNULL)
    out = subprocess.run([REF, "qv", tab, fa], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode().splitlines()
    ct = {int(f[1]): f for f in (l.split("\t") for l in out if l.startswith("CT\t"))}
    hist = (C.c_int64 * 1024)(*[int(ct[i][2]) for i in range(1024)])
    cnt = (C.c_int64 * 1024)(*[int(ct[i][3]) for i in range(1024)])
    qs = yak_amd.QstatT()
    L.yak_qv_solve(hist, cnt, 21, 0.00004, C.byref(qs))
    import math

    def f3(x):                                                # C's printf keeps the sign of a NaN, Python's % does not
        return ("-nan" if math.copysign(1.0, x) < 0 else "nan") if math.isnan(x) else "%.3f" % x
    mine = ["CT\t%d\t%d\t%d\t%s" % (i, hist[i], cnt[i], f3(qs.adj_cnt[i])) for i in range(1023, -1, -1)]
    mine += ["FR\t%.3g\t%.3g" % (qs.fpr_lower, qs.fpr_upper), "ER\t%d\t%s" % (qs.tot, f3(qs.err)), "CV\t%s" % f3(qs.cov),
             "QV\t%s\t%s" % (f3(qs.qv_raw), f3(qs.qv))]
    assert mine == [l for l in out if l[:2] in ("CT", "FR", "ER", "CV", "QV")]


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF), reason="prebuilt reference binary not present")
def test_yak_amd_qv_full_output_equals_reference(ya, tmp_path):
    """`yak-amd qv -p` against `yak qv -p`: every line, SQ lines as a sorted set"""
    G = 150000
    fq, fa, tab = str(tmp_path / "r.fq"), str(tmp_path / "a.fa"), str(tmp_path / "t.yak")
    subprocess.check_call([SYN, "-n", str(G * 35 // 150), "-l", "150", "-g", str(G), "-s", "29", "-o", fq])
    subprocess.check_call([SYN, "-a", "-n", "30", "-l", "8000", "-g", str(G), "-s", "29", "-e", "0.003", "-N", "0.0001", "-o", fa])
    subprocess.run([REF, "count", "-k21", "-b28", "-o", tab, fq], check=True, stderr=subprocess.DEVNULL)
    a = subprocess.run([REF, "qv", "-p", tab, fa], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode().splitlines()
    b = subprocess.run([YAM, "qv", "-p", tab, fa], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode().splitlines()
    assert sorted(l for l in a if l.startswith("SQ")) == sorted(l for l in b if l.startswith("SQ"))
    assert [l for l in a if not l.startswith("SQ")] == [l for l in b if not l.startswith("SQ")] and any(l.startswith("QV") for l in a)
