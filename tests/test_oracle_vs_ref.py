"""Where the reference binary is available (oracle/_ref/yak, built by `make -C oracle ref` from
/root/reference or carried prebuilt to the GPU box), run it next to the oracle CLI."""
import os
import subprocess

import pytest

from conftest import ROOT

REF = os.path.join(ROOT, "oracle", "_ref", "yak")
YKO = os.path.join(ROOT, "oracle", "yko")
SYN = os.path.join(ROOT, "tools", "yaksynth")

pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="prebuilt reference binary not present")

COMBOS = [["-k31"], ["-k31", "-b20"], ["-k31", "-b25"], ["-k21", "-K50k", "-t3"], ["-k41"], ["-k63", "-b24"],
          ["-k27", "-p12", "-b26"], ["-k31", "-b22", "-H40"]]


@pytest.mark.parametrize("args", COMBOS, ids=lambda a: "".join(a))
def test_cli_bytes_identical(args, tmp_path, oracle):
    fq = str(tmp_path / "r.fq")
    subprocess.check_call([SYN, "-n", "4000", "-l", "150", "-g", "20000", "-s", "77", "-o", fq])
    a, b = str(tmp_path / "a.yak"), str(tmp_path / "b.yak")
    subprocess.run([REF, "count"] + args + ["-o", a, fq], check=True, stderr=subprocess.DEVNULL)
    subprocess.run([YKO, "count"] + args + ["-o", b, fq], check=True, stderr=subprocess.DEVNULL)
    assert open(a, "rb").read() == open(b, "rb").read()


def test_two_input_files(tmp_path, oracle):
    f1, f2 = str(tmp_path / "1.fq"), str(tmp_path / "2.fq")
    subprocess.check_call([SYN, "-n", "4000", "-g", "20000", "-s", "5", "-o", f1])
    subprocess.check_call([SYN, "-n", "3000", "-g", "20000", "-s", "5", "-e", "0.01", "-o", f2])
    a, b = str(tmp_path / "a.yak"), str(tmp_path / "b.yak")
    subprocess.run([REF, "count", "-b24", "-o", a, f1, f2], check=True, stderr=subprocess.DEVNULL)
    subprocess.run([YKO, "count", "-b24", "-o", b, f1, f2], check=True, stderr=subprocess.DEVNULL)
    assert open(a, "rb").read() == open(b, "rb").read()


@pytest.mark.parametrize("pre_resize", [0, 1], ids=["plain", "resize_before_merge"])
def test_cntasm_sequence_on_the_oracle_equals_reference_cli(pre_resize, tmp_path, oracle):
    """pins the oracle's shrink / setcnt / merge / tighten (htab.c:102-110, 171-285) on `yak cntasm`
    (main.c:90-161): three assemblies, unique k-mers per sample, merged, shrunk, tightened, dumped"""
    import ctypes as C
    O = oracle.lib()
    O.yko_ch_merge.argtypes = [C.POINTER(oracle.Ch), C.POINTER(oracle.Ch), C.c_int, C.c_int, C.c_int]
    O.yko_ch_shrink.argtypes = [C.POINTER(oracle.Ch), C.c_int, C.c_int]
    O.yko_ch_tighten.argtypes = [C.POINTER(oracle.Ch)]
    fas = []
    for j, e in enumerate((0.0, 0.004, 0.008)):
        fa = str(tmp_path / f"asm{j}.fa")
        subprocess.check_call([SYN, "-a", "-n", "10", "-l", "15000", "-g", "100000", "-s", "9", "-e", str(e), "-N", "0.0002", "-o", fa])
        fas.append(fa)
    K = 21
    oo = oracle.copt(k=K, chunk=1900000000)
    ho = None
    for i, fa in enumerate(fas):
        g1 = O.yko_count_file(fa.encode(), C.byref(oo), None)
        if ho is None:
            ho = g1
            O.yko_ch_shrink(ho, 1, 1); O.yko_ch_setcnt(ho, 1)
        else:
            O.yko_ch_merge(ho, g1, 1, 1, pre_resize)
        if i == len(fas) - 1:
            O.yko_ch_shrink(ho, i + 1, 1023)
    O.yko_ch_tighten(ho)
    out = str(tmp_path / "ref.yak")
    subprocess.run([REF, "cntasm", f"-k{K}"] + (["-r"] if pre_resize else []) + ["-o", out] + fas, check=True, stderr=subprocess.DEVNULL)
    assert open(out, "rb").read() == oracle.dump_bytes(ho)
    O.yko_ch_destroy(ho)


@pytest.mark.parametrize("cmd", ["subtract", "isec"])
def test_subtract_isec_sequence_on_the_oracle_equals_reference_cli(cmd, tmp_path, oracle):
    """`yak subtract` / `yak isec` (main.c:217-284) against the oracle's restore + set operation + tighten"""
    import ctypes as C
    O = oracle.lib()
    O.yko_ch_subtract.argtypes = [C.POINTER(oracle.Ch)] * 2; O.yko_ch_isec.argtypes = [C.POINTER(oracle.Ch)] * 2
    O.yko_ch_tighten.argtypes = [C.POINTER(oracle.Ch)]
    tabs = []
    for j, (n, e) in enumerate(((9000, 0.004), (4000, 0.02))):
        fq, t = str(tmp_path / f"r{j}.fq"), str(tmp_path / f"t{j}.yak")
        subprocess.check_call([SYN, "-n", str(n), "-l", "150", "-g", "50000", "-s", "17", "-e", str(e), "-o", fq])
        subprocess.run([REF, "count", "-k25", "-o", t, fq], check=True, stderr=subprocess.DEVNULL)
        tabs.append(t)
    out = str(tmp_path / "ref.yak")
    subprocess.run([REF, cmd, "-o", out] + tabs, check=True, stderr=subprocess.DEVNULL)
    o0, o1 = O.yko_ch_restore(tabs[0].encode()), O.yko_ch_restore(tabs[1].encode())
    (O.yko_ch_subtract if cmd == "subtract" else O.yko_ch_isec)(o0, o1)
    O.yko_ch_tighten(o0)
    assert open(out, "rb").read() == oracle.dump_bytes(o0)
    O.yko_ch_destroy(o0); O.yko_ch_destroy(o1)


def _three_tables(tmp_path):
    tabs = []
    for j, (n, e, s) in enumerate(((9000, 0.004, 17), (6000, 0.01, 17), (5000, 0.006, 18))):
        fq, t = str(tmp_path / f"r{j}.fq"), str(tmp_path / f"t{j}.yak")
        subprocess.check_call([SYN, "-n", str(n), "-l", "150", "-g", "50000", "-s", str(s), "-e", str(e), "-o", fq])
        subprocess.run([REF, "count", "-k25", "-o", t, fq], check=True, stderr=subprocess.DEVNULL)
        tabs.append(t)
    return tabs


@pytest.mark.parametrize("family", ["triobin", "sexchr"])
def test_restore_core_flag_modes_equal_reference_library(family, tmp_path, oracle):
    """yak_ch_restore_core modes 2-6 (htab.c:396-476) called in the reference's own shared library
    (oracle/_ref/libyakref.so) against the oracle's restatement: flag sets ORed into one table"""
    import ctypes as C
    from oracle.pyoracle import REF_LIB
    R, O = C.CDLL(REF_LIB), oracle.lib()
    R.yak_ch_restore_core.restype = C.c_void_p
    R.yak_ch_dump.argtypes = [C.c_void_p, C.c_char_p]
    O.yko_ch_restore_core.restype = C.POINTER(oracle.Ch)
    O.yko_ch_restore_core.argtypes = [C.POINTER(oracle.Ch), C.c_char_p, C.c_int, C.c_int, C.c_int]
    tabs = _three_tables(tmp_path)
    steps = [(2, tabs[0]), (3, tabs[1])] if family == "triobin" else [(4, tabs[0]), (5, tabs[1]), (6, tabs[2])]
    hr, ho = None, None
    for mode, fn in steps:
        hr = R.yak_ch_restore_core(C.c_void_p(hr), fn.encode(), C.c_int(mode), C.c_int(2), C.c_int(5))
        ho = O.yko_ch_restore_core(ho, fn.encode(), mode, 2, 5)
        assert hr and ho
    out = str(tmp_path / "ref.yak")
    assert R.yak_ch_dump(C.c_void_p(hr), out.encode()) == 0
    data = open(out, "rb").read()
    assert data == oracle.dump_bytes(ho) and len(data) > 16 + 8 * 1024
    assert O.yko_ch_restore_core(None, tabs[0].encode(), 3, 2, 5) is None or not O.yko_ch_restore_core(None, tabs[0].encode(), 3, 2, 5)
