"""The drop-in boundary exercised by the reference's OWN caller.

`oracle/_ref/yak_on_amd` is the reference's unmodified main.c / qv.c / inspect.c / triobin.c / ... object
files linked against yak_amd/libyak_amd.so instead of count.c, htab.c, bbf.c and misc.c (recipe:
oracle/Makefile `ref`).  Every sub-command that goes through yak.h must then
produce the bytes `oracle/_ref/yak` (the whole reference) produces: reference main.c:13-64 (count),
90-161 (cntasm), 163-215 (qv), 217-284 (isec / subtract via inspect), 66-88 (recount).
"""
import os
import subprocess

import pytest

from conftest import ROOT

REF = os.path.join(ROOT, "oracle", "_ref", "yak")
ONAMD = os.path.join(ROOT, "oracle", "_ref", "yak_on_amd")
SYN = os.path.join(ROOT, "tools", "yaksynth")

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(ONAMD)), reason="prebuilt reference binaries not present")]


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    import yak_amd
    if yak_amd.lib().yakamd_device_count() < 1:
        pytest.skip("no MI355X visible")
    d = tmp_path_factory.mktemp("refcli")
    f = {k: str(d / v) for k, v in dict(fq="r.fq", fq2="r2.fq", fa="a.fa", asm="asm.fa").items()}
    subprocess.check_call([SYN, "-n", "6000", "-l", "150", "-g", "30000", "-s", "21", "-o", f["fq"]])
    subprocess.check_call([SYN, "-n", "3500", "-l", "150", "-g", "30000", "-s", "21", "-o", f["fq2"]])      # the first 3500 reads of the same set
    f["fq3"] = str(d / "r3.fq")
    subprocess.check_call([SYN, "-n", "5000", "-l", "150", "-g", "30000", "-s", "22", "-o", f["fq3"]])      # another genome
    subprocess.check_call([SYN, "-a", "-n", "20", "-l", "3000", "-g", "30000", "-s", "21", "-e", "0.003", "-N", "0.0005", "-o", f["fa"]])
    subprocess.check_call([SYN, "-a", "-n", "6", "-l", "5000", "-g", "30000", "-s", "21", "-e", "0.0", "-o", f["asm"]])
    f["dir"] = str(d)
    return f


def both(args, out_name, files, stdout=False):
    """run the same command line through the whole reference and through its caller files on the library"""
    res = []
    for exe, tag in ((REF, "ref"), (ONAMD, "amd")):
        out = os.path.join(files["dir"], f"{out_name}.{tag}")
        a = [x.replace("@OUT@", out) for x in args]
        r = subprocess.run([exe] + a, check=True, stdout=subprocess.PIPE if stdout else None, stderr=subprocess.PIPE)
        res.append(r.stdout if stdout else open(out, "rb").read())
    return res


@pytest.mark.parametrize("args", [["-k31"], ["-k31", "-b24"], ["-k21", "-b20", "-t3"], ["-k27", "-p11", "-b26", "-H6", "-K", "100k"]], ids=lambda a: "".join(a))
def test_count(args, files):
    a, b = both(["count"] + args + ["-o", "@OUT@", files["fq"]], "c" + "".join(args), files)
    assert a == b and len(a) > 16 + 8 * 1024


def test_count_second_file_for_pass2(files):
    a, b = both(["count", "-k31", "-b24", "-o", "@OUT@", files["fq"], files["fq2"]], "c2", files)
    assert a == b


def test_count_from_process_substitution(files):
    """the command line the reference's README gives for gzipped short reads -- `yak count -b.. -o out <(zcat r.fq.gz) <(zcat r.fq.gz)` -- through the
    whole reference and through its caller files on the library: the pipes arrive by name (/dev/fd/NN), every byte of them must be counted"""
    gz = os.path.join(files["dir"], "ps.fq.gz")
    subprocess.run(f"gzip -1 -c {files['fq']} > {gz}", shell=True, check=True)
    res = []
    for exe, tag in ((REF, "ref"), (ONAMD, "amd")):
        out = os.path.join(files["dir"], f"ps.{tag}")
        subprocess.run(["bash", "-c", f"{exe} count -k31 -b24 -t4 -o {out} <(zcat {gz}) <(zcat {gz})"], check=True, stderr=subprocess.PIPE, timeout=300)
        res.append(open(out, "rb").read())
    assert res[0] == res[1] and len(res[0]) > 16 + 8 * 1024
    # ... and the same bytes as from the file itself
    a, b = both(["count", "-k31", "-b24", "-o", "@OUT@", files["fq"]], "psf", files)
    assert a == res[0] == b
    # yak qv reading its sequences from a pipe as well (qv.c:115: the same gzopen / kseq route)
    tab = os.path.join(files["dir"], "ps.ref")
    outs = []
    for exe in (REF, ONAMD):
        r = subprocess.run(["bash", "-c", f"{exe} qv -t2 {tab} <(cat {files['fa']})"], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        outs.append(sorted(l for l in r.stdout.decode().splitlines() if not l.startswith("CC")))
    assert outs[0] == outs[1] and any(l.startswith("QV") for l in outs[0])


def test_qv_and_inspect(files):
    tab = both(["count", "-k27", "-b24", "-o", "@OUT@", files["fq"]], "qt", files)
    assert tab[0] == tab[1]
    t = os.path.join(files["dir"], "qt.ref")
    a, b = both(["qv", "-p", "-t2", t, files["fa"]], "qv", files, stdout=True)
    keep = lambda o: sorted(l for l in o.decode().splitlines() if not l.startswith("CC"))
    assert keep(a) == keep(b) and any(l.startswith("QV") for l in keep(a))
    a, b = both(["inspect", t], "insp", files, stdout=True)
    assert a == b


def test_cntasm_recount_isec(files):
    a, b = both(["cntasm", "-k27", "-o", "@OUT@", files["asm"], files["fa"]], "ca", files)
    assert a == b
    tab = both(["count", "-k27", "-o", "@OUT@", files["fq"]], "rt", files)
    assert tab[0] == tab[1]
    t = os.path.join(files["dir"], "rt.ref")
    a, b = both(["recount", "-o", "@OUT@", t, files["fq2"]], "rc", files)
    assert a == b
    t2 = both(["count", "-k27", "-o", "@OUT@", files["fq2"]], "rt2", files)
    assert t2[0] == t2[1]
    t2f = os.path.join(files["dir"], "rt2.ref")
    a, b = both(["inspect", t, t2f], "insp2", files, stdout=True)       # inspect.c: yak_ch_restore + yak_ch_hist + yak_ch_get per key
    assert a == b
    for cmd in ("isec", "subtract"):                                      # main.c:217-284
        a, b = both([cmd, "-o", "@OUT@", t, t2f], cmd, files)
        assert a == b and len(a) > 16 + 8 * 1024
    a, b = both(["print", "-c", t2f], "print", files, stdout=True)        # main.c:286-327: yak_ch_getseq per sub-table
    assert a == b and a.count(b"\n") > 1000


def test_triobin(files):
    """triobin.c: both parental tables through yak_ch_restore_core's flag modes, reads classified with yak_ch_get"""
    pat = both(["count", "-k21", "-o", "@OUT@", files["fq"]], "pat", files)
    mat = both(["count", "-k21", "-o", "@OUT@", files["fq3"]], "mat", files)
    assert pat[0] == pat[1] and mat[0] == mat[1]
    a, b = both(["triobin", "-c1", "-d2", "-t1", os.path.join(files["dir"], "pat.ref"), os.path.join(files["dir"], "mat.ref"), files["fa"]], "tb", files, stdout=True)
    assert a == b and len(a.splitlines()) >= 20
