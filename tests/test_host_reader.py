"""Host input stage: the product's FASTA/FASTQ reader -- general record reader
and the in-buffer fast path -- yields exactly the sequences the reference's kseq loop would
(count.c:93-96), checked against the oracle's reader (itself pinned on the reference through the
golden .yak files of the same literal inputs).  Host only: no device is touched."""
import gzip
import os
import random
import subprocess

import pytest

from conftest import GOLD, ROOT

SYN = os.path.join(ROOT, "tools", "yaksynth")


def same(fn, oracle, k=0):
    """general reader, in-buffer fast path, and the speculative parallel parser (several thread counts,
    windows small enough that guesses fall into every kind of line) all give the oracle's image"""
    import yak_amd
    want = oracle.read_image(fn, k)
    assert yak_amd.host_image(fn, k, fast=True) == want
    assert yak_amd.host_image(fn, k, fast=False) == want
    try:
        for thr, win in ((2, 0), (3, 70001), (8, 0), (5, 1 << 20), (7, 333 if os.path.getsize(fn) < 300000 else 40009)):
            os.environ["YAKAMD_PARSE_THREADS"] = str(thr)
            if win:
                os.environ["YAKAMD_PARSE_WINDOW"] = str(win)
            else:
                os.environ.pop("YAKAMD_PARSE_WINDOW", None)
            assert yak_amd.host_image(fn, k, fast=True) == want, (thr, win)
    finally:
        os.environ.pop("YAKAMD_PARSE_THREADS", None); os.environ.pop("YAKAMD_PARSE_WINDOW", None)
    return want


@pytest.mark.parametrize("name", ["edge.fx", "one3000.fa", "one3000x2.fa", "polya.fa"])
def test_literal_inputs(name, oracle):
    for k in (0, 5, 31):
        same(os.path.join(GOLD, "inputs", name), oracle, k)


def test_synthetic_fastq_fasta_and_gzip(oracle, tmp_path):
    fq, fa = str(tmp_path / "r.fq"), str(tmp_path / "c.fa")
    subprocess.check_call([SYN, "-n", "20000", "-l", "150", "-g", "100000", "-s", "3", "-o", fq])      # 6 MB: several 1 MiB buffers
    subprocess.check_call([SYN, "-a", "-n", "40", "-l", "70000", "-g", "100000", "-s", "3", "-o", fa])  # lines longer than... one buffer holds them
    a = same(fq, oracle, 31)
    assert a.count(b"\n") == 20000
    same(fa, oracle, 31)
    gz = str(tmp_path / "r.fq.gz")
    with gzip.open(gz, "wb") as f:
        f.write(open(fq, "rb").read())
    assert same(gz, oracle, 31) == a


def test_awkward_shapes(oracle, tmp_path):
    rnd = random.Random(11)

    def seq(n):
        return "".join(rnd.choice("ACGTN") for _ in range(n))
    recs = []
    for i in range(30000):                                     # ragged lengths: records straddle the buffer boundaries
        n = rnd.choice([0, 1, 5, 30, 31, 32, 150, 151, 400, 2000])
        s, style = seq(n), rnd.randrange(9)
        if style == 0:
            recs.append(f">f{i} comment here\n{s}\n")
        elif style == 1:                                       # wrapped FASTA
            recs.append(f">w{i}\n" + "".join(s[j:j + 60] + "\n" for j in range(0, n, 60)))
        elif style == 2:                                       # CRLF
            recs.append(f"@c{i}\r\n{s}\r\n+\r\n{'I' * n}\r\n")
        elif style == 3:                                       # quality line starting with '@', header with tab
            recs.append(f"@q{i}\tx\n{s}\n+\n{'@' + 'I' * (n - 1) if n else ''}\n")
        elif style == 4:                                       # wrapped FASTQ
            recs.append(f"@m{i}\n{s[:n // 2]}\n{s[n // 2:]}\n+m{i}\n{'I' * (n // 2)}\n{'I' * (n - n // 2)}\n")
        elif style == 5:                                       # blank lines and junk between records
            recs.append(f"@b{i}\n{s}\n+\n{'I' * n}\n\njunk line\n")
        elif style == 6:
            recs.append(f">g{i}\n\n{s}\n\n")
        else:
            recs.append(f"@r{i}\n{s}\n+\n{'I' * n}\n")
    body = "".join(recs)
    for tail in ("", "@last\nACGTACGTACGTACGTACGTACGTACGTACGTACGT", ">last\nACGTACGTACGTACGTACGTACGTACGTACGTACGT\n",
                 "@trunc\nACGTACGTACGTACGTACGTACGTACGTACGTACGT\n+\nIIII\n"):
        fn = str(tmp_path / "x.fx")
        open(fn, "w", newline="").write(body + tail)
        for k in (0, 31):
            same(fn, oracle, k)
    assert len(body) > 3 << 20
