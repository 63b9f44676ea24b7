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


def bases_runs(img):
    """the maximal runs of bases of an image, in stream order: all that count.c:28-43 sees of it (a position that is no base only ends a run)"""
    import re
    return re.sub(rb"[^ACGT]+", b"\n", img.upper().replace(b"U", b"T").replace(b"\0", b"A").replace(b"\1", b"C").replace(b"\2", b"G").replace(b"\3", b"T")).strip(b"\n")


def same(fn, oracle, k=0):
    """general reader, in-buffer fast path, and the speculative parallel parser (several thread counts,
    windows small enough that guesses fall into every kind of line) all give the oracle's image"""
    import yak_amd
    want = oracle.read_image(fn, k)
    assert yak_amd.host_image(fn, k, fast=True) == want
    assert yak_amd.host_image(fn, k, fast=False) == want
    runs = bases_runs(want)
    try:
        for thr, win in ((2, 0), (3, 70001), (8, 0), (5, 1 << 20), (7, 333 if os.path.getsize(fn) < 300000 else 40009)):
            os.environ["YAKAMD_PARSE_THREADS"] = str(thr)
            if win:
                os.environ["YAKAMD_PARSE_WINDOW"] = str(win)
            else:
                os.environ.pop("YAKAMD_PARSE_WINDOW", None)
            assert yak_amd.host_image(fn, k, fast=True) == want, (thr, win)
            pk = yak_amd.host_image_packed(fn, k)                  # what yak_count() really feeds: the windows packed by the parser threads
            assert pk is None or bases_runs(pk) == runs, (thr, win)
    finally:
        os.environ.pop("YAKAMD_PARSE_THREADS", None); os.environ.pop("YAKAMD_PARSE_WINDOW", None)
    return want


def same_gz(fn, want, k, tmp_path, chunks=(1 << 21, 40000, 3000), threads=(2, 8, 5), fronts=(64 << 20,)):
    """the file as ordinary gzip -- one member, several, with sync flushes -- through the parallel gzip reader (csrc/pgz.h) at chunk sizes
    from "one thread takes it all" down to "a chunk is a fraction of a deflate block", and with hardly any room in front of a batch for
    the record carried over: the image is the plain file's"""
    import zlib
    import yak_amd
    data = open(fn, "rb").read()
    forms = {"l6": gzip.compress(data, 6), "l1": gzip.compress(data, 1)}
    step = max(1, len(data) // 7)
    forms["members"] = b"".join(gzip.compress(data[i:i + step], 9) for i in range(0, len(data), step)) or gzip.compress(b"")
    c = zlib.compressobj(6, zlib.DEFLATED, 31)
    forms["flushed"] = b"".join(c.compress(data[i:i + 50000]) + c.flush(zlib.Z_SYNC_FLUSH) for i in range(0, len(data), 50000)) + c.flush()
    try:
        for name, raw in forms.items():
            gz = str(tmp_path / ("z_" + name + ".gz"))
            open(gz, "wb").write(raw)
            assert gzip.open(gz, "rb").read() == data
            for i, ch in enumerate(chunks):
                yak_amd.gz_tune(ch, 0, fronts[i % len(fronts)])
                os.environ["YAKAMD_PARSE_THREADS"] = str(threads[i % len(threads)])
                assert yak_amd.gz_inflate(gz, threads[i % len(threads)]) == data, (name, ch)
                assert yak_amd.host_image(gz, k, fast=True) == want, (name, ch)
                assert bases_runs(yak_amd.host_image_packed(gz, k)) == bases_runs(want), (name, ch)
    finally:
        os.environ.pop("YAKAMD_PARSE_THREADS", None)
        yak_amd.gz_tune(1 << 20, 4 << 20, 64 << 20)


@pytest.mark.parametrize("name", ["edge.fx", "one3000.fa", "one3000x2.fa", "polya.fa"])
def test_literal_inputs(name, oracle, tmp_path):
    for k in (0, 5, 31):
        want = same(os.path.join(GOLD, "inputs", name), oracle, k)
        same_gz(os.path.join(GOLD, "inputs", name), want, k, tmp_path, chunks=(1 << 21, 1024))


def test_synthetic_fastq_fasta_and_gzip(oracle, tmp_path):
    fq, fa = str(tmp_path / "r.fq"), str(tmp_path / "c.fa")
    subprocess.check_call([SYN, "-n", "20000", "-l", "150", "-g", "100000", "-s", "3", "-o", fq])      # 6 MB: several 1 MiB buffers
    subprocess.check_call([SYN, "-a", "-n", "40", "-l", "70000", "-g", "100000", "-s", "3", "-o", fa])  # lines longer than... one buffer holds them
    a = same(fq, oracle, 31)
    assert a.count(b"\n") == 20000
    same_gz(fq, a, 31, tmp_path)
    same_gz(fa, same(fa, oracle, 31), 31, tmp_path, fronts=(64 << 20, 1000, 0))
    gz = str(tmp_path / "r.fq.gz")
    with gzip.open(gz, "wb") as f:
        f.write(open(fq, "rb").read())
    assert same(gz, oracle, 31) == a


def test_named_pipes_deliver_the_whole_stream(oracle, tmp_path):
    """`yak count ... <(zcat reads.fq.gz)` (the reference's README) hands yak_count() a pipe by NAME (/dev/fd/NN): every byte must arrive, plain or
    gzipped, through the reader and through the reference-facing entry point's own image.  (Until round 6 the reader opened a second descriptor
    on a file zlib had called uncompressed -- behind the buffer zlib had already pulled out of the pipe: the first megabyte of the stream was lost.)"""
    import threading
    import yak_amd
    fq = str(tmp_path / "r.fq")
    subprocess.check_call([SYN, "-n", "20000", "-l", "150", "-g", "100000", "-s", "5", "-o", fq])      # 6 MB: several of zlib's buffers
    want = oracle.read_image(fq, 31)
    gz = str(tmp_path / "r.fq.gz")
    with gzip.open(gz, "wb", compresslevel=1) as f:
        f.write(open(fq, "rb").read())
    fifo = str(tmp_path / "pipe")
    os.mkfifo(fifo)
    for src in (fq, gz):
        for fast in (True, False):
            def feed():
                with open(fifo, "wb") as w:
                    w.write(open(src, "rb").read())
            t = threading.Thread(target=feed)
            t.start()
            try:
                got = yak_amd.host_image(fifo, 31, fast=fast)
            finally:
                t.join()
            assert got == want, (src, fast)
    # a short input whose writer has closed its end before the reader looks at the name a second time: every byte is in the pipe, nobody may open() the
    # FIFO again (that open would wait for a writer for ever: the parallel routes did so until round 6, for inputs that fit the pipe's buffer).  The reader
    # runs in a child with a deadline: a blocked open() must fail the test, not hang it
    import sys
    tiny = str(tmp_path / "tiny.fq")
    open(tiny, "wb").write(b"".join(open(fq, "rb").readlines()[:400]))
    tiny_gz = str(tmp_path / "tiny.fq.gz")
    with gzip.open(tiny_gz, "wb") as f:
        f.write(open(tiny, "rb").read())
    tiny_want = oracle.read_image(tiny, 31)
    out = str(tmp_path / "tiny.img")
    code = f"import sys; sys.path.insert(0, {ROOT!r}); import yak_amd; open({out!r}, 'wb').write(yak_amd.host_image({fifo!r}, 31))"
    for src in (tiny, tiny_gz):
        for thr in ("1", "4"):
            def feed():
                with open(fifo, "wb") as w:
                    w.write(open(src, "rb").read())
            pr = subprocess.Popen([sys.executable, "-c", code], env=dict(os.environ, YAKAMD_PARSE_THREADS=thr))
            t = threading.Thread(target=feed)
            t.start()
            t.join()
            try:
                pr.wait(timeout=60)
            except subprocess.TimeoutExpired:
                pr.kill()
                raise AssertionError(("the reader blocked on the FIFO", src, thr))
            assert pr.returncode == 0 and open(out, "rb").read() == tiny_want, (src, thr)


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
            want = same(fn, oracle, k)
        same_gz(fn, want, 31, tmp_path, chunks=(1 << 21, 30000), threads=(3, 8))
    assert len(body) > 3 << 20


def write_bgzf(fn, data, block=65280, level=6, eof_block=True, sizes=None):
    """block gzip as bgzip / htslib write it: members of <= 64 KiB with the 'BC' extra field holding the member size - 1"""
    import struct
    import zlib
    out, off, i = [], 0, 0
    while off < len(data) or (off == 0 and not out):
        n = sizes[i % len(sizes)] if sizes else block
        chunk = data[off:off + n]
        off += len(chunk); i += 1
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = c.compress(chunk) + c.flush()
        bsize = 12 + 6 + len(comp) + 8
        out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize - 1) + comp + struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))
        if not chunk:
            break
    if eof_block:
        out.append(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    open(fn, "wb").write(b"".join(out))


@pytest.mark.parametrize("no_libdeflate", [False, True], ids=["libdeflate_if_present", "zlib_inflate"])
def test_block_gzip_is_parsed_in_parallel(no_libdeflate, oracle, tmp_path, monkeypatch, capfd):
    """a BGZF file goes through the parallel parser (its threads inflate the blocks their segment touches); the image is the
    plain file's, whatever the block sizes, with and without the trailing empty block, and plain gzip still streams"""
    import yak_amd
    if no_libdeflate:
        monkeypatch.setenv("YAKAMD_NO_LIBDEFLATE", "1")
    fq = str(tmp_path / "r.fq")
    subprocess.check_call([SYN, "-n", "20000", "-l", "150", "-g", "100000", "-s", "5", "-o", fq])
    data = open(fq, "rb").read()
    want = oracle.read_image(fq, 31)
    for name, kw in (("a.fq.gz", {}), ("b.fq.gz", dict(sizes=[1, 65536, 700, 33333, 2], eof_block=False)), ("c.fq.gz", dict(block=4096, level=1))):
        gz = str(tmp_path / name)
        write_bgzf(gz, data, **kw)
        assert gzip.open(gz, "rb").read() == data               # what we wrote is valid gzip
        assert same(gz, oracle, 31) == want
    # the parallel path was really taken (and is really off for plain gzip)
    monkeypatch.setenv("YAKAMD_PARSE_THREADS", "4"); monkeypatch.setenv("YAKAMD_VERBOSE", "1")
    capfd.readouterr()
    assert yak_amd.host_image(str(tmp_path / "a.fq.gz"), 31, fast=True) == want
    assert "BGZF blocks inflated by the parser threads" in capfd.readouterr().err
    plain = str(tmp_path / "p.fq.gz")
    with gzip.open(plain, "wb") as f:
        f.write(data)
    assert yak_amd.host_image(plain, 31, fast=True) == want
    assert "BGZF" not in capfd.readouterr().err
    # a corrupt block ends the stream there instead of inventing data: flip a byte in the middle of the payload of a.fq.gz
    raw = bytearray(open(str(tmp_path / "a.fq.gz"), "rb").read())
    raw[len(raw) // 2] ^= 0x55
    bad = str(tmp_path / "bad.fq.gz"); open(bad, "wb").write(bytes(raw))
    got = yak_amd.host_image(bad, 31, fast=True)
    assert want.startswith(got[:got.rfind(b"\n", 0, len(got) - 1) + 1][:1000]) and len(got) < len(want)


def test_long_fasta_records_are_stripped_by_several_threads(oracle, tmp_path):
    """a record whose body passes 1 MB (a chromosome) is stripped of its line ends by the parser threads straight from the mapped file
    (FxReader::bulk_body): wrapped at 60 / 61 / ragged widths, CRLF, blank lines, a '+' or '@' line that ends the body the way kseq.h:209
    says, short records around it, no final newline -- every shape against the oracle's reader"""
    rnd = random.Random(5)

    def body(n, width, eol="\n", ragged=False):
        out, i = [], 0
        while i < n:
            w = rnd.randint(1, width) if ragged else width
            out.append("".join(rnd.choice("ACGT") for _ in range(min(w, n - i))) + eol)
            i += w
        return "".join(out)
    cases = {
        "wrap60.fa": ">a desc\n" + body(2_500_000, 60) + ">b\nACGTACGTAC\n>c\n" + body(1_300_000, 61) + ">d\nACGT\n",
        "crlf.fa": ">a\r\n" + body(1_400_000, 70, "\r\n") + ">b\r\n" + body(50, 70, "\r\n"),
        "ragged_blank.fa": ">a\n" + body(700_000, 90, ragged=True) + "\n\n" + body(900_000, 90, ragged=True) + ">b\n" + body(2_000_000, 80, ragged=True).rstrip("\n"),
        "plus_in_fasta.fa": ">a\n" + body(1_500_000, 60) + "+\n" + "I" * 100 + "\n>b\n" + body(1_200_000, 60) + "@c\n" + body(100, 60),
        "one_line.fa": ">a\n" + body(3_000_000, 3_000_000) + ">b\n" + body(10, 10),
    }
    for name, text in cases.items():
        fn = str(tmp_path / name)
        open(fn, "w", newline="").write(text)
        for k in (0, 31):
            want = same(fn, oracle, k)
        same_gz(fn, want, 31, tmp_path, chunks=(100000, 20000), threads=(4, 8), fronts=(64 << 20, 4096))   # the carried record outgrows the room in front


def test_gzip_reader_on_streams_that_are_not_text_or_not_whole(oracle, tmp_path):
    """the parallel gzip reader against zlib itself: bytes that are not text (no block start is ever accepted: the stitch decodes all of it),
    stored and fixed-Huffman blocks, a header with a name and a comment, trailing garbage (ignored, as gzread ignores it), a truncated file
    (every complete symbol is delivered, as gzread delivers it) -- and a flipped bit or a wrong CRC fail the reader instead of yielding bytes"""
    import io
    import zlib
    import yak_amd
    rnd = random.Random(3)
    text = "".join("@r%d\n%s\n+\n%s\n" % (i, "".join(rnd.choice("ACGT") for _ in range(100)), "".join(chr(33 + rnd.randrange(41)) for _ in range(100))) for i in range(12000)).encode()
    noise = bytes(rnd.getrandbits(8) for _ in range(200000))

    def z(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
        c = zlib.compressobj(level, zlib.DEFLATED, 31, 9, strategy)
        return c.compress(data) + c.flush()
    named = io.BytesIO()
    with gzip.GzipFile(filename="reads.fq", mode="wb", fileobj=named, compresslevel=6) as f:
        f.write(text)
    cases = {"noise": (z(noise + text[:300000] + noise), noise + text[:300000] + noise), "stored": (z(text, 0), text), "fixed": (z(text[:500000], 6, zlib.Z_FIXED), text[:500000]),
             "named": (named.getvalue(), text), "garbage": (z(text) + b"this is not a gzip member", text), "empty": (z(b""), b"")}
    try:
        for name, (raw, data) in cases.items():
            fn = str(tmp_path / (name + ".gz"))
            open(fn, "wb").write(raw)
            for ch, thr in ((1 << 21, 4), (50000, 8), (2000, 3)):
                yak_amd.gz_tune(ch, 0, -1)
                assert yak_amd.gz_inflate(fn, thr) == data, (name, ch)
        whole = z(text)
        for cut in (len(whole) // 3, len(whole) - 9, len(whole) - 3, 25):
            fn = str(tmp_path / "cut.gz")
            open(fn, "wb").write(whole[:cut])
            d = zlib.decompressobj(31)
            want = d.decompress(whole[:cut])                       # what inflate() can make of it
            for ch, thr in ((1 << 21, 4), (20000, 8)):
                yak_amd.gz_tune(ch, 0, -1)
                assert yak_amd.gz_inflate(fn, thr) == want, (cut, ch)
        for at, what in ((len(whole) - 6, "CRC32"), (len(whole) - 2, "ISIZE"), (len(whole) // 2, "")):
            bad = bytearray(whole); bad[at] ^= 0x10
            fn = str(tmp_path / "bad.gz")
            open(fn, "wb").write(bytes(bad))
            for ch, thr in ((1 << 21, 4), (20000, 8)):
                yak_amd.gz_tune(ch, 0, -1)
                with pytest.raises(OSError) as e:
                    yak_amd.gz_inflate(fn, thr)
                assert what in str(e.value)
                os.environ["YAKAMD_PARSE_THREADS"] = "4"
                with pytest.raises(OSError):
                    yak_amd.host_image(fn, 31, fast=True)
        # what zlib refuses behind a member boundary is refused here too: a second member whose header has the magic but a method other than 8 or
        # reserved flags ("unknown compression method" / "unknown header flags set"), and a second member whose first match reaches back into the FIRST member's
        # bytes ("invalid distance too far back": a member starts with an empty window)
        def bits(*fields):                                        # (value, n_bits, msb_first) packed the deflate way
            acc = n = 0
            for v, w, msb in fields:
                for i in range(w):
                    acc |= ((v >> (w - 1 - i if msb else i)) & 1) << n; n += 1
            return acc.to_bytes((n + 7) // 8, "little")
        reach_back = bits((1, 1, False), (1, 2, False), (0b0000001, 7, True), (0, 5, True), (0, 7, True))     # final fixed block: length 3 at distance 1, end of block
        hdr = b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03"
        small = z(text[:200000])
        for name, tail in (("method", b"\x1f\x8b\x07" + hdr[3:] + small[10:]), ("flags", b"\x1f\x8b\x08\x20" + hdr[4:] + small[10:]),
                           ("reach", hdr + reach_back + zlib.crc32(b"\n\n\n").to_bytes(4, "little") + (3).to_bytes(4, "little"))):
            raw = small + tail
            with pytest.raises(zlib.error):
                zlib.decompressobj(31).decompress(small[len(small):] + tail)            # zlib refuses the member by itself ...
            fn = str(tmp_path / ("second_" + name + ".gz"))
            open(fn, "wb").write(raw)
            d = zlib.decompressobj(31)
            assert d.decompress(raw) == text[:200000] and d.eof and d.unused_data == tail   # ... which is what gzread() meets behind the good one (Python's own gzip module parses headers itself and lets reserved flags pass)
            for ch, thr in ((1 << 21, 4), (20000, 8), (3000, 3)):
                yak_amd.gz_tune(ch, 0, -1)
                with pytest.raises(OSError):
                    yak_amd.gz_inflate(fn, thr)
        assert yak_amd.gz_inflate(str(tmp_path / "cut.gz")[:-6] + "nope.gz", 4) is None
        open(str(tmp_path / "plain.txt"), "wb").write(text[:100000])
        assert yak_amd.gz_inflate(str(tmp_path / "plain.txt"), 4) is None          # not gzip: the caller keeps its own path
    finally:
        os.environ.pop("YAKAMD_PARSE_THREADS", None)
        yak_amd.gz_tune(1 << 20, 4 << 20, 64 << 20)


@pytest.mark.parametrize("wide", [True, False], ids=["avx2_if_present", "table_only"])
def test_host_packer_follows_the_packed_image_format(wide, tmp_path):
    """yakamd_pack_bases_host (what yak_count()'s parser threads run on what they parsed) against the format include/yak_amd.h states for
    yakamd_feed_packed_dev: base j at bits 2 (j % 16) of code word j / 16 by seq_nt4_table (ACGT, acgt, U / u = T, raw 0..3), one validity bit
    per base, zero codes where the bit is zero -- on reads with Ns and record separators, on every byte value, at lengths around the 32-base
    words (the process is a fresh one: the packer picks its AVX2 path once)"""
    import sys
    code = r"""
import sys, random
sys.path.insert(0, %r)
import yak_amd
if len(sys.argv) > 1:
    yak_amd.lib().yakamd_test_set(b"YAKAMD_NO_AVX2", 1)       # before the packer is first called: it picks its path once
nt4 = {65: 0, 97: 0, 67: 1, 99: 1, 71: 2, 103: 2, 84: 3, 116: 3, 85: 3, 117: 3, 0: 0, 1: 1, 2: 2, 3: 3}
rnd = random.Random(2)
def want(b):
    n = len(b); nw = (n + 31) // 32
    codes = [0] * (2 * nw); valid = [0] * nw
    for j, x in enumerate(b):
        c = nt4.get(x)
        if c is not None:
            valid[j >> 5] |= 1 << (j & 31); codes[j >> 4] |= c << (2 * (j & 15))
    cb = (8 * nw + 15) & ~15
    out = b"".join(w.to_bytes(4, "little") for w in codes) + bytes(cb - 8 * nw) + b"".join(w.to_bytes(4, "little") for w in valid)
    return out
cases = [b"", b"A", b"ACGTN\nacgtn\nUu\x00\x01\x02\x03\x04", bytes(range(256)) * 3]
for n in (31, 32, 33, 63, 64, 65, 1000, 4099):
    cases.append(bytes(rnd.choice(b"ACGTACGTACGTACGTNacgt\n") for _ in range(n)))
cases.append(bytes(rnd.choice(b"ACGT") for _ in range(70000)))
cases.append(bytes(rnd.getrandbits(8) for _ in range(50000)))
for b in cases:
    got = yak_amd.pack_bases_host(b)
    assert len(got) == yak_amd.lib().yakamd_packed_bytes(len(b)) and got == want(b), len(b)
print("ok")
""" % ROOT
    assert subprocess.run([sys.executable, "-c", code] + ([] if wide else ["table"]), check=True, stdout=subprocess.PIPE).stdout.strip() == b"ok"


@pytest.mark.parametrize("seed", range(60))
def test_gzip_reader_randomised(seed, tmp_path):
    """seeded random streams -- text of random line shapes, runs, repeats at every distance, bytes that are no text, empty members, members of a few
    bytes -- deflated at random levels, strategies, window sizes and flush points, read back with random chunk sizes, thread counts and carry
    room: the bytes are zlib's"""
    import zlib
    import yak_amd
    rnd = random.Random(7000 + seed)

    def piece():
        kind = rnd.randrange(7)
        n = rnd.choice([0, 1, 7, 300, 5000, 70000, 400000])
        if kind == 0:
            return bytes(rnd.choice(b"ACGT") for _ in range(n))
        if kind == 1:
            return (b"@read/%d\n" % rnd.randrange(10 ** 6) + bytes(rnd.choice(b"ACGTN") for _ in range(100)) + b"\n+\n" + bytes(33 + rnd.randrange(41) for _ in range(100)) + b"\n") * (n // 200 + 1)
        if kind == 2:
            return bytes([rnd.choice(b"ACGT\n")]) * n                              # a run: distance 1
        if kind == 3:
            unit = bytes(rnd.choice(b"ACGTacgtN\n") for _ in range(rnd.choice([2, 3, 31, 258, 4000, 32768, 40000])))
            return (unit * (n // len(unit) + 1))[:n]
        if kind == 4:
            return bytes(rnd.getrandbits(8) for _ in range(min(n, 30000)))        # no text: a searched start may not be accepted in here
        if kind == 5:
            return b"".join(b">c%d\n" % i + bytes(rnd.choice(b"ACGT") for _ in range(60)) + b"\n" for i in range(n // 70))
        return bytes(rnd.choice(b"ACGTACGTACGT!#$%&Ixyz\t\r\n") for _ in range(n))
    members, raw = [], []
    for _ in range(rnd.choice([1, 1, 2, 5])):
        data = b"".join(piece() for _ in range(rnd.randrange(1, 9)))
        c = zlib.compressobj(rnd.choice([0, 1, 1, 4, 6, 6, 9]), zlib.DEFLATED, 16 + rnd.choice([9, 12, 15, 15]), rnd.choice([1, 8, 9]),
                             rnd.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]))
        out, at = [], 0
        while at < len(data):
            n = rnd.choice([1, 100, 65536, 10 ** 6])
            out.append(c.compress(data[at:at + n])); at += n
            if rnd.random() < 0.3:
                out.append(c.flush(rnd.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH])))
        out.append(c.flush())
        members.append(b"".join(out)); raw.append(data)
    fn = str(tmp_path / "r.gz")
    open(fn, "wb").write(b"".join(members))
    want = b"".join(raw)
    assert gzip.open(fn, "rb").read() == want
    try:
        for _ in range(6):
            yak_amd.gz_tune(rnd.choice([1024, 3000, 20000, 150000, 1 << 20]), 0, rnd.choice([0, 100, 64 << 20]))
            assert yak_amd.gz_inflate(fn, rnd.choice([1, 2, 3, 8, 13])) == want
    finally:
        yak_amd.gz_tune(1 << 20, 4 << 20, 64 << 20)


@pytest.mark.parametrize("seed", range(6))
def test_gzip_reader_on_damaged_streams_follows_zlib(seed, tmp_path):
    """one to three flipped bits anywhere in a one-member and a many-member file, 120 damaged files per seed: whatever zlib makes of the stream decides --
    it inflates to the end (damage in a header field nobody checks, behind the last member): the same bytes; it ends early without an error (trailing bytes
    that are no member): the same bytes; it reports an error (data, CRC32, ISIZE, a refused member header): the reader fails, it never hands out other bytes"""
    import zlib
    import yak_amd
    rnd = random.Random(9100 + seed)
    text = "".join("@r%d\n%s\n+\n%s\n" % (i, "".join(rnd.choice("ACGT") for _ in range(100)), "".join(chr(33 + rnd.randrange(41)) for _ in range(100))) for i in range(1500)).encode()
    forms = [gzip.compress(text, 6), b"".join(gzip.compress(text[i:i + 70000], 5) for i in range(0, len(text), 70000))]

    def gzread(raw):
        """(bytes, state) the way gzread() goes through members: 'ok', 'trunc' (the input ends inside a member: what there is is delivered), 'error'"""
        out, data, first = b"", raw, True
        while data:
            d = zlib.decompressobj(31)
            try:
                out += d.decompress(data)
            except zlib.error:
                if not first and not (len(data) >= 2 and data[0] == 0x1f and data[1] == 0x8b):
                    return out, "ok"                                   # not a member: trailing garbage, ignored
                return out, "error"
            if not d.eof:
                return out, "trunc"
            data, first = d.unused_data, False
        return out, "ok"
    try:
        for it in range(120):
            raw = bytearray(rnd.choice(forms))
            for _ in range(rnd.choice([1, 1, 1, 2, 3])):
                raw[rnd.randrange(len(raw))] ^= 1 << rnd.randrange(8)
            raw = bytes(raw)
            want, state = gzread(raw)
            fn = str(tmp_path / "d.gz")
            open(fn, "wb").write(raw)
            ch, thr = rnd.choice([(1 << 21, 4), (20000, 8), (3000, 3)])
            yak_amd.gz_tune(ch, 0, -1)
            try:
                got, err = yak_amd.gz_inflate(fn, thr), None
            except OSError as e:
                got, err = None, str(e)
            if raw[:3] != b"\x1f\x8b\x08" or raw[3] & 0xe0:
                assert got is None                                     # the first header is not one the reader takes (the caller's zlib path reports it)
            elif state == "ok":
                assert got == want, (seed, it, ch, err)
            elif state == "trunc":
                assert got == want or err is not None, (seed, it, ch)
            else:
                assert err is not None, (seed, it, ch, None if got is None else len(got))
    finally:
        yak_amd.gz_tune(1 << 20, 4 << 20, 64 << 20)
