#!/usr/bin/env python3
"""One ordinary single-member gzip file written by several processes, the way pigz does it: every piece of the input is deflated on its own
(raw deflate, ended with a sync flush, primed with the 32 KiB before it as dictionary, so matches cross the pieces), the pieces are
joined, a final empty block, the CRC32 and the length close the member.  For timing runs on boxes without pigz:
    pgzip.py [-l level] [-p procs] [-b piece_bytes] in out.gz"""
import argparse
import multiprocessing as mp
import struct
import zlib

A = None


def piece(i):
    with open(A.src, "rb") as f:
        lo = i * A.b
        f.seek(max(0, lo - 32768))
        prev = f.read(lo - max(0, lo - 32768))
        data = f.read(A.b)
    c = zlib.compressobj(A.l, zlib.DEFLATED, -15, 8, zlib.Z_DEFAULT_STRATEGY, prev) if prev else zlib.compressobj(A.l, zlib.DEFLATED, -15)
    return c.compress(data) + c.flush(zlib.Z_SYNC_FLUSH), zlib.crc32(data), len(data)


def main():
    global A
    ap = argparse.ArgumentParser()
    ap.add_argument("-l", type=int, default=6); ap.add_argument("-p", type=int, default=mp.cpu_count()); ap.add_argument("-b", type=int, default=8 << 20)
    ap.add_argument("src"); ap.add_argument("dst")
    A = ap.parse_args()
    import os
    n = os.path.getsize(A.src)
    crc, tot = 0, 0
    with mp.Pool(A.p) as pool, open(A.dst, "wb") as out:
        out.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03")
        for comp, c, ln in pool.imap(piece, range((n + A.b - 1) // A.b), chunksize=1):
            out.write(comp)
            crc = zlib.crc32(b"", crc) if ln == 0 else _combine(crc, c, ln)
            tot += ln
        out.write(b"\x03\x00" + struct.pack("<II", crc & 0xffffffff, tot & 0xffffffff))


def _combine(crc1, crc2, len2):
    """zlib's crc32_combine (GF(2) matrix squaring), which the zlib module does not export"""
    def times(mat, vec):
        s, i = 0, 0
        while vec:
            if vec & 1:
                s ^= mat[i]
            vec >>= 1; i += 1
        return s

    def square(mat):
        return [times(mat, mat[i]) for i in range(32)]
    if len2 <= 0:
        return crc1
    odd = [0xedb88320] + [1 << i for i in range(31)]
    even = square(odd)
    odd = square(even)
    while True:
        even = square(odd)
        if len2 & 1:
            crc1 = times(even, crc1)
        len2 >>= 1
        if not len2:
            break
        odd = square(even)
        if len2 & 1:
            crc1 = times(odd, crc1)
        len2 >>= 1
        if not len2:
            break
    return crc1 ^ crc2


if __name__ == "__main__":
    main()
