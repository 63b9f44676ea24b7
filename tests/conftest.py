import ctypes as C
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure(path, target):
    if not os.path.exists(os.path.join(ROOT, path)):
        subprocess.check_call(["make", "-s", "-C", ROOT, target])


# the settings the environment reaches; every other YAKAMD_ name is a test switch that only yakamd_test_set() reaches
PUBLIC_KNOBS = ['YAKAMD_VERBOSE', 'YAKAMD_DEVICE', 'YAKAMD_GPUS', 'YAKAMD_GPU_LIST', 'YAKAMD_AUTO_SWEEP_GB', 'YAKAMD_MGPU_CHUNK', 'YAKAMD_MGPU_NO_RCCL', 'YAKAMD_BATCH', 'YAKAMD_FAST_BUDGET', 'YAKAMD_NO_RETAIN', 'YAKAMD_RETAIN_GB', 'YAKAMD_PARSE_THREADS', 'YAKAMD_PARSE_WINDOW', 'YAKAMD_NO_LIBDEFLATE', 'YAKAMD_NO_PGZ', 'YAKAMD_NO_HOST_PACK']


@pytest.fixture
def knob(monkeypatch):
    """knob(name, value): a public knob goes into the environment, a test switch through the library's hook; both are undone after the test"""
    import yak_amd
    L = yak_amd.lib()

    def set_(name, value):
        if name in PUBLIC_KNOBS:
            monkeypatch.setenv(name, str(value))
        else:
            L.yakamd_test_set(name.encode(), int(value))
    yield set_
    L.yakamd_test_reset()


@pytest.fixture(scope="session")
def oracle():
    """the CPU restatement (oracle/liboracle.so) -- the checker, never the product"""
    _ensure("oracle/liboracle.so", "oracle")
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def synth():
    _ensure("tools/libyaksynth.so", "tools")
    L = C.CDLL(os.path.join(ROOT, "tools", "libyaksynth.so"))
    L.yaksynth_reads.restype = C.c_int64
    L.yaksynth_reads.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_uint64,
                                 C.c_double, C.c_double, C.c_int64, C.c_int]

    def make(n, l=150, g=None, s=1, e=0.005, N=0.0005, first=0, **_):
        g = g if g is not None else max(l, n * l // 30)
        buf = C.create_string_buffer(n * (l + 1))
        assert L.yaksynth_reads(buf, n, l, g, s, e, N, first, 4) == n * (l + 1)
        return buf.raw
    return make


@pytest.fixture(scope="session")
def manifest():
    return json.load(open(os.path.join(GOLD, "manifest.json")))


def parse_fastx_image(path):
    """tiny independent FASTA/FASTQ reader for the committed literal inputs -> memory image"""
    out = []
    lines = open(path, "rb").read().split(b"\n")
    i = 0
    while i < len(lines):
        ln = lines[i]
        if ln[:1] in (b">", b"@"):
            fastq = ln[:1] == b"@"
            i += 1
            seq = b""
            while i < len(lines) and lines[i][:1] not in (b">", b"@", b"+"):
                seq += lines[i].rstrip(b"\r")
                i += 1
            if fastq and i < len(lines) and lines[i][:1] == b"+":
                i += 1
                q = b""
                while i < len(lines) and len(q) < len(seq):
                    q += lines[i].rstrip(b"\r")
                    i += 1
                if len(q) != len(seq):
                    break
            out.append(seq)
        else:
            i += 1
    return b"".join(s + b"\n" for s in out)


def image_for_case(desc, synth):
    """memory image (reads separated by '\\n') of a golden case's input"""
    if "synth" in desc:
        return synth(**desc["synth"])
    return parse_fastx_image(os.path.join(GOLD, desc["file"]))


def args_to_opts(args):
    o = dict(k=31, pre=10, n_hash=4, bf_shift=0)
    for a in args:
        if a.startswith("-k"): o["k"] = int(a[2:])
        elif a.startswith("-p"): o["pre"] = int(a[2:])
        elif a.startswith("-b"): o["bf_shift"] = int(a[2:])
        elif a.startswith("-H"): o["n_hash"] = int(a[2:])
    return o
