"""GPU parity tests proper: the HIP path, called through the C ABI (include/yak.h +
include/yak_amd.h), must reproduce the reference's .yak bytes -- against the committed golden
vectors (made by the reference) and against the oracle on seeded inputs.  Bit-exact: integer work."""
import ctypes as C
import hashlib
import json
import os
import struct
import subprocess

import pytest

from conftest import GOLD, ROOT, args_to_opts, image_for_case

pytestmark = pytest.mark.gpu
CASES = sorted(json.load(open(os.path.join(GOLD, "manifest.json"))).keys())


@pytest.fixture(scope="module")
def ya():
    import yak_amd
    L = yak_amd.lib()
    assert L.yakamd_device_count() >= 1, "GPU tests need an MI355X; the engine has no CPU fallback"
    return yak_amd


@pytest.mark.parametrize("name", CASES)
def test_golden_vectors(name, ya, synth, manifest):
    desc = manifest[name]
    img = image_for_case(desc, synth)
    data, _ = ya.count_protocol_host(img, **args_to_opts(desc["args"]))
    assert len(data) == desc["size"]
    assert hashlib.md5(data).hexdigest() == desc["md5"]
    if desc["stored"]:
        assert data == open(os.path.join(GOLD, name + ".yak"), "rb").read()


OPTS = [dict(k=32), dict(k=33), dict(k=47), dict(k=63), dict(k=63, bf_shift=22), dict(k=40, pre=12, bf_shift=25), dict(k=31), dict(k=21), dict(k=11), dict(k=5), dict(k=31, pre=12), dict(k=31, pre=14, bf_shift=27),
        dict(k=31, bf_shift=19), dict(k=31, bf_shift=21), dict(k=31, bf_shift=26), dict(k=25, bf_shift=20, n_hash=9),
        dict(k=31, bf_shift=22, n_hash=70), dict(k=31, bf_shift=12), dict(k=31, bf_shift=10), dict(k=31, bf_shift=24, n_hash=1)]


@pytest.mark.parametrize("opt", OPTS, ids=lambda o: "-".join(f"{k}{v}" for k, v in o.items()))
def test_vs_oracle_short_reads(opt, ya, oracle, synth):
    img = synth(5000, g=25000, s=31)
    got, tot = ya.count_protocol_host(img, **opt)
    want, wtot = oracle.count_protocol_mem(img, **opt)
    assert tot == wtot
    assert got == want


def test_vs_oracle_long_contigs_with_n(ya, oracle, synth):
    img = synth(40, l=30000, g=400000, s=3, N=0.001)        # long-sequence path (SURVEY section 5, config 4 shape)
    for opt in (dict(k=21), dict(k=31, bf_shift=25), dict(k=55), dict(k=36, bf_shift=24)):
        assert ya.count_protocol_host(img, **opt)[0] == oracle.count_protocol_mem(img, **opt)[0]


def test_second_pass_on_a_different_file(ya, oracle, synth):
    a, b = synth(4000, g=20000, s=5), synth(3000, g=20000, s=5, e=0.02, first=100000)
    got, _ = ya.count_protocol_host(a, bf_shift=24, buf2=b)
    want, _ = oracle.count_protocol_mem(a, bf_shift=24, buf2=b)
    assert got == want


EDGE = {
    "empty": b"",
    "only_separators": b"\n\n\nNNNN\n",
    "shorter_than_k": b"ACGTACGT\nACG\n",
    "exactly_k": b"ACGTTGCAAGGCTTAACCGGTTAACCGGATC\n",
    "all_n": b"N" * 500 + b"\n",
    "lower_case_and_u": b"acgtugcaaggcuuaaccgguuaaccggaucacgatcgatcgatcagctagctagctagcatcgatcg\n",
    "raw_0123_bytes": bytes([0, 1, 2, 3] * 40) + b"\n" + bytes([3, 2, 1, 0] * 40) + b"\n",
    "iupac": b"ACGTRYKMACGTACGTACGTACGTACGTTGCATGCATGCATGCAACGTSWACGTACGTTGCATGCATGCATGCAAACCGGTT\n",
    "palindromes": b"ACGT" * 60 + b"\n" + b"AATT" * 60 + b"\n",
    "saturation": b"A" * 2500 + b"\n" + b"T" * 900 + b"\n" + b"ACGTTGCA" * 300 + b"\n",
    "no_trailing_separator": b"ACGTTGCAAGGCTTAACCGGTTAACCGGATCGGATTACAGGATTTACA",
    "long_read_one_n": b"ACGTTGCAAGGCTTAACCGGTTAACCGGATCGGATTACAGGATTTACAGGCATCGATCGGGATATCGCGCTAGCTAGGCTAN" + b"GATTACA" * 30 + b"\n",
}


@pytest.mark.parametrize("name", sorted(EDGE))
@pytest.mark.parametrize("opt", [dict(k=31), dict(k=7), dict(k=31, bf_shift=20), dict(k=45)], ids=["k31", "k7", "k31b20", "k45"])
def test_edge_inputs(name, opt, ya, oracle):
    img = EDGE[name]
    got, tot = ya.count_protocol_host(img, **opt)
    want, wtot = oracle.count_protocol_mem(img, **opt)
    assert (got, tot) == (want, wtot)


def test_grow_on_existing_key(ya, oracle, manifest):
    """the same read twice doubles the capacity of sub-tables sitting at 75 % load"""
    one = image_for_case(manifest["one_read"], None)
    two = image_for_case(manifest["one_read_x2"], None)
    a, _ = ya.count_protocol_host(one)
    b, _ = ya.count_protocol_host(two)

    def caps(d):
        out, off = [], 16
        for _ in range(1024):
            cap, n = struct.unpack_from("<II", d, off)
            out.append((cap, n)); off += 8 + 8 * n
        return out
    ca, cb = caps(a), caps(b)
    assert any(x == (4, 3) and y == (8, 3) for x, y in zip(ca, cb))
    assert any(x == (0, 0) for x in ca)
    assert a == oracle.count_protocol_mem(one)[0] and b == oracle.count_protocol_mem(two)[0]


@pytest.mark.parametrize("env", [dict(YAKAMD_BATCH="4096"), dict(YAKAMD_BATCH="8192", YAKAMD_LASTPUT_TAIL="100"),
                                 dict(YAKAMD_BATCH="65536", YAKAMD_LASTPUT_TAIL="1"), dict(YAKAMD_MULTI_BITS="10")],
                         ids=["batch4k", "batch8k_tail100", "batch64k_tail1", "multi10"])
def test_device_batching_is_invisible(env, ya, oracle, synth, monkeypatch, knob):
    """cutting the stream into many device batches (accumulator growth + rehash, per-batch bloom
    phases, last-put fallback scan, tiny `multi` filter) must not change a byte -- the reference's
    independence of -K/-t"""
    img = synth(3000, g=15000, s=12)
    for k, v in env.items():
        knob(k, v)
    for opt in (dict(k=31), dict(k=31, bf_shift=20), dict(k=31, bf_shift=24)):
        assert ya.count_protocol_host(img, **opt)[0] == oracle.count_protocol_mem(img, **opt)[0]


def test_cli_drop_in(ya, oracle, tmp_path):
    """the C caller (yak-amd count == reference main_count) on real files: FASTQ, gz, multi-line FASTA"""
    fq = str(tmp_path / "r.fq")
    subprocess.check_call([os.path.join(ROOT, "tools", "yaksynth"), "-n", "3000", "-g", "15000", "-s", "8", "-o", fq])
    subprocess.check_call(["gzip", "-kf", fq])
    for args, inp in ((["-k31"], fq), (["-k31", "-b22"], fq + ".gz"), (["-k5"], os.path.join(GOLD, "inputs", "edge.fx"))):
        a, b = str(tmp_path / "a.yak"), str(tmp_path / "b.yak")
        subprocess.run([os.path.join(ROOT, "yak_amd", "yak-amd"), "count"] + args + ["-o", a, inp], check=True, stderr=subprocess.DEVNULL)
        subprocess.run([os.path.join(ROOT, "oracle", "yko"), "count"] + args + ["-o", b, inp], check=True, stderr=subprocess.DEVNULL)
        assert open(a, "rb").read() == open(b, "rb").read()
    # block gzip (bgzip / htslib): members are inflated by the parser's threads, the stream -- and the .yak -- stay the same
    from test_host_reader import write_bgzf
    big = str(tmp_path / "big.fq")
    subprocess.check_call([os.path.join(ROOT, "tools", "yaksynth"), "-n", "12000", "-g", "60000", "-s", "9", "-o", big])
    write_bgzf(big + ".gz", open(big, "rb").read(), sizes=[65280, 40000, 1])
    a, b = str(tmp_path / "a2.yak"), str(tmp_path / "b2.yak")
    r = subprocess.run([os.path.join(ROOT, "yak_amd", "yak-amd"), "count", "-k31", "-b24", "-t4", "-o", a, big + ".gz"], check=True, stderr=subprocess.PIPE)
    subprocess.run([os.path.join(ROOT, "oracle", "yko"), "count", "-k31", "-b24", "-o", b, big], check=True, stderr=subprocess.DEVNULL)
    assert open(a, "rb").read() == open(b, "rb").read()


@pytest.mark.parametrize("env", [dict(), dict(YAKAMD_BATCH="65536"), dict(YAKAMD_CNT2_WGS="3"), dict(YAKAMD_RETAIN2="0"), dict(YAKAMD_RETAIN2="0", YAKAMD_BATCH="65536"),
                                 dict(YAKAMD_RETAIN_GB="0"), dict(YAKAMD_RETAIN2="0", YAKAMD_COUNT_OWN="0"),
                                 dict(YAKAMD_CNT2_SMALL="-1"), dict(YAKAMD_CNT2_SMALL="1000000", YAKAMD_S2_BITS="0", big="1"), dict(YAKAMD_CNT2_SMALL="1000000", YAKAMD_CNT2_WGS="2"),
                                 dict(YAKAMD_LC_FLAT="1"), dict(YAKAMD_S2_BITS="8", YAKAMD_P3_MIN="3", YAKAMD_P3_LOW="4")],
                         ids=["subbucket_records", "subbucket_records_many_batches", "subbucket_records_3_workgroups", "prefix_records", "prefix_records_many_batches",
                              "budget_refuses", "count_kernel_not_applicable",
                              "subbucket_records_big_lds_table", "subbucket_records_small_lds_table_overfull", "subbucket_records_small_lds_table",
                              "subbucket_records_flat_gather", "subbucket_records_level2_two_sweeps"])
@pytest.mark.parametrize("opt", [dict(k=31, bf_shift=24), dict(k=21, bf_shift=20), dict(k=31, bf_shift=22, n_hash=7)], ids=["k31b24", "k21b20", "k31b22H7"])
def test_second_pass_counts_the_records_the_first_pass_retained(opt, env, ya, oracle, synth, monkeypatch, knob):
    """main.c:53-57: both passes read the same input.  With yakamd_retain_input the create_new pass keeps its hashed k-mers on the
    device -- grouped by sub-bucket together with the keys every sub-bucket put into the table (k_cnt2) when the pass was one slice into
    an empty table, else grouped by prefix (k_img_count_own) -- and the count pass counts those (yakamd_count_retained), or reports
    that nothing usable was kept and takes the input again; either way the bytes are the oracle's, and the retained path must really
    have been taken where it applies"""
    L = ya.lib()
    env = dict(env)
    big = env.pop("big", None)                                 # ~390 keys per sub-bucket (one sub-bucket per sub-table): more than the small LDS table of k_cnt2 takes
    if big and opt["k"] != 31:
        pytest.skip("one size is enough")
    for k_, v in env.items():
        knob(k_, v)
    img = synth(40000, g=400000, s=23) if big else synth(9000, g=40000, s=19)
    want, wtot = oracle.count_protocol_mem(img, **opt)
    d = L.yakamd_dev_alloc(len(img) + 64)
    assert L.yakamd_memcpy_h2d(d, img, len(img)) == 0
    t = ya.Table(opt["k"], 10, opt.get("n_hash", 4), opt["bf_shift"])
    assert L.yakamd_retain_input(t.h, 1) == 0
    t.count_pass(1, [(d, len(img), 0)])
    kept = L.yakamd_retained_instances(t.h)
    assert (kept > 0) == (env.get("YAKAMD_RETAIN_GB") != "0")
    t.destroy_bf(); t.clear()
    assert L.yakamd_pass_begin(t.h, 0) == 0
    r = L.yakamd_count_retained(t.h)
    assert r == (0 if kept and "YAKAMD_COUNT_OWN" not in env else 1)
    if r:
        assert L.yakamd_feed_bases_dev(t.h, d, len(img), 0) == 0
    n_ins = L.yakamd_pass_end(t.h)
    assert n_ins == 0 and L.yakamd_retained_instances(t.h) == 0
    t.shrink(2, 1023)
    assert t.dump_bytes() == want and t.tot == wtot
    t.close()
    L.yakamd_dev_free(d)


def test_yak_count_reuses_the_first_pass_when_the_second_names_the_same_file(ya, oracle, tmp_path):
    """yak_count(fn, opt, NULL) of a filtered count keeps the k-mers; yak_count(fn, opt, h) on the SAME file counts them without
    reading it again (the log line says so); another file, a changed file or YAKAMD_NO_RETAIN take the ordinary path.  Same bytes"""
    fq, fq2 = str(tmp_path / "r.fq"), str(tmp_path / "r2.fq")
    yam, yko, syn = (os.path.join(ROOT, *p_) for p_ in (("yak_amd", "yak-amd"), ("oracle", "yko"), ("tools", "yaksynth")))
    subprocess.check_call([syn, "-n", "8000", "-g", "40000", "-s", "8", "-o", fq])
    subprocess.check_call([syn, "-n", "5000", "-g", "40000", "-s", "8", "-o", fq2])
    a, b = str(tmp_path / "a.yak"), str(tmp_path / "b.yak")
    for files, env, reused in (([fq], {}, True), ([fq, fq2], {}, False), ([fq], {"YAKAMD_NO_RETAIN": "1"}, False), ([fq, fq], {}, True)):
        r = subprocess.run([yam, "count", "-k31", "-b24", "-o", a] + files, check=True, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        subprocess.run([yko, "count", "-k31", "-b24", "-o", b] + files, check=True, stderr=subprocess.DEVNULL)
        assert open(a, "rb").read() == open(b, "rb").read()
        assert (b"kept on the device" in r.stderr) == reused


@pytest.mark.parametrize("bf", [0, 23])
def test_partitioned_exchange_path_on_one_gpu(bf, ya, oracle, synth):
    """same with the production exchange format: yakamd_partition_dev once per source, the owner
    receives per-source slices + per-prefix offsets through yakamd_feed_partitioned_dev"""
    L = ya.lib()
    world, P = 4, 1024
    per = P // world
    slices = [synth(1200, g=12000, s=5, first=r * 1200) for r in range(world)]
    nb = len(slices[0])
    parts_by_src = []
    for x in slices:                                     # what every source rank prepares
        d = L.yakamd_dev_alloc(nb); rec = L.yakamd_dev_alloc(nb * 16); hsh = L.yakamd_dev_alloc(nb * 8)
        assert L.yakamd_memcpy_h2d(d, x, nb) == 0
        bst = (C.c_uint64 * (P + 1))(); bst2 = (C.c_uint64 * (P + 1))()
        n = L.yakamd_partition_dev(31, 10, d, nb, rec, bst)
        assert n == bst[P] and n > 0
        assert L.yakamd_partition_hashes_dev(31, 10, d, nb, hsh, bst2) == n and list(bst2) == list(bst)
        parts_by_src.append((rec, list(bst), hsh))
        L.yakamd_dev_free(d)
    parts, tot = [], 0
    for r in range(world):
        lo, hi = r * per, (r + 1) * per
        t = ya.Table(31, 10, 4, bf)
        assert L.yakamd_set_shard(t.h, lo, hi) == 0

        def one_pass(create_new):
            assert L.yakamd_pass_begin(t.h, create_new) == 0
            for src, (rec, bst, hsh) in enumerate(parts_by_src):
                m = bst[hi] - bst[lo]
                offs = [0] * lo + [b - bst[lo] for b in bst[lo:hi + 1]] + [m] * (P - hi)
                ob = (C.c_uint64 * (P + 1))(*offs)
                if create_new:
                    feed = L.yakamd_feed_partitioned_lent_dev if (src + r) & 1 else L.yakamd_feed_partitioned_dev   # lent: used in place
                    assert feed(t.h, rec + 16 * bst[lo], m, ob, src * nb, nb) == 0
                else:                                   # count-existing pass: 8-byte hashes, same grouping
                    assert L.yakamd_count_partitioned_dev(t.h, hsh + 8 * bst[lo], m, ob) == 0
            n_ins = L.yakamd_pass_end(t.h)
            assert n_ins >= 0
            t.h.contents.tot += n_ins
        one_pass(1)
        if bf:
            t.destroy_bf(); t.clear(); one_pass(0); t.shrink(2, 1023)
        data = t.dump_bytes(); tot += t.tot; t.close()
        off = 16
        for p in range(P):
            cap, n = struct.unpack_from("<II", data, off)
            if lo <= p < hi:
                parts.append(data[off:off + 8 + 8 * n])
            off += 8 + 8 * n
    for rec, _, hsh in parts_by_src:
        L.yakamd_dev_free(rec); L.yakamd_dev_free(hsh)
    want, wtot = oracle.count_protocol_mem(b"".join(slices), k=31, bf_shift=bf)
    assert want[:16] + b"".join(parts) == want and tot == wtot


@pytest.mark.parametrize("bf", [0, 23])
def test_prefix_sharded_path_on_one_gpu(bf, ya, oracle, synth):
    """the multi-GPU data path with virtual ranks on one device: per-destination extraction
    (yakamd_extract_dev), owner-side yakamd_set_shard + yakamd_feed_hashed_dev with source-rank
    stream times; the concatenation of the ranks' sub-tables must equal the whole-input result"""
    from yak_amd import shard
    L = ya.lib()
    world, P = 4, 1024
    slices = [synth(1200, g=12000, s=5, first=r * 1200) for r in range(world)]
    nb = len(slices[0])
    d = []
    for x in slices:
        p = L.yakamd_dev_alloc(nb)
        assert p and L.yakamd_memcpy_h2d(p, x, nb) == 0
        d.append(p)
    xh, xt = L.yakamd_dev_alloc(nb * 8), L.yakamd_dev_alloc(nb * 4)
    parts, tot = [], 0
    for r in range(world):
        lo, hi = shard.owner_range(r, world, P)
        t = ya.Table(31, 10, 4, bf)
        assert L.yakamd_set_shard(t.h, lo, hi) == 0

        def one_pass(create_new):
            assert L.yakamd_pass_begin(t.h, create_new) == 0
            for src in range(world):
                n = L.yakamd_extract_dev(31, d[src], nb, xh, xt, 10, lo, hi, None)
                assert n >= 0
                if n:
                    assert L.yakamd_feed_hashed_dev(t.h, xh, xt, n, src * nb, nb) == 0
            n_ins = L.yakamd_pass_end(t.h)
            assert n_ins >= 0
            t.h.contents.tot += n_ins
        one_pass(1)
        if bf:
            t.destroy_bf(); t.clear(); one_pass(0); t.shrink(2, 1023)
        data = t.dump_bytes(); tot += t.tot; t.close()
        off = 16
        for p in range(P):
            cap, n = struct.unpack_from("<II", data, off)
            if lo <= p < hi:
                parts.append(data[off:off + 8 + 8 * n])
            off += 8 + 8 * n
    for p in d + [xh, xt]:
        L.yakamd_dev_free(p)
    want, wtot = oracle.count_protocol_mem(b"".join(slices), k=31, bf_shift=bf)
    assert want[:16] + b"".join(parts) == want and tot == wtot


@pytest.mark.parametrize("env", [dict(YAKAMD_FAST="0"), dict(YAKAMD_S2_BITS="0"), dict(YAKAMD_FAST_BUDGET="100000"),
                                 dict(YAKAMD_S2_BITS="3", YAKAMD_BATCH="32768"), dict(YAKAMD_PART_BITS="6"),
                                 dict(YAKAMD_S2_BITS="6"), dict(YAKAMD_S2_BITS="11", YAKAMD_CH2="4096"), dict(YAKAMD_S2_BITS="13"),
                                 dict(YAKAMD_P2_WC="0", YAKAMD_S2_BITS="6"),
                                 dict(YAKAMD_COUNT_OWN="0", YAKAMD_COUNT_LDS="0"), dict(YAKAMD_COUNT_OWN="0", YAKAMD_COUNT_LDS="0", YAKAMD_RNG_LOG="6"),
                                 dict(YAKAMD_COUNT_OWN="0", YAKAMD_COUNT_LDS="0", YAKAMD_RNG_LOG="5", YAKAMD_XLIST_CAP="0"),
                                 dict(YAKAMD_COUNT_OWN="0", YAKAMD_COUNT_LDS="0", YAKAMD_RNG_LOG="6", YAKAMD_XLIST_CAP="7"),
                                 dict(YAKAMD_COUNT_OWN="0", YAKAMD_COUNT_LDS="0", YAKAMD_COUNT_RNG="0"), dict(YAKAMD_COUNT_OWN="0"),
                                 dict(YAKAMD_OWN_LDS="18500", YAKAMD_OWN_MAXRB="12"), dict(YAKAMD_OWN_LDS="18500", YAKAMD_OWN_MAXRB="12", YAKAMD_XLIST_CAP="0"),
                                 dict(YAKAMD_OWN_LDS="19500", YAKAMD_OWN_MAXRB="12", YAKAMD_XLIST_CAP="5"), dict(YAKAMD_LC2="0"), dict(YAKAMD_LC2="0", YAKAMD_S2_BITS="4"),
                                 dict(YAKAMD_LC2_WGS="3"), dict(YAKAMD_YTAG="0"), dict(YAKAMD_YTAG="0", YAKAMD_OWN_LDS="18500", YAKAMD_OWN_MAXRB="12", YAKAMD_XLIST_CAP="5"), dict(YAKAMD_R2_SMALL_F="16"), dict(YAKAMD_R2_SMALL_F="1024"),
                                 dict(YAKAMD_REC8="0"), dict(YAKAMD_REC8_OUT="0"), dict(YAKAMD_REC8_OUT="0", YAKAMD_BATCH="16384"), dict(YAKAMD_BATCH="8192", YAKAMD_S2_BITS="3"),
                                 dict(YAKAMD_R2_SMALL_BITS="5"), dict(YAKAMD_R2_SMALL_BITS="5", YAKAMD_R2_SEG_LOG="10"), dict(YAKAMD_R2_SMALL_BITS="7", YAKAMD_R2_SEG_LOG="11"), dict(YAKAMD_REPLAY2="0"),
                                 dict(YAKAMD_R2_SMALL_BITS="5", YAKAMD_R2_SEG_LOG="10", YAKAMD_R2_PPART_G="3"), dict(YAKAMD_R2_SMALL_BITS="6", YAKAMD_R2_SEG_LOG="10", YAKAMD_R2_PPART_G="16", YAKAMD_R2_DBL="64"),
                                 dict(YAKAMD_FAST_BUDGET="3000000", YAKAMD_BATCH="65536"), dict(YAKAMD_FAST_BUDGET="1100000", YAKAMD_BATCH="65536"),
                                 dict(YAKAMD_FAST_BUDGET="40000000", YAKAMD_BATCH="1048576"),
                                 dict(YAKAMD_S2_BITS="0", YAKAMD_OVF_SCRATCH_WORDS="200000"), dict(YAKAMD_SLICE_SB="1", YAKAMD_BATCH="65536"),
                                 dict(YAKAMD_S2_BITS="8", YAKAMD_P3_MIN="3", YAKAMD_P3_LOW="4"), dict(YAKAMD_S2_BITS="9", YAKAMD_P3_MIN="5", YAKAMD_P3_LOW="3", YAKAMD_BATCH="65536"),
                                 dict(YAKAMD_S2_BITS="8", YAKAMD_P3_MIN="3", YAKAMD_P3_LOW="4", YAKAMD_REC8_OUT="0"), dict(YAKAMD_S2_BITS="8", YAKAMD_P3_MIN="3", YAKAMD_REC8="0"),
                                 dict(YAKAMD_S2_BITS="14", YAKAMD_CH2="4096"), dict(YAKAMD_LC_FLAT="1"), dict(YAKAMD_LC_FLAT="1", YAKAMD_S2_BITS="5", YAKAMD_BATCH="65536"), dict(YAKAMD_LC_FLAT="0"),
                                 dict(YAKAMD_TSORT="0"), dict(YAKAMD_TSORT="1"), dict(YAKAMD_TSORT="1", YAKAMD_TS_BITS="0"), dict(YAKAMD_TSORT="1", YAKAMD_TS_BITS="3", YAKAMD_LC_FLAT="1"), dict(YAKAMD_TSORT="1", YAKAMD_TS_BITS="6", YAKAMD_BATCH="65536"), dict(YAKAMD_TSORT="1", YAKAMD_TS_BITS="12", YAKAMD_CH2="4096"),
                                 dict(YAKAMD_TSORT="1", YAKAMD_TS_BITS="6", YAKAMD_TS_JOIN="3"), dict(YAKAMD_TSORT="1", YAKAMD_TS_BITS="4", YAKAMD_TS_JOIN="2", YAKAMD_TS_CAP="100"), dict(YAKAMD_TSORT="1", YAKAMD_TS_BITS="0", YAKAMD_TS_CAP="64"),
                                 dict(YAKAMD_POOL_FILL="166"), dict(YAKAMD_POOL_FILL="1", YAKAMD_POOL_VM="0"), dict(YAKAMD_POOL_VM_MIN="1048576", YAKAMD_POOL_FILL="166"), dict(YAKAMD_POOL_VM_MIN="4194304", YAKAMD_FAST_BUDGET="3000000", YAKAMD_BATCH="65536"),
                                 dict(YAKAMD_POOL_VM_MIN="1048576", YAKAMD_POOL_VM_ROOMY="0", YAKAMD_POOL_FILL="90"), dict(YAKAMD_POOL_VM_MIN="2097152", YAKAMD_POOL_VM_ROOMY="0", YAKAMD_FAST_BUDGET="3000000", YAKAMD_BATCH="65536")],
                         ids=["general_path", "lds_overflow_to_global", "budget_exceeded_midpass", "s2_3_multibatch", "part6_general",
                              "write_combined_level2", "write_combined_level2_wide", "write_combined_level2_segments", "plain_scatters",
                              "range_count_whole_table", "range_count_split", "range_count_cross_sweep", "range_count_short_list",
                              "count_with_device_atomics", "count_lds_rank_kernel",
                              "key_owning_count_32_slot_ranges", "key_owning_count_cross_sweep", "key_owning_count_short_list", "three_tier_lds_kernels", "three_tier_lds_kernels_crowded",
                              "lc2_three_persistent_workgroups", "pass2_plain_hashes", "pass2_plain_hashes_cross_sweep", "replay_prefix_16", "replay_prefix_1024",
                              "rec16_records", "tagged_in_rec16_out", "tagged_in_rec16_out_multibatch", "tagged_multibatch_s2_3",
                              "streaming_replay_from_32_slots", "streaming_replay_1k_slot_segments", "streaming_replay_2k_slot_segments", "k_replay_only",
                              "streaming_replay_keys_grouped_by_3_workgroups", "streaming_replay_keys_grouped_by_16_workgroups_6_wave_doubling", "pass_in_slices", "pass_in_single_batch_slices", "pass_in_two_slices",
                              "lds_overflow_to_global_in_groups", "slices_cut_by_sub_bucket_load",
                              "level2_two_sweeps", "level2_two_sweeps_plain_second_multibatch", "level2_two_sweeps_rec16_out", "level2_two_sweeps_rec16_in",
                              "level2_two_sweeps_16k_sub_buckets", "flat_gather", "flat_gather_multibatch", "one_workgroup_per_sub_table_gather",
                              "sort_stable_radix_passes", "sort_bitmap_ranks", "sort_one_bin_per_sub_table", "sort_8_bins_plain_scatter", "sort_64_bins_multibatch", "sort_4096_bins_in_segments",
                              "sort_bins_joined_by_8", "sort_joined_bins_beyond_the_stage", "sort_one_bin_in_windows",
                              "every_buffer_prefilled_with_0xa5", "every_buffer_zeroed_superblocks_only", "mapped_ranges_from_1_mib_prefilled", "mapped_ranges_from_4_mib_pass_in_slices",
                              "mapped_ranges_taken_apart_prefilled", "mapped_ranges_taken_apart_pass_in_slices"])
def test_every_insert_path_is_exact(env, ya, oracle, synth, monkeypatch, knob):
    """the exclusive-ownership LDS path, its global-scratch overflow variant, the accumulator path
    and the mid-pass switch between them all give the reference bytes"""
    img = synth(20000, g=90000, s=21)
    for k, v in env.items():
        knob(k, v)
    for opt in (dict(k=31), dict(k=31, bf_shift=22), dict(k=31, bf_shift=28), dict(k=21, bf_shift=20)):
        got, tot = ya.count_protocol_host(img, **opt)
        want, wtot = oracle.count_protocol_mem(img, **opt)
        assert (got == want, tot) == (True, wtot), opt


@pytest.mark.parametrize("env", [dict(), dict(YAKAMD_S2_BITS="5"), dict(YAKAMD_S2_BITS="8", YAKAMD_BATCH="65536")], ids=["auto", "s2_5", "s2_8_multibatch"])
def test_low_complexity_bursts(env, ya, oracle, synth, monkeypatch, knob):
    """homopolymer and short-period reads put thousands of consecutive k-mers into ONE partition
    bucket: the write-combining stacks overflow and the single-record path at the end of each run
    is taken; mixed with ordinary reads so the aligned groups and the singles share runs"""
    import random
    rnd = random.Random(5)
    parts = []
    for i in range(400):
        parts.append(rnd.choice([b"A" * 150, b"T" * 150, b"AC" * 75, b"ACG" * 50, b"AAAAC" * 30, b"G" * 149 + b"N"]))
    img = synth(6000, g=40000, s=33)
    mixed = img[:len(img) // 2] + b"N" + b"N".join(parts) + b"N" + img[len(img) // 2:] + b"N" + b"N".join(parts[:50])
    for k, v in env.items():
        knob(k, v)
    for opt in (dict(k=31), dict(k=31, bf_shift=28), dict(k=15, bf_shift=27), dict(k=33)):
        got, tot = ya.count_protocol_host(mixed, **opt)
        want, wtot = oracle.count_protocol_mem(mixed, **opt)
        assert (got == want, tot) == (True, wtot), opt


@pytest.mark.parametrize("env", [dict(), dict(YAKAMD_REPLAY_LDS="0"), dict(YAKAMD_REPLAY_LDS="8192"), dict(YAKAMD_REPLAY_LDS="8192", YAKAMD_PAR_REPLAY="0"),
                                 dict(YAKAMD_COUNT_OWN="0", YAKAMD_COUNT_LDS="0", YAKAMD_RNG_LOG="10"), dict(YAKAMD_COUNT_OWN="0", YAKAMD_COUNT_LDS="0", YAKAMD_RNG_LOG="7", YAKAMD_XLIST_CAP="100"),
                                 dict(YAKAMD_R2_SMALL_BITS="10", YAKAMD_R2_SEG_LOG="11"), dict(YAKAMD_R2_SMALL_BITS="9", YAKAMD_R2_SEG_LOG="10"), dict(YAKAMD_REPLAY2="0"),
                                 dict(YAKAMD_R2_SMALL_BITS="9", YAKAMD_R2_SEG_LOG="10", YAKAMD_R2_PPART_G="4"), dict(YAKAMD_R2_SMALL_BITS="8", YAKAMD_R2_SEG_LOG="11", YAKAMD_R2_DBL="62"), dict(YAKAMD_R2_SMALL_BITS="8", YAKAMD_R2_SEG_LOG="11", YAKAMD_R2_DBL="54"),
                                 dict(YAKAMD_OWN_LDS="30000", YAKAMD_OWN_MAXRB="12"), dict(YAKAMD_OWN_LDS="21000", YAKAMD_OWN_MAXRB="12", YAKAMD_XLIST_CAP="64"), dict(YAKAMD_OWN_LDS="21000", YAKAMD_OWN_MAXRB="12", YAKAMD_XLIST_CAP="64", YAKAMD_YTAG="0"), dict(YAKAMD_R2_SMALL_F="32"), dict(YAKAMD_REPLAY_LDS="32768"),
                                 dict(YAKAMD_REPLAY_LDS="2048"), dict(YAKAMD_REPLAY_LDS="1024", YAKAMD_REPLAY_THREADS="256"), dict(YAKAMD_REPLAY_LDS="2048", YAKAMD_DBG="256")],
                         ids=["lds_ranks", "global_ranks", "lds_16bit_ranks", "serial_doubling", "pass2_by_slot_ranges", "pass2_ranges_list_overflow",
                              "streaming_replay_2k_slot_segments", "streaming_replay_from_512_slots_1k_slot_segments", "k_replay_for_16k_slots",
                              "streaming_replay_keys_grouped_by_4_workgroups", "doubling_6_waves_2_walks", "doubling_5_waves_4_walks",
                              "pass2_key_owning_ranges", "pass2_key_owning_ranges_list_overflow", "pass2_key_owning_ranges_plain_hashes", "replay_prefix_32", "lds_keys_for_small_stages",
                              "segmented_lds_ranks", "segmented_lds_ranks_small", "global_ranks_for_large_stages"])
def test_replay_variants_on_large_subtables(env, ya, oracle, synth, monkeypatch, knob):
    """~7 M distinct k-mers (1x coverage): every sub-table grows to 16 Ki slots, so the layout replay
    goes through LDS-resident keys, 32- and 16-bit LDS owner ranks, global ranks, and the parallel
    doubling with its LDS base phase -- each variant must give the reference bytes"""
    img = synth(50000, g=8_000_000, s=77, e=0.0, N=0.0)
    for k, v in env.items():
        knob(k, v)
    dbg0 = (C.c_uint32 * 4)(); dbg1 = (C.c_uint32 * 4)()
    ya.lib().yakamd_debug_counters(dbg0)
    for opt in (dict(k=31), dict(k=27, bf_shift=30)):
        got, tot = ya.count_protocol_host(img, **opt)
        want, wtot = oracle.count_protocol_mem(img, **opt)
        assert (got == want, tot) == (True, wtot), opt
    ya.lib().yakamd_debug_counters(dbg1)
    if env.get("YAKAMD_REPLAY2") == "0":
        assert dbg1[2] == dbg0[2]
    elif "YAKAMD_R2_SMALL_BITS" in env:                       # 8 Ki-slot sub-tables: the streaming replay did them, and never handed one back
        assert dbg1[2] > dbg0[2] and dbg1[3] == dbg0[3], list(dbg1)


@pytest.mark.parametrize("seed", range(32))
def test_randomised_differential(seed, ya, oracle, synth, monkeypatch, knob):
    """seeded random points of the option space (k, prefix length, filter size, number of probes, read
    length, error and N rates, coverage) and of the engine's knobs: the protocol's bytes against the oracle"""
    import random
    rnd = random.Random(1000 + seed)
    k = rnd.choice([5, 11, 15, 21, 27, 31, 31, 31, 32, 33, 47, 63])
    pre = rnd.choice([10, 10, 10, 11, 12, 13, 14])
    bf = rnd.choice([0, 0, pre + 9, pre + 10, pre + 12, pre + 16, pre + 22, pre + 3])
    n_hash = rnd.choice([1, 2, 4, 4, 4, 7, 33])
    L_ = rnd.choice([max(k, 30), 75, 150, 150, 400])
    n = rnd.randrange(40, 6000)
    g = rnd.choice([300, 2000, 20000, n * L_ // 4 + 100])
    img = synth(n, l=L_, g=max(g, L_), s=seed + 5, e=rnd.choice([0.0, 0.005, 0.05]), N=rnd.choice([0.0, 0.0005, 0.02]))
    for key, vals in (("YAKAMD_BATCH", [None, None, "4096", "65536", "1048576"]), ("YAKAMD_S2_BITS", [None, None, None, "0", "2", "5", "9"]),
                      ("YAKAMD_REPLAY_LDS", [None, None, "0", "1024", "4096", "32768"]), ("YAKAMD_COUNT_LDS", [None, None, "0"]),
                      ("YAKAMD_COUNT_OWN", [None, None, "0"]), ("YAKAMD_OWN_LDS", [None, None, "18500", "24000"]), ("YAKAMD_OWN_MAXRB", [None, "12"]), ("YAKAMD_LC2", [None, None, None, "0"]),
                      ("YAKAMD_R2_SMALL_BITS", [None, "5", "6", "8"]), ("YAKAMD_R2_SEG_LOG", [None, "10", "11", "12"]), ("YAKAMD_REPLAY2", [None, None, None, "0"]), ("YAKAMD_REC8", [None, None, "0"]), ("YAKAMD_REC8_OUT", [None, None, "0"]),
                      ("YAKAMD_RNG_LOG", [None, "5", "8"]), ("YAKAMD_FAST", [None, None, None, "0"])):
        v = rnd.choice(vals)
        if v is not None:
            knob(key, v)
    got, tot = ya.count_protocol_host(img, k=k, pre=pre, n_hash=n_hash, bf_shift=bf)
    want, wtot = oracle.count_protocol_mem(img, k=k, pre=pre, n_hash=n_hash, bf_shift=bf)
    assert (got == want, tot) == (True, wtot), dict(k=k, pre=pre, bf=bf, n_hash=n_hash, n=n, L=L_, g=g)
