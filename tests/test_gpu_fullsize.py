"""Full-size goldens from the reference itself, run by the driver's GPU tier.

tests/golden/cfg2_full.json and cfg45_full.json hold the md5 + size of the `.yak` files the REFERENCE binary
(oracle/_ref/yak, compiled from /root/reference, tests/gen_golden_full.py) wrote for workloads tools/yaksynth
regenerates from a seed: nothing of the reference is needed at run time.  bench.py compares the device result
with them (`verify.equals_reference`) and exits non-zero on a mismatch; here each workload is one bench step.
  * no filter, 10 M reads: 217 M distinct k-mers, one pass, every instance a put-call (htab.c:66-69);
  * 30 M reads with the filter: 4.5 G stream positions, a pass counted in two slices (> 2^32 - 16 positions);
  * cfg4 at 1 Gb and 2 Gb (10 / 20 x 100 Mb contigs, k = 21, count.c:28-43,120-125): sub-tables of 2 / 4 Mi slots, streaming replay
    through five doublings beyond the LDS-resident sizes;
  * cfg5: `yak qv -p` CT histogram of 75 K x 20 kb reads against the cfg2 table (qv.c:34-135).
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("YAKAMD_SKIP_FULLSIZE") == "1", reason="YAKAMD_SKIP_FULLSIZE=1")]


def bench_line(*args):
    import yak_amd
    if yak_amd.lib().yakamd_device_count() < 1:
        pytest.skip("no MI355X visible")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-qv", "--no-pcie", "--no-packed", "--no-nofilter"] + list(args),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])


def test_no_filter_10m_reads_equals_reference():
    d = bench_line("--config", "nofilter")
    assert d["verify"]["equals_reference"] is True and d["verify"]["batch_independent"] is True
    assert d["final_distinct"] > 200e6


def test_30m_reads_equal_reference():
    """4.5 G stream positions: one slice of tagged records (their times are ranks, not positions), pass 2 on the records pass 1 retained"""
    d = bench_line("--reads", "30000000")
    assert d["verify"]["equals_reference"] is True
    assert d["roofline"]["pass2_input"].startswith("records retained")


def test_30m_reads_sliced_pass_equals_reference():
    """the same with {hash, position} records, whose 32-bit positions force two slices: state carries over exactly as between the reference's chunks"""
    d = bench_line("--reads", "30000000", "--knob", "YAKAMD_REC8=0")
    assert d["verify"]["equals_reference"] is True


def test_cfg4_1gb_assembly_equals_reference():
    d = bench_line("--config", "cfg4", "--contigs", "10", "--contig-len", "100000000")
    v = d["verify"]
    assert v["equals_reference"] is True and v["sum_sizes_equals_tot"] and v["count_mass_equals_instances"] and v["load_rule"]
    assert v["yak_size_bytes"] == 7998181472                   # = 16 + 8 P + 8 D, the reference's file size


def test_cfg4_2gb_assembly_equals_reference():
    """twice that: 20 x 100 Mb, 2.0 G distinct 21-mers in 4 Mi-slot sub-tables, 4096 sub-buckets per sub-table (the level-2 scatter in two
    sweeps), a 16 GB .yak -- the largest size the reference itself could be run on in the build container (35 GB of tables in 62 GB)"""
    d = bench_line("--config", "cfg4", "--contigs", "20", "--contig-len", "100000000")
    v = d["verify"]
    assert v["equals_reference"] is True and v["sum_sizes_equals_tot"] and v["count_mass_equals_instances"] and v["load_rule"]
    assert v["yak_size_bytes"] == 15992705192


def test_cfg5_qv_histogram_equals_reference():
    d = bench_line("--config", "cfg5")
    assert d["verify"]["equals_reference"] is True and d["verify"]["kmers"] == d["verify"]["kmers_expected"]


def test_cfg4_1gb_in_sweeps_over_prefix_ranges():
    """the same 1 Gb assembly through yak_count() as four ranks on one device (the way sizes beyond one pass -- 5 Gb -- are
    counted: `--config cfg4 --contigs 50 --sweeps 8`): the distinct count is the reference's, the counts add up to the
    instances, two chunkings of the stream agree"""
    d = bench_line("--config", "cfg4", "--contigs", "10", "--contig-len", "100000000", "--sweeps", "4")
    v = d["verify"]
    assert v["count_mass_equals_instances"] and v["sum_hist_equals_tot"] and v["chunking_independent"]
    assert v["distinct"] == 999771658 and v["yak_size_bytes"] == 7998181472


def test_cfg4_5gb_assembly_in_sweeps():
    """BASELINE configs[3] at its full size (50 contigs x 100 Mb, k = 21, 5 G distinct k-mers: a 40 GB .yak): yak_count() in the 2 sweeps over
    prefix ranges the library picks by itself for a file of that size, and in 4.  The reference needs ~80 GB of host memory at this size; the
    golden md5 comes from the ORACLE run in 8 prefix ranges (`yko count -R lo:hi`, tests/gen_golden_full.py --cfg4-ranges 50), a procedure that
    reproduces the reference's own md5 at 2 Gb (tests/golden/cfg45_full.json: cfg4_20x100000000.oracle_prefix_ranges_reproduce_it).  The
    2-sweep run is compared byte for byte (md5 of the 40 GB dump); both runs keep the size-independent properties"""
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "cfg45_full.json"))).get("cfg4_50x100000000")
    for sweeps in ("2",):                                      # (4 and 8 sweeps were the earlier rounds' way through this size: tests/tools, profiles/r03_*)
        d = bench_line("--config", "cfg4", "--contigs", "50", "--sweeps", sweeps, *(() if sweeps == "2" else ("--no-verify",)))
        v = d["verify"]
        assert v["count_mass_equals_instances"] and v["sum_hist_equals_tot"] and v["chunking_independent"]
        assert v["distinct"] == 4994315360 and v["yak_size_bytes"] == 16 + 8 * 1024 + 8 * 4994315360
        if sweeps == "2" and gold:
            assert v["equals_golden"] is True and v["yak_md5"] == gold["md5"] and gold["size"] == v["yak_size_bytes"]


@pytest.mark.parametrize("rank", [0, 5, 7])
def test_cfg3_rank_share_equals_oracle(rank):
    """BASELINE configs[2] (600 M x 150 bp reads, prefix-sharded over 8 GPUs), the only configuration that does not fit one GPU: ONE rank's share of it
    does.  The rank receives, round after round, the records of its 128 sub-tables from all 8 sources' chunks of the 600 M-read stream -- exactly what
    the exchange delivers -- and the bytes it contributes to the job's .yak file ({capacity, size, keys in slot order} of its sub-tables,
    htab.c:385-389) must be those the ORACLE computed over the whole stream (oracle/yko_synth, tests/gen_golden_cfg3.py ->
    tests/golden/cfg3_full.json holds all eight ranks' shares since round 6 -- what `bench.py --gpus 8` compares the whole job's table with; three of them run
    here, ~50 s each).  Reference: count.c:129-143 (a sub-table is a function of its own put-calls in stream order)."""
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "cfg3_full.json")))
    d = bench_line("--config", "cfg3shard", "--rank", str(rank))
    v, ref = d["verify"], gold["ranges"][f"{128 * rank}:{128 * rank + 128}"]
    assert v["share_equals_oracle"] is True and v["share_md5"] == ref["md5"] and v["share_bytes"] == ref["size"] and v["distinct"] == ref["distinct"]
    assert v["count_mass_equals_instances"] and v["sum_hist_equals_tot"]
    assert d["instances_received_per_pass"] == ref["instances"]          # no count saturates at 30x: the mass of the counts is the instances
    assert d["peak_hbm_bytes_in_the_real_job"] <= 0.9 * 288e9


def test_cfg3_rank_shares_at_1m_reads_equal_oracle():
    """the same procedure at 1 M reads (8 x 125 000, G = 5 Mb), every rank that the golden holds: seconds instead of a minute"""
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "cfg3_full.json")))["procedure_check_1M_reads"]
    for rng, ref in gold.items():
        lo = int(rng.split(":")[0])
        d = bench_line("--config", "cfg3shard", "--rank", str(lo // 128), "--reads", "125000", "--batch-reads", "50000")
        v = d["verify"]
        assert v["share_equals_oracle"] is True and v["share_md5"] == ref["md5"] and v["distinct"] == ref["distinct"], (rng, v)
