"""The N > 1 path of bench.py on ONE device: two ranks (torch.distributed.run, gloo instead of RCCL so
that both may use GPU 0) shard the prefixes, exchange their k-mers and count; the job's result must
equal the single-rank run on the same logical input (rank 0's reads followed by rank 1's)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def run_bench(args, nproc=1, port=29547):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable]
    if nproc > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1", "--master-port", str(port)]
    cmd += [os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, env=env, timeout=900, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    out = r.stdout.decode()
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("bf", [34, 0])
def test_two_ranks_equal_one_rank(bf):
    common = ["--bf-shift", str(bf), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-verify", "--no-qv", "--job-md5"]
    two = run_bench(["--gpus", "2", "--driver", "torch", "--backend", "gloo", "--reads", "150000"] + common, nproc=2, port=29547 + bf)
    one = run_bench(["--reads", "300000"] + common)
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["final_distinct"] == one["final_distinct"] and one["final_distinct"] > 0
    assert two["job_yak_md5"] == one["job_yak_md5"] and one["job_yak_md5"]        # the .yak bytes, not just the size
    assert two["kmer_instances_per_s"] > 0 and two["scaling"] == "weak"


def test_single_rank_exchange_path_matches_direct_path():
    """--force-exchange: partition + (self) all-to-all + pre-partitioned feed, against the direct feed"""
    common = ["--reads", "400000", "--bf-shift", "35", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-verify", "--no-qv", "--job-md5"]
    a = run_bench(common)
    b = run_bench(common + ["--force-exchange"])
    c = run_bench(common + ["--force-exchange", "--no-overlap"])
    assert a["final_distinct"] == b["final_distinct"] == c["final_distinct"] > 0
    assert a["job_yak_md5"] == b["job_yak_md5"] == c["job_yak_md5"]


@pytest.mark.parametrize("n,bf,extra", [(2, 0, []), (2, 34, []), (4, 0, ["--batch-reads", "70000"])], ids=["two_ranks_no_filter", "two_ranks_filtered_protocol", "four_ranks_several_rounds"])
def test_bench_n_gpus_runs_the_librarys_own_driver(n, bf, extra):
    """`bench.py --gpus N` as the driver launches it (torch.distributed.run, N processes): rank 0 runs the library's C driver
    (yakamd_count_multi_dev) over the N ranks -- on this one-GPU box they share device 0, so nothing is exchanged -- and the sharded table's
    .yak bytes equal those of ONE table fed the same stream (the bench compares them itself under --job-md5 and fails loudly)"""
    args = ["--gpus", str(n), "--reads", "150000", "--steps", "1", "--warmup", "0", "--job-md5"] + (["--bf-shift", str(bf)] if bf else []) + extra
    r = run_bench(args, nproc=n, port=29580 + n + bf)
    assert r["n_gpus"] == n and r["verify"]["equals_one_table"] and r["final_distinct"] > 0
    assert r["config"]["driver"].startswith("C: yakamd_count_multi_dev") and r["config"]["exchange"].startswith("none")
    assert r["config"]["bf_shift"] == bf and r["config"]["reads_per_gpu"] == 150000


@pytest.mark.parametrize("knobs,exchange", [(["YAKAMD_MGPU_LOOPBACK=1"], "grouped ncclSend/ncclRecv call pattern served"), (["YAKAMD_MGPU_LOOPBACK=1", "YAKAMD_MGPU_NO_OVERLAP=1"], "grouped"),
                                            (["YAKAMD_MGPU_LOOPBACK=1", "YAKAMD_MGPU_LOOPBACK_FAIL=3"], "hipMemcpyPeerAsync"), (["YAKAMD_MGPU_NO_RCCL=1"], "hipMemcpyPeerAsync")],
                         ids=["loopback_overlapped", "loopback_stage_by_stage", "a_failed_group_repeats_as_peer_copies", "peer_copies"])
def test_bench_n_gpus_with_a_slot_per_rank_runs_the_exchange(knobs, exchange):
    """the device-resident driver with every rank a slot of its own on device 0 (YAKAMD_MGPU_SLOT_PER_RANK): rounds of two chunks, the partition of round
    b + 1 overlapped with the exchange and feed of round b, records crossing between the slots' send and receive buffers -- and the sharded table's bytes
    still equal ONE table's (bench.py --job-md5 fails loudly otherwise).  The line carries weak_base, cpu_baseline and (null without a counter pass of
    this very command) roofline.traffic with its source."""
    args = ["--gpus", "2", "--reads", "150000", "--batch-reads", "40000", "--steps", "1", "--warmup", "0", "--job-md5", "--knob", "YAKAMD_MGPU_SLOT_PER_RANK=1"]
    for kv in knobs:
        args += ["--knob", kv]
    r = run_bench(args, nproc=2, port=29611 + len(knobs) + len(exchange))
    assert r["n_gpus"] == 2 and r["verify"]["equals_one_table"] and r["final_distinct"] > 0
    assert r["config"]["exchange"].startswith(exchange), r["config"]["exchange"]
    assert r["config"]["rounds"] == 4
    wb = r["weak_base"]
    assert wb["reads"] == 150000 and wb["ms_per_step"] > 0 and wb["final_distinct"] > 0 and wb["base_ms_over_job_ms"] > 0
    cb = r["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["value"] > 0 and cb["cores"] >= 1
    assert "traffic" in r["roofline"] and r["roofline"]["traffic_source"]
