"""Several GPUs behind yak_count() (YAKAMD_GPUS; SURVEY 8e, reference count.c:129-143): the library deals the
input to N ranks in chunks, every rank groups its k-mers by prefix, one exchange per round moves them to the
owner.  A one-GPU box runs the N ranks on the same device (YAKAMD_GPU_LIST=0,0,...: device copies instead of
RCCL); the bytes must equal the single-GPU run's, which the other tests pin on the oracle and the reference."""
import os
import subprocess

import pytest

from conftest import ROOT

YAM = os.path.join(ROOT, "yak_amd", "yak-amd")
YKO = os.path.join(ROOT, "oracle", "yko")
SYN = os.path.join(ROOT, "tools", "yaksynth")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def reads(tmp_path_factory):
    import yak_amd
    if yak_amd.lib().yakamd_device_count() < 1:
        pytest.skip("no MI355X visible")
    d = tmp_path_factory.mktemp("mgpu")
    fq, fa = str(d / "r.fq"), str(d / "c.fa")
    subprocess.check_call([SYN, "-n", "20000", "-l", "150", "-g", "100000", "-s", "31", "-o", fq])
    subprocess.check_call([SYN, "-a", "-n", "12", "-l", "40000", "-g", "100000", "-s", "31", "-e", "0.001", "-N", "0.0003", "-o", fa])
    return dict(fq=fq, fa=fa, dir=str(d))


@pytest.mark.parametrize("n_gpu,chunk", [(2, "300000"), (4, "150000"), (8, "100000"), (2, None)])
@pytest.mark.parametrize("args,inp", [(["-k31", "-b24"], "fq"), (["-k31"], "fq"), (["-k21", "-b22", "-t1"], "fa")], ids=["reads_b24", "reads_nofilter", "contigs_b22_stream_reader"])
def test_multi_gpu_count_equals_single(n_gpu, chunk, args, inp, reads):
    want, got = os.path.join(reads["dir"], "one.yak"), os.path.join(reads["dir"], "multi.yak")
    subprocess.run([YKO, "count"] + args + ["-o", want, reads[inp]], check=True, stderr=subprocess.DEVNULL)      # the oracle's file
    env = dict(os.environ, YAKAMD_GPUS=str(n_gpu), YAKAMD_GPU_LIST=",".join(["0"] * n_gpu))
    if chunk:
        env["YAKAMD_MGPU_CHUNK"] = chunk
    r = subprocess.run([YAM, "count"] + args + ["-o", got, reads[inp]], check=True, env=env, stderr=subprocess.PIPE)
    assert f"{n_gpu} GPUs".encode() in r.stderr
    assert open(got, "rb").read() == open(want, "rb").read()


@pytest.mark.parametrize("args,extra", [(["-k31", "-b24"], ["-X", "YAKAMD_MGPU_REC16=1"]), (["-k41", "-b24"], []), (["-k31", "-b24"], ["-X", "YAKAMD_FAST=0"])],
                         ids=["rec16_exchange", "k41_no_tagged_records", "general_path_refuses_tagged_uses_rec16"])
def test_multi_gpu_exchange_formats(args, extra, reads):
    """the exchange moves 8-byte tagged records where k / pre / the pass allow them; the 16-byte {hash, position} format
    must stay exact: forced, for k >= 32, and when the owners' passes are not on the exclusive-ownership path"""
    want, got = os.path.join(reads["dir"], "one2.yak"), os.path.join(reads["dir"], "multi2.yak")
    subprocess.run([YKO, "count"] + args + ["-o", want, reads["fq"]], check=True, stderr=subprocess.DEVNULL)
    env = dict(os.environ, YAKAMD_GPUS="2", YAKAMD_GPU_LIST="0,0", YAKAMD_MGPU_CHUNK="300000")
    subprocess.run([YAM] + extra + ["count"] + args + ["-o", got, reads["fq"]], check=True, env=env, stderr=subprocess.PIPE)   # -X: a test switch of the library
    assert open(got, "rb").read() == open(want, "rb").read()


def test_count_from_named_pipes(reads):
    """`yak count -b.. -o out <(zcat reads.gz) <(zcat reads.gz)` -- the way the reference's README feeds gzipped short reads: both passes read a pipe
    given by NAME.  The bytes are the file's (one GPU, and the multi-GPU reader's route); the reader lost the first megabyte of such a stream until round 6."""
    want, got = os.path.join(reads["dir"], "onep.yak"), os.path.join(reads["dir"], "pipe.yak")
    gz = os.path.join(reads["dir"], "r.fq.gz")
    subprocess.run(f"gzip -1 -c {reads['fq']} > {gz}", shell=True, check=True)
    for args in ("-k31 -b24", "-k21"):
        subprocess.run([YKO, "count"] + args.split() + ["-o", want, reads["fq"]], check=True, stderr=subprocess.DEVNULL)
        for src, env in ((f"cat {reads['fq']}", {}), (f"zcat {gz}", {}), (f"cat {reads['fq']}", dict(YAKAMD_GPUS="2", YAKAMD_GPU_LIST="0,0", YAKAMD_MGPU_CHUNK="300000"))):
            subprocess.run(["bash", "-c", f"{YAM} count {args} -o {got} <({src}) <({src})"], check=True, env=dict(os.environ, **env), stderr=subprocess.PIPE, timeout=300)
            assert open(got, "rb").read() == open(want, "rb").read(), (args, src, env)
    # a producer that is done before the library looks at the name a second time (a short input: all of it sits in the pipe): no second open() of the pipe
    tiny, tw = os.path.join(reads["dir"], "tiny.fq"), os.path.join(reads["dir"], "tiny.yak")
    subprocess.run(f"head -n 400 {reads['fq']} > {tiny}", shell=True, check=True)
    subprocess.run([YKO, "count", "-k21", "-o", tw, tiny], check=True, stderr=subprocess.DEVNULL)
    for src in (f"cat {tiny}", f"gzip -c {tiny}"):
        for t in ("-t1", "-t8"):
            subprocess.run(["bash", "-c", f"{YAM} count -k21 {t} -o {got} <({src})"], check=True, stderr=subprocess.PIPE, timeout=120)
            assert open(got, "rb").read() == open(tw, "rb").read(), (src, t)
    # standard input (count.c:151: gzdopen(0)): redirected from a file, from a pipe, gzipped through a pipe
    for cmd in (f"{YAM} count -k21 -o {got} - < {reads['fq']}", f"cat {reads['fq']} | {YAM} count -k21 -o {got} -", f"cat {gz} | {YAM} count -k21 -t1 -o {got} -"):
        subprocess.run(["bash", "-c", cmd], check=True, stderr=subprocess.PIPE, timeout=300)
        assert open(got, "rb").read() == open(want, "rb").read(), cmd


def test_large_unfiltered_plain_files_are_counted_in_sweeps(reads):
    """no filter + a plain file beyond YAKAMD_AUTO_SWEEP_GB: yak_count() takes the input as N ranks on its one device (an
    assembly of several Gb does not fit one pass); a filtered count, or YAKAMD_GPUS set, leaves the rule off.  Bytes unchanged."""
    want, got = os.path.join(reads["dir"], "one3.yak"), os.path.join(reads["dir"], "multi3.yak")
    env = dict(os.environ, YAKAMD_AUTO_SWEEP_GB="0.000001")
    env.pop("YAKAMD_GPUS", None); env.pop("YAKAMD_GPU_LIST", None)
    for args, swept in ((["-k21"], True), (["-k21", "-b22"], False)):
        subprocess.run([YKO, "count"] + args + ["-o", want, reads["fa"]], check=True, stderr=subprocess.DEVNULL)
        r = subprocess.run([YAM, "count"] + args + ["-o", got, reads["fa"]], check=True, env=env, stderr=subprocess.PIPE)
        assert (b"sweeps over prefix ranges" in r.stderr) == swept
        assert open(got, "rb").read() == open(want, "rb").read()
    r = subprocess.run([YAM, "count", "-k21", "-o", got, reads["fa"]], check=True, env=dict(env, YAKAMD_GPUS="1"), stderr=subprocess.PIPE)
    assert b"sweeps over prefix ranges" not in r.stderr
    # a process that does not own the device memory of the few-sweep plan yet takes more sweeps (YAKAMD_COLD_GB: the estimated peak of a sweep's share
    # against what the driver hands out without charging); 0 switches that off.  Same bytes either way.
    subprocess.run([YKO, "count", "-k21", "-o", want, reads["fa"]], check=True, stderr=subprocess.DEVNULL)
    for cold, n in (("0.000001", b"counting in 16 sweeps"), ("0", b"counting in 2 sweeps")):
        r = subprocess.run([YAM, "count", "-k21", "-o", got, reads["fa"]], check=True, env=dict(env, YAKAMD_COLD_GB=cold), stderr=subprocess.PIPE)
        assert n in r.stderr
        assert open(got, "rb").read() == open(want, "rb").read()


def test_a_sequence_longer_than_a_chunk_is_split_with_an_overlap(reads):
    """contigs of 40 kb dealt in chunks of 16 KiB: a chunk then ends inside a sequence and the next one starts k - 1 bases earlier
    (chromosomes beyond YAKAMD_MGPU_CHUNK = 256 Mb take this path in a swept count); the bytes stay the single pass's"""
    want, got = os.path.join(reads["dir"], "one4.yak"), os.path.join(reads["dir"], "multi4.yak")
    for args in (["-k21"], ["-k31", "-b22"], ["-k41"]):
        subprocess.run([YKO, "count"] + args + ["-o", want, reads["fa"]], check=True, stderr=subprocess.DEVNULL)
        for n_gpu in (2, 4):
            env = dict(os.environ, YAKAMD_GPUS=str(n_gpu), YAKAMD_GPU_LIST=",".join(["0"] * n_gpu), YAKAMD_MGPU_CHUNK="16384")
            subprocess.run([YAM, "count"] + args + ["-o", got, reads["fa"]], check=True, env=env, stderr=subprocess.PIPE)
            assert open(got, "rb").read() == open(want, "rb").read()
    # ... and with a budget that fills a rank's slice several times over: a continuation chunk (its first k - 1 positions lie before the end
    # of the chunk in front of it) must be able to start the next slice (fast_admit compared t, not t + k - 1, with t_end)
    for args in (["-k21"], ["-k31", "-b22"]):
        subprocess.run([YKO, "count"] + args + ["-o", want, reads["fa"]], check=True, stderr=subprocess.DEVNULL)
        for budget in ("400000", "150000"):
            env = dict(os.environ, YAKAMD_GPUS="2", YAKAMD_GPU_LIST="0,0", YAKAMD_MGPU_CHUNK="16384", YAKAMD_FAST_BUDGET=budget)
            subprocess.run([YAM, "count"] + args + ["-o", got, reads["fa"]], check=True, env=env, stderr=subprocess.PIPE)
            assert open(got, "rb").read() == open(want, "rb").read(), (args, budget)


RIG = {"loopback": ["YAKAMD_MGPU_LOOPBACK=1"], "loopback_round_2_fails": ["YAKAMD_MGPU_LOOPBACK=1", "YAKAMD_MGPU_LOOPBACK_FAIL=2"], "peer_copies": ["YAKAMD_MGPU_NO_RCCL=1"]}


@pytest.mark.parametrize("rig", sorted(RIG))
@pytest.mark.parametrize("n_gpu,args,inp", [(2, ["-k31", "-b24"], "fq"), (4, ["-k31"], "fq"), (2, ["-k21", "-b22", "-t1"], "fa"), (2, ["-k41", "-b24"], "fq")],
                         ids=["2_reads_b24", "4_reads_nofilter", "2_contigs_stream_reader", "2_k41_rec16"])
def test_slots_on_one_device_exchange_like_distinct_devices(rig, n_gpu, args, inp, reads):
    """YAKAMD_MGPU_SLOT_PER_RANK (test switch): the N ranks on device 0 each get a slot of their own -- chunk, send and receive buffers, exchange and copy
    streams, staging events per slot -- and every round EXCHANGES between the slots, the code a box with several GPUs runs (S > 1: receive layout, grouped
    ncclSend / ncclRecv, the repeat of a failed round as peer copies, feeds out of the receive buffers).  `loopback`: the grouped calls are served by the
    library's in-process rig, which fails a group whose receives do not find sends of the same count in posting order (what hangs the real library);
    `loopback_round_2_fails`: the second group fails without moving a byte and the round, and all later ones, go as peer copies; `peer_copies`: no collective
    library at all.  The .yak bytes are the oracle's."""
    want, got = os.path.join(reads["dir"], "one6.yak"), os.path.join(reads["dir"], "multi6.yak")
    subprocess.run([YKO, "count"] + args + ["-o", want, reads[inp]], check=True, stderr=subprocess.DEVNULL)
    env = dict(os.environ, YAKAMD_GPUS=str(n_gpu), YAKAMD_GPU_LIST=",".join(["0"] * n_gpu), YAKAMD_MGPU_CHUNK="150000")
    x = [a for kv in ["YAKAMD_MGPU_SLOT_PER_RANK=1"] + RIG[rig] for a in ("-X", kv)]
    r = subprocess.run([YAM] + x + ["count"] + args + ["-o", got, reads[inp]], check=True, env=env, stderr=subprocess.PIPE)
    said = {"loopback": b"in-process test rig", "loopback_round_2_fails": b"peer copies from now on", "peer_copies": b"peer copies)"}[rig]
    assert said in r.stderr, r.stderr[-1500:]
    assert b"nothing exchanged" not in r.stderr
    assert open(got, "rb").read() == open(want, "rb").read()


def test_multi_gpu_on_distinct_devices(reads):
    """two REAL devices (the staging events of yak_count_multi belong to one device each; with ranks sharing device 0, as every
    other test here runs them, an event made on the wrong device goes unnoticed).  Skipped on a one-GPU box."""
    import yak_amd
    n_dev = yak_amd.lib().yakamd_device_count()
    if n_dev < 2:
        pytest.skip("needs two MI355X")
    want, got = os.path.join(reads["dir"], "one5.yak"), os.path.join(reads["dir"], "multi5.yak")
    for args in (["-k31", "-b24"], ["-k31"]):
        subprocess.run([YKO, "count"] + args + ["-o", want, reads["fq"]], check=True, stderr=subprocess.DEVNULL)
        for n_gpu in sorted({2, min(n_dev, 8)}):
            if 1024 % n_gpu:
                continue
            env = dict(os.environ, YAKAMD_GPUS=str(n_gpu), YAKAMD_MGPU_CHUNK="300000")
            env.pop("YAKAMD_GPU_LIST", None)
            r = subprocess.run([YAM, "count"] + args + ["-o", got, reads["fq"]], check=True, env=env, stderr=subprocess.PIPE)
            assert f"{n_gpu} GPUs".encode() in r.stderr
            assert open(got, "rb").read() == open(want, "rb").read(), (args, n_gpu)
