#!/bin/bash
# round 4, last measurements: the default line (with its host-side figures), the CLI end to end, the 5 Gb assembly in two sweeps
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04final2; mkdir -p $O
timeout 600 python -m pytest tests/test_qv.py -x -q -m gpu 2>&1 | tail -4 > $O/tests_qv.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
T=/tmp/e2e; mkdir -p $T
tools/yaksynth -n 10000000 -l 150 -g 50000000 -s 42 -t 32 -o $T/r.fq
python3 tests/tools/pgzip.py -l 6 -p 32 $T/r.fq $T/r.fq.gz
echo "nproc $(nproc); $(ls -la $T/r.fq | awk '{print $5}') bytes of FASTQ, $(ls -la $T/r.fq.gz | awk '{print $5}') as gzip -6 (one member, tests/tools/pgzip.py)" > $O/e2e_cli.txt
run() { # label, file, env...
  lab=$1; f=$2; shift 2
  for i in 1 2 3 4 5; do
    sleep 3   # (a process that starts while the driver still clears the 58 GB the one before it gave back pays for that in its own allocations)
    s=$(date +%s.%N); env "$@" YAKAMD_VERBOSE=1 yak_amd/yak-amd count -k31 -b37 -t32 -o $T/o.yak $f 2>$T/err.txt; e=$(date +%s.%N)
    grep "reader:\|gzip:\|in total\|dump:" $T/err.txt | cut -c1-260 >> $O/e2e_cli.txt
    python3 -c "print('$lab wall %.3f s' % ($e - $s))" >> $O/e2e_cli.txt
  done
  md5sum $T/o.yak >> $O/e2e_cli.txt
}
run plain $T/r.fq A=1
run gz $T/r.fq.gz A=1
run gz_gzread $T/r.fq.gz YAKAMD_NO_PGZ=1 2>/dev/null
YAKAMD_VERBOSE=1 timeout 900 python bench.py --config cfg4 --contigs 50 --warmup 1 > $O/bench_cfg4_5gb_sweeps2.json 2> $O/bench_cfg4_5gb.err
grep "ranks: input\|pool after" $O/bench_cfg4_5gb.err | head -8 | cut -c1-200 > $O/cfg4_5gb_stages.txt
cat $O/tests_qv.txt; grep "wall\|o.yak" $O/e2e_cli.txt; python3 -c "
import json
d=json.loads([l for l in open('$O/bench_default.json') if l.startswith('{')][-1]); print('default', d['ms_per_step'], d['value'], d['pcie_inclusive_ms'], d['pcie_inclusive_host_pack'], d['e2e_cli'], d['verify'])
d=json.loads([l for l in open('$O/bench_cfg4_5gb_sweeps2.json') if l.startswith('{')][-1]); print('5gb', d['ms_per_step'], d['seconds_second_chunking'], d['seconds_jobs_before_the_timed_one'], d['verify']['equals_golden'])"
