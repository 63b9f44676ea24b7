#!/bin/bash
# host stage on the GPU box: the CLI end to end on a plain FASTQ file and on the same file as ordinary gzip
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04host; mkdir -p $O
echo "nproc $(nproc)" > $O/e2e.txt
if [ "$1" = "tests" ]; then timeout 900 python -m pytest tests/test_gpu_api.py tests/test_ref_cli_on_amd.py tests/test_multi_c.py -x -q 2>&1 | tail -5 > $O/tests.txt; fi
T=/tmp/e2e; mkdir -p $T
tools/yaksynth -n 10000000 -l 150 -g 50000000 -s 42 -t 32 -o $T/r.fq
ls -la $T/r.fq >> $O/e2e.txt
for i in 1 2 3; do
  sleep 2
  s=$(date +%s.%N); YAKAMD_VERBOSE=1 yak_amd/yak-amd count -k31 -b37 -t32 -o $T/o.yak $T/r.fq 2>$T/err.txt; e=$(date +%s.%N)
  grep -v "processed" $T/err.txt | tail -14 | cut -c1-250 >> $O/e2e.txt; python3 -c "print('plain wall %.3f s' % ($e - $s))" >> $O/e2e.txt
done
md5sum $T/o.yak >> $O/e2e.txt
s=$(date +%s.%N); python3 tests/tools/pgzip.py -l 6 -p 32 $T/r.fq $T/r.fq.gz; e=$(date +%s.%N); python3 -c "print('pgzip %.1f s' % ($e - $s))" >> $O/e2e.txt
ls -la $T/r.fq.gz >> $O/e2e.txt
for i in 1 2 3; do
  sleep 2
  s=$(date +%s.%N); YAKAMD_VERBOSE=1 yak_amd/yak-amd count -k31 -b37 -t32 -o $T/o2.yak $T/r.fq.gz 2>$T/err.txt; e=$(date +%s.%N)
  grep -v "processed" $T/err.txt | tail -14 | cut -c1-250 >> $O/e2e.txt; python3 -c "print('gz wall %.3f s' % ($e - $s))" >> $O/e2e.txt
done
md5sum $T/o2.yak >> $O/e2e.txt
s=$(date +%s.%N); YAKAMD_NO_PGZ=1 yak_amd/yak-amd count -k31 -b37 -t32 -o $T/o3.yak $T/r.fq.gz 2>/dev/null; e=$(date +%s.%N); python3 -c "print('gz through gzread wall %.3f s' % ($e - $s))" >> $O/e2e.txt
md5sum $T/o3.yak >> $O/e2e.txt
cat $O/tests.txt 2>/dev/null; tail -50 $O/e2e.txt
