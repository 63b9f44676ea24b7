#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04host; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_ref_cli_on_amd.py tests/test_multi_c.py -x -q 2>&1 | tail -5 > $O/tests5.txt
T=/tmp/e2e; mkdir -p $T
tools/yaksynth -n 10000000 -l 150 -g 50000000 -s 42 -t 32 -o $T/r.fq
python3 tests/tools/pgzip.py -l 6 -p 32 $T/r.fq $T/r.fq.gz
echo "nproc $(nproc)" > $O/e2e5.txt
run() { # label, file, env...
  lab=$1; f=$2; shift 2
  for i in 1 2 3 4 5; do
    sleep 3   # (a process that starts while the driver still clears the 58 GB the one before it gave back pays for that in its own allocations)
    s=$(date +%s.%N); env "$@" YAKAMD_VERBOSE=1 yak_amd/yak-amd count -k31 -b37 -t32 -o $T/o.yak $f 2>$T/err.txt; e=$(date +%s.%N)
    grep "reader:\|in total;\|dump:" $T/err.txt | cut -c1-260 >> $O/e2e5.txt
    python3 -c "print('$lab wall %.3f s' % ($e - $s))" >> $O/e2e5.txt
  done
  md5sum $T/o.yak >> $O/e2e5.txt
}
run plain $T/r.fq A=1
run gz $T/r.fq.gz A=1
cat $O/tests5.txt; grep "wall\|o.yak" $O/e2e5.txt
