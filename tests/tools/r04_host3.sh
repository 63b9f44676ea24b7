#!/bin/bash
# the CLI end to end, five runs of each form of the input, with the reader's own account of its time
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04host; mkdir -p $O
echo "nproc $(nproc)" > $O/e2e3.txt
T=/tmp/e2e; mkdir -p $T
tools/yaksynth -n 10000000 -l 150 -g 50000000 -s 42 -t 32 -o $T/r.fq
python3 tests/tools/pgzip.py -l 6 -p 32 $T/r.fq $T/r.fq.gz
run() { # label, file, env...
  lab=$1; f=$2; shift 2
  for i in 1 2 3 4 5; do
    s=$(date +%s.%N); env "$@" YAKAMD_VERBOSE=1 yak_amd/yak-amd count -k31 -b37 -t32 -o $T/o.yak $f 2>$T/err.txt; e=$(date +%s.%N)
    grep "reader:\|in total;\|dump:\|pool after" $T/err.txt | cut -c1-260 >> $O/e2e3.txt
    python3 -c "print('$lab wall %.3f s' % ($e - $s))" >> $O/e2e3.txt
  done
  md5sum $T/o.yak >> $O/e2e3.txt
}
run plain $T/r.fq A=1
run gz $T/r.fq.gz A=1
run plain_nopack $T/r.fq YAKAMD_NO_HOST_PACK=1
grep "wall\|o.yak" $O/e2e3.txt
