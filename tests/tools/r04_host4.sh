#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04host; mkdir -p $O
T=/tmp/e2e; mkdir -p $T
tools/yaksynth -n 10000000 -l 150 -g 50000000 -s 42 -t 32 -o $T/r.fq
for i in 1 2 3; do
    s=$(date +%s.%N); YAKAMD_VERBOSE=2 yak_amd/yak-amd count -k31 -b37 -t32 -o $T/o.yak $T/r.fq 2>$T/err.txt; e=$(date +%s.%N)
    grep -v "processed" $T/err.txt | cut -c1-300 > $O/e2e4_$i.txt
    python3 -c "print('plain wall %.3f s' % ($e - $s))" >> $O/e2e4_$i.txt
done
cat $O/e2e4_3.txt
