#!/bin/bash
cd $GRAFT_REPO_ROOT
T=/tmp/e2e; mkdir -p $T
[ -f $T/r.fq ] || tools/yaksynth -n 10000000 -l 150 -g 50000000 -s 42 -t 32 -o $T/r.fq
for i in 1 2 3; do
  sleep 3
  s=$(date +%s.%N); YAKAMD_VERBOSE=1 yak_amd/yak-amd count -k31 -b37 -t32 -o $T/o.yak $T/r.fq 2>$T/err.txt; e=$(date +%s.%N)
  grep -v "processed" $T/err.txt | tail -12; python3 -c "print('wall %.3f s' % ($e - $s))"
done
md5sum $T/o.yak
if [ "$1" = "gz" ]; then
  [ -f $T/r.fq.gz ] || gzip -1 -k $T/r.fq
  for i in 1 2; do sleep 3; s=$(date +%s.%N); yak_amd/yak-amd count -k31 -b37 -t32 -o $T/o2.yak $T/r.fq.gz 2>$T/err.txt; e=$(date +%s.%N); grep -v processed $T/err.txt | tail -3; python3 -c "print('gz wall %.3f s' % ($e - $s))"; done
  md5sum $T/o2.yak
fi
