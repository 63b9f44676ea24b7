#!/bin/bash
# quick device parity sweep: product CLI vs oracle CLI on synthetic reads (debug helper)
set -u
cd "$(dirname "$0")/.."
T=${TMPDIR:-/tmp}/ykq; mkdir -p $T
S=tools/yaksynth; O=oracle/yko; A=yak_amd/yak-amd
$S -n 12000 -l 150 -g 60000 -s 7 -o $T/s.fq
$S -n 300 -l 5000 -g 200000 -s 9 -a -N 0.002 -o $T/l.fa
$S -n 100000 -l 150 -g 500000 -s 42 -o $T/c1.fq
printf ">a\nACGTNNACGTACGTTTGACCA\r\n>b desc\nAC\n\nGT\n@c\nACGTAGGCATTACGGACTA\n+\nIIIIIIIIIIIIIIIIIII\n@d\nACGTAGGCATTACGGACTAGG\n+\nIIII\n" > $T/e.fx
fail=0
run(){ name=$1; shift; $O count "$@" -o $T/o_$name.yak 2>/dev/null; timeout 300 $A count "$@" -o $T/a_$name.yak 2>$T/a_$name.err; rc=$?
  if cmp -s $T/o_$name.yak $T/a_$name.yak; then echo "OK   $name $(stat -c %s $T/o_$name.yak)"; else echo "DIFF $name rc=$rc"; tail -3 $T/a_$name.err; fail=1; fi; }
run edge -k5 $T/e.fx
run nb -k31 $T/s.fq
run k21 -k21 $T/s.fq
run k15 -k15 $T/l.fa
run c1 -k31 -K64m $T/c1.fq
run K -k31 -K100k $T/s.fq
run p12 -k27 -p12 $T/s.fq
run b24 -k31 -b24 $T/s.fq
run b20 -k31 -b20 $T/s.fq
run b30 -k31 -b30 $T/s.fq
run b15 -k31 -b15 $T/s.fq
run b19H7 -k31 -b19 -H7 $T/s.fq
run b22H40 -k31 -b22 -H40 $T/s.fq
run c1b -k31 -b26 $T/c1.fq
run p12b -k27 -p12 -b26 $T/s.fq
exit $fail
