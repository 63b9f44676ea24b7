#!/bin/bash
# end-to-end CLI comparison on a FASTQ file: yak-amd count vs the reference binary (if present) vs the oracle
set -u
cd "$(dirname "$0")/../.."
T=${TMPDIR:-/tmp}/yke2e; mkdir -p $T
N=${1:-2000000}
tools/yaksynth -n $N -l 150 -g $((N*5)) -s 42 -t 16 -o $T/r.fq
ls -la $T/r.fq | awk '{print "fastq bytes", $5}'
s=$(date +%s.%N); yak_amd/yak-amd count -k31 -b35 -K1g -t16 -o $T/a.yak $T/r.fq 2> $T/a.err; e=$(date +%s.%N); python3 -c "print('yak-amd wall %.2f s' % ($e - $s))"; tail -2 $T/a.err
if [ -x oracle/_ref/yak ]; then s=$(date +%s.%N); oracle/_ref/yak count -k31 -b35 -t32 -o $T/r.yak $T/r.fq 2> $T/r.err; e=$(date +%s.%N); python3 -c "print('reference -t32 wall %.2f s' % ($e - $s))"; tail -1 $T/r.err; cmp $T/a.yak $T/r.yak && echo "BYTES IDENTICAL to reference"; fi
