#!/bin/bash
# where the time of `yak-amd count` goes on the benchmark reads written as FASTQ (host input stage)
cd $GRAFT_REPO_ROOT
T=${TMPDIR:-/tmp}/yke2e; mkdir -p $T
N=${1:-10000000}
[ -f $T/r.fq ] || tools/yaksynth -n $N -l 150 -g $((N*5)) -s 42 -t 32 -o $T/r.fq
ls -la $T/r.fq | awk '{print "fastq bytes", $5}'
run() { local label=$1; shift; s=$(date +%s.%N); env "$@" yak_amd/yak-amd count -k31 -b37 -o $T/a.yak $T/r.fq 2> $T/a.err; e=$(date +%s.%N); python3 -c "print('$label: %.2f s' % ($e - $s))"; }
run warm YAKAMD_PARSE_THREADS=32
run "t32 pinned win1g" YAKAMD_PARSE_THREADS=32
grep -E "yak_count::|main" $T/a.err | head -12
run "t32 pageable win1g" YAKAMD_PARSE_THREADS=32 YAKAMD_PIN=0
run "t32 pinned win256m" YAKAMD_PARSE_THREADS=32 YAKAMD_PARSE_WINDOW=268435456
grep -E "yak_count::" $T/a.err | tail -4
run "t32 pinned win128m" YAKAMD_PARSE_THREADS=32 YAKAMD_PARSE_WINDOW=134217728
run "t16 pinned win256m" YAKAMD_PARSE_THREADS=16 YAKAMD_PARSE_WINDOW=268435456
run "t8 pinned win256m" YAKAMD_PARSE_THREADS=8 YAKAMD_PARSE_WINDOW=268435456
md5sum $T/a.yak
s=$(date +%s.%N); python3 -c "
import sys; sys.path.insert(0,'.')
import yak_amd, os
os.environ['YAKAMD_PARSE_THREADS']='32'; os.environ['YAKAMD_PARSE_DISCARD']='1'; os.environ['YAKAMD_VERBOSE']='1'
yak_amd.host_image('$T/r.fq', 31, True)"; e=$(date +%s.%N); python3 -c "print('parse only (32 threads): %.2f s' % ($e - $s))"
