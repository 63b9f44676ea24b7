#!/bin/bash
cd $GRAFT_REPO_ROOT
T=${TMPDIR:-/tmp}/yke2e; mkdir -p $T
N=${1:-10000000}
[ -f $T/r.fq ] || tools/yaksynth -n $N -l 150 -g $((N*5)) -s 42 -t 32 -o $T/r.fq
run() { local label=$1; local out=$2; shift; shift; s=$(date +%s.%N); env "$@" yak_amd/yak-amd count -k31 -b37 -o $out $T/r.fq 2> $T/a.err; e=$(date +%s.%N); python3 -c "print('$label: %.2f s' % ($e - $s))"; grep -E "sequences in total|dumpped|Real time|CMD" $T/a.err | sed 's/^/    /'; }
run "warm" /dev/null YAKAMD_PARSE_THREADS=32
run "devnull t32" /dev/null YAKAMD_PARSE_THREADS=32
run "devnull t32 nopin" /dev/null YAKAMD_PARSE_THREADS=32 YAKAMD_PIN=0
run "file t32" $T/a.yak YAKAMD_PARSE_THREADS=32
run "file again t32" $T/a.yak YAKAMD_PARSE_THREADS=32
run "devnull t32 win256" /dev/null YAKAMD_PARSE_THREADS=32 YAKAMD_PARSE_WINDOW=268435456
run "devnull t64 win512" /dev/null YAKAMD_PARSE_THREADS=64 YAKAMD_PARSE_WINDOW=536870912
s=$(date +%s.%N); python3 -c "
import ctypes; ctypes.CDLL('yak_amd/libyak_amd.so').yakamd_device_count()"; e=$(date +%s.%N); python3 -c "print('python + dlopen + device count: %.2f s' % ($e - $s))"
