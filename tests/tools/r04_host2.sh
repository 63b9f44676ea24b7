#!/bin/bash
# host stage variants on the GPU box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04host; mkdir -p $O
echo "nproc $(nproc)" > $O/e2e2.txt
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_ref_cli_on_amd.py -x -q -k "packed or cli or Cli or ref" 2>&1 | tail -5 > $O/tests2.txt
bash tests/quick_gpu.sh > $O/quick.txt 2>&1
T=/tmp/e2e; mkdir -p $T
tools/yaksynth -n 10000000 -l 150 -g 50000000 -s 42 -t 32 -o $T/r.fq
python3 tests/tools/pgzip.py -l 6 -p 32 $T/r.fq $T/r.fq.gz
run() { # label, file, env...
  lab=$1; f=$2; shift 2
  for i in 1 2; do
    sleep 1
    s=$(date +%s.%N); env "$@" YAKAMD_VERBOSE=1 yak_amd/yak-amd count -k31 -b37 -t32 -o $T/o.yak $f 2>$T/err.txt; e=$(date +%s.%N)
    if [ $i = 2 ]; then grep "processed\|in total\|dump:\|gzip:" $T/err.txt | cut -c1-160 >> $O/e2e2.txt; fi
    python3 -c "print('$lab wall %.3f s' % ($e - $s))" >> $O/e2e2.txt
  done
  md5sum $T/o.yak >> $O/e2e2.txt
}
run plain $T/r.fq A=1
run plain_nopack $T/r.fq YAKAMD_NO_HOST_PACK=1
run plain_win1g $T/r.fq YAKAMD_PARSE_WINDOW=1073741824
run gz $T/r.fq.gz A=1
run gz_chunk1m $T/r.fq.gz YAKAMD_GZ_CHUNK=1048576
run gz_chunk512k $T/r.fq.gz YAKAMD_GZ_CHUNK=524288
cat $O/tests2.txt; tail -3 $O/quick.txt; grep "wall\|o.yak" $O/e2e2.txt
