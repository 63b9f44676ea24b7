T=/tmp/qvt; mkdir -p $T
tools/yaksynth -n 600 -l 150 -g 2500 -s 5 -o $T/r.fq
tools/yaksynth -a -n 30 -l 1000 -g 2500 -s 5 -e 0.01 -N 0.001 -o $T/a.fa
oracle/yko count -k31 -b24 -o $T/t.yak $T/r.fq 2>/dev/null
for o in "-p -E" "-p -l 1000 -f 0.8" "-l 2000" "-K 5k -p"; do
  oracle/yko qv $o $T/t.yak $T/a.fa | sort > $T/o.txt
  yak_amd/yak-amd qv $o $T/t.yak $T/a.fa 2>$T/err.txt | sort > $T/a.txt
  if cmp -s $T/o.txt $T/a.txt; then echo "OK qv $o ($(wc -l < $T/a.txt) lines)"; else echo "DIFF qv $o"; tail -2 $T/err.txt; diff $T/o.txt $T/a.txt | head -5; fi
done
tools/yaksynth -n 100000 -l 150 -g 500000 -s 42 -o $T/c1.fq
tools/yaksynth -a -n 50 -l 20000 -g 500000 -s 42 -e 0.002 -o $T/c1.fa
oracle/yko count -k21 -o $T/c1.yak $T/c1.fq 2>/dev/null
oracle/yko qv -p $T/c1.yak $T/c1.fa | sort > $T/o.txt; yak_amd/yak-amd qv -p $T/c1.yak $T/c1.fa 2>/dev/null | sort > $T/a.txt
cmp -s $T/o.txt $T/a.txt && echo "OK qv k21 long contigs" || echo "DIFF k21"
