#!/bin/bash
# the sweep rule of a process that does not own its device memory yet (YAKAMD_COLD_GB): API / multi tests, the 5 Gb assembly with the library's own rule
# (first job, then warm), and `yak-amd count` -- one job per process, the way yak count is used -- on the 2 Gb and 5 Gb FASTA with the rule on and off
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06cold}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_multi_c.py tests/test_gpu_multirank.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
YAKAMD_VERBOSE=1 timeout 400 python bench.py --config cfg4 --contigs 50 --warmup 1 > $O/bench_cfg4_5gb.json 2> $O/bench_cfg4_5gb.err
python3 - $O/bench_cfg4_5gb.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], "ms", round(d["ms_per_step"], 2), d["config"].get("sweeps_of_every_job"), {k: x for k, x in (d.get("verify") or {}).items() if isinstance(x, bool)}, {k: d[k] for k in d if k.startswith(("first_job", "peak_hbm_bytes")) and "note" not in k})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
grep "pool after\|sweeps over" $O/bench_cfg4_5gb.err | tail -4 | cut -c1-400
T=${TMPDIR:-/tmp}/ykcold; mkdir -p $T
for n in 20 50; do
  tools/yaksynth -T -n $n -l 100000000 -s 42 -w 60 -t 32 -o $T/asm$n.fa
  for cold in 110 0; do
    sleep 8
    s=$(date +%s.%N); YAKAMD_COLD_GB=$cold yak_amd/yak-amd count -k21 -t32 -o $T/a.yak $T/asm$n.fa 2> $T/a.err; e=$(date +%s.%N)
    python3 -c "print('yak-amd count -k21 on $n x 100 Mb, YAKAMD_COLD_GB=$cold: wall %.2f s' % ($e - $s))"; grep "sweeps over\|yak_count::" $T/a.err | tail -2 | cut -c1-250; md5sum $T/a.yak | cut -c1-32
  done
  rm -f $T/asm$n.fa $T/a.yak
done
