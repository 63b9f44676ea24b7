#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02g; mkdir -p $O
echo "== default bench (all legs)"; s=$(date +%s); timeout 900 python bench.py > $O/default.json 2>$O/default.err; echo "rc $? wall $(( $(date +%s) - s )) s"; tail -3 $O/default.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02g/default.json"))
r = d["roofline"]
print("step", d["ms_per_step"], "value", d["value"], "pass1_frac", r["pass1_frac"], "p1 core", r["pass1_extract_insert_frac"], "pass2_frac", r["pass2_frac"], "step_frac", r["step_frac"])
print("pcie", d["pcie_inclusive_ms"], "e2e", d["e2e_cli"], "\ncpu", d["cpu_baseline"])
print("qv", d["qv_lookup_probe"])
PY
echo "== nofilter"; timeout 600 python bench.py --config nofilter --no-cpu-baseline --no-pcie --steps 2 > $O/nofilter.json 2>$O/nofilter.err; echo rc $?; tail -2 $O/nofilter.err
python -c "
import json; d=json.load(open('gpurun_out/r02g/nofilter.json')); print(d['ms_per_step'], d['phase_ms_last_step']['pass1'], d['verify'], d['roofline']['step_frac'])"
echo "== cfg4 small"; timeout 600 python bench.py --config cfg4 --contigs 4 --contig-len 50000000 --steps 1 --warmup 1 > $O/cfg4s.json 2>$O/cfg4s.err; echo rc $?; tail -3 $O/cfg4s.err; cat $O/cfg4s.json | head -c 1500; echo
echo "== cfg5"; timeout 600 python bench.py --config cfg5 --steps 2 --warmup 1 > $O/cfg5.json 2>$O/cfg5.err; echo rc $?; tail -3 $O/cfg5.err; cat $O/cfg5.json | head -c 1500; echo
echo "== 2 ranks on one GPU (gloo), self-spawn"; timeout 600 python bench.py --gpus 2 --backend gloo --reads 400000 --batch-reads 150000 --job-md5 --no-cpu-baseline --steps 1 --warmup 0 > $O/two.json 2>$O/two.err; echo rc $?; tail -3 $O/two.err
timeout 300 python bench.py --reads 800000 --job-md5 --no-cpu-baseline --no-qv --no-pcie --steps 1 --warmup 0 > $O/one.json 2>$O/one.err; echo rc $?
python -c "
import json; a=json.load(open('gpurun_out/r02g/two.json')); b=json.load(open('gpurun_out/r02g/one.json')); print('two-rank md5', a['job_yak_md5'], 'one-rank md5', b['job_yak_md5'], a['job_yak_md5']==b['job_yak_md5'], a['ms_per_step'], b['ms_per_step'])"
