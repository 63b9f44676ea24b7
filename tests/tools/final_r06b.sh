#!/bin/bash
# round-6 final measurements, part B (after part A's counter files were copied to profiles/): the bench lines, kernel stats of default / nofilter / cfg4 1 Gb /
# cfg3shard, the layout dispatch trace, the N = 2 line on one device (slots + test rig), the CLI end to end, the GPU test tier
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06final; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gpu_test_tier.txt 2>&1; tail -4 $O/gpu_test_tier.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default_steps20_warmup5.json 2> /dev/null
Q="--no-cpu-baseline --no-pcie --no-qv --no-packed --no-nofilter"
timeout 600 python bench.py --config nofilter $Q > $O/bench_nofilter.json 2> /dev/null
timeout 600 python bench.py --config cfg4 --contigs 10 --contig-len 100000000 > $O/bench_cfg4_1gb.json 2> /dev/null
timeout 600 python bench.py --config cfg4 --contigs 20 --contig-len 100000000 > $O/bench_cfg4_2gb.json 2> /dev/null
timeout 600 python bench.py --config cfg5 > $O/bench_cfg5.json 2> /dev/null
timeout 600 python bench.py --reads 30000000 $Q > $O/bench_30m.json 2> /dev/null
timeout 600 python bench.py --no-retain $Q --no-verify > $O/bench_noretain.json 2> /dev/null
YAKAMD_VERBOSE=1 timeout 900 python bench.py --config cfg3shard --warmup 1 > $O/bench_cfg3shard.json 2> $O/bench_cfg3shard.err
sleep 5
YAKAMD_VERBOSE=1 timeout 900 python bench.py --config cfg4 --contigs 50 --warmup 1 > $O/bench_cfg4_5gb_sweeps8.json 2> $O/bench_cfg4_5gb.err
sleep 5
timeout 900 python bench.py --config cfg4 --contigs 50 --sweeps 2 --warmup 1 --no-verify > $O/bench_cfg4_5gb_sweeps2.json 2> /dev/null
grep "ranks: input\|pool after" $O/bench_cfg4_5gb.err | head -4 > $O/cfg4_5gb_stages.txt
grep "pool after\|level-2 partition\|k_lc2\|slice of the pass" $O/bench_cfg3shard.err | tail -8 > $O/cfg3shard_stages.txt
sleep 5
timeout 900 python bench.py --gpus 2 --reads 37500000 --steps 2 --warmup 1 > $O/bench_gpus2.json 2> $O/bench_gpus2.err
timeout 900 python bench.py --gpus 2 --reads 37500000 --steps 2 --warmup 1 --no-verify --no-cpu-baseline --no-weak-base --knob YAKAMD_MGPU_SLOT_PER_RANK=1 --knob YAKAMD_MGPU_LOOPBACK=1 > $O/bench_gpus2_slots.json 2> $O/bench_gpus2_slots.err
for f in default default_steps20_warmup5 nofilter cfg4_1gb cfg4_2gb cfg5 30m noretain cfg3shard cfg4_5gb_sweeps8 cfg4_5gb_sweeps2 gpus2 gpus2_slots; do python3 - $O/bench_$f.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d.get("roofline", {})
    v = d.get("verify") or {}
    print(sys.argv[1].split("/")[-1], "ms", round(d["ms_per_step"], 2), "value", round(d["value"] / 1e6, 1), "M/s", {k: round(x, 4) for k, x in r.items() if ("frac" in k or k == "hbm_util") and isinstance(x, float)}, "traffic", r.get("traffic"), {k: x for k, x in v.items() if isinstance(x, bool)}, {k: d[k] for k in d if k.startswith(("first_job", "peak_hbm_bytes")) and "note" not in k})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
for cfg in "default:" "nofilter:--config nofilter" "cfg4_1gb:--config cfg4 --contigs 10 --contig-len 100000000" "cfg3shard:--config cfg3shard"; do
  name=${cfg%%:*}; args=${cfg#*:}
  S="--steps 3 --warmup 1"; [ $name = cfg3shard ] && S="--warmup 1"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$name -- python bench.py $S --no-cpu-baseline --no-verify --no-pcie --no-packed --no-nofilter --no-qv $args > $O/bench_profiled_$name.json 2>/dev/null
  cp $(find $O/trace_$name -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$name.csv
  rm -rf $O/trace_$name
done
python3 - $O/kernel_stats_default.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.4: print(r["Name"][:56].ljust(56), r["Calls"].rjust(5), "%9.3f ms avg" % (float(r["AverageNs"])/1e6), "%9.2f ms tot" % (float(r["TotalDurationNs"])/1e6), r["Percentage"]+"%")
PY
R2OUT=r06final/r2_nofilter bash tests/tools/trace_r2.sh --config nofilter > /dev/null 2>&1; cp gpurun_out/r06final/r2_nofilter/r2_dispatches.txt $O/r2_dispatches_nofilter.txt
timeout 300 bash tests/tools/r04_e2e.sh gz > $O/e2e_cli.txt 2>&1
ls $O
