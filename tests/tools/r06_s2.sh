#!/bin/bash
# (1) the LDS conflict floor of random accesses (pmc_mb_lds.sh), (2) hipMemCreate from several threads (mb_vmm5), (3) a cfg3 rank share with the level-2
# partition forced into ONE sweep of 2^13 / 2^12 sub-buckets per sub-table (k_lc2 then takes sub-buckets of 8 K / 17 K records in rounds of 768)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06s2}; mkdir -p $O
bash tests/tools/pmc_mb_lds.sh $(basename $O) > $O/mb_lds.log 2>&1
hipcc --offload-arch=gfx950 -O2 tests/tools/mb/mb_vmm5.hip -o /tmp/mb_vmm5 -lpthread 2> /dev/null && timeout 300 /tmp/mb_vmm5 200 > $O/mb_vmm5.txt 2>&1
for s in 0 13 12; do
  if [ $s = 0 ]; then E=""; else E="YAKAMD_S2_BITS=$s"; fi
  env $E YAKAMD_VERBOSE=1 timeout 400 python bench.py --config cfg3shard --warmup 1 > $O/bench_cfg3shard_s$s.json 2> $O/bench_cfg3shard_s$s.err
  python3 - $O/bench_cfg3shard_s$s.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], "ms", round(d["ms_per_step"], 2), d.get("rank_seconds"), {k: x for k, x in (d.get("verify") or {}).items() if isinstance(x, bool)})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
  grep -h "k_lc2:\|two sweeps\|passed on" $O/bench_cfg3shard_s$s.err | tail -6
done
cat $O/mb_vmm5.txt; cat $O/mb_lds_counters.txt
