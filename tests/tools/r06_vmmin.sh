#!/bin/bash
# cfg3shard and the 5 Gb job with the chunk tier's threshold at 2 and 4 GiB (default 1): the ~1.1 GB per-round buffers of a rank's feeds back in the superblocks
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06vmmin}; mkdir -p $O
for m in 2147483648 4294967296; do
  sleep 5
  YAKAMD_VERBOSE=1 timeout 600 python bench.py --config cfg3shard --warmup 1 --knob YAKAMD_POOL_VM_MIN=$m > $O/bench_cfg3shard_$m.json 2> $O/bench_cfg3shard_$m.err
  python3 - $O/bench_cfg3shard_$m.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], "ms", round(d["ms_per_step"], 2), d.get("rank_seconds"), d.get("first_job_rank_seconds"), {k: x for k, x in (d.get("verify") or {}).items() if isinstance(x, bool)})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
  grep "pool after" $O/bench_cfg3shard_$m.err | tail -2 | cut -c1-520
done
sleep 5
YAKAMD_VERBOSE=1 timeout 600 python bench.py --config cfg4 --contigs 50 --warmup 1 --no-verify --knob YAKAMD_POOL_VM_MIN=2147483648 > $O/bench_cfg4_5gb.json 2> $O/bench_cfg4_5gb.err
python3 - $O/bench_cfg4_5gb.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], "ms", round(d["ms_per_step"], 2), d["config"].get("sweeps_of_every_job"), {k: d[k] for k in d if k.startswith(("first_job_ms", "peak_hbm_bytes")) and "note" not in k})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
grep "pool after" $O/bench_cfg4_5gb.err | tail -2 | cut -c1-520
