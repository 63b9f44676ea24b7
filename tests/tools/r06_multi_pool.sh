#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06mpool}; mkdir -p $O
R=${READS:-37500000}
SL="--knob YAKAMD_MGPU_SLOT_PER_RANK=1 --knob YAKAMD_MGPU_LOOPBACK=1"
for v in overlap nooverlap; do
  X=""; [ $v = nooverlap ] && X="--knob YAKAMD_MGPU_NO_OVERLAP=1"
  YAKAMD_VERBOSE=1 timeout 900 python bench.py --gpus 2 --reads $R --steps 2 --warmup 1 --no-verify --no-cpu-baseline --no-weak-base $SL $X $EXTRA > $O/bench_$v.json 2> $O/bench_$v.err
  python3 -c "
import json,sys
d=json.loads([l for l in open('$O/bench_$v.json') if l.startswith('{')][-1]); print('$v', round(d['ms_per_step'],1), d.get('first_job_ms'))"
  grep "pool after\|rounds in" $O/bench_$v.err | cut -c1-330 | tail -8
done
