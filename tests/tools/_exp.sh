timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py 2>/tmp/b.err | tee gpurun_out/r01i_bench_default.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value', round(d['value']/1e6,1), 'M distinct/s; ms', round(d['ms_per_step'],1), 'frac', round(d['roofline']['frac'],3), d['roofline']['kernel'], 'cpu', round(d['cpu_baseline']['value']/1e6,2), d['cpu_baseline']['kind'], 'qv', round(d['qv_lookup_probe']['lookups_per_s']/1e9,1))"
tail -2 /tmp/b.err
