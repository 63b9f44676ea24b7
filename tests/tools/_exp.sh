timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('ms', round(d['ms_per_step'],1), {k:(round(v,4) if isinstance(v,float) else v) for k,v in r.items() if k not in ('all_kernels','model','traffic_source')})
for k in r['all_kernels']: print(k['kernel'][:20], round(k['ms'],2), round(k['achieved_GBs']), round(k['frac'],3), k.get('traffic_bytes'))
"
