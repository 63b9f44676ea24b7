run(){ env "$@" timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$*', 'ms', round(d['ms_per_step'],1), d['verify'].get('yak_md5'), d.get('replay_doublings_parallel_vs_serial_fallback'), [(k['kernel'][:9], round(k['ms'],2)) for k in d['roofline']['all_kernels']][-1:])
"; }
run A=1
YAKAMD_DBG=32 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify 2>&1 | grep "replay block" | tail -2
