run(){ env "$@" timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$*', 'ms', round(d['ms_per_step'],1), d['verify'].get('yak_md5'), d['phase_ms_last_step']['pass1'], [(k['kernel'][:9], round(k['ms'],2)) for k in d['roofline']['all_kernels']][2:3])
"; }
run A=1
bash tests/quick_gpu.sh 2>&1 | grep -c "^OK"
