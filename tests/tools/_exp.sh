run(){ env "$@" timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$*', 'ms', round(d['ms_per_step'],1), [(k['kernel'][:7], round(k['ms'],2)) for k in d['roofline']['all_kernels'][:3]])
"; }
run YAKAMD_CH2=16384
run YAKAMD_CH2=65536
run YAKAMD_CH2=131072
run YAKAMD_CH2=262144
