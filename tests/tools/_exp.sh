run(){ env "$@" timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$*', 'ms', round(d['ms_per_step'],1), [(k['kernel'][:9], round(k['ms'],2)) for k in d['roofline']['all_kernels']][-1:])
"; }
run YAKAMD_REPLAY_LDS=16384 YAKAMD_REPLAY_THREADS=1024
run YAKAMD_REPLAY_LDS=16384 YAKAMD_REPLAY_THREADS=512
run YAKAMD_REPLAY_LDS=8192 YAKAMD_REPLAY_THREADS=512
run YAKAMD_REPLAY_LDS=8192 YAKAMD_REPLAY_THREADS=256
run YAKAMD_REPLAY_LDS=32768 YAKAMD_REPLAY_THREADS=512
