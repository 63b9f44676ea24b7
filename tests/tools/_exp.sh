for fe in "--force-exchange" "--force-exchange --no-overlap"; do
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-qv $fe 2>/tmp/fe.err | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$fe', 'ms', round(d['ms_per_step'],1), d['final_distinct'], d['phase_wall_ms_last_step'])
" || tail -5 /tmp/fe.err
done
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 700 bash tests/tools/two_ranks_one_gpu.sh 1000000 2>&1 | tail -3
