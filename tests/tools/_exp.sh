timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "partitioned_exchange or prefix_sharded" 2>&1 | tail -3
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 700 bash tests/tools/two_ranks_one_gpu.sh 1000000 2>&1 | tail -4
for fe in "" "--force-exchange"; do
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify $fe 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$fe', 'ms', round(d['ms_per_step'],1), d['final_distinct'], d['phase_wall_ms_last_step'], [(k['kernel'][:9], round(k['ms'],2)) for k in d['roofline']['all_kernels']])
"
done
