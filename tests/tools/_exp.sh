timeout 800 python -m pytest tests/test_qv.py -m gpu -x -q 2>&1 | tail -2
timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms', round(d['ms_per_step'],1), d['qv_lookup_probe'])
"
