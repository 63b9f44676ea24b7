#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02i; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -x -q 2>&1 | tail -15
YAKAMD_VERBOSE=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-qv --no-pcie 2>$O/b.err | grep '^{' > $O/b.json; grep -E "streaming replay" $O/b.err | tail -2
python - $O <<'PY'
import json,sys
d = json.load(open(sys.argv[1] + "/b.json")); p=d["phase_ms_last_step"]
print("cfg2 step", round(d["ms_per_step"],2), p["pass1"], d["phase_wall_ms_last_step"], d["verify"])
PY
timeout 300 python bench.py --config nofilter --no-cpu-baseline --no-pcie --steps 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().splitlines()[-1]); print('nofilter', d['ms_per_step'], d['phase_ms_last_step']['pass1'], d['verify'])"
timeout 300 python bench.py --config cfg4 --contigs 4 --contig-len 50000000 --steps 1 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().splitlines()[-1]); print('cfg4s', d['ms_per_step'], d['phase_ms_last_step'], d['verify'])"
timeout 600 python bench.py --config cfg4 --contigs 10 --contig-len 100000000 --steps 1 --warmup 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().splitlines()[-1]); print('cfg4 1G', d['ms_per_step'], d['phase_ms_last_step'], d['verify'])"
