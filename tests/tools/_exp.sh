YAKAMD_DBG=32 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify 2>&1 | grep "replay block" | tail -2
