timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sharded or partitioned_exchange" 2>&1 | tail -3
CASE0=1 NCASE=2 timeout 250 python tests/tools/dbg_shardscale.py 2>&1 | grep -v "lds tier\|amdgpu.ids" | tail -2
