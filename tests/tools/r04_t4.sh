#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04t4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_multi_c.py -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
cat $O/pytest.txt
