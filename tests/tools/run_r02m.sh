#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r02m}; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -40 > $O/t.log
tail -40 $O/t.log
