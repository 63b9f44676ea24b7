#!/bin/bash
# parity tests + A/B runs given as arguments
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4
bash tests/tools/ab.sh "$@"
