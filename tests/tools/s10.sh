#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/s10; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -30 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
