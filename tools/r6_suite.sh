#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6s}; rm -rf $O; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 1700 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "rc $?" >> $O/pytest_gpu.log; tail -14 $O/pytest_gpu.log | cut -c1-300
