#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6s}; rm -rf $O; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest_gpu.log 2>&1; echo "rc $?" >> $O/pytest_gpu.log; tail -30 $O/pytest_gpu.log | cut -c1-300
