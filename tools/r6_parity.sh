#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6i}; rm -rf $O; mkdir -p $O
HWG_PARITY_SUMMARY=$GRAFT_REPO_ROOT/$O/parity_summary.txt timeout 1500 python -m pytest tests/test_trainer_lessons_gpu.py tests/test_pipeline_gpu.py tests/test_modules_gpu.py -m gpu -q --durations=25 > $O/pytest.log 2>&1; tail -40 $O/pytest.log | cut -c1-200
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "conv_fwd_bwd or random" > $O/pytest_ops.log 2>&1; tail -3 $O/pytest_ops.log
PROBE="8,64,512,64,49,1,1,0,0;8,64,512,64,9,1,1,0,0;4,64,1024,64,25,1,1,0,0" timeout 300 python tools/conv_probe.py 2>&1 | tail -3
