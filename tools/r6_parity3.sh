#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6par5}; rm -rf $O; mkdir -p $O
HWG_PARITY_SUMMARY=$GRAFT_REPO_ROOT/$O/parity_summary.txt timeout 900 python -m pytest tests/test_trainer_lessons_gpu.py -q -k "teacher_forced and tf_full" > $O/parity_tests.log 2>&1
tail -3 $O/parity_tests.log | cut -c1-300
grep -n "forced:" $O/parity_summary.txt | cut -c1-260
grep -n "forcing passes\|added by a forcing" $O/parity_summary.txt | cut -c1-300
