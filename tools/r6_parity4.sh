#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6par9}; rm -rf $O; mkdir -p $O
for cfg in "2e-6 3"; do
  set -- $cfg
  echo "== nudge $1 passes $2"
  HWG_TF_NUDGE=$1 HWG_TF_FORCING_PASSES=$2 HWG_PARITY_SUMMARY=$GRAFT_REPO_ROOT/$O/parity_summary_$1_$2.txt timeout 900 python -m pytest tests/test_trainer_lessons_gpu.py -q -k "teacher_forced and (tf_full or tf_trained)" > $O/parity_tests_$1_$2.log 2>&1
  tail -3 $O/parity_tests_$1_$2.log | cut -c1-300
  grep -n "forced pass\|outside its\|forcing passes" $O/parity_summary_$1_$2.txt | cut -c1-330
done
