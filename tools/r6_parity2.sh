#!/bin/bash
# the parity summary of the round (the same selection as tools/collect_profiles_r6.sh) -> gpurun_out/<tag>/parity_summary.txt
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6par}; rm -rf $O; mkdir -p $O
HWG_PARITY_SUMMARY=$GRAFT_REPO_ROOT/$O/parity_summary.txt timeout 1500 python -m pytest tests/test_trainer_lessons_gpu.py tests/test_pretrain_trainers_gpu.py tests/test_pipeline_gpu.py tests/test_ops_gpu.py tests/test_modules_gpu.py tests/test_ddp_gpu.py -q -k "teacher_forced or match_reference_per_tensor or forcing or pretrain or reference_trainer or gate_flips or adversarial or full_size or module_parity or two_ranks" > $O/parity_tests.log 2>&1
tail -4 $O/parity_tests.log | cut -c1-300
grep -n "compared" $O/parity_summary.txt | grep -v "u0.count" | head -8 | cut -c1-220
grep -n "forced pass\|outside its" $O/parity_summary.txt | cut -c1-260
