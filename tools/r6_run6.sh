#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6m}; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_trainer_gpu.py tests/test_trainer_lessons_gpu.py tests/test_ddp_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log | cut -c1-300
for v in "HWG_BALANCE_SETS=1" "HWG_BALANCE_SETS=0" "HWG_BALANCE_SETS=1"; do
  echo "== $v"
  env $v HWG_BENCH_NO_MINNEC=1 timeout 300 python bench.py --steps 70 --warmup 7 --no-cpu-baseline --no-gen --no-other-workloads 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        j = json.loads(l); print(j['value'], (j.get('whole_cycles') or {}).get('value'), j.get('per_lesson_ms'))"
done > $O/ab.txt 2>&1
cat $O/ab.txt
HWG_BENCH_NO_MINNEC=1 timeout 300 python bench.py --workload iam_gan_b1a1_w512 --steps 70 --warmup 7 --no-cpu-baseline --no-gen --no-other-workloads 2>/dev/null | cut -c1-300
