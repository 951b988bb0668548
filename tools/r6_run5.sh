#!/bin/bash
# in-kernel partial sums + one-launch normalisations: bit-identity tests, A/B of the step, census
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6j}; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "split_partials or one_launch or forced or balanced or two_tap or merged or norm or adain or conv" > $O/pytest_new.log 2>&1; tail -5 $O/pytest_new.log | cut -c1-400
for v in "HWG_SPLIT_INKERNEL=1 HWG_NORM_FUSED=1" "HWG_SPLIT_INKERNEL=0 HWG_NORM_FUSED=0" "HWG_SPLIT_INKERNEL=1 HWG_NORM_FUSED=0" "HWG_SPLIT_INKERNEL=0 HWG_NORM_FUSED=1" "HWG_SPLIT_INKERNEL=1 HWG_NORM_FUSED=1"; do
  echo "== $v"
  env $v HWG_BENCH_NO_MINNEC=1 timeout 300 python bench.py --steps 70 --warmup 7 --no-cpu-baseline --no-gen --no-other-workloads 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        j = json.loads(l); print(j['value'], (j.get('whole_cycles') or {}).get('value'), j.get('per_lesson_ms'))"
done > $O/ab.txt 2>&1
cat $O/ab.txt
bash tools/collect_census.sh $O/census > $O/census.log 2>&1; head -60 $O/census/launch_census.txt | cut -c1-130
