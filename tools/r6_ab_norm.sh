#!/bin/bash
# same-box A/B of the 1024-thread normalisation workgroups (HWG_NORM_BIG=0 off / default on): norm tests first, then two traced bench runs each way
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_modules_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python tools/norm_bw.py > $O/norm_bw_big.txt 2>&1; HWG_NORM_BIG=0 timeout 300 python tools/norm_bw.py > $O/norm_bw_small.txt 2>&1
for rep in 1 2; do
for v in 0 256; do
  HWG_NORM_BIG=$v HWG_BENCH_NO_MINNEC=1 timeout 300 python bench.py --steps 70 --warmup 7 --no-cpu-baseline --no-gen --no-other-workloads > $O/bench_${v}_$rep.json 2>/dev/null
  python - <<PY
import json
j=json.loads([l for l in open("$O/bench_${v}_$rep.json") if l.startswith('{"metric"')][-1])
print("NORM_BIG=$v rep $rep value", j["value"], "whole", j["whole_cycles"]["value"], j["per_lesson_ms"])
PY
done
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 256; do
  HWG_NORM_BIG=$v HWG_BENCH_NO_MINNEC=1 timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt$v -o kt -f csv -- python3 bench.py --steps 28 --warmup 7 --no-cpu-baseline --no-gen --no-other-workloads > $O/kt$v.log 2>&1
  find $O/kt$v -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$v.csv \;
  python tools/launch_census.py $O/kernel_stats_$v.csv $O/kt$v.log > $O/census_$v.txt 2>&1
  rm -rf $O/kt$v
  head -1 $O/census_$v.txt; grep -E "moments|apply_|finalize" $O/census_$v.txt
done
