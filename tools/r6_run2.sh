#!/bin/bash
# round 6: full GPU suite, default bench line (+ other workloads), launch census with per-grid classes
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6b}; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
HWG_CONV_DUMP=$O/conv_shapes.txt timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
bash tools/collect_census.sh $O/census > $O/census.log 2>&1; head -3 $O/census/launch_census.txt
