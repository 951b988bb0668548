#!/bin/bash
# round 6, first GPU call: full GPU suite, call trace, default bench line, launch census
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6a; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout 300 python tools/call_trace.py $O/call_trace.txt > $O/call_trace.log 2>&1; tail -2 $O/call_trace.log
HWG_CONV_DUMP=$O/conv_shapes.txt timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
bash tools/collect_census.sh $O/census > $O/census.log 2>&1; head -5 $O/census/launch_census.txt
