#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6e}; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
HWG_BENCH_NO_MINNEC=1 timeout 300 python bench.py --workload rimes_gan_b4a2_w256_1024 --steps 42 --warmup 7 --no-cpu-baseline --no-gen --no-other-workloads > $O/rimes.json 2> $O/rimes.err; grep -c "stays on the eager" $O/rimes.err; grep "stays on the eager" $O/rimes.err | cut -c1-400 | sort | uniq -c | head -20
HWG_CONV_DUMP=$O/conv_shapes.txt timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
bash tools/collect_census.sh $O/census > $O/census.log 2>&1; head -3 $O/census/launch_census.txt
