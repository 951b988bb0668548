#!/bin/bash
# the driver's command and a 140-step run, on whatever box the lease gives (bench lines of several boxes: profiles/r06_bench_boxes.txt)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6b}; rm -rf $O; mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.log
timeout 600 python bench.py --steps 140 --warmup 14 --no-cpu-baseline --no-other-workloads > $O/bench_line_long.json 2>> $O/bench_err.log
timeout 600 python bench.py > $O/bench_line_default.json 2>> $O/bench_err.log
for f in $O/bench_line*.json; do grep '^{"metric"' $f | tail -1 | cut -c1-330; done
