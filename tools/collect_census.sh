#!/bin/bash
# launch census of the default bench workload: rocprofv3 kernel trace -> launches / average / time per step by kernel   (GPU box)
set -u
OUT=$GRAFT_REPO_ROOT/${1:-gpurun_out/census}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
HWG_BENCH_NO_MINNEC=1 timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -f csv -- python3 bench.py --steps 28 --warmup 7 --no-cpu-baseline --no-gen --no-other-workloads > $OUT/kt.log 2>&1
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
KT=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
[ -n "$KT" ] && python tools/kernel_by_grid.py $KT 300 > $OUT/kernel_by_grid.txt 2>&1
python tools/launch_census.py $OUT/kernel_stats.csv $OUT/kt.log > $OUT/launch_census.txt 2>&1
rm -rf $OUT/kt
head -100 $OUT/launch_census.txt | cut -c1-150
