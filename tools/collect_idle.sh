#!/bin/bash
# kernel trace of the default bench workload -> GPU idle per lesson (tools/gpu_idle.py); gpurun_out/idle/
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-idle}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
shift
HWG_BENCH_NO_MINNEC=1 HWG_BENCH_NO_PROF=1 timeout 400 rocprofv3 --kernel-trace -d $OUT/kt -o kt -f csv -- python3 bench.py --steps 28 --warmup 7 --no-cpu-baseline --no-gen "$@" > $OUT/kt.log 2>&1
tail -1 $OUT/kt.log | cut -c1-200
python tools/gpu_idle.py $(find $OUT/kt -name "*kernel_trace.csv" | head -1) | tee $OUT/idle.txt
rm -rf $OUT/kt
