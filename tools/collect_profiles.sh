#!/bin/bash
# Run on the GPU box (gpurun): bench line, rocprofv3 kernel stats of the same command, PMC passes; outputs under gpurun_out/profiles_run/
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_run
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 400 python bench.py --steps 28 --warmup 7 > $OUT/bench_line.json 2> $OUT/bench_err.log
HWG_CONV_DUMP=$OUT/conv_shapes.txt timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/ktrace -o kt -f csv -- python3 bench.py --steps 28 --warmup 7 --no-cpu-baseline > $OUT/ktrace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pf -- python3 bench.py --steps 7 --warmup 2 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o pw -- python3 bench.py --steps 7 --warmup 2 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
python tools/pmc_traffic.py $OUT/pmc_fetch/pf_results.db $OUT/pmc_write/pw_results.db > $OUT/pmc_traffic.json 2>&1
rm -rf $OUT/pmc_fetch $OUT/pmc_write
ls -la $OUT $OUT/ktrace | head -30
tail -1 $OUT/bench_line.json | cut -c1-300
