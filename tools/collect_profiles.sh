#!/bin/bash
# Run on the GPU box (gpurun): bench line, rocprofv3 kernel stats of the same command, PMC passes (whole step + per conv shape);
# outputs under gpurun_out/profiles_run/ - copy the ones to keep into profiles/ (tools/prof_summary.py names them per round)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_run
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 500 python bench.py --steps 28 --warmup 7 > $OUT/bench_line.json 2> $OUT/bench_err.log
HWG_CONV_DUMP=$OUT/conv_shapes.txt timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/ktrace -o kt -f csv -- python3 bench.py --steps 28 --warmup 7 --no-cpu-baseline --no-gen > $OUT/ktrace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pf -- python3 bench.py --steps 7 --warmup 2 --no-cpu-baseline --no-gen > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o pw -- python3 bench.py --steps 7 --warmup 2 --no-cpu-baseline --no-gen > $OUT/pmc_write.log 2>&1
python tools/pmc_traffic.py $OUT/pmc_fetch/pf_results.db $OUT/pmc_write/pw_results.db > $OUT/pmc_traffic.json 2>&1
rm -rf $OUT/pmc_fetch $OUT/pmc_write
# per-shape traffic of the conv kernels: the bench run's top shapes replayed one by one under the same two counters
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/shapes/fetch -o sf -- python3 tools/pmc_shapes.py run $OUT/conv_shapes.txt > $OUT/shapes_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/shapes/write -o sw -- python3 tools/pmc_shapes.py run $OUT/conv_shapes.txt > $OUT/shapes_write.log 2>&1
python tools/pmc_shapes.py parse $OUT/conv_shapes.txt $OUT/shapes $OUT/pmc_shapes.json > $OUT/shapes_parse.log 2>&1
rm -rf $OUT/shapes
ls -la $OUT $OUT/ktrace | head -30
tail -3 $OUT/shapes_parse.log $OUT/shapes_fetch.log
tail -1 $OUT/bench_line.json | cut -c1-300
