#!/bin/bash
# per-shape HBM traffic of the conv kernels (the step's top shapes replayed one by one under FETCH_SIZE / WRITE_SIZE), from a committed shape table
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6shapes; rm -rf $OUT; mkdir -p $OUT
T=${1:-profiles/r06_conv_shapes_b4a2_w512.txt}
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/shapes/fetch -o sf -- python3 tools/pmc_shapes.py run $T > $OUT/shapes_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/shapes/write -o sw -- python3 tools/pmc_shapes.py run $T > $OUT/shapes_write.log 2>&1
python tools/pmc_shapes.py parse $T $OUT/shapes $OUT/pmc_shapes.json > $OUT/shapes_parse.log 2>&1
rm -rf $OUT/shapes
tail -3 $OUT/shapes_parse.log; grep -i "error\|Traceback" -A3 $OUT/shapes_fetch.log | head
