#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6g}; rm -rf $O; mkdir -p $O
for v in "X=1" "HWG_C1_ROWS=0" "HWG_NORM_BIG=0" "HWG_C1_ROWS=0 HWG_NORM_BIG=0" "HWG_WGRAD_C1_ROWS=0 HWG_C1_ROWS=0 HWG_NORM_BIG=0 HWG_COL2IM_LDS=0"; do
  echo "== $v"; env $v timeout 600 python -m pytest tests/test_pipeline_gpu.py -m gpu -x -q -k "count_lesson_recogniser" -s 2>&1 | grep -E "error |passed|failed|no schedule" | cut -c1-260
done > $O/count_variants.txt 2>&1
cat $O/count_variants.txt
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "mlp_chain or c1 or conv_fwd_bwd or random or onerow" > $O/pytest.log 2>&1; tail -4 $O/pytest.log | cut -c1-300
PROBE="$(tr '\n' ';' < tools/probes/probe_r6_c1.txt)" timeout 300 python tools/conv_probe.py > $O/probe_c1wgrad.txt 2>&1; cat $O/probe_c1wgrad.txt | tail -9
