#!/bin/bash
# full GPU suite, col2im A/B, then the round-6 artefact collection
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6i; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log | cut -c1-300
for v in 1 0; do
  echo "== HWG_COL2IM_LDS=$v"
  HWG_COL2IM_LDS=$v PROBE="$(grep -v "^#" tools/probes/probe_r6_c1.txt | tr "\n" ";")" timeout 300 python tools/conv_probe.py 2>&1 | tail -9
done > $O/col2im_ab.txt 2>&1
cat $O/col2im_ab.txt
bash tools/collect_profiles_r6.sh > $O/collect.log 2>&1
tail -60 $O/collect.log
