#!/bin/bash
# PMC passes over a list of single-conv probes (tools/conv_probe.py): usage  pmc_probe.sh <probe file> <out dir>   (GPU box)
set -u
PF=$1; OUT=$GRAFT_REPO_ROOT/$2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PROBE="$(grep -v '^#' $PF | tr '\n' ';')"
python tools/conv_probe.py > $OUT/plain.txt 2>&1
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/p$i -o p -- python3 tools/conv_probe.py > $OUT/p$i.log 2>&1
done
python tools/pmc_probe.py $(find $OUT -name "*.db") > $OUT/pmc_probe.txt 2>&1
rocprofv3 -L > $OUT/avail.txt 2>&1 || rocprofv3 --list-avail > $OUT/avail.txt 2>&1
find $OUT -name "*.db" -delete
cat $OUT/plain.txt; cat $OUT/pmc_probe.txt | cut -c1-400
