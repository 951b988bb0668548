cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
HWG_WINO_WGRAD=2 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/wg6 -o wg -- python3 tools/pmc_shapes.py time tools/data/_three.txt > gpurun_out/wg6.log 2>&1
python3 - <<'PY'
import sqlite3,glob
db=glob.glob('gpurun_out/wg6/**/*.db', recursive=True)[0]
c=sqlite3.connect(db).cursor()
for r in c.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%wino_wgrad_kernel%' group by counter_name"): print(r[0], r[1], round(r[2]))
PY
grep "us  x" gpurun_out/wg6.log | cut -c1-60; python -m pytest tests/test_ops_gpu.py -q -m gpu -k winograd_weight 2>&1 | tail -1; PMC_SHAPES_KIND=wgrad HWG_WINO_WGRAD=2 python tools/pmc_shapes.py time tools/data/_three.txt | grep "us  x" | cut -c1-60
