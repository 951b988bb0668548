cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PMC_SHAPES_KIND=wino_conv python tools/pmc_shapes.py time tools/data/conv_shapes_b4a2_w512.txt > gpurun_out/ab_w7.log 2>&1
tail -n 1 gpurun_out/ab_w7.log
python bench.py --steps 28 --warmup 7 --no-cpu-baseline --no-gen 2>/dev/null | cut -c1-120
python bench.py --steps 28 --warmup 7 --no-cpu-baseline --no-gen 2>/dev/null | cut -c1-120
