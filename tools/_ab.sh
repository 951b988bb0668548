cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "winograd_weight" 2>&1 | tail -2
python bench.py --steps 28 --warmup 7 --no-cpu-baseline --no-gen 2>/dev/null | cut -c1-120
python bench.py --steps 28 --warmup 7 --no-cpu-baseline --no-gen 2>/dev/null | cut -c1-120
