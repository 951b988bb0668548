#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6q}; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "mlp_chain or c1 or conv_fwd_bwd or random or onerow" > $O/pytest.log 2>&1; tail -12 $O/pytest.log | cut -c1-300
PROBE="$(tr '\n' ';' < tools/probes/probe_r6_c1fwd.txt)" timeout 300 python tools/conv_probe.py > $O/probe_c1fwd.txt 2>&1; cat $O/probe_c1fwd.txt | tail -12
PROBE="$(tr '\n' ';' < tools/probes/probe_r6_onerow.txt)" timeout 300 python tools/conv_probe.py > $O/probe_onerow.txt 2>&1; tail -3 $O/probe_onerow.txt
HWG_BENCH_NO_MINNEC=1 timeout 300 python bench.py --workload rimes_gan_b4a2_w256_1024 --steps 42 --warmup 7 --no-cpu-baseline --no-gen --no-other-workloads > $O/rimes.json 2> $O/rimes.err; grep -c "stays on the eager" $O/rimes.err; python -c "
import json;j=json.loads([l for l in open('$O/rimes.json') if l.startswith('{\"metric\"')][-1]);print(j['value'],j['whole_cycles'],j['replay'])"
