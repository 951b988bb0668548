"""How much slack does the host have? Runs bench.py's timed loop with every C-ABI call delayed by a busy-wait of DELAY_US microseconds
(first argument; the rest are bench.py's arguments). Where the step is GPU-bound the throughput does not move until the added host time
exceeds the slack; where it is host-bound it drops by the added time.   python tools/host_slack.py 4 --steps 20 --warmup 5"""
import sys, time
sys.path.insert(0, '.')
delay = float(sys.argv[1]) * 1e-6
sys.argv = [sys.argv[0]] + sys.argv[2:] + ["--no-cpu-baseline", "--no-gen"]
import os
os.environ["HWG_BENCH_NO_MINNEC"] = "1"; os.environ["HWG_BENCH_NO_PROF"] = "1"
from handwriting_line_generation_amd import _lib as L
_call = L.call
count = [0]
if delay > 0:
    def call(name, *a):
        count[0] += 1
        t = time.perf_counter() + delay
        while time.perf_counter() < t:
            pass
        return _call(name, *a)
    L.call = call
import bench
bench.main()
print("delayed calls", count[0], file=sys.stderr)
