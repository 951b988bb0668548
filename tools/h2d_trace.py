import sys, collections, atexit, traceback
sys.path.insert(0, '.')
sys.argv = ['bench.py', '--steps', '14', '--warmup', '7', '--no-cpu-baseline', '--no-gen']
import torch
from handwriting_line_generation_amd import ops
cnt = collections.Counter(); byt = collections.Counter()
orig = ops.h2d
def traced(host, device, dtype=None):
    f = sys._getframe(1)
    key = "%s:%d" % (f.f_code.co_filename.split('/')[-1], f.f_lineno)
    cnt[key] += 1
    try: byt[key] += torch.as_tensor(host).numel() * torch.as_tensor(host).element_size()
    except Exception: pass
    return orig(host, device, dtype)
ops.h2d = traced
import os
os.environ["HWG_BENCH_NO_MINNEC"] = "1"
def report():
    tot = sum(cnt.values())
    sys.stderr.write("h2d calls total %d (21 steps incl. warmup)\n" % tot)
    for k, v in cnt.most_common(40): sys.stderr.write("  %-40s %5d  avg %7.0f B\n" % (k, v, byt[k] / max(v, 1)))
atexit.register(report)
exec(compile(open('bench.py').read(), 'bench.py', 'exec'))
