import os, sys, time, json, torch
sys.path.insert(0, '.')
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)), 'torch threads', torch.get_num_threads(), flush=True)
for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
    if os.path.exists(f): print(f, open(f).read().strip(), flush=True)
os.system("grep -m1 'model name' /proc/cpuinfo")
from oracle import cycle_ref
from handwriting_line_generation_amd.model import HWWithStyle, Autoencoder
cfg = json.load(open('tests/golden/model_config_iam.json'))
m = HWWithStyle(cfg); ae = Autoencoder({'type': '2tight', 'hwr': 80})
tr = {k for k, p in m.named_parameters() if p.requires_grad}
enc = {k[8:]: v for k, v in ae.state_dict().items() if k.startswith('encoder.')}
for nt in (int(sys.argv[1]),):
    torch.set_num_threads(nt)
    t = time.time()
    r = cycle_ref.time_cycles(m.state_dict(), tr, enc, 4, 2, 256, 12, budget_s=1, max_cycles=1)
    print('threads', nt, r, flush=True)
