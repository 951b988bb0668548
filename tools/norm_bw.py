"""Effective bandwidth of the normalisation kernels on the step's largest tensors (forward: 2 reads + 1 write, backward: 5 reads + 1 write)."""
import sys, torch
sys.path.insert(0, '.')
from handwriting_line_generation_amd import ops
dev = torch.device('cuda:0')
for (N, H, W, C, kind) in [(8, 64, 512, 64, 'bn'), (8, 32, 256, 128, 'bn'), (4, 32, 514, 128, 'gn'), (4, 66, 1026, 64, 'gn'), (16, 64, 512, 64, 'gn'), (8, 16, 128, 256, 'bn'), (8, 64, 488, 16, 'gn')]:
    x = torch.randn(N, H, W, C, device=dev, requires_grad=True)
    g = torch.ones(C, device=dev, requires_grad=True); b = torch.zeros(C, device=dev, requires_grad=True)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    f = (lambda: ops.batch_norm_train(x, g, b, rm, rv, 0.1, 1e-5, act=ops.ACT_RELU)) if kind == 'bn' else (lambda: ops.group_norm(x, 8, g, b, 1e-5, act=ops.ACT_RELU))
    y = f(); dy = torch.randn_like(y)
    for _ in range(3): y = f(); y.backward(dy)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    for _ in range(20): y = f()
    e[1].record()
    for _ in range(20):
        y = f(); y.backward(dy)
    e[2].record(); torch.cuda.synchronize()
    tf = e[0].elapsed_time(e[1]) / 20 * 1e3; tb = e[1].elapsed_time(e[2]) / 20 * 1e3 - tf
    mb = x.numel() * 4 / 1e6
    print("%s %-22s %6.1f MB  fwd %6.1f us (%.2f TB/s at 3 passes)  bwd %6.1f us (%.2f TB/s at 6 passes)" % (kind, (N, H, W, C), mb, tf, 3 * mb / tf, tb, 6 * mb / tb))
