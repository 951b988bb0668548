"""Effective bandwidth of the normalisation kernels on the step's tensors (forward: 2 reads + 1 write, backward: 5 reads + 1 write), measured on
the C-ABI calls themselves: 50 back-to-back launches of hwg_norm_fwd / hwg_norm_bwd between two events, so the figure is GPU time per call
(moments + apply kernels), not the host's autograd overhead (round 3's version timed `y.backward()` from Python and was host-bound on the
16 MB tensors: 0.6 TB/s there said nothing about the kernels)."""
import sys, torch
sys.path.insert(0, '.')
from handwriting_line_generation_amd import _lib as L, ops
dev = torch.device('cuda:0')
st = torch.cuda.current_stream().cuda_stream
ITERS = 50
for (N, H, W, C, kind) in [(8, 64, 512, 64, 'bn'), (8, 32, 256, 128, 'bn'), (4, 32, 514, 128, 'gn'), (4, 66, 1026, 64, 'gn'), (16, 64, 512, 64, 'gn'), (8, 16, 128, 256, 'bn'),
                           (8, 64, 488, 16, 'gn'), (8, 8, 129, 512, 'bn'), (8, 64, 488, 16, 'in')]:
    mode = {'in': ops.NORM_IN, 'gn': ops.NORM_GN, 'bn': ops.NORM_BN}[kind]
    HW = H * W
    x = torch.randn(N, H, W, C, device=dev); y = torch.empty_like(x); dy = torch.randn_like(x); dx = torch.empty_like(x)
    g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev); dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    mean = torch.empty(N, C, device=dev); rstd = torch.empty(N, C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    ws = torch.empty(L.query("hwg_norm_workspace", N, HW, C), dtype=torch.uint8, device=dev)
    aff = (g, b) if kind != 'in' else (None, None)

    def fwd():
        L.call("hwg_norm_fwd", x, y, N, HW, C, mode, 8, 1e-5, aff[0], aff[1], 0, None, ops.ACT_RELU, 0.0, mean, rstd,
               rm if kind == 'bn' else None, rv if kind == 'bn' else None, 0.1, ws, ws.numel(), st)

    def bwd():
        L.call("hwg_norm_bwd", dy, x, y, dx, N, HW, C, mode, 8, aff[0], aff[1], 0, None, ops.ACT_RELU, 0.0, mean, rstd,
               dg if kind != 'in' else None, db if kind != 'in' else None, 1, ws, ws.numel(), st)
    res = []
    for fn in (fwd, bwd):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(ITERS): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / ITERS * 1e3)
    mb = x.numel() * 4 / 1e6
    print("%s %-22s %6.1f MB  fwd %6.1f us (%.2f TB/s at 3 passes)  bwd %6.1f us (%.2f TB/s at 6 passes)" % (kind, (N, H, W, C), mb, res[0], 3 * mb / res[0], res[1], 6 * mb / res[1]))
