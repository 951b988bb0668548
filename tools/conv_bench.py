"""Per-shape throughput of the MFMA conv kernels on the layer shapes of the IAM GAN step (8 lines of 64x512)."""
import ctypes, sys, torch
sys.path.insert(0, '.')
from handwriting_line_generation_amd import _lib as L, ops

SHAPES = [  # name, N,H,W,C,K,R,S, stride, pad, transposed
    ("G0.conv2 256>256", 8, 4, 122, 256, 256, 3, 3, (1, 1), (1, 1), 0),
    ("G1.conv1 256>128", 8, 8, 122, 256, 128, 3, 3, (1, 1), (1, 1), 0),
    ("G1.conv2 128>128", 8, 8, 122, 128, 128, 3, 3, (1, 1), (1, 1), 0),
    ("G2.conv1 128>64", 8, 16, 122, 128, 64, 3, 3, (1, 1), (1, 1), 0),
    ("G2.conv2 64>64", 8, 16, 122, 64, 64, 3, 3, (1, 1), (1, 1), 0),
    ("G3.conv1 T4x4s2 64>32", 8, 16, 122, 64, 32, 4, 4, (2, 2), (1, 1), 1),
    ("G3.conv2 32>32", 8, 32, 244, 32, 32, 3, 3, (1, 1), (1, 1), 0),
    ("G4.conv1 T4x4s2 32>16", 8, 32, 244, 32, 16, 4, 4, (2, 2), (1, 1), 1),
    ("G4.conv2 16>16", 8, 64, 488, 16, 16, 3, 3, (1, 1), (1, 1), 0),
    ("D.convs1.0 64>64", 8, 58, 488, 64, 64, 3, 3, (1, 1), (0, 1), 0),
    ("D.convs1.0 64>64 N16", 16, 58, 512, 64, 64, 3, 3, (1, 1), (0, 1), 0),
    ("D.convs1.3 64>128", 8, 28, 244, 64, 128, 3, 3, (1, 1), (0, 1), 0),
    ("D.convs2.0 128>128", 8, 26, 244, 128, 128, 3, 3, (1, 1), (0, 1), 0),
    ("D.convs3.0 128>128", 8, 12, 122, 128, 128, 3, 3, (1, 1), (0, 1), 0),
    ("D.convs3.4 128>256", 8, 5, 61, 128, 256, 3, 3, (1, 1), (0, 1), 0),
    ("H.conv1 64>128", 8, 32, 256, 64, 128, 3, 3, (1, 1), (1, 1), 0),
    ("H.conv2 128>256", 8, 16, 128, 128, 256, 3, 3, (1, 1), (1, 1), 0),
    ("H.conv3 256>256", 8, 16, 128, 256, 256, 3, 3, (1, 1), (1, 1), 0),
    ("H.conv4 256>512", 8, 8, 129, 256, 512, 3, 3, (1, 1), (1, 1), 0),
    ("H.conv5 512>512", 8, 8, 129, 512, 512, 3, 3, (1, 1), (0, 0), 0),
    ("H.conv6 512>512", 8, 3, 127, 512, 512, 3, 3, (1, 1), (0, 0), 0),
    ("H.cnn1d 512>512 d2", 8, 1, 125, 512, 512, 1, 3, (1, 1), (0, 2), 0),
    ("S.down1 4x4s2 64>128", 4, 66, 1026, 64, 128, 4, 4, (2, 2), (0, 0), 0),
    ("S.down2 128>128", 4, 32, 514, 128, 128, 3, 3, (1, 1), (0, 0), 0),
    ("S.down3 4x4s2 128>256", 4, 32, 514, 128, 256, 4, 4, (2, 2), (0, 0), 0),
    ("S.down4 256>256", 4, 15, 258, 256, 256, 3, 3, (1, 1), (0, 0), 0),
    ("S.down5 4x4s21 256>256", 4, 13, 258, 256, 256, 4, 4, (2, 1), (0, 0), 0),
]

def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

import os
AB = dict(kv.split("=") for kv in os.environ.get("CONV_BENCH_AB", "").split(";") if kv)     # e.g. CONV_BENCH_AB="HWG_CONV_PF=1": second fwd timing under these knobs
dev = torch.device('cuda:0')
st = torch.cuda.current_stream().cuda_stream
tot = {"fwd": [0, 0], "wgrad": [0, 0]}
print("%-26s %9s %8s | fwd us  TF/s | wgrad us TF/s" % ("layer", "M", "GFLOP"))
for (name, N, H, W, C, K, R, S, stride, pad, tr) in SHAPES:
    dil = (1, 1)
    if tr:
        P = (H - 1) * stride[0] - 2 * pad[0] + R; Q = (W - 1) * stride[1] - 2 * pad[1] + S
        pix = N * H * W
    else:
        P = (H + 2 * pad[0] - R) // stride[0] + 1; Q = (W + 2 * pad[1] - S) // stride[1] + 1
        pix = N * P * Q
    fl = 2.0 * pix * K * C * R * S
    x = torch.randn(N, H, W, C, device=dev); wp = torch.randn(R * S, K, C, device=dev) * 0.05
    y = torch.empty(N, P, Q, K, device=dev)
    d = ops._desc(N, H, W, C, K, R, S, stride, pad, dil, P, Q, tr)
    needf = L.query("hwg_conv_fwd_workspace", d.ptr); wsf = torch.empty(max(needf, 16), dtype=torch.uint8, device=dev)
    t = bench(lambda: L.call("hwg_conv_fwd", d.ptr, x, wp, None, y, 0, wsf, wsf.numel(), st))
    alt = ""
    if AB:
        with ops.tuning(**AB):
            needf = L.query("hwg_conv_fwd_workspace", d.ptr); wsf = torch.empty(max(needf, 16), dtype=torch.uint8, device=dev)
            t2 = bench(lambda: L.call("hwg_conv_fwd", d.ptr, x, wp, None, y, 0, wsf, wsf.numel(), st))
            alt = " | %s: %7.1f us %5.1f TF plan %s" % (AB, t2 * 1e6, fl / t2 / 1e12, ops.last_plan())
        L.call("hwg_conv_fwd", d.ptr, x, wp, None, y, 0, wsf if needf else None, needf, st) if False else None
    # weight gradient of the same layer: anchor = output side
    if tr:
        dw_desc = ops._desc(N, P, Q, K, C, R, S, stride, pad, dil, H, W); u, v = x, y
        dwt = torch.empty(C, K, R, S, device=dev); sa, sb = K * R * S, R * S
    else:
        dw_desc = ops._desc(N, H, W, C, K, R, S, stride, pad, dil, P, Q); u, v = y, x
        dwt = torch.empty(K, C, R, S, device=dev); sa, sb = C * R * S, R * S
    need = L.query("hwg_conv_wgrad_workspace", dw_desc.ptr); ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
    tw = bench(lambda: L.call("hwg_conv_wgrad", dw_desc.ptr, u, v, dwt, sa, sb, S, 1, 0, None, 0, ws, ws.numel(), st))
    tot["fwd"][0] += fl; tot["fwd"][1] += t; tot["wgrad"][0] += fl; tot["wgrad"][1] += tw
    print("%-26s %9d %8.2f | %7.1f %5.1f | %7.1f %5.1f%s" % (name, pix, fl / 1e9, t * 1e6, fl / t / 1e12, tw * 1e6, fl / tw / 1e12, alt))
for k, (f, t) in tot.items():
    print("TOTAL %-6s %.1f GFLOP in %.2f ms = %.1f TFLOP/s (%.1f%% of 157.3)" % (k, f / 1e9, t * 1e3, f / t / 1e12, f / t / 1e12 / 157.3 * 100))
