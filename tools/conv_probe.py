"""Timing probe for single conv launches: each line of PROBE (or argv) is  N,H,W,C,K,R,S,ph,pw[,sh,sw,transposed]  optionally followed by
KEY=VALUE tuning knobs (HWG_CONV_FORCE=64,64,32,1 ...). Prints event-timed microseconds per launch (hwg_conv_fwd only)."""
import os, sys, torch
sys.path.insert(0, '.')
from handwriting_line_generation_amd import _lib as L, ops

dev = torch.device('cuda:0')
st = torch.cuda.current_stream().cuda_stream


def bench(fn, iters=int(os.environ.get('PROBE_ITERS', '60'))):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


lines = [l for l in (sys.argv[1:] or os.environ.get("PROBE", "").split(";")) if l.strip()]


def warm_clocks(ms=400.0):
    """the first ~50 ms of MFMA work after idle run at a lower clock (a probe line is only ~6 ms long): keep the matrix cores busy first"""
    N, H, W, C, K = 8, 32, 256, 256, 256
    x = torch.randn(N, H, W, C, device=dev); y = torch.empty(N, H, W, K, device=dev); wp = torch.randn(9, K, C, device=dev) * 0.05
    d = ops._desc(N, H, W, C, K, 3, 3, (1, 1), (1, 1), (1, 1), H, W, 0)
    need = L.query("hwg_conv_fwd_workspace", d.ptr); ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    t = 0.0
    while t < ms:
        e0.record()
        for _ in range(50): L.call("hwg_conv_fwd", d.ptr, x, wp, None, y, 0, ws, ws.numel(), st)
        e1.record(); torch.cuda.synchronize()
        t += e0.elapsed_time(e1)


if not os.environ.get("PROBE_NO_WARM"):
    warm_clocks()
for line in lines:
    parts = line.split()
    v = [int(t) for t in parts[0].split(",")]
    N, H, W, C, K, R, S, ph, pw = v[:9]
    sh, sw, tr = (v[9:12] + [1, 1, 0][len(v[9:12]):])
    env = dict(kv.split("=", 1) for kv in parts[1:])
    wino = env.pop("WINO", None)
    wgrad = env.pop("WGRAD", None)
    s2 = env.pop("S2", None)
    if tr:
        P = (H - 1) * sh - 2 * ph + R; Q = (W - 1) * sw - 2 * pw + S; pix = N * H * W
    else:
        P = (H + 2 * ph - R) // sh + 1; Q = (W + 2 * pw - S) // sw + 1; pix = N * P * Q
    fl = 2.0 * pix * K * C * R * S
    x = torch.randn(N, H, W, C, device=dev); y = torch.empty(N, P, Q, K, device=dev)
    with ops.tuning(**env):
        d = ops._desc(N, H, W, C, K, R, S, (sh, sw), (ph, pw), (1, 1), P, Q, tr)
        if wgrad:          # weight gradient of the (non-transposed) layer: engine chosen like ops._make_wgrad_plan does
            dy = torch.randn(N, P, Q, K, device=dev); dw = torch.empty(K, C, R, S, device=dev); db = torch.empty(K, device=dev)
            use_wino = ((R == 3 and S == 3) or (R == 4 and S == 4 and (sh, sw) == (2, 2))) and bool(L.query("hwg_wino_wgrad_preferred", d.ptr))
            fn = "hwg_wino_wgrad" if use_wino else "hwg_conv_wgrad"
            need = L.query(fn + "_workspace", d.ptr); ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
            t = bench(lambda: L.call(fn, d.ptr, dy, x, dw, C * R * S, R * S, S, 1, 0, db, 0, ws, ws.numel(), st))
        elif s2:           # F(3x3,2x2) path of the 4x4 stride-2 pad-0 layers (forward, or tr = 1: their data gradient)
            Kc, Cc = (C, K) if tr else (K, C)
            wp = torch.randn(L.query("hwg_wino_s2_weight_floats", Kc, Cc, tr), device=dev) * 0.05
            need = L.query("hwg_wino_s2_workspace", d.ptr); ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
            t = bench(lambda: L.call("hwg_wino_s2_conv", d.ptr, x, wp, None, y, 0, ws, ws.numel(), st))
        elif wino:
            wp = torch.randn((C + 15) // 16, 16, (K + 15) // 16 * 16, 16, device=dev) * 0.05
            need = L.query("hwg_wino_conv_workspace", d.ptr); ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
            t = bench(lambda: L.call("hwg_wino_conv_fwd", d.ptr, x, wp, None, y, 0, ws, ws.numel(), st))
        else:
            wp = torch.randn(R * S, K, C, device=dev) * 0.05
            need = L.query("hwg_conv_fwd_workspace", d.ptr); ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
            t = bench(lambda: L.call("hwg_conv_fwd", d.ptr, x, wp, None, y, 0, ws, ws.numel(), st))
        print("%-44s %-40s %8.1f us %6.1f TF  plan %s" % (parts[0], " ".join(parts[1:]), t * 1e6, fl / t / 1e12, ops.last_plan()), flush=True)
