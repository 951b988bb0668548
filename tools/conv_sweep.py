"""Sweep tile / split-K schedules of hwg_conv_fwd over the layer shapes one training step really launches.

Input: a per-shape dump written by `HWG_CONV_DUMP=file python bench.py ...`. For every forward/data-gradient shape all candidate
schedules are timed through the C-ABI (HWG_CONV_FORCE="bm,bn,bk,nsplit" is the library's tuning hook) and compared with the
library's own choice. Used to derive the cost model in conv_mfma.hip:plan_conv().
"""
import ctypes, os, sys, torch
sys.path.insert(0, '.')
from handwriting_line_generation_amd import _lib as L, ops

dump = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/conv_shapes5.txt"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 14
shapes = []
for line in open(dump):
    if line.startswith("#") or "conv_mfma_kernel" not in line:
        continue
    head, tup = line.split("conv_mfma_kernel")
    ms, n, avg, tf = head.split()
    shapes.append((eval(tup), int(n), float(avg)))

dev = torch.device('cuda:0')
st = torch.cuda.current_stream().cuda_stream


def bench(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


TILES = [(128, 128), (128, 64), (64, 64), (128, 32)]
SPLITS = [1, 2, 3, 4, 6, 8, 12, 16]
tot_def = tot_best = 0.0
import json
ALL = []
print("# launches/step  default_us  best_us  best_cfg  TF_default TF_best  shape")
for (N, H, W, C, K, R, S, stride, pad, dil, tr), n, avg_in_step in shapes:
    if tr:
        P = (H - 1) * stride[0] - 2 * pad[0] + R; Q = (W - 1) * stride[1] - 2 * pad[1] + S
        pix = N * H * W
    else:
        P = (H + 2 * pad[0] - dil[0] * (R - 1) - 1) // stride[0] + 1; Q = (W + 2 * pad[1] - dil[1] * (S - 1) - 1) // stride[1] + 1
        pix = N * P * Q
    fl = 2.0 * pix * K * C * R * S
    x = torch.randn(N, H, W, C, device=dev); wp = torch.randn(R * S, K, C, device=dev) * 0.05
    y = torch.empty(N, P, Q, K, device=dev)
    d = ops._desc(N, H, W, C, K, R, S, stride, pad, dil, P, Q, tr)

    def run():
        need = L.query("hwg_conv_fwd_workspace", d.ptr)
        ws = ops.workspace(need, dev) if need else None
        L.call("hwg_conv_fwd", d.ptr, x, wp, None, y, 0, ws, need, st)

    os.environ["HWG_CONV_FORCE"] = "64,64,16,1"; ops.tuning_reload()
    run(); y_ref = y.clone()
    os.environ.pop("HWG_CONV_FORCE", None); ops.tuning_reload()
    t_def = bench(run)
    worst = float((y - y_ref).abs().max())
    best = (t_def, "default")
    res = {}
    for bm, bn in TILES:
        if bn > 32 and K <= 32: continue
        for ns in SPLITS:
            if ns > 1 and pix * K * ns * 4 > (1 << 30): continue
            os.environ["HWG_CONV_FORCE"] = "%d,%d,%d,%d" % (bm, bn, 32 if C % 32 == 0 else 16, ns); ops.tuning_reload()
            t = bench(run, 6)
            worst = max(worst, float((y - y_ref).abs().max()))
            res[(bm, bn, ns)] = t
            if t < best[0]: best = (t, "%dx%d/%d" % (bm, bn, ns))
    os.environ.pop("HWG_CONV_FORCE", None); ops.tuning_reload()
    tot_def += t_def * n / steps; tot_best += best[0] * n / steps
    top = sorted(res.items(), key=lambda kv: kv[1])[:4]
    ALL.append({"shape": [N, H, W, C, K, R, S, list(stride), list(pad), list(dil), tr], "P": P, "Q": Q, "launches_per_step": n / steps, "default_us": t_def * 1e6,
                "results_us": {"%d,%d,%d" % k: v * 1e6 for k, v in res.items()}})
    print("%6.1f %9.1f %9.1f %-12s %6.1f %6.1f  %s | %s" % (n / steps, t_def * 1e6, best[0] * 1e6, best[1], fl / t_def / 1e12, fl / best[0] / 1e12,
          (N, H, W, C, K, R, S, stride, pad, dil, tr, "maxdiff %.1e" % worst), " ".join("%dx%d/%d:%.0f" % (k[0], k[1], k[2], v * 1e6) for k, v in top)), flush=True)
if len(sys.argv) > 3:
    json.dump(ALL, open(sys.argv[3], "w"))
print("TOTAL per step: default %.3f ms, best-per-shape %.3f ms" % (tot_def * 1e3, tot_best * 1e3))
