"""Sweep tile / pixel-split schedules of hwg_conv_wgrad over the weight-gradient shapes of one training step (see conv_sweep.py)."""
import ctypes, json, os, sys, torch
sys.path.insert(0, '.')
from handwriting_line_generation_amd import _lib as L, ops

dump = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 14
shapes = []
for line in open(dump):
    if line.startswith("#") or "wgrad_mfma_kernel" not in line:
        continue
    head, tup = line.split("wgrad_mfma_kernel")
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


TARGETS = [256, 512, 768, 1024, 1536, 2048, 3072]
ALL = []
tot_def = tot_best = 0.0
for (N, H, W, C, K, R, S, stride, pad, dil, _), n, avg in shapes:
    P = (H + 2 * pad[0] - dil[0] * (R - 1) - 1) // stride[0] + 1; Q = (W + 2 * pad[1] - dil[1] * (S - 1) - 1) // stride[1] + 1
    fl = 2.0 * N * P * Q * K * C * R * S
    u = torch.randn(N, P, Q, K, device=dev); v = torch.randn(N, H, W, C, device=dev)
    dw = torch.empty(K, C, R, S, device=dev)
    d = ops._desc(N, H, W, C, K, R, S, stride, pad, dil, P, Q)

    def run():
        need = L.query("hwg_conv_wgrad_workspace", d.ptr)
        ws = ops.workspace(need, dev)
        L.call("hwg_conv_wgrad", d.ptr, u, v, dw, C * R * S, R * S, S, 1, 0, None, 0, ws, ws.numel(), st)

    os.environ.pop("HWG_WGRAD_FORCE", None); ops.tuning_reload()
    t_def = bench(run)
    ref = dw.clone()
    res = {}
    worst = 0.0
    for cfg in (0, 1):
        if cfg == 0 and not (K >= 128 and C >= 128): continue
        if cfg == 1 and not (K > 32 or C > 32): continue
        for tg in TARGETS:
            os.environ["HWG_WGRAD_FORCE"] = "%d,%d" % (cfg, tg); ops.tuning_reload()
            if L.query("hwg_conv_wgrad_workspace", d.ptr) > (3 << 30): continue
            res["%d,%d" % (cfg, tg)] = bench(run, 6)
            worst = max(worst, float((dw - ref).abs().max() / ref.abs().max()))
    os.environ.pop("HWG_WGRAD_FORCE", None); ops.tuning_reload()
    if not res: res = {"default": t_def}
    bk = min(res, key=res.get)
    tot_def += t_def * n / steps; tot_best += min(res[bk], t_def) * n / steps
    ALL.append({"shape": [N, H, W, C, K, R, S, list(stride), list(pad), list(dil)], "P": P, "Q": Q, "launches_per_step": n / steps, "default_us": t_def * 1e6,
                "results_us": {k: t * 1e6 for k, t in res.items()}})
    print("%6.1f %9.1f %9.1f %-10s %6.1f %6.1f %s rel %.1e | %s" % (n / steps, t_def * 1e6, res[bk] * 1e6, bk, fl / t_def / 1e12, fl / res[bk] / 1e12,
          (N, H, W, C, K, R, S, stride, pad, dil), worst, " ".join("%s:%.0f" % (k, t * 1e6) for k, t in res.items())), flush=True)
if len(sys.argv) > 3:
    json.dump(ALL, open(sys.argv[3], "w"))
print("TOTAL per step: default %.3f ms, best-per-shape %.3f ms" % (tot_def * 1e3, tot_best * 1e3))
