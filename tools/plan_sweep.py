"""GPU: how good are the planner's choices on the step's own launch mix?

  python tools/plan_sweep.py <conv_shapes.txt from HWG_CONV_DUMP> [top]

For every 3x3 stride-1 forward / data-gradient shape of the dump (most expensive first) the launch is timed as the planner would run it
(engine by hwg_wino_preferred, schedule by the cost models) and with every forced alternative (Winograd kernel variant x channel split,
direct tile x split). Prints chosen vs best and the time the step would save with a perfect planner - the lines with a ratio well above 1
are where the cost models are wrong.
"""
import ast
import os
import re
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from handwriting_line_generation_amd import _lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def read(path, top):
    rows = {}
    for line in open(path):
        m = re.match(r"\s*([\d.]+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+(\S+)\s+(\(.*\))(?:\s+[\d.eE+-]+)?\s*$", line)
        if not m:
            continue
        ms, n, avg, tf, kind, shape = m.groups()
        sh = ast.literal_eval(shape)
        if kind not in ("wino_conv_kernel", "conv_mfma_kernel") or sh[10] != 0:
            continue
        if sh[5:7] != (3, 3) or sh[7] != (1, 1) or sh[9] != (1, 1) or sh[3] % 16 or sh[4] < 16:
            continue
        key = sh[:10]
        r = rows.setdefault(key, [0.0, 0, set()])
        r[0] += float(ms)
        r[1] += int(n)
        r[2].add(sh[11])
    out = sorted(((v[0], v[1], k, "/".join(sorted(v[2]))) for k, v in rows.items()), reverse=True)
    return out[:top]


def main():
    path = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    saved = total = 0.0
    for ms, n, sh, nets in read(path, top):
        N, H, W, C, K, R, S, stride, pad, dil = sh
        P, Q = H + 2 * pad[0] - 2, W + 2 * pad[1] - 2
        x = torch.randn(N, H, W, C, device=dev)
        y = torch.empty(N, P, Q, K, device=dev)
        ww = torch.randn(L.query("hwg_wino_weight_floats", K, C), device=dev) * 0.05
        wd = torch.randn(9, K, C, device=dev) * 0.05
        res = {}

        def run(label, wino, **env):
            with ops.tuning(**env):
                d = ops._desc(N, H, W, C, K, 3, 3, (1, 1), pad, (1, 1), P, Q, 0)
                if wino == "free":
                    wino = bool(L.query("hwg_wino_supported", d.ptr) and L.query("hwg_wino_preferred", d.ptr))
                if wino:
                    need = L.query("hwg_wino_conv_workspace", d.ptr)
                    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
                    t = bench(lambda: L.call("hwg_wino_conv_fwd", d.ptr, x, ww, None, y, 0, ws, ws.numel(), st))
                else:
                    need = L.query("hwg_conv_fwd_workspace", d.ptr)
                    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
                    t = bench(lambda: L.call("hwg_conv_fwd", d.ptr, x, wd, None, y, 0, ws, ws.numel(), st))
                res[label] = (t, ops.last_plan())
            return t

        chosen = run("chosen", "free")
        chosen_plan = res["chosen"][1]
        cfgs = [2] if K <= 16 else ([0, 7] if K <= 48 else ([1, 5, 6, 7] if K <= 64 else [1, 5, 6]))
        for cfg in cfgs:
            if cfg == 5 and C < 64:
                continue
            for ns in (1, 2, 4, 8):
                if ns > 1 and C // 16 // ns < 2:
                    break
                run("wino %d,%d" % (cfg, ns), True, HWG_WINO_FORCE="%d,%d" % (cfg, ns))
        run("direct model", False)
        for tile in ("64,64", "128,64", "64,128", "128,128"):
            for ns in (1, 2, 4):
                run("direct %s,%d" % (tile, ns), False, HWG_CONV_FORCE="%s,%d,%d" % (tile, 32 if C % 32 == 0 else 16, ns))
        del res["chosen"]
        best = min(res, key=lambda k: res[k][0])
        bt = res[best][0]
        total += chosen * n
        saved += max(chosen - bt, 0.0) * n
        flag = "  <<<" if chosen > 1.08 * bt else ""
        print("%-46s x%3d %-14s chosen %7.1f us %-14s best %7.1f us  %-18s ratio %.2f%s" % (
            sh[:5] + (pad,), n, nets, chosen, chosen_plan, bt, best, chosen / bt, flag), flush=True)
    print("weighted: chosen %.3f ms, perfect planner would save %.3f ms (%.1f%%) over the dump's sampled steps" % (
        total * 1e-3, saved * 1e-3, 100.0 * saved / max(total, 1e-9)))


if __name__ == "__main__":
    main()
