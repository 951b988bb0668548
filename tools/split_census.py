"""GPU: which forward / data-gradient convolutions of the step run split over the contraction (and so pay a conv_split_reduce launch)?
  python tools/split_census.py profiles/r04_conv_shapes_b4a2_w512.txt
Every wino_conv / conv_mfma shape of the dump is launched once as the planner would run it; prints launches per step, split factor, network."""
import ast, os, re, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from handwriting_line_generation_amd import _lib as L, ops  # noqa: E402
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
STEPS = 14.0
rows = []
for line in open(sys.argv[1]):
    m = re.match(r"\s*([\d.]+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+(\S+)\s+(\(.*\))\s*$", line)
    if not m:
        continue
    ms, n, avg, tf, kind, shape = m.groups()
    sh = ast.literal_eval(shape)
    if kind not in ("wino_conv_kernel", "conv_mfma_kernel") or sh[3] % 16:
        continue
    N, H, W, C, K, R, S, stride, pad, dil, mode, net = sh
    tr = 1 if mode == 1 else 0
    if tr:
        P = (H - 1) * stride[0] - 2 * pad[0] + dil[0] * (R - 1) + 1; Q = (W - 1) * stride[1] - 2 * pad[1] + dil[1] * (S - 1) + 1
    else:
        P = (H + 2 * pad[0] - dil[0] * (R - 1) - 1) // stride[0] + 1; Q = (W + 2 * pad[1] - dil[1] * (S - 1) - 1) // stride[1] + 1
    d = ops._desc(N, H, W, C, K, R, S, stride, pad, dil, P, Q, tr)
    x = torch.randn(N, H, W, C, device=dev); y = torch.empty(N, P, Q, K, device=dev)
    if kind == "wino_conv_kernel":
        wp = torch.randn(L.query("hwg_wino_weight_floats", K, C), device=dev) * 0.05
        need = L.query("hwg_wino_conv_workspace", d.ptr); ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
        L.call("hwg_wino_conv_fwd", d.ptr, x, wp, None, y, 0, ws, ws.numel(), st)
    else:
        wp = torch.randn(R * S, K, C, device=dev) * 0.05
        need = L.query("hwg_conv_fwd_workspace", d.ptr); ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
        L.call("hwg_conv_fwd", d.ptr, x, wp, None, y, 0, ws, ws.numel(), st)
    eng, cfg, ns = ops.last_plan()
    rows.append((int(n) / STEPS, ns, float(avg), kind, sh))
torch.cuda.synchronize()
tot = sum(r[0] for r in rows if r[1] > 1)
print("# %.1f of %.1f conv launches per step run split over the contraction" % (tot, sum(r[0] for r in rows)))
for r in sorted((r for r in rows if r[1] > 1), key=lambda r: -r[0]):
    print("%6.2f/step  split %2d  %7.1f us  %-17s %s" % r)
