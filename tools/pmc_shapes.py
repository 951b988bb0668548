"""Per-shape HBM traffic of the convolution kernels (rocprofv3 PMC), for the launch mix of a bench run.

  1. python bench.py ... with HWG_CONV_DUMP=<shapes.txt>           (per-shape launch table of the timed region)
  2. rocprofv3 --kernel-trace --pmc FETCH_SIZE -d <dir>/fetch -o out -- python3 tools/pmc_shapes.py run <shapes.txt>
     rocprofv3 --kernel-trace --pmc WRITE_SIZE -d <dir>/write -o out -- python3 tools/pmc_shapes.py run <shapes.txt>
  3. python tools/pmc_shapes.py parse <shapes.txt> <dir> <out.json>
  (python tools/pmc_shapes.py time <shapes.txt>: the same replay timed with HIP events - A/B of a kernel change on the step's launch mix)

`run` replays the top shapes one after the other (REPS launches each, a marker kernel in between) through the same C entry points the
step uses; `parse` cuts the dispatch trace at the markers and sums the counters of everything launched for a shape (main kernel plus its
split / partial reduce pass). FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 reports half the bytes of wide coalesced reads),
both counters are in KiB. Algorithmic bytes = 4 * (input + weights + output) elements - what a single pass over the operands costs.
"""
import ast
import ctypes
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REPS = 3
TOP = 64


def read_shapes(path):
    rows = []
    for line in open(path):
        if line.startswith("#"):
            continue
        m = re.match(r"\s*([\d.]+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+(\S+)\s+(\(.*\))(?:\s+[\d.eE+-]+)?\s*$", line)
        if not m:
            continue
        ms, n, avg, tf, kind, shape = m.groups()
        if "reduce" in kind or "direct" in kind:
            continue
        rows.append((float(ms), int(n), kind, ast.literal_eval(shape)))
    rows.sort(key=lambda r: -r[0])
    return rows[:TOP]


def algorithmic_bytes(kind, sh):
    N, H, W, C, K, R, S, stride, pad, dil = sh[:10]
    if sh[10] == "wgrad":
        P = (H + 2 * pad[0] - dil[0] * (R - 1) - 1) // stride[0] + 1
        Q = (W + 2 * pad[1] - dil[1] * (S - 1) - 1) // stride[1] + 1
        return 4.0 * (N * P * Q * K + N * H * W * C + K * C * R * S)
    if sh[10] == 1:     # fractionally strided: output is larger than the input
        P = (H - 1) * stride[0] - 2 * pad[0] + R
        Q = (W - 1) * stride[1] - 2 * pad[1] + S
    else:
        P = (H + 2 * pad[0] - dil[0] * (R - 1) - 1) // stride[0] + 1
        Q = (W + 2 * pad[1] - dil[1] * (S - 1) - 1) // stride[1] + 1
    return 4.0 * (N * H * W * C + N * P * Q * K + K * C * R * S)


def run(path, timing=False):
    import torch
    from handwriting_line_generation_amd import _lib as L
    from handwriting_line_generation_amd import ops
    dev = torch.device("cuda:0")
    st = ops._stream()
    marker = torch.empty(4, dtype=torch.float32, device=dev)
    g = torch.Generator().manual_seed(0)
    total = 0.0
    only = os.environ.get("PMC_SHAPES_KIND")
    for ms, n, kind, sh in read_shapes(path):
        if only and only not in kind:
            continue
        N, H, W, C, K, R, S, stride, pad, dil = sh[:10]
        wgrad = sh[10] == "wgrad"
        transposed = 0 if wgrad else int(sh[10])
        if transposed:
            P = (H - 1) * stride[0] - 2 * pad[0] + R
            Q = (W - 1) * stride[1] - 2 * pad[1] + S
        else:
            P = (H + 2 * pad[0] - dil[0] * (R - 1) - 1) // stride[0] + 1
            Q = (W + 2 * pad[1] - dil[1] * (S - 1) - 1) // stride[1] + 1
        d = ops._desc(N, H, W, C, K, R, S, stride, pad, dil, P, Q, transposed)
        x = torch.randn(N, H, W, C, generator=g).to(dev)
        if wgrad:
            u = torch.randn(N, P, Q, K, generator=g).to(dev)
            dw = torch.empty(K, C, R, S, dtype=torch.float32, device=dev)
            if L.query("hwg_wino_wgrad_preferred", d.ptr):      # as ops.py chooses
                need = L.query("hwg_wino_wgrad_workspace", d.ptr)
                ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
                call = lambda: L.call("hwg_wino_wgrad", d.ptr, u, x, dw, C * R * S, R * S, S, 1, 0, None, 0, ws, ws.numel(), st)  # noqa: E731
            else:
                need = L.query("hwg_conv_wgrad_workspace", d.ptr)
                ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
                call = lambda: L.call("hwg_conv_wgrad", d.ptr, u, x, dw, C * R * S, R * S, S, 1, 0, None, 0, ws, ws.numel(), st)  # noqa: E731
        elif kind == "wino_conv_kernel" and R == 4 and S == 4:      # F(3x3,2x2) on the space-to-depth image (4x4 stride 2, either direction)
            # (the filter image is sized by the FORWARD layer's (K, C): for the data-gradient form that is this product's (C, K))
            w = torch.randn(L.query("hwg_wino_s2_weight_floats", C if transposed else K, K if transposed else C, transposed), generator=g).to(dev)
            y = torch.empty(N, P, Q, K, dtype=torch.float32, device=dev)
            need = L.query("hwg_wino_s2_workspace", d.ptr)
            ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
            call = lambda: L.call("hwg_wino_s2_conv", d.ptr, x, w, None, y, 0, ws, ws.numel(), st)  # noqa: E731
        elif (kind == "wino_conv_kernel" if not os.environ.get("PMC_FREE_CHOICE") else
              (not transposed and L.query("hwg_wino_supported", d.ptr) and L.query("hwg_wino_preferred", d.ptr))):
            w = torch.randn(L.query("hwg_wino_weight_floats", K, C), generator=g).to(dev)
            y = torch.empty(N, P, Q, K, dtype=torch.float32, device=dev)
            need = L.query("hwg_wino_conv_workspace", d.ptr)
            ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
            call = lambda: L.call("hwg_wino_conv_fwd", d.ptr, x, w, None, y, 0, ws, ws.numel(), st)  # noqa: E731
        else:
            w = torch.randn(R * S, K, C, generator=g).to(dev)
            y = torch.empty(N, P, Q, K, dtype=torch.float32, device=dev)
            need = L.query("hwg_conv_fwd_workspace", d.ptr)
            ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
            call = lambda: L.call("hwg_conv_fwd", d.ptr, x, w, None, y, 0, ws, ws.numel(), st)  # noqa: E731
        if timing:
            for _ in range(3):
                call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                call()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 20
            total += us * n
            print("%8.1f us  x%3d  (bench %8.1f)  %-18s %s" % (us, n, ms * 1e3 / n, kind, sh), flush=True)
            continue
        L.call("hwg_randn", marker, 4, 1, 0, st)          # marker kernel: start of this shape's group
        for _ in range(REPS):
            call()
        torch.cuda.synchronize()
    if timing:
        print("weighted total %.3f ms (bench: %.3f ms)" % (total * 1e-3, sum(r[0] for r in read_shapes(path))))
        return
    L.call("hwg_randn", marker, 4, 1, 0, st)


def _dispatches(db, counter):
    import sqlite3
    cur = sqlite3.connect(db).cursor()
    q = "select dispatch_id, kernel_name, sum(value) from counters_collection where counter_name = ? group by dispatch_id order by dispatch_id"
    return [(name, val) for _, name, val in cur.execute(q, (counter,))]


def parse(path, pmc_dir, out_json):
    shapes = read_shapes(path)
    table = []
    per = {}
    for which in ("fetch", "write"):
        dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(os.path.join(pmc_dir, which)) for f in fs if f.endswith(".db")]
        assert dbs, "no rocprofv3 database under %s/%s" % (pmc_dir, which)
        groups, cur = [], None
        for name, val in _dispatches(dbs[0], "FETCH_SIZE" if which == "fetch" else "WRITE_SIZE"):
            if "randn" in name:
                cur = []
                groups.append(cur)
            elif cur is not None and ("conv" in name or "wgrad" in name or "wino" in name):
                cur.append((name, val))
        groups = groups[:len(shapes)]
        assert len(groups) == len(shapes), "%s: %d marker groups for %d shapes" % (which, len(groups), len(shapes))
        per[which] = groups
    for (ms, n, kind, sh), gf, gw in zip(shapes, per["fetch"], per["write"]):
        fetch = 2.0 * 1024.0 * sum(v for _, v in gf) / REPS        # KiB -> bytes, x2 (gfx950 half-counting of wide reads)
        write = 1024.0 * sum(v for _, v in gw) / REPS
        alg = algorithmic_bytes(kind, sh)
        clean = lambda k: k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0]   # noqa: E731
        table.append({"kind": kind, "shape": repr(sh), "launches_in_bench": n, "ms_in_bench": ms, "kernels": sorted({clean(k) for k, _ in gf}),
                      "hbm_fetch_bytes": round(fetch), "hbm_write_bytes": round(write), "algorithmic_bytes": round(alg),
                      "ratio": round((fetch + write) / alg, 2)})
    with open(out_json, "w") as f:
        json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tools/pmc_shapes.py, %d launches per shape" % REPS,
                   "correction": "FETCH_SIZE x 2 x 1024, WRITE_SIZE x 1024 (MI355X_MICROARCH.md, HBM section)", "shapes": table}, f, indent=1)
    print("wrote", out_json, len(table), "shapes; traffic / algorithmic bytes: median %.2f" % sorted(t["ratio"] for t in table)[len(table) // 2])


if __name__ == "__main__":
    if sys.argv[1] in ("run", "time"):
        run(sys.argv[2], timing=sys.argv[1] == "time")
    else:
        parse(sys.argv[2], sys.argv[3], sys.argv[4])
