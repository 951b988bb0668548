"""GPU: timing of the character-expert kernels (csrc/expert_bank.hip, segment sums of style_ops.hip) at the bench step's load:
~970 recognised windows over ~55 of the 79 experts (bench JSON `style_extractor_load`), the five layer shapes of CharExtractor."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from handwriting_line_generation_amd import _lib as L, ops  # noqa: E402
from handwriting_line_generation_amd.model import expert_bank  # noqa: E402

dev = torch.device("cuda:0")
if os.environ.get("WGRAD_ROWS"):
    expert_bank.WGRAD_TILE_ROWS = int(os.environ["WGRAD_ROWS"])
st = ops._stream()


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


def main():
    rs = np.random.RandomState(3)
    E, n, B = 79, int(os.environ.get("WINDOWS", 970)), 8
    present = rs.permutation(E)[:55]
    p = 1.0 / (1.0 + np.arange(55)) ** 0.8
    cls = np.sort(present[rs.choice(55, size=n, p=p / p.sum())]).astype(np.int64)
    plan = expert_bank.make_plan(cls, dev)
    tab = lambda ts: ops.h2d(np.array([t.data_ptr() for t in ts], dtype=np.int64), dev)  # noqa: E731
    total = 0.0
    for name, R, Cin, Cout, S in (("conv1a", 5, 256, 128, 3), ("conv1b", 5, 128, 256, 3), ("conv2", 5, 256, 256, 1), ("fc0", 1, 256, 256, 1), ("fc1", 1, 256, 128, 1)):
        pad = S // 2
        W = [torch.randn(Cout, Cin, S, device=dev) * 0.05 for _ in range(E)]
        Bs = [torch.randn(Cout, device=dev) * 0.1 for _ in range(E)]
        gW = [torch.zeros_like(w) for w in W]
        gB = [torch.zeros_like(b) for b in Bs]
        wptr, bptr, gwptr, gbptr = tab(W), tab(Bs), tab(gW), tab(gB)
        tseg, trow, nt, _ = expert_bank.plan_tiles(plan, R, dev)
        wseg, wrow, wnt, wrun = expert_bank.plan_tiles(plan, R, dev, expert_bank.WGRAD_TILE_ROWS)
        x = torch.randn(n, R, Cin, device=dev)
        dy = torch.randn(n, R, Cout, device=dev)
        y = torch.empty(n, R, Cout, device=dev)
        dx = torch.empty(n, R, Cin, device=dev)
        ws = torch.empty(max(L.query("hwg_grouped_conv1d_wgrad_workspace", wnt, Cin, Cout, S), 16), dtype=torch.uint8, device=dev)
        tf = bench(lambda: L.call("hwg_grouped_conv1d_fwd", x, plan["seg_start"], plan["seg_eid"], tseg, trow, nt, wptr, bptr, y, R, Cin, Cout, S, pad, st))
        td = bench(lambda: L.call("hwg_grouped_conv1d_dgrad", dy, plan["seg_start"], plan["seg_eid"], tseg, trow, nt, wptr, dx, R, Cin, Cout, S, pad, st))
        tw = bench(lambda: L.call("hwg_grouped_conv1d_wgrad", dy, x, plan["seg_start"], plan["seg_eid"], plan["G"], wseg, wrow, wrun, wnt,
                                  expert_bank.WGRAD_TILE_ROWS, gwptr, gbptr, R, Cin, Cout, S, pad, ws, ws.numel(), st))
        wbytes = 55 * Cout * Cin * S * 4 / 1e6
        print("%-7s R=%d %3d->%3d S=%d  tiles %3d / %3d   fwd %6.1f us  dgrad %6.1f us  wgrad+reduce %6.1f us   (present experts' weights %.1f MB)" % (
            name, R, Cin, Cout, S, nt, wnt, tf, td, tw, wbytes), flush=True)
        total += tf + td + tw
    C = 256
    rows = torch.randn(n, C, device=dev)
    gp = tab([torch.zeros(C, device=dev) for _ in range(E)])
    ta = bench(lambda: L.call("hwg_segment_accumulate_ptr", rows, plan["seg_start"], plan["seg_eid"], plan["G"], gp, C, st))
    v = torch.randn(n, 128, device=dev)
    wgt = torch.rand(n, device=dev)
    seg = torch.from_numpy(rs.randint(0, B, size=n).astype(np.int32)).to(dev)
    out = torch.empty(B, 128, device=dev)
    wsum = torch.empty(B, device=dev)
    tm = bench(lambda: L.call("hwg_segment_weighted_mean", v, wgt, seg, n, 128, B, out, wsum, st))
    print("segment_accumulate_ptr (C=256) %6.1f us   segment_weighted_mean (C=128, B=8) %6.1f us" % (ta, tm))
    print("grouped layers total %.1f us" % total)


if __name__ == "__main__":
    main()
