"""GPU tuning aid: Winograd F(2x2,3x3) kernels (csrc/conv_wino.hip) vs a fp64 CPU convolution and vs the direct MFMA engine.
usage: python tools/wino_check.py [quick]"""
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from handwriting_line_generation_amd import _lib as L  # noqa: E402
from handwriting_line_generation_amd import ops  # noqa: E402

dev = torch.device("cuda:0")

# (N, H, W, C, K, pad_h, pad_w, tag)
SHAPES = [
    (2, 9, 13, 16, 16, 1, 1, "tiny odd"),
    (1, 6, 10, 32, 48, 0, 1, "K=48"),
    (3, 7, 7, 64, 80, 2, 2, "pad 2 (dgrad of pad 0)"),
    (16, 58, 512, 64, 64, 0, 1, "D convs1.0"),
    (16, 28, 256, 64, 128, 0, 1, "D convs1.3"),
    (16, 26, 256, 128, 128, 0, 1, "D convs2.0"),
    (16, 12, 128, 128, 128, 0, 1, "D convs3.0"),
    (16, 5, 64, 128, 256, 0, 1, "D convs3.4"),
    (8, 4, 122, 256, 256, 1, 1, "G b0 conv2"),
    (8, 8, 122, 256, 128, 1, 1, "G b1 conv1"),
    (8, 8, 122, 128, 128, 1, 1, "G b1 conv2"),
    (8, 16, 122, 64, 64, 1, 1, "G b2 conv2"),
    (8, 32, 244, 32, 32, 1, 1, "G b3 conv2"),
    (8, 64, 488, 16, 16, 1, 1, "G b4 conv2"),
    (8, 16, 128, 256, 256, 1, 1, "HWR conv3"),
    (8, 8, 129, 512, 512, 0, 0, "HWR conv5"),
    (8, 32, 256, 64, 128, 1, 1, "HWR conv1"),
]


CONFIGS = [None, "0", "1", "3", "4", "3,2", "3,4", "4,2", "4,4"]
if os.environ.get("WINO_CONFIGS"):
    CONFIGS = os.environ["WINO_CONFIGS"].split(";")


def time_it(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    g = torch.Generator().manual_seed(0)
    st = ops._stream()
    worst = 0.0
    only = os.environ.get("WINO_SHAPES")
    for (N, H, W, C, K, ph, pw, tag) in SHAPES:
        if only and not any(o in tag for o in only.split(";")):
            continue
        P, Q = H + 2 * ph - 2, W + 2 * pw - 2
        x = torch.randn(N, H, W, C, generator=g)
        w = torch.randn(K, C, 3, 3, generator=g) / (C * 9) ** 0.5
        b = torch.randn(K, generator=g)
        xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
        d = ops._desc(N, H, W, C, K, 3, 3, (1, 1), (ph, pw), (1, 1), P, Q, 0)
        assert L.query("hwg_wino_supported", d.ptr), tag
        u = torch.empty(L.query("hwg_wino_weight_floats", K, C), dtype=torch.float32, device=dev)
        L.call("hwg_wino_pack_weight", wd, u, K, C, C * 9, 9, 3, 1, 0, st)
        y = torch.empty(N, P, Q, K, dtype=torch.float32, device=dev)
        need = L.query("hwg_wino_conv_workspace", d.ptr)
        ws = ops.workspace(max(need, 16), dev)

        def run_wino():
            L.call("hwg_wino_conv_fwd", d.ptr, xd, u, bd, y, 0, ws, ws.numel(), st)

        def run_direct():
            ops.WINOGRAD = False
            try:
                return ops.conv2d(xd, wd, bd, 1, (ph, pw))
            finally:
                ops.WINOGRAD = True
        run_wino()
        torch.cuda.synchronize()
        big = N * P * Q * K * C > 2e9
        if big and not quick:
            # spot-check against fp64 on a slice of the batch
            ref = F.conv2d(x[:1].permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=(ph, pw)).permute(0, 2, 3, 1)
            got = y[:1].cpu().double()
        elif not big:
            ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=(ph, pw)).permute(0, 2, 3, 1)
            got = y.cpu().double()
        else:
            ref = got = None
        yd = run_direct()
        err = float((got - ref).abs().max() / ref.abs().max()) if ref is not None else float("nan")
        errd = float((yd.cpu().double() - y.cpu().double()).abs().max() / y.abs().max())
        worst = max(worst, errd if err != err else err)
        td = time_it(run_direct)
        fl = 2.0 * N * P * Q * K * C * 9
        res = []
        for force in CONFIGS:
            if force is None:
                os.environ.pop("HWG_WINO_FORCE", None); ops.tuning_reload()
            else:
                os.environ["HWG_WINO_FORCE"] = force; ops.tuning_reload()
            need2 = L.query("hwg_wino_conv_workspace", d.ptr)
            ws = ops.workspace(max(need2, 16), dev)
            y.zero_()
            run_wino()
            e2 = float((yd.double() - y.double()).abs().max() / yd.abs().max())
            worst = max(worst, e2 if not os.environ.get('HWG_WINO_DBG') else 0.0)
            res.append("%s:%6.1fus/%5.1fTF" % (force or "plan", time_it(run_wino), fl / time_it(run_wino) * 1e-6))
        os.environ.pop("HWG_WINO_FORCE", None); ops.tuning_reload()
        print("%-22s N%d %dx%d C%d K%d err64 %.1e |w-d| %.1e direct %6.1fus/%5.1fTF  %s" %
              (tag, N, H, W, C, K, err, errd, td, fl / td * 1e-6, "  ".join(res)), flush=True)
    print("worst relative error %.2e" % worst)
    assert worst < 1e-4 or os.environ.get('HWG_WINO_DBG')


if __name__ == "__main__":
    main()
