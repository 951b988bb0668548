"""Per-kernel numerics: every HIP op (through the C-ABI) against a plain PyTorch fp32 CPU reference of the same op.
Tolerance: 1e-4 of the reference's max magnitude (BASELINE.json north_star: fp32 1e-4); integer outputs exact."""
import os
import zlib

import numpy as np

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _close(got, ref, name, tol=TOL):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, "%s: shape %s vs %s" % (name, tuple(got.shape), tuple(ref.shape))
    scale = max(ref.abs().max().item(), 1e-6)
    err = (got - ref).abs().max().item()
    assert err <= tol * scale + 1e-6, "%s: max err %.3e (scale %.3e)" % (name, err, scale)


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


CONV_CASES = [
    # name, N,H,W,C,K,R,S, stride, pad, dil, transposed
    ("mfma128x128_bk32", 4, 32, 260, 32, 128, 3, 3, (1, 1), (1, 1), (1, 1), False),
    ("mfma128x64_bk16", 4, 32, 260, 16, 64, 3, 3, (1, 1), (0, 1), (1, 1), False),
    ("mfma128x32", 2, 20, 70, 32, 16, 3, 3, (1, 1), (1, 1), (1, 1), False),
    ("mfma64x64_smallM", 2, 4, 50, 128, 128, 3, 3, (1, 1), (1, 1), (1, 1), False),
    ("mfma_k80_edge", 2, 1, 61, 64, 80, 1, 3, (1, 1), (0, 0), (1, 1), False),
    ("mfma_c208", 2, 1, 40, 208, 128, 1, 3, (1, 1), (0, 1), (1, 1), False),
    ("dil4_conv1d", 2, 1, 60, 64, 64, 1, 3, (1, 1), (0, 4), (1, 4), False),
    ("stride2_4x4", 2, 16, 40, 32, 64, 4, 4, (2, 2), (1, 1), (1, 1), False),
    ("stride21_4x4", 2, 8, 30, 32, 32, 4, 4, (2, 1), (0, 1), (1, 1), False),
    ("k63_5x5", 1, 9, 20, 16, 48, 5, 5, (1, 1), (2, 2), (1, 1), False),
    ("c1_7x7", 2, 20, 50, 1, 64, 7, 7, (1, 1), (0, 3), (1, 1), False),
    ("c1_5x5", 2, 16, 40, 1, 32, 5, 5, (1, 1), (2, 2), (1, 1), False),
    ("c1_7x7_many_chunks", 4, 64, 256, 1, 64, 7, 7, (1, 1), (0, 3), (1, 1), False),
    ("c1_4x4_s2_k80", 2, 33, 70, 1, 80, 4, 4, (2, 2), (1, 1), (1, 1), False),
    ("c1_3x3", 2, 16, 41, 1, 64, 3, 3, (1, 1), (1, 1), (1, 1), False),
    ("to1_3x3", 3, 3, 20, 256, 1, 3, 3, (1, 1), (0, 1), (1, 1), False),
    ("to1_1x1", 3, 1, 20, 256, 1, 1, 1, (1, 1), (0, 0), (1, 1), False),
    ("to2_1x1", 3, 1, 20, 32, 2, 1, 1, (1, 1), (0, 0), (1, 1), False),
    ("to1_16", 2, 16, 40, 16, 1, 1, 1, (1, 1), (0, 0), (1, 1), False),
    ("to1_3x3_c64", 2, 20, 50, 64, 1, 3, 3, (1, 1), (1, 1), (1, 1), False),          # the nine-loads-in-flight 3x3 head kernel (decoder image head)
    ("to1_3x3_c32_pad01", 3, 9, 31, 32, 1, 3, 3, (1, 1), (0, 1), (1, 1), False),
    ("to1_3x3_c48", 2, 7, 23, 48, 1, 3, 3, (1, 1), (1, 1), (1, 1), False),           # (12 of the 16 lanes of a pixel hold channels)
    ("convT_lift_4x3", 2, 1, 30, 208, 256, 4, 3, (1, 1), (0, 1), (1, 1), True),
    ("convT_s2_4x4", 2, 8, 20, 64, 32, 4, 4, (2, 2), (1, 1), (1, 1), True),
    ("convT_s1_3x3", 2, 6, 20, 32, 32, 3, 3, (1, 1), (1, 1), (1, 1), True),
    ("convT_6x3", 2, 1, 12, 32, 64, 6, 3, (1, 1), (0, 0), (1, 1), True),
    ("convT_to1", 2, 8, 20, 32, 1, 3, 3, (1, 1), (1, 1), (1, 1), True),
    # one row on the narrow side: the data gradient of a pad-0 layer whose output has ONE row, and the forward of a transposed layer with
    # one input row, both run as their stride-(R, 1) twins (row classes with 1 x S taps each)
    ("onerow_out_3x3_c64", 2, 3, 40, 64, 48, 3, 3, (1, 1), (0, 1), (1, 1), False),
    ("onerow_out_6x3_c32", 3, 6, 25, 32, 64, 6, 3, (1, 1), (0, 0), (1, 1), False),
    ("onerow_out_3x3_c512", 2, 3, 30, 512, 512, 3, 3, (1, 1), (0, 0), (1, 1), False),
    ("onerow_out_4x4_s21", 2, 4, 41, 64, 64, 4, 4, (2, 1), (0, 0), (1, 1), False),
    ("onerow_out_4x4_s21_h5", 2, 5, 41, 64, 64, 4, 4, (2, 1), (0, 0), (1, 1), False),      # a dead fifth input row (the style extractor's last block)
    ("onerow_out_4x4_s22", 2, 4, 40, 32, 48, 4, 4, (2, 2), (0, 1), (1, 1), False),
    # Winograd F(2x2,3x3) path (3x3 / stride 1 / dilation 1, >= 16 output channels): odd sizes, every padding the networks use
    # (0/1 forward, 2 = data gradient of pad 0), channel counts off the tile sizes, the split-channel schedule, 16-wide layers
    ("wino_odd_pad1", 3, 9, 13, 16, 16, 3, 3, (1, 1), (1, 1), (1, 1), False),
    ("wino_pad0_k48", 2, 7, 30, 32, 48, 3, 3, (1, 1), (0, 0), (1, 1), False),
    ("wino_pad01_c48_k80", 2, 6, 21, 48, 80, 3, 3, (1, 1), (0, 1), (1, 1), False),
    ("wino_split_c256", 1, 4, 30, 256, 64, 3, 3, (1, 1), (1, 1), (1, 1), False),
    ("wino_k16_wide", 2, 32, 120, 16, 16, 3, 3, (1, 1), (1, 1), (1, 1), False),
    ("wino_c24_padded", 2, 5, 11, 24, 32, 3, 3, (1, 1), (1, 1), (1, 1), False),
    ("wino_big_tiles", 4, 20, 130, 64, 128, 3, 3, (1, 1), (0, 1), (1, 1), False),
    ("wino_1row", 2, 3, 40, 128, 256, 3, 3, (1, 1), (0, 1), (1, 1), False),
    # narrow layers (K, C <= 32, >= 16 k output pixels): the all-taps 16x16x4 weight-gradient kernel, 1 / 2 / 4 channel blocks, ragged blocks
    ("narrow_16x16_3x3", 2, 64, 160, 16, 16, 3, 3, (1, 1), (1, 1), (1, 1), False),
    ("narrow_16to32_4x4s2", 3, 128, 200, 16, 32, 4, 4, (2, 2), (1, 1), (1, 1), False),
    ("narrow_32x32_3x3_pad0", 2, 42, 258, 32, 32, 3, 3, (1, 1), (0, 0), (1, 1), False),
    ("narrow_c20_k24", 2, 64, 161, 20, 24, 3, 3, (1, 1), (1, 1), (1, 1), False),
    # single-channel first layers with 64 filters and >= 4096 output pixels: weight gradient with the input rows staged in LDS (c1rows_*, the
    # default) - ragged last segment, pads 0 / 1 / 2 / 3 - and on the VALU weight-gradient kernel (wgrad_c1_kernel, HWG_WGRAD_C1=1;
    # off by default: measured slower than the taps-as-N MFMA kernel) - ragged last 64-pixel group, pad 0 / 1 / 3, dilation, few and many ranges
    ("c1rows_3x3", 3, 40, 67, 1, 64, 3, 3, (1, 1), (1, 1), (1, 1), False),
    ("c1rows_5x5_pad0", 2, 36, 131, 1, 64, 5, 5, (1, 1), (0, 0), (1, 1), False),
    ("c1rows_7x7", 2, 70, 150, 1, 64, 7, 7, (1, 1), (0, 3), (1, 1), False),
    ("c1rows_7x7_pad2", 5, 33, 129, 1, 64, 7, 7, (1, 1), (2, 2), (1, 1), False),
    ("c1valu_3x3", 3, 40, 67, 1, 64, 3, 3, (1, 1), (1, 1), (1, 1), False),
    ("c1valu_5x5_pad0", 2, 36, 131, 1, 64, 5, 5, (1, 1), (0, 0), (1, 1), False),
    ("c1valu_7x7_dil2", 2, 70, 150, 1, 64, 7, 7, (1, 1), (3, 6), (1, 2), False),
    ("c1valu_3x3_big", 8, 64, 256, 1, 64, 3, 3, (1, 1), (1, 1), (1, 1), False),
    # RIMES (78 classes): channel counts that are not multiples of 4
    ("rimes_convT_lift_206", 2, 1, 20, 206, 64, 4, 3, (1, 1), (0, 1), (1, 1), True),
    ("rimes_conv1d_to78", 2, 1, 30, 64, 78, 1, 3, (1, 1), (0, 0), (1, 1), False),
    ("rimes_conv1d_334", 2, 1, 30, 334, 32, 1, 5, (1, 1), (0, 2), (1, 1), False),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_fwd_bwd(cuda, case):
    from handwriting_line_generation_amd import ops
    name = case[0]
    if name.startswith("wino_"):
        # the library picks Winograd or the direct kernels per geometry from its cost models; these cases must run the Winograd kernels
        with ops.tuning(HWG_WINO="2"):
            return _conv_case(cuda, ops, case, expect_fwd_engine=6)
    if name.startswith("c1rows_") or name == "c1_7x7_many_chunks":       # the default for >= 4096 output pixels: input rows staged in LDS (wgrad_c1_rows_kernel)
        return _conv_case(cuda, ops, case, expect_wgrad_cfg=15)
    if name.startswith("c1valu_"):
        with ops.tuning(HWG_WGRAD_C1="1"):
            return _conv_case(cuda, ops, case, expect_wgrad_cfg=14)
    if name.startswith("narrow_"):
        with ops.tuning(HWG_WGRAD_NARROW="2"):      # the all-taps narrow-layer weight-gradient kernel also for 2 / 4 channel blocks
            return _conv_case(cuda, ops, case, expect_wgrad_cfg=100 + case[6])
    return _conv_case(cuda, ops, case)


def _conv_case(cuda, ops, case, expect_fwd_engine=None, expect_wgrad_cfg=None):
    name, N, H, W, C, K, R, S, stride, pad, dil, transposed = case
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)   # (hash() changes with PYTHONHASHSEED)
    x = torch.randn(N, C, H, W, generator=g)
    if transposed:
        w = torch.randn(C, K, R, S, generator=g) * (1.0 / (C * R * S) ** 0.5)
    else:
        w = torch.randn(K, C, R, S, generator=g) * (1.0 / (C * R * S) ** 0.5)
    b = torch.randn(K, generator=g)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    if transposed:
        yr = F.conv_transpose2d(xr, wr, br, stride=stride, padding=pad, dilation=dil)
    else:
        yr = F.conv2d(xr, wr, br, stride=stride, padding=pad, dilation=dil)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)

    xg = nhwc(x).to(cuda).requires_grad_(True)
    wg = w.to(cuda).requires_grad_(True)
    bg = b.to(cuda).requires_grad_(True)
    if transposed:
        yg = ops.conv_transpose2d(xg, wg, bg, stride=stride, padding=pad, dilation=dil)
    else:
        yg = ops.conv2d(xg, wg, bg, stride=stride, padding=pad, dilation=dil)
    if expect_fwd_engine is not None:
        assert ops.last_plan()[0] == expect_fwd_engine, "%s ran engine %s" % (name, ops.last_plan())
    _close(nchw(yg), yr, name + ".y")
    yg.backward(nhwc(gy).to(cuda))
    if expect_wgrad_cfg is not None:      # the weight gradient is the last convolution-family launch of the backward pass
        assert ops.last_plan()[:2] == (1, expect_wgrad_cfg), "%s: weight gradient ran %s" % (name, ops.last_plan())
    _close(nchw(xg.grad), xr.grad, name + ".dx")
    _close(wg.grad, wr.grad, name + ".dw")
    _close(bg.grad, br.grad, name + ".db")


# Fractionally strided convolutions whose kernel is a multiple of the stride run with their parity classes merged into the GEMM's N dimension
# (ConvK.mode 2: one stride-1 (R/sh x S/sw)-tap convolution to classes * K channels, written depth-to-space). Every geometry family the
# networks have - 4x4 stride 2 with padding 0 / 1, stride (2, 1), 6x3 stride (3, 1), output_padding, ragged channel counts, one input row -
# merged vs by-class vs torch's CPU transposed convolution, with bias, also with the K loop split.
MERGE_CASES = [(2, 8, 20, 64, 32, 4, 4, (2, 2), (1, 1), (0, 0)), (2, 7, 19, 32, 64, 4, 4, (2, 2), (0, 0), (0, 0)), (3, 5, 33, 48, 80, 4, 4, (2, 1), (0, 0), (0, 0)),
               (2, 5, 21, 32, 24, 4, 4, (2, 2), (1, 1), (1, 1)), (2, 1, 30, 64, 40, 6, 3, (3, 1), (0, 1), (0, 0)), (1, 9, 17, 16, 16, 2, 2, (2, 2), (0, 0), (0, 0)),
               (2, 6, 40, 128, 64, 4, 4, (2, 2), (0, 0), (1, 0)), (2, 4, 12, 32, 48, 4, 6, (2, 3), (1, 2), (0, 0)), (4, 1, 62, 256, 256, 4, 4, (2, 1), (0, 0), (0, 0))]


@pytest.mark.parametrize("force", ["", "128,128,32,2", "64,64,32,3"])
@pytest.mark.parametrize("case", MERGE_CASES, ids=lambda c: "x".join(str(v) for v in c[:7]) + "_s%d%d_p%d%d_o%d%d" % (c[7] + c[8] + c[9]))
def test_transposed_conv_merged_classes(cuda, case, force):
    from handwriting_line_generation_amd import ops
    N, H, W, C, K, R, S, stride, pad, opad = case
    g = torch.Generator().manual_seed(77)
    x = torch.randn(N, C, H, W, generator=g); w = torch.randn(C, K, R, S, generator=g) / (C * R * S / (stride[0] * stride[1])) ** 0.5; b = torch.randn(K, generator=g)
    yr = F.conv_transpose2d(x, w, b, stride=stride, padding=pad, output_padding=opad)
    outs = {}
    for merge in ("2", "0"):
        env = dict(HWG_CONV_MERGE=merge)
        if force:
            env["HWG_CONV_FORCE"] = force
        with ops.tuning(**env):
            y = ops.conv_transpose2d(nhwc(x).to(cuda), w.to(cuda), b.to(cuda), stride=stride, padding=pad, output_padding=opad)
            lp = ops.last_plan()
            assert lp[0] == 0 and (lp[1] >= 1000000) == (merge == "2"), lp
            outs[merge] = y
    _close(nchw(outs["2"]), yr, "merged classes %s %s" % (case, force), tol=2e-5)
    _close(nchw(outs["0"]), yr, "by class %s %s" % (case, force), tol=2e-5)
    with ops.tuning(HWG_CONV_MERGE="2"):          # and as a layer: the gradients of a transposed layer / of the stride-2 convolution whose data gradient it is
        xg, wg, bg = nhwc(x).to(cuda).requires_grad_(True), w.to(cuda).requires_grad_(True), b.to(cuda).requires_grad_(True)
        xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        gy = torch.randn(yr.shape, generator=g)
        F.conv_transpose2d(xr, wr, br, stride=stride, padding=pad, output_padding=opad).backward(gy)
        ops.conv_transpose2d(xg, wg, bg, stride=stride, padding=pad, output_padding=opad).backward(nhwc(gy).to(cuda))
        _close(nchw(xg.grad), xr.grad, "merged layer dx %s" % (case,), tol=2e-5)
        _close(wg.grad, wr.grad, "merged layer dw %s" % (case,), tol=2e-5)
        if opad == (0, 0):
            xc = torch.randn(yr.shape, generator=g).requires_grad_(True)
            wc = w.clone()                           # F.conv2d(x [.,K,..], weight [C, K, R, S]) -> C channels
            yc = F.conv2d(xc, wc, None, stride=stride, padding=pad)
            if yc.shape[2:] == x.shape[2:]:
                gc = torch.randn(yc.shape, generator=g)
                yc.backward(gc)
                xcg = nhwc(xc.detach()).to(cuda).requires_grad_(True)
                ops.conv2d(xcg, wc.to(cuda), None, stride, pad).backward(nhwc(gc).to(cuda))
                _close(nchw(xcg.grad), xc.grad, "stride-s conv dx through merged classes %s" % (case,), tol=2e-5)


# Split-K / channel-split launches without a reduce launch (hwg_split_arrive_wave / _block, hwg_common.h): the wavefront or workgroup that delivers a
# tile's last partial image sums them in split order. Same arithmetic as conv_split_reduce_kernel, so the outputs must be the SAME BITS as with
# HWG_SPLIT_INKERNEL=0 - for every tile config of the direct kernel, the by-class and merged transposed modes, the Winograd kernels that take
# the counters (cfg 0 / 1 / 2 / 7 and the 64 x 64 DMA kernel) and the two-tap kernel; with bias, accumulating into an existing output, and
# repeatedly (the counters must be back at zero after every launch, also after launches of other geometries in between).
SPLIT_CASES = [("direct", (2, 9, 40, 64, 96, 3, 3, (1, 1), (1, 1), (1, 1), 0), dict(HWG_WINO="0", HWG_CONV_FORCE="64,64,32,4")),
               ("direct", (2, 9, 40, 64, 96, 3, 3, (1, 1), (1, 1), (1, 1), 0), dict(HWG_WINO="0", HWG_CONV_FORCE="128,128,32,2")),
               ("direct", (3, 1, 126, 128, 80, 1, 3, (1, 1), (0, 2), (1, 2), 0), dict(HWG_CONV_FORCE="128,64,32,3")),
               ("direct", (3, 1, 126, 128, 80, 1, 3, (1, 1), (0, 4), (1, 4), 0), dict(HWG_CONV_FORCE="128,32,16,8")),
               ("direct", (2, 7, 19, 128, 64, 4, 4, (2, 2), (0, 0), (1, 1), 1), dict(HWG_CONV_MERGE="0", HWG_CONV_FORCE="64,64,32,2")),
               ("direct", (2, 7, 19, 128, 64, 4, 4, (2, 2), (0, 0), (1, 1), 1), dict(HWG_CONV_MERGE="2", HWG_CONV_FORCE="128,128,32,2")),
               ("direct", (2, 1, 30, 128, 40, 6, 3, (3, 1), (0, 1), (1, 1), 1), dict(HWG_CONV_MERGE="0", HWG_CONV_FORCE="64,64,16,3")),
               ("wino", (2, 13, 37, 64, 80, 3, 3, (1, 1), (1, 1), (1, 1), 0), dict(HWG_WINO="2", HWG_WINO_FORCE="0,2")),
               ("wino", (1, 9, 66, 48, 208, 3, 3, (1, 1), (0, 1), (1, 1), 0), dict(HWG_WINO="2", HWG_WINO_FORCE="1,3")),
               ("wino", (3, 7, 21, 128, 32, 3, 3, (1, 1), (2, 2), (1, 1), 0), dict(HWG_WINO="2", HWG_WINO_FORCE="2,4")),
               ("wino", (2, 5, 19, 96, 64, 3, 3, (1, 1), (0, 0), (1, 1), 0), dict(HWG_WINO="2", HWG_WINO_FORCE="7,2")),
               ("wino", (2, 13, 37, 64, 80, 3, 3, (1, 1), (1, 1), (1, 1), 0), dict(HWG_WINO="2", HWG_WINO_FORCE="6,4", HWG_WINO_BAL="-1")),
               ("wino", (2, 8, 129, 512, 256, 3, 3, (1, 1), (1, 1), (1, 1), 0), dict(HWG_WINO="2", HWG_WINO_FORCE="6,2", HWG_WINO_BAL="-1")),
               ("wino", (3, 8, 26, 64, 128, 4, 4, (2, 2), (0, 0), (1, 1), 0), dict(HWG_WINO_S2="2", HWG_WINO_FORCE="6,2")),
               ("wino", (2, 6, 130, 128, 96, 4, 4, (2, 2), (0, 0), (1, 1), 0), dict(HWG_WINO_S2="2", HWG_WINO_FORCE="6,4"))]


@pytest.mark.parametrize("case", SPLIT_CASES, ids=lambda c: "x".join(str(v) for v in c[1][:7]) + "_" + "_".join(v.replace(",", ".") for v in c[2].values()))
def test_split_partials_summed_by_the_last_arrival_are_the_reduce_kernels_bits(cuda, case):
    from handwriting_line_generation_amd import ops
    kind, (N, H, W, C, K, R, S, stride, pad, dil, transposed), env = case
    g = torch.Generator().manual_seed(123)
    x = nhwc(torch.randn(N, C, H, W, generator=g)).to(cuda)
    w = (torch.randn((C, K, R, S) if transposed else (K, C, R, S), generator=g) / (C * R * S) ** 0.5).to(cuda)
    b = torch.randn(K, generator=g).to(cuda)
    other = nhwc(torch.randn(1, 32, 5, 70, generator=g)).to(cuda); wo = torch.randn(48, 32, 3, 3, generator=g).to(cuda)

    def layer(xin, win, bias):
        if transposed:
            return ops.conv_transpose2d(xin, win, bias, stride=stride, padding=pad)
        return ops.conv2d(xin, win, bias, stride=stride, padding=pad, dilation=dil)

    outs = {}
    for mode in ("0", "1"):
        with ops.tuning(HWG_SPLIT_INKERNEL=mode, **env):
            y = layer(x, w, b)
            lp = ops.last_plan()
            assert lp[2] > 1, "%s did not run split: %s" % (case, lp)
            ys = [y, layer(x, w, None)]
            ops.conv2d(other, wo, None, 1, (1, 1))              # another geometry on the same stream and counters
            ys.append(layer(x, w, b))
            # through the layer twice on one input: the data gradient may split too, and the second one ACCUMULATES into the first (`accumulate`)
            xg = x.clone().requires_grad_(True)
            (layer(xg, w, b) + layer(xg, w, None)).backward(torch.ones_like(y))
            ys.append(xg.grad)
            outs[mode] = [t.clone() for t in ys]
    torch.cuda.synchronize()
    for k, (a_, b_) in enumerate(zip(outs["0"], outs["1"])):
        assert torch.isfinite(b_).all()
        assert torch.equal(a_, b_), "%s output %d: %.3e" % (case, k, float((a_ - b_).abs().max()))
    assert torch.equal(outs["1"][0], outs["1"][2])


def test_winograd_agrees_with_direct_engine(cuda):
    """the same 3x3 layer through the F(2x2,3x3) kernels and through the direct implicit-GEMM kernels"""
    from handwriting_line_generation_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 11, 37, 64, generator=g).to(cuda)
    w = (torch.randn(96, 64, 3, 3, generator=g) / 24).to(cuda)
    gy = torch.randn(3, 11, 37, 96, generator=g).to(cuda)
    outs = []
    for mode, engine in (("2", 6), ("0", 0)):        # "2": always Winograd, "0": never (default: the library's cost models choose per geometry)
        with ops.tuning(HWG_WINO=mode, HWG_WINO_WGRAD=mode):
            xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
            y = ops.conv2d(xg, wg, None, 1, 1)
            assert ops.last_plan()[0] == engine
            y.backward(gy)
            outs.append((y.detach(), xg.grad, wg.grad))
    for a, b, n in zip(outs[0], outs[1], ("y", "dx", "dw")):
        _close(a, b, "winograd vs direct " + n, tol=2e-5)


WINO_WGRAD_CASES = [(2, 12, 40, 64, 64, 1, 1), (3, 11, 37, 64, 96, 1, 1), (2, 9, 33, 80, 208, 0, 1), (1, 7, 130, 128, 64, 2, 2), (2, 3, 16, 256, 128, 0, 0),
                    (4, 16, 64, 32, 48, 1, 1), (1, 5, 9, 16, 16, 1, 0)]


@pytest.mark.parametrize("case", WINO_WGRAD_CASES, ids=lambda c: "x".join(map(str, c)))
def test_winograd_weight_gradient(cuda, case):
    """F(3x3, 2x2) weight gradient (conv_wino_wgrad.hip) against autograd in fp64: odd tile counts, zero / wide padding, ragged channel blocks,
    accumulation into an existing gradient"""
    import os
    from handwriting_line_generation_amd import ops
    N, H, W, C, K, ph, pw = case
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, C, H, W, generator=g); w = torch.randn(K, C, 3, 3, generator=g) / (3 * C ** 0.5); b = torch.randn(K, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = F.conv2d(xr, wr, br, 1, (ph, pw))
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy.double())
    with ops.tuning(HWG_WINO_WGRAD="2"):
        xg = x.permute(0, 2, 3, 1).contiguous().to(cuda).requires_grad_(True)
        wg, bg = w.to(cuda).requires_grad_(True), b.to(cuda).requires_grad_(True)
        for rep in range(2):          # the second backward accumulates through autograd
            y = ops.conv2d(xg, wg, bg, 1, (ph, pw))
            y.backward(gy.permute(0, 2, 3, 1).contiguous().to(cuda))
            assert ops.last_plan()[0] == 7, "the Winograd weight-gradient kernel was not the one launched: %s" % (ops.last_plan(),)
    _close(wg.grad, 2 * wr.grad.float(), "wino wgrad dw", tol=2e-5)
    _close(bg.grad, 2 * br.grad.float(), "wino wgrad db", tol=2e-5)
    _close(xg.grad.permute(0, 3, 1, 2), 2 * xr.grad.float(), "wino wgrad dx", tol=2e-5)


REDUCE_CASES = [  # N, H, W, C, K, R, S, stride, pad, wino-wgrad mode; the row kernel takes filters of >= 65536 (k, c) pairs cut into <= 16 ranges
    (2, 12, 40, 256, 256, 3, 3, 1, 1, "2"), (2, 12, 40, 256, 256, 3, 3, 1, 1, "0"), (2, 5, 40, 512, 512, 3, 3, 1, 1, "1"), (1, 9, 33, 272, 304, 3, 3, 1, 0, "2"),
    (2, 20, 66, 256, 256, 4, 4, 2, 0, "0"), (2, 1, 127, 512, 512, 1, 3, 1, (0, 1), "0"), (1, 12, 30, 256, 256, 5, 5, 1, 2, "0"),
    (2, 12, 40, 64, 64, 3, 3, 1, 1, "2"), (3, 7, 19, 20, 36, 3, 3, 1, 1, "0")]


@pytest.mark.parametrize("case", REDUCE_CASES, ids=lambda c: "x".join(str(v) for v in c[:7]))
def test_weight_gradient_partial_image_reduce_variants(cuda, case):
    """the row-contiguous reduce of the split weight-gradient partials (wgrad_reduce_rows_kernel) against the tap-at-a-time one it
    replaces: same sums (the lane-split variants add in another fixed order: 1e-6), first launch and accumulating launch, with the bias sums"""
    from handwriting_line_generation_amd import ops
    N, H, W, C, K, R, S, stride, pad, wmode = case
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, H, W, C, generator=g).to(cuda)
    w = (torch.randn(K, C, R, S, generator=g) / (R * S * C) ** 0.5).to(cuda)
    b = torch.randn(K, generator=g).to(cuda)
    outs = []
    for rows in ("0", "1"):
        with ops.tuning(HWG_WGRAD_REDUCE_ROWS=rows, HWG_WINO_WGRAD=wmode):
            wg, bg = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
            for rep in range(2):
                y = ops.conv2d(x, wg, bg, stride, pad)
                gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(9 + rep)).to(cuda)
                y.backward(gy)
                if rep == 0:
                    first = (wg.grad.clone(), bg.grad.clone())
            outs.append((first[0], first[1], wg.grad.clone(), bg.grad.clone()))
    for a, r_, name in zip(outs[1], outs[0], ("dw", "db", "dw accumulated", "db accumulated")):
        _close(a, r_, "reduce variants " + name, tol=2e-6)


@pytest.mark.parametrize("shape", [(16, 58, 512, 64, 64, 0, 1), (8, 8, 129, 512, 512, 0, 0), (4, 32, 514, 128, 128, 0, 0)], ids=lambda c: "x".join(map(str, c)))
def test_winograd_engines_agree_at_bench_sizes(cuda, shape):
    """full-size layers of the bench step (discriminator 64 ch at 58x512, recogniser 512 ch, style extractor 128 ch): the Winograd forward,
    data-gradient and weight-gradient kernels against the direct implicit-GEMM engine on the same tensors (relative L2; the CPU reference
    would take minutes at these sizes)"""
    import os
    from handwriting_line_generation_amd import ops
    N, H, W, C, K, ph, pw = shape
    g = torch.Generator().manual_seed(17)
    x = torch.randn(N, H, W, C, generator=g).to(cuda)
    w = (torch.randn(K, C, 3, 3, generator=g) / (3 * C ** 0.5)).to(cuda)
    b = torch.randn(K, generator=g).to(cuda)
    outs = []
    for wino in ("2", "0"):
        with ops.tuning(HWG_WINO=wino, HWG_WINO_WGRAD=wino):
            xg, wg, bg = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
            y = ops.conv2d(xg, wg, bg, 1, (ph, pw))
            gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(3)).to(cuda)
            y.backward(gy)
            outs.append((y.detach(), xg.grad, wg.grad, bg.grad))
    for a, r, n in zip(outs[0], outs[1], ("y", "dx", "dw", "db")):
        err = float((a.double() - r.double()).norm() / r.double().norm())
        assert err < 1e-5, "winograd vs direct %s at %s: rel L2 %.2e" % (n, shape, err)


@pytest.mark.parametrize("shape", [(4, 20, 130, 64, 128, 0, 1), (2, 8, 129, 512, 256, 1, 1)], ids=lambda c: "x".join(map(str, c)))
def test_winograd_dma_kernel_repeatable_under_contention(cuda, shape):
    """The 64x64 Winograd kernel streams its filters by LDS DMA with hand-placed wait counts and raw barriers: 60 launches, while a second
    stream keeps other kernels running next to it (different timing every launch), must all produce the bit pattern of the first one
    (a missing wait shows up as run-to-run differences long before it shows up as a wrong mean)"""
    import os
    from handwriting_line_generation_amd import ops
    N, H, W, C, K, ph, pw = shape
    g = torch.Generator().manual_seed(23)
    x = torch.randn(N, H, W, C, generator=g).to(cuda)
    w = (torch.randn(K, C, 3, 3, generator=g) / (3 * C ** 0.5)).to(cuda)
    noise_a = torch.randn(1 << 22, device=cuda)
    side = torch.cuda.Stream()
    with ops.tuning(HWG_WINO="2", HWG_WINO_FORCE="6"):
        first = None
        for rep in range(60):
            with torch.cuda.stream(side):
                for _ in range(1 + rep % 3):
                    noise_a = noise_a * 1.0000001 + 1e-9
            y = ops.conv2d(x, w, None, 1, (ph, pw))
            if first is None:
                assert ops.last_plan()[:2] == (6, 6)
                first = y.clone()
            else:
                assert torch.equal(y, first), "launch %d differs from launch 0 in %d elements" % (rep, int((y != first).sum()))
        torch.cuda.synchronize()


# ---- every Winograd schedule, forced, against the CPU oracle -------------------------------------------------------------------------
# The library picks one of five forward schedules (tile / kernel variant) and a channel-split factor per geometry from a cost model, so
# which kernel a small test shape exercises is an accident of that model. Here each schedule x split factor is FORCED (HWG_WINO_FORCE)
# onto ragged shapes - 80 / 208 output channels (partial 16-blocks), 48 input channels (3 chunks: uneven splits), odd tile counts,
# padding 0 / 1 / 2 - and compared with torch's fp32 CPU convolution; hwg_last_plan() proves the forced schedule is what was launched.
WINO_FORCED_SHAPES = [(2, 13, 37, 64, 80, 1, 1), (1, 9, 66, 48, 208, 0, 1), (3, 7, 21, 128, 32, 2, 2), (2, 5, 19, 96, 64, 0, 0), (1, 6, 10, 16, 16, 1, 0)]


@pytest.mark.parametrize("nsplit", [1, 2, 4])
@pytest.mark.parametrize("cfg", [0, 1, 2, 5, 6, 7])
def test_winograd_forced_schedules_vs_cpu(cuda, cfg, nsplit):
    from handwriting_line_generation_amd import ops
    ran = 0
    for shape in WINO_FORCED_SHAPES:
        N, H, W, C, K, ph, pw = shape
        # the 12-wave producer / consumer kernel is only ever planned for >= 64 contraction channels (forward: C, data gradient: K);
        # 16 output channels only have the 128 x 16 tile
        if (cfg == 5 and min(C, K) < 64) or (cfg != 2 and min(C, K) <= 16):
            continue
        g = torch.Generator().manual_seed(31 + cfg)
        x = torch.randn(N, C, H, W, generator=g); w = torch.randn(K, C, 3, 3, generator=g) / (3 * C ** 0.5); b = torch.randn(K, generator=g)
        xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
        yr = F.conv2d(xr, wr, br, 1, (ph, pw))
        gy = torch.randn(yr.shape, generator=g)
        yr.backward(gy)
        with ops.tuning(HWG_WINO="2", HWG_WINO_FORCE="%d,%d" % (cfg, nsplit), HWG_WINO_WGRAD="0"):
            xg = nhwc(x).to(cuda).requires_grad_(True)
            wg, bg = w.to(cuda).requires_grad_(True), b.to(cuda).requires_grad_(True)
            yg = ops.conv2d(xg, wg, bg, 1, (ph, pw))
            want = (6, cfg, min(nsplit, C // 16))
            assert ops.last_plan() == want, "forward of %s ran %s, forced %s" % (shape, ops.last_plan(), want)
            dx = ops.conv2d(nhwc(gy).to(cuda), wg.detach().flip(2, 3).transpose(0, 1).contiguous(), None, 1, (2 - ph, 2 - pw)) if K % 16 == 0 else None
            if dx is not None:      # the data gradient as its own forced Winograd launch (correlation with the mirrored, transposed filter)
                assert ops.last_plan() == (6, cfg, min(nsplit, K // 16)), ops.last_plan()
                _close(nchw(dx), xr.grad, "forced %d,%d %s dx(direct call)" % (cfg, nsplit, shape), tol=2e-5)
            yg.backward(nhwc(gy).to(cuda))
        name = "forced cfg %d split %d %s" % (cfg, nsplit, shape)
        _close(nchw(yg), yr, name + ".y", tol=2e-5)
        _close(nchw(xg.grad), xr.grad, name + ".dx", tol=2e-5)
        _close(wg.grad, wr.grad, name + ".dw", tol=2e-5)
        _close(bg.grad, br.grad, name + ".db", tol=2e-5)
        ran += 1
    assert ran >= 2


# The balanced schedule of the 64 x 64 DMA kernel (whole-tile rounds + the leftover tiles cut into equal runs of (tile, chunk) units, pieces
# summed by wino_bal_reduce_kernel): forced workgroup counts that cut tiles into 1, 2 and 3+ pieces, runs that span several tiles, a leading
# whole-tile region, ragged channel counts, with bias - against torch's CPU convolution, and bit-repeatable.
WINO_BAL_CASES = [((2, 13, 37, 64, 80, 1, 1), "3"), ((2, 13, 37, 64, 80, 1, 1), "7"), ((2, 13, 37, 64, 80, 1, 1), "40"), ((1, 9, 66, 48, 208, 0, 1), "5"),
                  ((1, 9, 66, 48, 208, 0, 1), "11"), ((2, 5, 19, 96, 64, 0, 0), "4"), ((2, 5, 19, 96, 64, 0, 0), "6"), ((8, 66, 130, 32, 64, 1, 1), "9,256"),
                  ((8, 66, 130, 32, 64, 1, 1), "26,256"), ((8, 66, 130, 32, 64, 1, 1), "64,0"), ((2, 8, 129, 512, 256, 1, 1), "256")]


@pytest.mark.parametrize("case", WINO_BAL_CASES, ids=lambda c: "x".join(map(str, c[0])) + "_G" + c[1].replace(",", "_lead"))
def test_winograd_balanced_schedule_vs_cpu(cuda, case):
    from handwriting_line_generation_amd import ops
    (N, H, W, C, K, ph, pw), force = case
    g = torch.Generator().manual_seed(57)
    x = torch.randn(N, C, H, W, generator=g); w = torch.randn(K, C, 3, 3, generator=g) / (3 * C ** 0.5); b = torch.randn(K, generator=g)
    yr = F.conv2d(x, w, b, 1, (ph, pw))
    gy = torch.randn(yr.shape, generator=g)
    dxr = torch.nn.grad.conv2d_input(x.shape, w, gy, 1, (ph, pw))
    with ops.tuning(HWG_WINO="2", HWG_WINO_BAL=force, HWG_WINO_WGRAD="0"):
        xg, wg, bg = nhwc(x).to(cuda), w.to(cuda), b.to(cuda)
        y = ops.conv2d(xg, wg, bg, 1, (ph, pw))
        assert ops.last_plan() == (6, 6, -int(force.split(",")[0])), ops.last_plan()
        y2 = ops.conv2d(xg, wg, bg, 1, (ph, pw))
        assert torch.equal(y, y2)
        _close(nchw(y), yr, "balanced %s %s y" % (force, case[0]), tol=2e-5)
        if K % 16 == 0:
            dx = ops.conv2d(nhwc(gy).to(cuda), wg.flip(2, 3).transpose(0, 1).contiguous(), None, 1, (2 - ph, 2 - pw))
            lp = ops.last_plan()
            assert lp[0] == 6 and (lp[1] == 6 and lp[2] < 0) == (C > 48), lp      # (the 64 x 64 kernel needs > 48 output channels)
            _close(nchw(dx), dxr, "balanced %s %s dx" % (force, case[0]), tol=2e-5)
    with ops.tuning(HWG_WINO="2", HWG_WINO_FORCE="6,1", HWG_WINO_BAL="-1", HWG_WINO_WGRAD="0"):
        y1 = ops.conv2d(xg, wg, bg, 1, (ph, pw))
        assert ops.last_plan() == (6, 6, 1)
    assert float((y - y1).abs().max()) < 2e-5 * float(yr.abs().max())


# F(3x3,2x2) for the 4x4 stride-2 pad-0 layers (conv_wino.hip, hwg_wino_s2_*): forward through the space-to-depth gather, data gradient through
# the depth-to-space scatter, through the C-ABI directly - ragged tile counts (outputs not multiples of 3), channel counts off the 64-wide
# tiles, bias, uniform channel splits and the balanced schedule - against torch's CPU convolution.
WINO_S2_CASES = [(2, 12, 20, 16, 64, ""), (1, 10, 38, 32, 80, ""), (3, 8, 26, 64, 128, "HWG_WINO_FORCE=6,2"), (2, 14, 44, 48, 64, "HWG_WINO_BAL=7"),
                 (2, 6, 130, 128, 96, "HWG_WINO_FORCE=6,4"), (1, 16, 64, 32, 256, "HWG_WINO_BAL=40,0"), (4, 66, 130, 64, 128, "")]


@pytest.mark.parametrize("case", WINO_S2_CASES, ids=lambda c: "x".join(map(str, c[:5])) + ("_" + c[5].replace("HWG_WINO_", "").replace("=", "").replace(",", "_") if c[5] else ""))
def test_winograd_two_tap_kernel_for_4x4_stride2_layers(cuda, case):
    from handwriting_line_generation_amd import ops, _lib as L
    N, H, W, C, K, env = case
    g = torch.Generator().manual_seed(91)
    x = torch.randn(N, C, H, W, generator=g); w = torch.randn(K, C, 4, 4, generator=g) / (4 * C ** 0.5); b = torch.randn(K, generator=g)
    yr = F.conv2d(x, w, b, stride=2)
    P, Q = yr.shape[2:]
    gy = torch.randn(yr.shape, generator=g)
    dxr = F.conv_transpose2d(gy, w, None, stride=2)
    assert dxr.shape[2:] == (H, W)
    st = torch.cuda.current_stream().cuda_stream
    with ops.tuning(HWG_WINO_S2="2", **dict(kv.split("=") for kv in env.split())):
        wd = w.to(cuda)
        # forward
        d = ops._desc(N, H, W, C, K, 4, 4, (2, 2), (0, 0), (1, 1), P, Q, 0)
        assert L.query("hwg_wino_s2_supported", d.ptr) == 1
        u = torch.empty(L.query("hwg_wino_s2_weight_floats", K, C, 0), device=cuda)
        L.call("hwg_wino_s2_pack_weight", wd, u, K, C, C * 16, 16, 0, st)
        need = L.query("hwg_wino_s2_workspace", d.ptr); ws = torch.empty(max(need, 16), dtype=torch.uint8, device=cuda)
        xg = nhwc(x).to(cuda); y = torch.full((N, P, Q, K), float("nan"), device=cuda)
        L.call("hwg_wino_s2_conv", d.ptr, xg, u, b.to(cuda), y, 0, ws, ws.numel(), st)
        lp = ops.last_plan()
        assert lp[:2] == (6, 36), lp
        _close(nchw(y), yr, "two-tap winograd forward %s" % (case,), tol=2e-5)
        y2 = torch.ones_like(y)
        L.call("hwg_wino_s2_conv", d.ptr, xg, u, None, y2, 1, ws, ws.numel(), st)          # accumulate, no bias
        _close(nchw(y2), yr - b.view(1, -1, 1, 1) + 1.0, "two-tap winograd forward accumulate %s" % (case,), tol=2e-5)
        # data gradient (described as the fractionally strided product: input dy, output dx)
        dd = ops._desc(N, P, Q, K, C, 4, 4, (2, 2), (0, 0), (1, 1), H, W, 1)
        if H == 2 * P + 2 and W == 2 * Q + 2 and K % 16 == 0 and 4 * C > 48:
            assert L.query("hwg_wino_s2_supported", dd.ptr) == 1
            ud = torch.empty(L.query("hwg_wino_s2_weight_floats", K, C, 1), device=cuda)
            L.call("hwg_wino_s2_pack_weight", wd, ud, K, C, C * 16, 16, 1, st)
            need = L.query("hwg_wino_s2_workspace", dd.ptr); ws = torch.empty(max(need, 16), dtype=torch.uint8, device=cuda)
            dx = torch.full((N, H, W, C), float("nan"), device=cuda)
            L.call("hwg_wino_s2_conv", dd.ptr, nhwc(gy).to(cuda), ud, None, dx, 0, ws, ws.numel(), st)
            assert ops.last_plan()[:2] == (6, 36)
            _close(nchw(dx), dxr, "two-tap winograd data gradient %s" % (case,), tol=2e-5)
        else:
            assert L.query("hwg_wino_s2_supported", dd.ptr) == 0
        # and as a layer (ops.conv2d: forward, data gradient, weight / bias gradients; a second pass after an in-place weight update must see it)
        xl, wl, bl = nhwc(x).to(cuda).requires_grad_(True), torch.nn.Parameter(w.to(cuda)), torch.nn.Parameter(b.to(cuda))
        xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        for rep in range(2):
            yl = ops.conv2d(xl, wl, bl, 2, 0)
            assert ops.last_plan()[:2] == (6, 36), ops.last_plan()
            yrr = F.conv2d(xr, wr, br, stride=2)
            _close(nchw(yl), yrr, "two-tap layer y (pass %d) %s" % (rep, case), tol=2e-5)
            xl.grad = wl.grad = bl.grad = xr.grad = wr.grad = br.grad = None
            yl.backward(nhwc(gy).to(cuda)); yrr.backward(gy)
            assert ops.last_plan()[:2] == (7, 36), ops.last_plan()      # the weight gradient: F(2x2 taps, 3x3 gradient tiles) on the Winograd weight-gradient kernel
            _close(nchw(xl.grad), xr.grad, "two-tap layer dx %s" % (case,), tol=2e-5)
            _close(wl.grad, wr.grad, "two-tap layer dw %s" % (case,), tol=2e-5)
            _close(bl.grad, br.grad, "two-tap layer db %s" % (case,), tol=2e-5)
            with torch.no_grad():
                wl.mul_(0.5); wr.mul_(0.5)


@pytest.mark.parametrize("first", ["even", "odd"])
def test_two_tap_layer_choice_is_keyed_by_the_input_width_parity(cuda, first):
    """4x4 stride-2 layers meet inputs of BOTH width parities at the same (N, P, Q) (the style extractor on real lines, width_bucket 0): width
    2Q+2 has the F(3x3,2x2) data-gradient geometry, width 2Q+3 does not (its last column meets no tap). The per-geometry engine choice must
    not leak from one to the other, whichever comes first (the cache key lacked P, Q)."""
    from handwriting_line_generation_amd import ops
    g = torch.Generator().manual_seed(17)
    N, C, K, H = 2, 32, 64, 14
    w = torch.randn(K, C, 4, 4, generator=g) / (4 * C ** 0.5); b = torch.randn(K, generator=g)
    widths = (44, 45) if first == "even" else (45, 44)         # same Q = 21 for both
    with ops.tuning(HWG_WINO_S2="2"):
        wl, bl = torch.nn.Parameter(w.to(cuda)), torch.nn.Parameter(b.to(cuda))
        for W in widths + widths:
            x = torch.randn(N, C, H, W, generator=g)
            xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
            yr = F.conv2d(xr, wr, br, stride=2)
            assert yr.shape[3] == 21
            gy = torch.randn(yr.shape, generator=g)
            xl = nhwc(x).to(cuda).requires_grad_(True)
            wl.grad = bl.grad = None
            yl = ops.conv2d(xl, wl, bl, 2, 0)
            _close(nchw(yl), yr, "width %d forward" % W, tol=2e-5)
            yl.backward(nhwc(gy).to(cuda)); yr.backward(gy)
            _close(nchw(xl.grad), xr.grad, "width %d dx" % W, tol=2e-5)
            _close(wl.grad, wr.grad, "width %d dw" % W, tol=2e-5)
            _close(bl.grad, br.grad, "width %d db" % W, tol=2e-5)


# One full-size layer per network (the bench step's own geometries), every engine, against torch's CPU convolution in fp64: the kernels that
# carry the headline number are compared with the oracle directly, not only with each other.
FULL_SIZE_LAYERS = [("D convs1.0", (16, 58, 512, 64, 64, 0, 1)), ("HWR conv5", (8, 8, 129, 512, 512, 0, 0)), ("style down.2", (4, 32, 514, 128, 128, 0, 0))]


@pytest.mark.parametrize("engine", ["model", "wino6", "wino1", "wino5", "direct"])
@pytest.mark.parametrize("layer", FULL_SIZE_LAYERS, ids=[l[0].replace(" ", "_") for l in FULL_SIZE_LAYERS])
def test_full_size_layers_vs_fp64(cuda, layer, engine):
    from handwriting_line_generation_amd import ops
    N, H, W, C, K, ph, pw = layer[1]
    g = torch.Generator().manual_seed(41)
    x = torch.randn(N, C, H, W, generator=g); w = torch.randn(K, C, 3, 3, generator=g) / (3 * C ** 0.5); b = torch.randn(K, generator=g)
    ref = _full_size_reference(layer[1], x, w, b)
    env = {"model": {}, "direct": dict(HWG_WINO="0", HWG_WINO_WGRAD="0"), "wino6": dict(HWG_WINO="2", HWG_WINO_FORCE="6", HWG_WINO_WGRAD="2"),
           "wino1": dict(HWG_WINO="2", HWG_WINO_FORCE="1", HWG_WINO_WGRAD="2"), "wino5": dict(HWG_WINO="2", HWG_WINO_FORCE="5", HWG_WINO_WGRAD="2")}[engine]
    with ops.tuning(**env):
        xg = nhwc(x).to(cuda).requires_grad_(True)
        wg, bg = w.to(cuda).requires_grad_(True), b.to(cuda).requires_grad_(True)
        y = ops.conv2d(xg, wg, bg, 1, (ph, pw))
        fwd_plan = ops.last_plan()
        y.backward(nhwc(ref["gy"]).to(cuda))
        wgrad_plan = ops.last_plan()
    if engine.startswith("wino"):
        assert fwd_plan[:2] == (6, int(engine[4:])) and wgrad_plan[0] == 7, (fwd_plan, wgrad_plan)
    elif engine == "direct":
        assert fwd_plan[0] == 0 and wgrad_plan[0] == 1, (fwd_plan, wgrad_plan)
    worst = 0.0
    rels = {}
    for got, key in ((nchw(y), "y"), (nchw(xg.grad), "dx"), (wg.grad, "dw"), (bg.grad, "db")):
        want = ref[key]
        rel = float((got.detach().cpu().double() - want).norm() / want.norm())
        mx = float((got.detach().cpu().double() - want).abs().max() / want.abs().max())
        worst = max(worst, mx)
        rels[key] = rel
        assert rel < 3e-6 and mx < 1e-4, "%s [%s, forward %s, weight gradient %s] %s: rel L2 %.2e, max/max %.2e vs fp64" % (
            layer[0], engine, fwd_plan, wgrad_plan, key, rel, mx)
    line = "%-13s [%-6s] forward plan %s, weight-gradient plan %s: rel L2 vs fp64  y %.2e  dx %.2e  dw %.2e  db %.2e; worst max-norm %.2e" % (
        layer[0], engine, fwd_plan, wgrad_plan, rels["y"], rels["dx"], rels["dw"], rels["db"], worst)
    print("\n" + line)
    if os.environ.get("HWG_PARITY_SUMMARY"):     # per-engine forward / gradient error of the full-size layers (Winograd F(2x2,3x3) vs direct vs fp64)
        with open(os.environ["HWG_PARITY_SUMMARY"], "a") as fh:
            fh.write("full-size layer " + line + "\n")


_FULL_REF = {}


@pytest.mark.parametrize("engine", ["model", "wino_s2", "direct"])
def test_full_size_stride2_layer_vs_fp64(cuda, engine):
    """the style extractor's first down-sampling convolution at the bench step's size (4 x 66 x 1026, 64 -> 128, 4x4 stride 2: the layer the
    F(3x3,2x2) kernel was built for), forward / data gradient / weight gradient against torch's CPU convolution in fp64, both engines"""
    from handwriting_line_generation_amd import ops
    N, H, W, C, K = 4, 66, 1026, 64, 128
    g = torch.Generator().manual_seed(47)
    x = torch.randn(N, C, H, W, generator=g); w = torch.randn(K, C, 4, 4, generator=g) / (4 * C ** 0.5); b = torch.randn(K, generator=g)
    key = ("s2", N, H, W, C, K)
    if key not in _FULL_REF:
        xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
        torch.set_num_threads(max(torch.get_num_threads(), min(16, os.cpu_count() or 1)))
        yr = F.conv2d(xr, wr, br, 2)
        gy = torch.randn(yr.shape, generator=torch.Generator().manual_seed(49))
        yr.backward(gy.double())
        _FULL_REF[key] = {"y": yr.detach(), "dx": xr.grad, "dw": wr.grad, "db": br.grad, "gy": gy}
    ref = _FULL_REF[key]
    env = {"model": {}, "wino_s2": dict(HWG_WINO_S2="2"), "direct": dict(HWG_WINO_S2="0")}[engine]
    with ops.tuning(**env):
        xg = nhwc(x).to(cuda).requires_grad_(True)
        wg, bg = torch.nn.Parameter(w.to(cuda)), torch.nn.Parameter(b.to(cuda))
        y = ops.conv2d(xg, wg, bg, 2, 0)
        fwd_plan = ops.last_plan()
        y.backward(nhwc(ref["gy"]).to(cuda))
    if engine == "wino_s2":
        assert fwd_plan[:2] == (6, 36), fwd_plan
    elif engine == "direct":
        assert fwd_plan[0] == 0, fwd_plan
    rels = {}
    for got, k in ((nchw(y), "y"), (nchw(xg.grad), "dx"), (wg.grad, "dw"), (bg.grad, "db")):
        want = ref[k]
        rel = float((got.detach().cpu().double() - want).norm() / want.norm())
        mx = float((got.detach().cpu().double() - want).abs().max() / want.abs().max())
        rels[k] = rel
        assert rel < 3e-6 and mx < 1e-4, "4x4 stride-2 layer [%s, forward %s] %s: rel L2 %.2e, max/max %.2e vs fp64" % (engine, fwd_plan, k, rel, mx)
    line = "style down 4x4s2 [%-7s] forward plan %s: rel L2 vs fp64  y %.2e  dx %.2e  dw %.2e  db %.2e" % (engine, fwd_plan, rels["y"], rels["dx"], rels["dw"], rels["db"])
    print("\n" + line)
    if os.environ.get("HWG_PARITY_SUMMARY"):
        with open(os.environ["HWG_PARITY_SUMMARY"], "a") as fh:
            fh.write("full-size layer " + line + "\n")




def _full_size_reference(shape, x, w, b):
    """fp64 CPU autograd reference of a layer, computed once per layer (a few seconds each) and shared by the engine variants"""
    if shape not in _FULL_REF:
        ph, pw = shape[5], shape[6]
        xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
        torch.set_num_threads(max(torch.get_num_threads(), min(16, os.cpu_count() or 1)))
        yr = F.conv2d(xr, wr, br, 1, (ph, pw))
        gy = torch.randn(yr.shape, generator=torch.Generator().manual_seed(43))
        yr.backward(gy.double())
        _FULL_REF[shape] = {"y": yr.detach(), "dx": xr.grad, "dw": wr.grad, "db": br.grad, "gy": gy}
    return _FULL_REF[shape]


@pytest.mark.parametrize("rows,C", [(1, 1), (300, 1), (262144, 1), (70000, 2), (1000, 3), (5000, 5), (4097, 49), (30000, 64)])
def test_colsum(cuda, rows, C):
    """column sums (bias gradients outside the fused paths): narrow (C <= 3), scalar and float4 kernels, one and many chunks, accumulate"""
    from handwriting_line_generation_amd import ops
    g = torch.Generator().manual_seed(rows + C)
    x = torch.randn(rows, C, generator=g)
    ref = x.double().sum(0)
    xd = x.to(cuda)
    out = ops.colsum(xd)
    tol = 1e-6 * max(1.0, float(x.abs().double().sum(0).max()))
    assert float((out.cpu().double() - ref).abs().max()) <= tol
    base = torch.randn(C, generator=g)
    out2 = ops.colsum(xd, out=base.to(cuda), accumulate=True)
    assert float((out2.cpu().double() - (ref + base.double())).abs().max()) <= tol + 1e-6 * float(base.abs().max())


def test_linear(cuda):
    from handwriting_line_generation_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(5, 128, generator=g); w = torch.randn(256, 128, generator=g) * 0.1; b = torch.randn(256, generator=g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = F.linear(xr, wr, br)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xg, wg, bg = (t.to(cuda).requires_grad_(True) for t in (x, w, b))
    yg = ops.linear(xg, wg, bg)
    _close(yg, yr, "linear.y")
    yg.backward(gy.to(cuda))
    _close(xg.grad, xr.grad, "linear.dx"); _close(wg.grad, wr.grad, "linear.dw"); _close(bg.grad, br.grad, "linear.db")


NORM_CASES = [
    ("gn8_lrelu", 3, 10, 33, 64, "gn", 8, "lrelu", 0.1, False),
    ("gn8_relu_mask", 3, 6, 20, 32, "gn", 8, "relu", 0.0, True),
    ("gn4_c16", 2, 5, 17, 16, "gn", 4, "none", 0.0, False),
    ("gn8_big", 2, 58, 256, 64, "gn", 8, "lrelu", 0.1, False),
    ("gn8_c512_1d", 2, 1, 60, 512, "gn", 8, "relu", 0.0, False),
    ("bn_relu", 4, 8, 30, 256, "bn", 1, "relu", 0.0, False),
    ("bn1d", 4, 1, 61, 512, "bn", 1, "relu", 0.0, False),
    ("in_plain", 2, 7, 19, 32, "in", 1, "none", 0.0, False),
]


@pytest.mark.parametrize("case", NORM_CASES, ids=[c[0] for c in NORM_CASES])
def test_norm_fwd_bwd(cuda, case):
    from handwriting_line_generation_amd import ops
    name, N, H, W, C, kind, groups, act, slope, use_mask = case
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, C, H, W, generator=g) * 2 + 0.5
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g)
    mask = None
    if use_mask:
        mask = (torch.rand(N, C, generator=g) > 0.3).float() / 0.7
    xr = x.clone().requires_grad_(True); gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    rm = torch.zeros(C); rv = torch.ones(C)
    if kind == "gn":
        yr = F.group_norm(xr, groups, gr, br, 1e-5)
    elif kind == "bn":
        yr = F.batch_norm(xr, rm, rv, gr, br, True, 0.1, 1e-5)
    else:
        yr = F.instance_norm(xr, eps=1e-5)
    if mask is not None:
        yr = yr * mask[:, :, None, None]
    if act == "relu":
        yr = F.relu(yr)
    elif act == "lrelu":
        yr = F.leaky_relu(yr, slope)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)

    xg = nhwc(x).to(cuda).requires_grad_(True)
    gg = gamma.to(cuda).requires_grad_(True); bg = beta.to(cuda).requires_grad_(True)
    actc = {"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "lrelu": ops.ACT_LRELU}[act]
    mg = mask.to(cuda) if mask is not None else None
    if kind == "gn":
        yg = ops.group_norm(xg, groups, gg, bg, 1e-5, mask=mg, act=actc, slope=slope)
    elif kind == "bn":
        rmg = torch.zeros(C, device=cuda); rvg = torch.ones(C, device=cuda)
        yg = ops.batch_norm_train(xg, gg, bg, rmg, rvg, 0.1, 1e-5, act=actc, slope=slope)
        _close(rmg, rm, name + ".running_mean"); _close(rvg, rv, name + ".running_var")
    else:
        yg = ops.instance_norm(xg, 1e-5)
    _close(nchw(yg), yr, name + ".y")
    yg.backward(nhwc(gy).to(cuda))
    _close(nchw(xg.grad), xr.grad, name + ".dx", tol=2e-4)
    if kind != "in":
        _close(gg.grad, gr.grad, name + ".dgamma", tol=2e-4); _close(bg.grad, br.grad, name + ".dbeta", tol=2e-4)


@pytest.mark.parametrize("shape", [(2, 4, 61, 256), (2, 16, 40, 64), (3, 64, 100, 16)])
def test_adain_epilogue(cuda, shape):
    from handwriting_line_generation_amd import ops
    N, H, W, C = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, C, H, W, generator=g)
    noise = torch.randn(N, C, H, W, generator=g)
    nw = torch.randn(1, C, 1, 1, generator=g) * 0.5
    gamma = torch.randn(N, C, generator=g) + 1
    beta = torch.randn(N, C, generator=g)
    scale = (2.0 / C) ** 0.5
    xr, nwr, gr, br = (t.clone().requires_grad_(True) for t in (x, nw, gamma, beta))
    t = xr + (nwr * scale) * noise
    u = F.leaky_relu(t, 0.2)
    yr = gr[:, :, None, None] * F.instance_norm(u, eps=1e-5) + br[:, :, None, None]
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xg = nhwc(x).to(cuda).requires_grad_(True)
    nwg = nw.to(cuda).requires_grad_(True)
    gg = gamma.to(cuda).requires_grad_(True); bg = beta.to(cuda).requires_grad_(True)
    yg = ops.adain_epilogue(xg, nhwc(noise).to(cuda), nwg, gg, bg, scale, 0.2, 1e-5)
    _close(nchw(yg), yr, "adain.y")
    yg.backward(nhwc(gy).to(cuda))
    _close(nchw(xg.grad), xr.grad, "adain.dx", tol=2e-4)
    _close(nwg.grad, nwr.grad, "adain.dnoise_w", tol=2e-4)
    _close(gg.grad, gr.grad, "adain.dgamma", tol=2e-4); _close(bg.grad, br.grad, "adain.dbeta", tol=2e-4)


# The per-sample normalisations run their moments pass and their apply pass in ONE launch per direction (norm_*_fused_kernel, norm_act.hip: same two
# bodies, a barrier over the sample's workgroups in between). Same code, same partial sums in the same order: the outputs, the saved statistics
# and every gradient must be the SAME BITS as with HWG_NORM_FUSED=0 - GroupNorm with shared affine (parameter gradients folded into the first
# sample's workgroups: they wait for the whole grid), with mask and gates recomputed in the backward pass, InstanceNorm, the generator epilogue
# (noise from a tensor / drawn in the kernel), 1 to 64 chunks per sample, 1 to 16 samples, repeated launches (the counters must return to zero).
FUSED_NORM_CASES = [("gn", 3, 10, 33, 64, 8, "lrelu", True), ("gn", 3, 6, 20, 32, 8, "relu", True), ("gn", 2, 58, 256, 64, 8, "lrelu", False),
                    ("gn", 2, 1, 60, 512, 8, "relu", False), ("gn", 4, 32, 514, 128, 8, "relu", False), ("gn", 16, 6, 62, 128, 8, "none", False),
                    ("in", 2, 7, 19, 32, 1, "none", False), ("in", 8, 64, 488, 16, 1, "none", False), ("gn", 1, 64, 1024, 16, 4, "lrelu", True),
                    ("adain", 2, 4, 61, 256, 1, "", False), ("adain", 8, 64, 488, 16, 1, "", False), ("adain", 3, 16, 122, 128, 1, "", False)]


@pytest.mark.parametrize("case", FUSED_NORM_CASES, ids=lambda c: "%s_%dx%dx%dx%d_%s" % (c[0], c[1], c[2], c[3], c[4], c[6] or "epi"))
def test_one_launch_normalisation_is_the_two_launch_arithmetic(cuda, case):
    from handwriting_line_generation_amd import ops
    kind, N, H, W, C, groups, act, use_mask = case
    g = torch.Generator().manual_seed(17)
    x = (torch.randn(N, H, W, C, generator=g) * 2 + 0.5).to(cuda)
    gy = torch.randn(N, H, W, C, generator=g).to(cuda)
    gamma = (torch.rand(C, generator=g) + 0.5).to(cuda); beta = torch.randn(C, generator=g).to(cuda)
    mask = ((torch.rand(N, C, generator=g) > 0.3).float() / 0.7).to(cuda) if use_mask else None
    noise = torch.randn(N, H, W, C, generator=g).to(cuda); nw = (torch.randn(1, C, 1, 1, generator=g) * 0.5).to(cuda)
    sg = (torch.randn(N, C, generator=g) + 1).to(cuda); sb = torch.randn(N, C, generator=g).to(cuda)
    actc = {"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "lrelu": ops.ACT_LRELU, "": 0}[act]
    outs = {}
    for mode in ("0", "1"):
        with ops.tuning(HWG_NORM_FUSED=mode):
            res = []
            for rep in range(3):
                xg = x.clone().requires_grad_(True)
                if kind == "gn":
                    gg, bg = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
                    y = ops.group_norm(xg, groups, gg, bg, 1e-5, mask=mask, act=actc, slope=0.1)
                    y.backward(gy)
                    res += [y.detach(), xg.grad, gg.grad, bg.grad]
                elif kind == "in":
                    y = ops.instance_norm(xg, 1e-5)
                    y.backward(gy)
                    res += [y.detach(), xg.grad]
                else:
                    nwg, gg, bg = nw.clone().requires_grad_(True), sg.clone().requires_grad_(True), sb.clone().requires_grad_(True)
                    y = ops.adain_epilogue(xg, noise, nwg, gg, bg, (2.0 / C) ** 0.5, 0.2, 1e-5)
                    y.backward(gy)
                    res += [y.detach(), xg.grad, nwg.grad, gg.grad, bg.grad]
            outs[mode] = [t.clone() for t in res]
    torch.cuda.synchronize()
    for k, (a_, b_) in enumerate(zip(outs["0"], outs["1"])):
        assert torch.isfinite(b_).all(), "%s output %d" % (case, k)
        assert torch.equal(a_, b_), "%s output %d: %.3e" % (case, k, float((a_ - b_).abs().max()))


def test_one_launch_normalisations_on_three_streams_at_once(cuda):
    """the arrival counters are per stream: the same normalisations and split convolutions enqueued on three streams at once (as the trainer's
    concurrent style passes do) give the bits of the one-stream run"""
    from handwriting_line_generation_amd import ops
    g = torch.Generator().manual_seed(23)
    xs = [(torch.randn(4, 32, 200 + 37 * i, 128, generator=g) + 0.3).to(cuda) for i in range(3)]
    gamma = (torch.rand(128, generator=g) + 0.5).to(cuda); beta = torch.randn(128, generator=g).to(cuda)
    w = (torch.randn(128, 128, 1, 3, generator=g) / 20).to(cuda)

    def work(x):
        y = x
        for _ in range(6):
            y = ops.group_norm(y, 8, gamma, beta, 1e-5, act=ops.ACT_RELU)
            y = ops.conv2d(y, w, None, 1, (0, 1))
        return y

    with torch.no_grad(), ops.tuning(HWG_CONV_FORCE="64,64,32,4"):
        want = [work(x) for x in xs]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream() for _ in xs]
        got = [None] * len(xs)
        for rep in range(3):
            for i, (x, st) in enumerate(zip(xs, streams)):
                st.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(st):
                    got[i] = work(x)
            for st in streams:
                torch.cuda.current_stream().wait_stream(st)
            torch.cuda.synchronize()
            for a_, b_ in zip(want, got):
                assert torch.equal(a_, b_)


def test_bias_act_and_tanh(cuda):
    from handwriting_line_generation_amd import ops
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 32, 5, 9, generator=g); b = torch.randn(32, generator=g)
    mask = (torch.rand(2, 32, generator=g) > 0.2).float() / 0.8
    xr = x.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    yr = F.leaky_relu((xr + br[None, :, None, None]) * mask[:, :, None, None], 0.1)
    gy = torch.randn(yr.shape, generator=g); yr.backward(gy)
    xg = nhwc(x).to(cuda).requires_grad_(True); bg = b.to(cuda).requires_grad_(True)
    yg = ops.bias_act(xg, bg, mask.to(cuda), ops.ACT_LRELU, 0.1)
    _close(nchw(yg), yr, "bias_act.y"); yg.backward(nhwc(gy).to(cuda))
    _close(nchw(xg.grad), xr.grad, "bias_act.dx"); _close(bg.grad, br.grad, "bias_act.db")
    # odd channel count takes the scalar path
    x3 = torch.randn(2, 7, 3, generator=g)
    y3 = ops.relu(x3.to(cuda))
    _close(y3, F.relu(x3), "relu.scalar")
    xt = torch.randn(3, 50, generator=g).requires_grad_(True)
    yt = torch.tanh(xt); gt = torch.randn(3, 50, generator=g); yt.backward(gt)
    xtg = xt.detach().to(cuda).requires_grad_(True)
    ytg = ops.tanh(xtg); _close(ytg, yt, "tanh.y"); ytg.backward(gt.to(cuda)); _close(xtg.grad, xt.grad, "tanh.dx")


def test_pixelnorm(cuda):
    from handwriting_line_generation_amd import ops
    g = torch.Generator().manual_seed(7)
    x = torch.randn(6, 128, generator=g).requires_grad_(True)
    y = x / torch.sqrt(torch.mean(x ** 2, dim=1, keepdim=True) + 1e-8)
    gy = torch.randn(6, 128, generator=g); y.backward(gy)
    xg = x.detach().to(cuda).requires_grad_(True)
    yg = ops.pixel_norm(xg); _close(yg, y, "pixelnorm.y"); yg.backward(gy.to(cuda)); _close(xg.grad, x.grad, "pixelnorm.dx")


def test_pools_resample(cuda):
    from handwriting_line_generation_amd import ops
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 32, 11, 23, generator=g)

    def check(fn_ref, fn_gpu, name, xx=x):
        xr = xx.clone().requires_grad_(True)
        yr = fn_ref(xr); gy = torch.randn(yr.shape, generator=g); yr.backward(gy)
        xg = nhwc(xx).to(cuda).requires_grad_(True)
        yg = fn_gpu(xg); _close(nchw(yg), yr, name + ".y"); yg.backward(nhwc(gy).to(cuda)); _close(nchw(xg.grad), xr.grad, name + ".dx")

    check(lambda t: F.avg_pool2d(t, 2), lambda t: ops.avg_pool2d(t, 2), "avgpool2")
    check(lambda t: F.avg_pool2d(t, (1, 2)), lambda t: ops.avg_pool2d(t, (1, 2)), "avgpool1x2")
    check(lambda t: F.max_pool2d(t, 2, 2), lambda t: ops.max_pool2d(t, 2, 2), "maxpool2")
    check(lambda t: F.max_pool2d(t, (2, 2), (2, 1), (0, 1)), lambda t: ops.max_pool2d(t, (2, 2), (2, 1), (0, 1)), "maxpool_s21_p01")
    check(lambda t: F.max_pool2d(t, (1, 2), (1, 2)), lambda t: ops.max_pool2d(t, (1, 2), (1, 2)), "maxpool1d")
    check(lambda t: F.interpolate(t, scale_factor=(2, 1), mode="nearest"), lambda t: ops.upsample_nearest(t, (2, 1)), "upsample21")
    k = torch.tensor([[1., 2, 1], [2, 4, 2], [1, 2, 1]]) / 16
    check(lambda t: F.conv2d(t, k.view(1, 1, 3, 3).repeat(32, 1, 1, 1), padding=1, groups=32), ops.blur3, "blur3")
    check(lambda t: F.pad(t, (2, 3, 1, 1), mode="replicate"), lambda t: ops.pad2d(t, 2, 3, 1, 1, "replicate"), "pad_replicate")
    check(lambda t: F.pad(t, (0, 5), value=-1.0), lambda t: ops.pad2d(t, 0, 5, 0, 0, "constant", -1.0), "pad_const")
    x1 = torch.randn(2, 1, 8, 13, generator=g)
    check(lambda t: F.pad(t, (0, 4, 0, 0), mode="replicate"), lambda t: ops.pad2d(t, 0, 4, 0, 0, "replicate"), "pad_replicate_c1", x1)
    check(lambda t: F.max_pool2d(t, 2, 2), lambda t: ops.max_pool2d(t, 2, 2), "maxpool_c1", x1)


def test_fused_activation_pools_are_bit_identical_to_the_separate_passes(cuda):
    """hwg_act_avgpool_* (discriminator: SN conv -> Dropout2d -> LeakyReLU -> AvgPool2d, model/discriminator_ap.py:84-131) and hwg_maxpool_relu_*
    (recogniser: conv -> ReLU -> MaxPool2d, model/cnn_only_hwr.py:31-43) against the separate kernels they replace - same bits, forward and
    backward - and against torch on the CPU (fp32, tolerance 1e-5)."""
    from handwriting_line_generation_amd import ops
    g = torch.Generator().manual_seed(81)
    for (N, C, H, W, k) in ((3, 32, 11, 23, (2, 2)), (2, 64, 1, 37, (1, 2)), (2, 6, 9, 10, (2, 2))):
        x = torch.randn(N, C, H, W, generator=g)
        mask = (torch.rand(N, C, generator=g) > 0.2).float() / 0.8
        for m in (None, mask):
            xa = nhwc(x).to(cuda).requires_grad_(True); xb = nhwc(x).to(cuda).requires_grad_(True)
            md = None if m is None else m.to(cuda)
            ya = ops.avg_pool2d(ops.bias_act(xa, None, md, ops.ACT_LRELU, 0.1), k)
            yb = ops.act_avg_pool2d(xb, k, md, ops.ACT_LRELU, 0.1)
            assert torch.equal(ya, yb)
            gy = torch.randn(ya.shape, generator=g).to(cuda)
            ya.backward(gy); yb.backward(gy)
            assert torch.equal(xa.grad, xb.grad)
            xr = x.clone().requires_grad_(True)
            yr = F.avg_pool2d(F.leaky_relu(xr * (1.0 if m is None else m[:, :, None, None]), 0.1), k)
            yr.backward(nchw(gy.cpu()))
            _close(nchw(yb), yr, "act_avgpool.y", tol=1e-5); _close(nchw(xb.grad), xr.grad, "act_avgpool.dx", tol=1e-5)
    for (N, C, H, W, args) in ((2, 32, 12, 22, (2, 2)), (2, 64, 8, 21, ((2, 2), (2, 1), (0, 1))), (3, 16, 1, 30, ((1, 2), (1, 2))), (2, 1, 8, 14, (2, 2))):
        x = torch.randn(N, C, H, W, generator=g)
        x[0, :, :2, :4] = -x[0, :, :2, :4].abs()          # whole windows <= 0: nothing may flow back through them
        xa = nhwc(x).to(cuda).requires_grad_(True); xb = nhwc(x).to(cuda).requires_grad_(True)
        ya = ops.bias_act(ops.max_pool2d(xa, *args), None, None, ops.ACT_RELU)
        yb = ops.max_pool2d(xb, *args, relu=True)
        assert torch.equal(ya, yb)
        gy = torch.randn(ya.shape, generator=g).to(cuda)
        ya.backward(gy); yb.backward(gy)
        assert torch.equal(xa.grad, xb.grad)
        xr = x.clone().requires_grad_(True)
        yr = F.max_pool2d(F.relu(xr), *args)               # the reference's order: ReLU first
        yr.backward(nchw(gy.cpu()))
        _close(nchw(yb), yr, "maxpool_relu.y", tol=1e-6); _close(nchw(xb.grad), xr.grad, "maxpool_relu.dx", tol=1e-6)


def test_cat_onehot_layout(cuda):
    from handwriting_line_generation_amd import ops
    g = torch.Generator().manual_seed(9)
    a = torch.randn(2, 1, 7, 80, generator=g); s = torch.randn(2, 128, generator=g)
    ar = a.clone().requires_grad_(True); sr = s.clone().requires_grad_(True)
    yr = torch.cat((ar, sr[:, None, None, :].expand(-1, 1, 7, -1)), dim=3)
    gy = torch.randn(yr.shape, generator=g); yr.backward(gy)
    ag = a.to(cuda).requires_grad_(True); sg = s.to(cuda).requires_grad_(True)
    yg = ops.cat_channels([ag, sg], (2, 1, 7))
    _close(yg, yr, "cat.y"); yg.backward(gy.to(cuda)); _close(ag.grad, ar.grad, "cat.da"); _close(sg.grad, sr.grad, "cat.ds")
    lab = torch.randint(0, 80, (9, 3), generator=g, dtype=torch.int32)
    oh = ops.onehot_rows(lab.to(cuda), 80)
    ref = F.one_hot(lab.long().t(), 80).float().view(3, 1, 9, 80)
    assert torch.equal(oh.cpu(), ref)
    x = torch.randn(2, 5, 4, 6, generator=g).requires_grad_(True)  # NCHW
    xg = x.detach().to(cuda).requires_grad_(True)
    yg = ops.to_nhwc(xg); _close(yg, x.permute(0, 2, 3, 1), "to_nhwc")
    zg = ops.to_nchw(yg); _close(zg, x, "roundtrip")
    zg.sum().backward(); _close(xg.grad, torch.ones_like(x), "layout.grad")


def test_log_softmax_ctc(cuda):
    from handwriting_line_generation_amd import ops
    g = torch.Generator().manual_seed(10)
    B, T, C, Lm = 3, 40, 80, 12
    x = torch.randn(B, 1, T, C, generator=g)
    xr = x.clone().requires_grad_(True)
    lp = F.log_softmax(xr.view(B, T, C), dim=2).permute(1, 0, 2)
    targets = torch.randint(1, C, (B, Lm), generator=g)
    targets[1, 3] = targets[1, 2]  # repeated character
    tl = torch.tensor([12, 7, 1]); il = torch.tensor([T, T, T])
    loss_r = F.ctc_loss(lp, targets, il, tl)
    loss_r.backward()
    xg = x.to(cuda).requires_grad_(True)
    lpg = ops.log_softmax_tbc(xg)
    _close(lpg, lp, "log_softmax")
    loss_g = ops.ctc_loss(lpg, targets, il, tl)
    _close(loss_g, loss_r, "ctc.loss")
    loss_g.backward()
    _close(xg.grad, xr.grad, "ctc.dlogits", tol=2e-4)
    # impossible alignment -> infinite loss is reported as 0 (model/loss.py:28-30)
    tl2 = torch.tensor([12, 12, 12]); il2 = torch.tensor([5, 5, 5])
    bad = ops.ctc_loss(lpg.detach(), targets, il2, tl2)
    assert bad.item() == 0.0


def test_losses_spectral(cuda):
    from handwriting_line_generation_amd import ops
    g = torch.Generator().manual_seed(12)
    a = torch.randn(4, 1, 64, 100, generator=g); b = torch.randn(4, 1, 64, 100, generator=g)
    for nm, fr, fg in (("l1", F.l1_loss, ops.l1_loss), ("mse", F.mse_loss, ops.mse_loss)):
        ar = a.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
        lr_ = fr(ar, br); (lr_ * 0.5).backward()
        ag = a.to(cuda).requires_grad_(True); bg = b.to(cuda).requires_grad_(True)
        lg = fg(ag, bg); _close(lg, lr_, nm); (lg * 0.5).backward()
        _close(ag.grad, ar.grad, nm + ".da"); _close(bg.grad, br.grad, nm + ".db")
    p = torch.randn(8, 64, generator=g)
    for mode, ref in ((ops.LOSS_MEAN, lambda t: t.mean()), (ops.LOSS_HINGE_REAL, lambda t: F.relu(1.0 - t).mean()),
                      (ops.LOSS_HINGE_FAKE, lambda t: F.relu(1.0 + t).mean())):
        pr = p.clone().requires_grad_(True); lr_ = ref(pr); lr_.backward()
        pg = p.to(cuda).requires_grad_(True); lg = ops.mean_loss(pg, mode); _close(lg, lr_, "mean%d" % mode); lg.backward()
        _close(pg.grad, pr.grad, "mean%d.grad" % mode)
    # spectral norm: one power iteration + gradient through sigma
    w = torch.randn(64, 32, 3, 3, generator=g); u = F.normalize(torch.randn(64, generator=g), dim=0); v = F.normalize(torch.randn(288, generator=g), dim=0)
    wr = w.clone().requires_grad_(True)
    wm = wr.view(64, -1)
    v2 = torch.mv(wm.t().detach(), u); v2 = v2 / (v2.norm() + 1e-12)
    u2 = torch.mv(wm.detach(), v2); u2 = u2 / (u2.norm() + 1e-12)
    sigma = u2.dot(wm.mv(v2))
    wsn = wr / sigma
    gw = torch.randn(w.shape, generator=g); wsn.backward(gw)
    wg = w.to(cuda).requires_grad_(True); ug = u.to(cuda); vg = v.to(cuda)
    wsng = ops.spectral_normalize(wg, ug, vg)
    _close(ug, u2, "sn.u"); _close(vg, v2, "sn.v"); _close(wsng, wsn, "sn.w")
    wsng.backward(gw.to(cuda)); _close(wg.grad, wr.grad, "sn.dw", tol=2e-4)


def test_dtw_matches_python_restatement(cuda):
    """bit-exact integer output vs the oracle's restatement of correct_pred"""
    from handwriting_line_generation_amd import ops
    from oracle import seq_oracle
    g = torch.Generator().manual_seed(13)
    for (T, B, Lr) in ((30, 3, 9), (20, 2, 30), (61, 4, 12)):
        pred = F.log_softmax(torch.randn(T, B, 20, generator=g) * 3, dim=2)
        label = torch.randint(1, 20, (Lr, B), generator=g)
        label[Lr - 2:, 0] = 0  # padded tail
        ref = seq_oracle.correct_pred(pred, label)
        got, lens = ops.dtw_align(pred.to(cuda), label.to(cuda))
        assert got.dtype == torch.int64 and torch.equal(got.cpu(), ref), "dtw mismatch T=%d L=%d" % (T, Lr)
        gt_ref, pos_ref = seq_oracle.gt_counts(ref, label)
        gt, meta = ops.gt_counts(got, label.to(cuda))
        assert torch.equal(gt.cpu(), gt_ref) and int(meta[0].item()) == pos_ref and int(meta[1].item()) == 0


def test_dtw_reference_known_answers_on_hip(cuda):
    """the reference's own correct_pred outputs (tests/golden/seq_kat.npz, recorded by tools/gen_golden.py from model/hw_with_style.py:18-74:
    exact ties resolved by first minimum, T=122 / L=30 full size, paths longer than T) straight through hwg_dtw_align: bit exact"""
    import os
    import numpy as np
    from handwriting_line_generation_amd import ops
    kat = np.load(os.path.join(os.path.dirname(__file__), "golden", "seq_kat.npz"))
    for n in range(5):
        pred = torch.from_numpy(kat["dtw%d_pred" % n]); label = torch.from_numpy(kat["dtw%d_label" % n])
        got, lens = ops.dtw_align(pred.to(cuda).contiguous(), label.to(cuda))
        ref = torch.from_numpy(kat["dtw%d_out" % n])
        assert got.dtype == torch.int64 and torch.equal(got.cpu(), ref), "reference KAT dtw%d differs on the HIP kernel" % n


def test_dtw_edge_shapes_exact(cuda):
    """degenerate geometries of the alignment, recorded from the reference's correct_pred (tests/golden/seq_kat_edges.npz): one prediction
    step, fewer steps than the blank-interleaved label (the path runs along the label axis and is longer than T), a single character,
    all-equal costs (every minimum is a tie), one-hot predictions, a zero-padded label tail"""
    import os
    import numpy as np
    from handwriting_line_generation_amd import ops
    kat = np.load(os.path.join(os.path.dirname(__file__), "golden", "seq_kat_edges.npz"))
    n = 0
    while "dtw%d_pred" % n in kat:
        pred = torch.from_numpy(kat["dtw%d_pred" % n]); label = torch.from_numpy(kat["dtw%d_label" % n]); ref = torch.from_numpy(kat["dtw%d_out" % n])
        got, lens = ops.dtw_align(pred.to(cuda).contiguous(), label.to(cuda))
        assert torch.equal(got.cpu(), ref), "dtw edge case %d (T=%d, L=%d)" % (n, pred.shape[0], label.shape[0])
        assert int(lens.max()) == ref.shape[0]
        n += 1
    assert n >= 8


def test_style_helpers_rng(cuda):
    from handwriting_line_generation_amd import ops
    g = torch.Generator().manual_seed(14)
    x = torch.randn(2, 30, 16, generator=g)
    ib = torch.tensor([0, 0, 1, 1], dtype=torch.int32); ip = torch.tensor([0, 7, 29, 8], dtype=torch.int32)
    xr = x.clone().requires_grad_(True)
    pads = F.pad(xr, (0, 0, 2, 2))
    pr = torch.stack([pads[ib[i], ip[i]:ip[i] + 5] for i in range(4)])
    gy = torch.randn(pr.shape, generator=g); pr.backward(gy)
    xg = x.to(cuda).requires_grad_(True)
    pg = ops.gather_windows(xg, ib.to(cuda), ip.to(cuda), 2)
    _close(pg.view(4, 5, 16), pr, "windows"); pg.backward(gy.view(4, 1, 5, 16).to(cuda)); _close(xg.grad, xr.grad, "windows.dx")
    v = torch.randn(4, 16, generator=g); wgt = torch.rand(4, generator=g); seg = torch.tensor([0, 0, 2, 0], dtype=torch.int32)
    vr = v.clone().requires_grad_(True)
    tot = torch.zeros(3, 16); ws = torch.zeros(3)
    for i in range(4):
        tot[seg[i]] = tot[seg[i]] + wgt[i] * vr[i]; ws[seg[i]] += wgt[i]
    ref = torch.where(ws[:, None] != 0, tot / ws[:, None], tot)
    gy = torch.randn(3, 16, generator=g); ref.backward(gy)
    vg = v.to(cuda).requires_grad_(True)
    og = ops.segment_weighted_mean(vg, wgt.to(cuda), seg.to(cuda), 3)
    _close(og, ref, "segmean"); og.backward(gy.to(cuda)); _close(vg.grad, vr.grad, "segmean.dv")
    rng = ops.DeviceRNG(1)
    z = rng.randn((1000, 1000), cuda)
    assert abs(z.mean().item()) < 5e-3 and abs(z.std().item() - 1) < 5e-3
    z2 = rng.randn((1000, 1000), cuda)
    assert not torch.equal(z, z2)
    m = rng.dropmask((1000, 1000), 0.1, cuda)
    assert abs((m == 0).float().mean().item() - 0.1) < 5e-3 and abs(m.max().item() - 1 / 0.9) < 1e-6
    am = ops.argmax_rows(x.view(60, 16).to(cuda))
    assert torch.equal(am.cpu().long(), x.view(60, 16).argmax(1))


@pytest.mark.parametrize("n,B,C", [(970, 8, 128), (3, 8, 16), (9000, 4, 64)])
def test_segment_weighted_mean_at_scale(cuda, n, B, C):
    """per-line score-weighted mean of the window styles: the list kernel (members of a line compacted in LDS, in order) at the bench step's
    load, a tiny call with empty lines, and a call too long for LDS (the walk-all-items kernel); forward and backward against the
    reference's accumulation loop in fp64"""
    from handwriting_line_generation_amd import ops
    g = torch.Generator().manual_seed(n + C)
    v = torch.randn(n, C, generator=g)
    wgt = torch.rand(n, generator=g)
    seg = torch.randint(0, B, (n,), generator=g, dtype=torch.int32)
    if n == 3:
        seg = torch.tensor([1, 1, 5], dtype=torch.int32)
    vr = v.double().requires_grad_(True)
    tot = torch.zeros(B, C, dtype=torch.float64)
    tot = tot.index_add(0, seg.long(), wgt.double()[:, None] * vr)
    ws = torch.zeros(B, dtype=torch.float64).index_add(0, seg.long(), wgt.double())
    ref = torch.where(ws[:, None] != 0, tot / ws[:, None].clamp_min(1e-300), tot)
    gy = torch.randn(B, C, generator=g)
    ref.backward(gy.double())
    vg = v.to(cuda).requires_grad_(True)
    out = ops.segment_weighted_mean(vg, wgt.to(cuda), seg.to(cuda), B)
    _close(out, ref.float(), "segment mean", tol=5e-6)
    out.backward(gy.to(cuda))
    _close(vg.grad, vr.grad.float(), "segment mean dv", tol=5e-6)


@pytest.mark.parametrize("R,Cin,Cout,S", [(5, 256, 128, 3), (5, 128, 256, 3), (5, 256, 256, 1), (1, 256, 128, 1), (3, 64, 96, 3), (5, 32, 16, 3), (1, 16, 48, 1)])
def test_grouped_expert_layers(cuda, R, Cin, Cout, S):
    """grouped per-expert Conv1d (hwg_grouped_conv1d_*) against torch conv1d run expert by expert; one long run spans several row tiles"""
    import numpy as np
    from handwriting_line_generation_amd import _lib as L, ops
    from handwriting_line_generation_amd.model import expert_bank
    pad = S // 2
    g = torch.Generator().manual_seed(5)
    E = 6
    Ws = [torch.randn(Cout, Cin, S, generator=g) * 0.05 for _ in range(E)]
    Bs = [torch.randn(Cout, generator=g) * 0.1 for _ in range(E)]
    cls = np.array([0] * 3 + [2] * 41 + [3] * 1 + [5] * 14, dtype=np.int64)   # expert 1 and 4 absent; run of 41 windows -> 4 tiles at R=5
    n = cls.size
    x = torch.randn(n, R, Cin, generator=g)
    dy = torch.randn(n, R, Cout, generator=g)
    # reference
    xr = x.clone().requires_grad_(True)
    Wr = [w.clone().requires_grad_(True) for w in Ws]
    Br = [b.clone().requires_grad_(True) for b in Bs]
    yr = torch.stack([F.conv1d(xr[i].t().unsqueeze(0), Wr[cls[i]], Br[cls[i]], padding=pad)[0].t() for i in range(n)])
    (yr * dy).sum().backward()
    # device
    dev = cuda
    Wd = [w.to(dev) for w in Ws]; Bd = [b.to(dev) for b in Bs]
    gW = [torch.full_like(w, 0.5) for w in Wd]; gB = [torch.full_like(b, 0.25) for b in Bd]   # kernels must ADD to these
    tab = lambda ts: ops.h2d(np.array([t.data_ptr() for t in ts], dtype=np.int64), dev)
    wptr, bptr, gwptr, gbptr = tab(Wd), tab(Bd), tab(gW), tab(gB)
    plan = expert_bank.make_plan(cls, dev)
    tseg, trow, nt, _ = expert_bank.plan_tiles(plan, R, dev)
    wrows = expert_bank.WGRAD_TILE_ROWS if R == 5 else 16   # small tiles too: several partial images per run
    wseg, wrow, wnt, wrun = expert_bank.plan_tiles(plan, R, dev, wrows)
    xd = x.to(dev); dyd = dy.to(dev)
    y = torch.empty(n, R, Cout, device=dev); dx = torch.empty(n, R, Cin, device=dev)
    st = ops._stream()
    L.call("hwg_grouped_conv1d_fwd", xd, plan["seg_start"], plan["seg_eid"], tseg, trow, nt, wptr, bptr, y, R, Cin, Cout, S, pad, st)
    L.call("hwg_grouped_conv1d_dgrad", dyd, plan["seg_start"], plan["seg_eid"], tseg, trow, nt, wptr, dx, R, Cin, Cout, S, pad, st)
    ws = ops.workspace(L.query("hwg_grouped_conv1d_wgrad_workspace", wnt, Cin, Cout, S), dev)
    L.call("hwg_grouped_conv1d_wgrad", dyd, xd, plan["seg_start"], plan["seg_eid"], plan["G"], wseg, wrow, wrun, wnt, wrows, gwptr, gbptr, R, Cin, Cout, S, pad,
           ws, ws.numel(), st)
    _close(y, yr, "grouped fwd")
    _close(dx, xr.grad, "grouped dgrad")
    for e in range(E):
        if Wr[e].grad is None:
            assert torch.all(gW[e] == 0.5) and torch.all(gB[e] == 0.25), "absent expert %d was touched" % e
        else:
            _close(gW[e] - 0.5, Wr[e].grad, "grouped wgrad e%d" % e)
            _close(gB[e] - 0.25, Br[e].grad, "grouped bias grad e%d" % e)


def test_style_path_parameters_get_gradients_from_a_constant_style(cuda):
    """text-only "gen" lessons feed the generator a sampled style that does not require grad (trainer :984): the style MLP and the AdaIN
    affines must still receive their parameter gradients, exactly as torch's Linear modules do"""
    from handwriting_line_generation_amd import ops
    g = torch.Generator().manual_seed(12)
    B, I = 4, 128
    ref_lin = [torch.nn.Linear(I, o) for o in (64, 32)]
    dev_lin = [torch.nn.Linear(I, o).to(cuda) for o in (64, 32)]
    ref_chain = [torch.nn.Linear(I, I) for _ in range(3)]
    dev_chain = [torch.nn.Linear(I, I).to(cuda) for _ in range(3)]
    for a, b in zip(dev_lin + dev_chain, ref_lin + ref_chain):
        a.load_state_dict(b.state_dict())
    x = torch.randn(B, I, generator=g)                      # no requires_grad anywhere on the input side
    h = x
    for m in ref_chain:
        h = F.leaky_relu(m(h), 0.2)
    sum(m(h).pow(2).sum() for m in ref_lin).backward()
    hd = ops.MLPChain(dev_chain, 0.2)(x.to(cuda))
    assert hd.requires_grad, "the chain's output must require grad through its parameters"
    sum(torch.cat(pr, 1).pow(2).sum() for pr in ops.LinearBank(dev_lin, halves=2)(hd)).backward()
    for k, (a, b) in enumerate(zip(dev_lin + dev_chain, ref_lin + ref_chain)):
        assert a.weight.grad is not None and a.bias.grad is not None, "layer %d got no gradient" % k
        _close(a.weight.grad, b.weight.grad, "dW %d" % k)
        _close(a.bias.grad, b.bias.grad, "db %d" % k)


def test_linear_bank_and_mlp_chain(cuda):
    """generator style path: the AdaIN affine bank (ops.LinearBank) and the style-embedding chain (ops.MLPChain) vs torch Linear modules"""
    from handwriting_line_generation_amd import ops
    g = torch.Generator().manual_seed(11)
    B, I = 8, 128
    outs = [512, 512, 256, 64, 32]
    ref_lin = [torch.nn.Linear(I, o) for o in outs]
    for m in ref_lin:
        m.weight.data = torch.randn(m.weight.shape, generator=g) * 0.1
        m.bias.data = torch.randn(m.bias.shape, generator=g)
    dev_lin = [torch.nn.Linear(I, o).to(cuda) for o in outs]
    for a, b in zip(dev_lin, ref_lin):
        a.load_state_dict(b.state_dict())
    x = torch.randn(B, I, generator=g)
    xr = x.clone().requires_grad_(True)
    xd = x.to(cuda).requires_grad_(True)
    ws = [torch.randn(B, o // 2, generator=g) for o in outs for _ in range(2)]
    # reference: gamma = first half of the columns, beta = second half; one output (beta of layer 3) deliberately unused
    ref_parts = []
    for m in ref_lin:
        y = m(xr)
        ref_parts += [y[:, : y.shape[1] // 2], y[:, y.shape[1] // 2:]]
    skip = 7
    sum(((p * w).sum() for k, (p, w) in enumerate(zip(ref_parts, ws)) if k != skip)).backward()
    bank = ops.LinearBank(dev_lin, halves=2)
    pairs = bank(xd)
    dev_parts = [t for pr in pairs for t in pr]
    for k, (a, b) in enumerate(zip(dev_parts, ref_parts)):
        _close(a, b, "bank output %d" % k)
    sum(((p * w.to(cuda)).sum() for k, (p, w) in enumerate(zip(dev_parts, ws)) if k != skip)).backward()
    _close(xd.grad, xr.grad, "bank dx")
    for k, (a, b) in enumerate(zip(dev_lin, ref_lin)):
        _close(a.weight.grad, b.weight.grad, "bank dW %d" % k)
        _close(a.bias.grad, b.bias.grad, "bank db %d" % k)

    D, Lc = 128, 6
    ref_chain = [torch.nn.Linear(D, D) for _ in range(Lc)]
    for m in ref_chain:
        m.weight.data = torch.randn(D, D, generator=g) * 0.15
        m.bias.data = torch.randn(D, generator=g) * 0.3
    dev_chain = [torch.nn.Linear(D, D).to(cuda) for _ in range(Lc)]
    for a, b in zip(dev_chain, ref_chain):
        a.load_state_dict(b.state_dict())
    z = torch.randn(B, D, generator=g)
    zr = z.clone().requires_grad_(True); zd = z.to(cuda).requires_grad_(True)
    h = zr
    for m in ref_chain:
        h = F.leaky_relu(m(h), 0.2)
    wout = torch.randn(B, D, generator=g)
    (h * wout).sum().backward()
    chain = ops.MLPChain(dev_chain, 0.2)
    hd = chain(zd)
    _close(hd, h, "chain output")
    (hd * wout.to(cuda)).sum().backward()
    _close(zd.grad, zr.grad, "chain dx")
    for k, (a, b) in enumerate(zip(dev_chain, ref_chain)):
        _close(a.weight.grad, b.weight.grad, "chain dW %d" % k)
        _close(a.bias.grad, b.bias.grad, "chain db %d" % k)
    # the backward pass as two launches (hwg_mlp_chain_bwd_split, the default) against the single-workgroup kernel: same bits, twice in a row
    # (the parameter gradients are ADDED into their buffers)
    assert ops.MLP_CHAIN_SPLIT
    got = {}
    for split in (True, False):
        ops.MLP_CHAIN_SPLIT = split
        try:
            for m in dev_chain:
                m.weight.grad = m.bias.grad = None
            zz = z.to(cuda).requires_grad_(True)
            for _ in range(2):
                (chain(zz) * wout.to(cuda)).sum().backward()
            got[split] = [zz.grad.clone()] + [m.weight.grad.clone() for m in dev_chain] + [m.bias.grad.clone() for m in dev_chain]
        finally:
            ops.MLP_CHAIN_SPLIT = True
    names_ = ["dx"] + ["dW%d" % k for k in range(Lc)] + ["db%d" % k for k in range(Lc)]
    diffs = {n: float((a - b).abs().max()) for n, a, b in zip(names_, got[True], got[False]) if not torch.equal(a, b)}
    assert not diffs, "two-launch backward differs from the single-workgroup kernel: %s" % diffs


def _random_conv_cases(n=36, seed=2026):
    import random
    rnd = random.Random(seed)
    cases = []
    while len(cases) < n:
        transposed = rnd.random() < 0.25
        C = rnd.choice([1, 3, 16, 16, 32, 48, 64, 80, 128, 208, 256])
        K = rnd.choice([1, 2, 16, 32, 64, 78, 80, 128, 256])
        R = rnd.choice([1, 1, 3, 3, 4, 5, 7]); S = rnd.choice([1, 3, 3, 4, 5, 7])
        if transposed:
            st = rnd.choice([(1, 1), (2, 2), (2, 1)]); dil = (1, 1)
            R = max(R, st[0]); S = max(S, st[1])
            pad = (rnd.randint(0, min(1, R - 1)), rnd.randint(0, min(1, S - 1)))
        else:
            st = rnd.choice([(1, 1), (1, 1), (2, 2), (2, 1)])
            dil = rnd.choice([(1, 1), (1, 1), (1, 2), (2, 1)]) if st == (1, 1) else (1, 1)
            pad = (rnd.randint(0, R // 2 + 1), rnd.randint(0, S // 2 + 1))
        N = rnd.randint(1, 5); H = rnd.randint(1, 19); W = rnd.randint(1, 75)
        if not transposed:
            if H + 2 * pad[0] < dil[0] * (R - 1) + 1 or W + 2 * pad[1] < dil[1] * (S - 1) + 1:
                continue
        if transposed and ((H - 1) * st[0] - 2 * pad[0] + R < 1 or (W - 1) * st[1] - 2 * pad[1] + S < 1):
            continue      # empty output
        if (C == 1 and K <= 2) or N * H * W * max(C, K) > 3_000_000:
            continue
        cases.append(("rnd%d_C%dK%d_%dx%d_s%s_p%s_d%s%s" % (len(cases), C, K, R, S, st, pad, dil, "_T" if transposed else ""), N, H, W, C, K, R, S, st, pad, dil,
                      transposed))
    return cases


import os as _os
RANDOM_CONV_CASES = _random_conv_cases(int(_os.environ.get("HWG_TEST_RANDOM_CONVS", "36")), int(_os.environ.get("HWG_TEST_RANDOM_SEED", "2026")))


@pytest.mark.parametrize("case", RANDOM_CONV_CASES, ids=[c[0] for c in RANDOM_CONV_CASES])
def test_conv_random_geometry(cuda, case):
    """seeded random conv / conv-transpose geometries (odd sizes, 1-pixel maps, RIMES channel counts, dilation, every schedule branch)"""
    test_conv_fwd_bwd(cuda, case)


def test_deferred_wgrad_reduce_is_bit_identical(cuda):
    """ops.DEFER_REDUCE: the sums of the weight-gradient partial images of a whole backward pass are queued and made by ONE table-driven
    launch per 32 gradients (hwg_wgrad_defer_flush). Every reduce schedule of the library (vec4, scalar / taps-as-N, 4- and 32-lane, row-contiguous
    with and without lanes, Winograd weight gradient) and a tensor that receives TWO gradients in one pass (a layer applied twice: summed in
    queue order by consecutive launches) must come out bit-identical to the launch-per-gradient path, bias gradients included."""
    from handwriting_line_generation_amd import ops
    g = torch.Generator().manual_seed(21)
    layers = [  # N, H, W, C, K, R, S, pad
        (2, 6, 70, 512, 512, 3, 3, (1, 1)),      # row-contiguous reduce (>= 65536 pairs)
        (2, 8, 40, 256, 256, 3, 3, (1, 1)),      # row-contiguous, lanes
        (4, 20, 64, 64, 64, 3, 3, (0, 1)),       # Winograd weight gradient, many ranges
        (4, 16, 48, 128, 128, 3, 3, (1, 1)),
        (2, 30, 100, 1, 64, 7, 7, (0, 3)),       # taps-as-N (scalar reduce)
        (2, 1, 60, 256, 256, 1, 3, (0, 1)),      # 1-D layer
        (2, 24, 80, 16, 16, 3, 3, (1, 1)),       # narrow all-taps kernel
        (2, 12, 40, 32, 64, 4, 4, (1, 1)),
    ]
    ws = []
    for (N, H, W, C, K, R, S, pad) in layers:
        w = torch.nn.Parameter((torch.randn(K, C, R, S, generator=g) * 0.05).to(cuda))
        b = torch.nn.Parameter((torch.randn(K, generator=g) * 0.05).to(cuda))
        ws.append((w, b))
    xs = [torch.randn(N, H, W, C, generator=g).to(cuda) for (N, H, W, C, K, R, S, pad) in layers]
    x_twice = torch.randn(3, 10, 33, 128, generator=g).to(cuda)        # layer 3 applied a second time, on another geometry

    def run(defer):
        for w, b in ws:
            w.grad = None; b.grad = None
        loss = 0
        for (w, b), x, (N, H, W, C, K, R, S, pad) in zip(ws, xs, layers):
            xx = x.clone().requires_grad_(True)
            stride = (2, 2) if R == 4 else (1, 1)
            y = ops.conv2d(xx, w, b, stride, pad)
            loss = loss + (y * y).sum() * 1e-3
        y2 = ops.conv2d(x_twice.clone().requires_grad_(True), ws[3][0], ws[3][1], (1, 1), (1, 1))
        loss = loss + (y2 * y2).sum() * 1e-3
        ops.DEFER_REDUCE = defer
        try:
            loss.backward()
        finally:
            ops.DEFER_REDUCE = False
            before = ops._defer["launches"]
            ops.join_side_stream()
        torch.cuda.synchronize()
        return [(w.grad.clone(), b.grad.clone()) for w, b in ws], ops._defer["launches"] - before

    ref, n0 = run(False)
    got, n1 = run(True)
    assert n0 == 0 and 2 <= n1 <= 3, (n0, n1)          # one launch + one more for the second gradient of the shared layer
    for i, ((rw, rb), (gw, gb)) in enumerate(zip(ref, got)):
        assert torch.equal(rw, gw), "layer %d weight gradient differs: max %.3e" % (i, float((rw - gw).abs().max()))
        assert torch.equal(rb, gb), "layer %d bias gradient differs" % i
        assert float(rw.abs().max()) > 0


def test_tape_backward_sets_equal_separate_backward_passes(cuda):
    """ops.Tape.backward_sets: S upstream gradients through a recorded sub-network in ONE pass (data-gradient convolutions, blur and
    resampling on S x N samples; AdaIN per set on slices of a preallocated result; the S weight gradients of a layer as one grouped launch,
    hwg_conv_wgrad_sets / hwg_wino_wgrad_sets, incl. a transposed layer whose sets differ in the gathered tensor) with every set's parameter
    gradients redirected into its own flat buffer (ops.grad_set) - against S separate autograd backward passes of the same ops."""
    import numpy as np
    from handwriting_line_generation_amd import ops
    from handwriting_line_generation_amd.trainer.flat_params import FlatParams
    g = torch.Generator().manual_seed(5)
    N, H, W, C0, C1, C2 = 2, 8, 36, 64, 64, 32

    def P(*shape, s=0.05):
        return torch.nn.Parameter((torch.randn(*shape, generator=g) * s).to(cuda))
    w1, b1 = P(C1, C0, 3, 3), P(C1)                   # 3x3 (Winograd weight gradient at these sizes or the direct one: the planner's choice)
    nw = P(1, C1, 1, 1, s=0.3)
    wt, bt = P(C1, C2, 4, 4), P(C2)                   # transposed, stride 2
    w3, b3 = P(16, C2, 3, 3), P(16)                   # narrow output
    w4 = P(1, 16, 1, 1, s=0.5)                        # K = 1 head (direct kernels: no grouped path)
    params = [w1, b1, nw, wt, bt, w3, b3, w4]
    flat = FlatParams(params, {"main": params})
    x = torch.randn(N, H, W, C0, generator=g).to(cuda)
    gam, bet = (torch.randn(N, C1, generator=g) * 0.5 + 1).to(cuda), (torch.randn(N, C1, generator=g) * 0.1).to(cuda)
    noise = torch.randn(N, H, W, C1, generator=g).to(cuda)

    def net(xin, gm, bt_):
        h = ops.conv2d(xin, w1, b1, 1, 1)
        h = ops.adain_epilogue(h, noise, nw, gm, bt_, 0.7, 0.2)
        h = ops.conv_transpose2d(h, wt, bt, stride=2, padding=1)
        h = ops.blur3(h)
        h = ops.conv2d(h, w3, b3, 1, 1)
        h = ops.upsample_nearest(h, (2, 1))
        return ops.tanh(ops.conv2d(h, w4, None))
    S = 3
    # reference: one autograd backward pass per set
    xa = x.clone().requires_grad_(True); ga = gam.clone().requires_grad_(True); ba = bet.clone().requires_grad_(True)
    ya = net(xa, ga, ba)
    ups = [torch.randn(ya.shape, generator=g).to(cuda) for _ in range(S)]
    want = []
    for s_ in range(S):
        flat.zero_grad("main")
        xa.grad = ga.grad = ba.grad = None
        ya.backward(ups[s_], retain_graph=True)
        ops.join_side_stream()
        want.append((flat.flat_grad.clone(), flat.touched.copy(), xa.grad.clone(), ga.grad.clone(), ba.grad.clone()))
    # tape: forward recorded once, the three gradients through it together; sets 0 and 1 into stash buffers, set 2 into the parameters' own gradients
    flat.zero_grad("main")
    tape = ops.Tape()
    xt, gt, bt2 = tape.watch(x.clone()), tape.watch(gam.clone()), tape.watch(bet.clone())
    ops.TAPE = tape
    try:
        with torch.no_grad():
            yt = net(xt, gt, bt2)
    finally:
        ops.TAPE = None
    assert torch.equal(yt, ya.detach())
    bufs = [(torch.zeros_like(flat.flat_grad), np.zeros(flat.nt, dtype=bool)) for _ in range(S - 1)]
    targets = [bufs[0], bufs[1], None]
    res = tape.backward_sets(yt, ups, targets)
    ops.join_side_stream()
    torch.cuda.synchronize()
    for s_ in range(S):
        fg, touched, dx, dg, db = want[s_]
        got_flat = bufs[s_][0] if s_ < S - 1 else flat.flat_grad
        got_mask = bufs[s_][1] if s_ < S - 1 else flat.touched
        assert np.array_equal(got_mask, touched), "set %d: touched pattern" % s_
        for k, pi in enumerate(flat.order):
            a = fg[int(flat.offsets[k]): int(flat.offsets[k]) + int(flat.numel[k])]
            b = got_flat[int(flat.offsets[k]): int(flat.offsets[k]) + int(flat.numel[k])]
            _close(b, a, "set %d parameter %d gradient" % (s_, pi), tol=2e-5)
        _close(res[id(xt)][s_], dx, "set %d dx" % s_, tol=2e-5)
        _close(res[id(gt)][s_], dg, "set %d dgamma" % s_, tol=2e-5)
        _close(res[id(bt2)][s_], db, "set %d dbeta" % s_, tol=2e-5)
    assert float(bufs[0][0].abs().max()) > 0 and not torch.equal(bufs[0][0], bufs[1][0])


def test_ctypes_fallback_without_the_call_thunks(cuda):
    """_lib falls back to plain ctypes calls (with a warning) when _hwgcall.so cannot be used - thunks built for another interpreter, or not
    built; HWG_NO_THUNKS=1 forces that path: a convolution forward / backward through it must equal the thunk path's bits."""
    import subprocess
    import sys
    code = (
        "import torch, warnings\n"
        "warnings.simplefilter('ignore')\n"
        "from handwriting_line_generation_amd import _lib, ops\n"
        "g = torch.Generator().manual_seed(1)\n"
        "x = torch.randn(2, 9, 20, 32, generator=g).cuda().requires_grad_(True)\n"
        "w = (torch.randn(48, 32, 3, 3, generator=g) * 0.1).cuda().requires_grad_(True)\n"
        "b = torch.randn(48, generator=g).cuda().requires_grad_(True)\n"
        "y = ops.conv2d(x, w, b, 1, 1)\n"
        "(y * y).sum().backward()\n"
        "torch.cuda.synchronize()\n"
        "print(_lib._hwgcall is None, float(y.double().sum()), float(w.grad.double().abs().sum()), float(x.grad.double().abs().sum()), float(b.grad.double().sum()))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for no_thunks in ("", "1"):
        env = dict(os.environ, PYTHONPATH=root, HWG_NO_THUNKS=no_thunks)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1].split())
    assert outs[0][0] == "False" and outs[1][0] == "True", outs
    assert outs[0][1:] == outs[1][1:], outs


def test_spectral_backward_of_several_layers_in_two_launches_is_bit_identical(cuda):
    """hwg_spectral_bwd_multi (the discriminator's spectral-norm layers walking backward together) against one hwg_spectral_bwd per layer:
    the same bits in every destination, accumulating into non-zero gradients, for layer sizes on both sides of the partial-count steps"""
    from handwriting_line_generation_amd import _lib as L, ops
    g = torch.Generator().manual_seed(9)
    shapes = [(64, 1 * 9), (128, 64 * 9), (256, 128 * 9), (128, 256 * 3), (1, 256 * 9), (16, 7)]
    layers = []
    for R, K in shapes:
        t = lambda *s: torch.randn(*s, generator=g).to(cuda)     # noqa: E731
        layers.append(dict(dwsn=t(R, K), wbar=t(R, K), u=t(R), v=t(K), sigma=(torch.rand(1, generator=g) + 0.5).to(cuda), dst0=t(R, K), R=R, K=K))
    st = torch.cuda.current_stream().cuda_stream
    single = []
    for l in layers:
        dst = l["dst0"].clone()
        ws = ops.workspace(L.query("hwg_spectral_workspace", l["R"], l["K"]), cuda)
        L.call("hwg_spectral_bwd", l["dwsn"], l["wbar"], l["u"], l["v"], l["sigma"], dst, l["R"], l["K"], 1, ws, ws.numel(), st)
        single.append(dst)
    rec = np.zeros(len(layers), dtype=ops._SN_BWD_REC)
    multi = [l["dst0"].clone() for l in layers]
    for k, l in enumerate(layers):
        rec[k] = (l["dwsn"].data_ptr(), l["wbar"].data_ptr(), l["u"].data_ptr(), l["v"].data_ptr(), l["sigma"].data_ptr(), multi[k].data_ptr(), l["R"], l["K"], 1, 0)
    ws = ops.workspace(L.query("hwg_spectral_bwd_multi_workspace", len(layers)), cuda)
    L.call("hwg_spectral_bwd_multi", rec.ctypes.data, len(layers), ws, ws.numel(), st)
    torch.cuda.synchronize()
    for k, (a, b) in enumerate(zip(single, multi)):
        assert torch.equal(a, b), "layer %d %s" % (k, shapes[k])
        # and the arithmetic itself against the closed form in fp64
        l = layers[k]
        sg = l["sigma"].double()
        want = l["dst0"].double() + l["dwsn"].double() / sg - (l["dwsn"].double() * l["wbar"].double()).sum() / sg ** 2 * torch.outer(l["u"].double(), l["v"].double())
        assert float((b.double() - want).abs().max()) < 2e-5 * float(want.abs().max())


def test_tape_guard_rejects_torch_level_ops_on_taped_tensors(cuda, monkeypatch):
    """a taped forward tracks Function calls and whole-memory reshapes only; with HWG_TAPE_CHECK on, any other torch-level op on a
    taped tensor raises instead of silently dropping the gradient path through its result"""
    from handwriting_line_generation_amd import ops
    from handwriting_line_generation_amd._lib import HwgError
    monkeypatch.setattr(ops, "TAPE_CHECK", True)
    x = torch.randn(2, 4, 4, 16, device=cuda)
    tape = ops.Tape()
    xin = tape.watch(x)
    with ops.taping(tape), torch.no_grad():
        y = ops.relu(xin)
        z = y.view(2, -1)                          # a reshape of a taped tensor: adopted by the next op
        assert z.shape == (2, 256) and y.is_contiguous() and y.dim() == 4
        w = torch.empty(3, device=cuda) * 2.0      # torch-level work on tensors the tape does not know is none of its business
        with pytest.raises(HwgError, match="torch-level op"):
            y * 2.0
        with pytest.raises(HwgError, match="torch-level op"):
            torch.cat([y, y], dim=0)
        with pytest.raises(HwgError, match="torch-level op"):
            z[:, :4].sum()
    assert ops.TAPE is None and len(tape.nodes) == 1 and w.shape == (3,)
