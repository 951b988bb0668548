"""CNN-only CTC recogniser on HIP kernels (reference: model/cnn_only_hwr.py:7-107).

7 conv3x3(+ReLU) with BatchNorm after conv2/4/6, four max-pools (the last two pool height only), collapsing a
64-row line to one row, then four dilated 1-D convs + BN + ReLU and a 512->n_class conv with LogSoftmax.
Output is [T,B,n_class] log-probabilities, T = W/4 - 6. BatchNorm follows `self.training` - the GAN trainer
leaves the "frozen" recogniser in train mode, so it normalises with batch statistics.
"""
from torch import nn

from .. import ops
from .layers import BatchNorm1d, BatchNorm2d, Conv1d, Conv2d, GroupNorm, Marker, group_count


class CNNOnlyHWR(nn.Module):
    CHANNELS = [64, 128, 256, 256, 512, 512, 512]
    PADS = [1, 1, 1, 1, 1, 0, 0]
    NORMED = (2, 4, 6)

    def __init__(self, nclass, nc=1, cnnOutSize=512, nh=512, leakyRelu=False, norm="group", small=False, pad=False):
        super().__init__()
        if leakyRelu or small:
            raise NotImplementedError("leakyRelu/small recogniser variants are not used by any shipped config")
        if pad == "less":
            self.pad_cols = 64
        elif pad:
            self.pad_cols = 128
        else:
            self.pad_cols = 0
        self.norm_kind = "group" if (norm is not None and "group" in norm) else ("batch" if norm else None)
        cnn = nn.Sequential()
        for i, ch in enumerate(self.CHANNELS):
            cin = nc if i == 0 else self.CHANNELS[i - 1]
            cnn.add_module("conv%d" % i, Conv2d(cin, ch, 3, 1, self.PADS[i]))
            if i in self.NORMED and self.norm_kind == "group":
                cnn.add_module("groupnorm%d" % i, GroupNorm(group_count(ch), ch))
            elif i in self.NORMED and self.norm_kind == "batch":
                cnn.add_module("batchnorm%d" % i, BatchNorm2d(ch))
            cnn.add_module("relu%d" % i, Marker("relu"))
            if i in (0, 1):
                cnn.add_module("pooling%d" % i, Marker("maxpool 2x2"))
            elif i == 3:
                cnn.add_module("pooling2", Marker("maxpool (2,2)/(2,1)/(0,1)"))
            elif i == 5:
                cnn.add_module("pooling3", Marker("maxpool (2,2)/(2,1)/(0,1)"))
        self.cnn = cnn
        size1d = 512
        mk = (lambda: GroupNorm(group_count(size1d), size1d)) if norm == "group" else (lambda: BatchNorm1d(size1d))
        layers = []
        for dil, pad_ in ((2, 2), (4, 4), (1, 0), (8, 8)):
            layers += [Conv1d(size1d, size1d, 3, 1, pad_, dil), mk(), Marker("relu")]
        layers += [Conv1d(size1d, nclass, 3, 1, 0, 1), Marker("log softmax")]
        self.cnn1d = nn.Sequential(*layers)

    def _conv_block(self, i, x, pool=None):
        """conv (+norm) + ReLU, followed by the max-pool `pool` = (kernel, stride, padding) of the reference's Sequential when there is one"""
        conv = getattr(self.cnn, "conv%d" % i)
        if i in self.NORMED and self.norm_kind is not None:
            h = conv(x)
            norm = getattr(self.cnn, ("groupnorm%d" if self.norm_kind == "group" else "batchnorm%d") % i)
            h = norm(h, "relu")
            return ops.max_pool2d(h, *pool) if pool else h
        # the bias is added in the conv epilogue (its gradient then rides along in the weight-gradient kernel); ReLU is one elementwise pass
        h = ops.conv2d(x, conv.weight, conv.bias, conv.stride, conv.padding, conv.dilation)
        if pool:
            # ReLU after the pool instead of before it: max and ReLU are both monotone, relu(max(w)) == max(relu(w)) exactly, and the two
            # orders route the gradient to the same window element wherever it is not zero anyway (a window whose maximum is <= 0 passes
            # nothing back in either order) - same bits forward and backward, with the ReLU passes over a half / quarter of the pixels
            return ops.max_pool2d(h, *pool, relu=True)          # (the ReLU rides along in the pooling kernels: hwg_maxpool_relu_*)
        return ops.bias_act(h, None, None, ops.ACT_RELU)

    logit_offset = None

    def forward(self, input, style=None):
        from .. import replay
        if replay.ENABLED:
            # one C-side replay of the recorded launch list instead of ~45 + ~70 Python -> torch -> C-ABI round trips (replay.py); None = not
            # eligible / not recorded yet / rejected by its self-check: the eager path below
            out = replay.hwr_forward(self, input)
            if out is not None:
                return out
        with ops.scope("HWR"):
            return self._forward(input, style)

    def _forward(self, input, style=None):
        """input NCHW [B,1,64,W] -> [T,B,n_class]"""
        x = ops.to_nhwc(input)
        if self.pad_cols:
            x = ops.pad2d(x, self.pad_cols, self.pad_cols, 0, 0, "constant", 0.0)
        x = self._conv_block(0, x, (2, 2))
        x = self._conv_block(1, x, (2, 2))
        x = self._conv_block(2, x)
        x = self._conv_block(3, x, ((2, 2), (2, 1), (0, 1)))
        x = self._conv_block(4, x)
        x = self._conv_block(5, x, ((2, 2), (2, 1), (0, 1)))
        x = self._conv_block(6, x)
        B, H, W, C = x.shape
        if H != 1:
            # the reference flattens (c,h) into channels; only height-1 features are meaningful for the shipped 64-px configs
            raise ValueError("recogniser expects 64-pixel-high lines (feature height %d != 1)" % H)
        for k in range(0, 12, 3):
            x = self.cnn1d[k](x)
            x = self.cnn1d[k + 1](x, "relu")
        x = self.cnn1d[12](x)
        if self.logit_offset is not None:
            # bench-only ("peaked recogniser" workload): a fixed pattern added to the logits so that a randomly initialised
            # recogniser predicts like a trained one (mostly blanks, confident characters) and the character experts see a realistic load;
            # every kernel of the recogniser still runs, forward and backward
            off = self.logit_offset(x.shape[0], x.shape[2], x.shape[3], x.device)
            x = ops.add(x, off)
        return ops.log_softmax_tbc(x)
