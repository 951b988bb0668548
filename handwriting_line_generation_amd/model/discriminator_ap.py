"""Two-head spectral-norm patch discriminator on HIP kernels (reference: model/discriminator_ap.py:11-161).

NHWC throughout; every vertical padding is 0 so a 64-row line shrinks 64->58->56->28->26->24->12->10->5->3(->1).
Spectral-normalised layers keep the reference's parameter names (`<layer>.module.{bias,weight_u,weight_v,weight_bar}`)
and run one power iteration on every forward, train or eval, mutating u and v (discriminator_ap.py:20-32,63-65).
"""
import torch
from torch import nn

from .. import ops, rng
from .layers import Conv2d, Dropout2d, GroupNorm, Marker, group_count


class _SNParams(nn.Module):
    pass


class SpectralConv2d(nn.Module):
    def __init__(self, in_ch, out_ch, kernel, padding):
        super().__init__()
        proto = nn.Conv2d(in_ch, out_ch, kernel, stride=1, padding=padding)  # for the default initialisation only
        self.kernel, self.padding = proto.kernel_size, proto.padding
        m = _SNParams()
        m.bias = nn.Parameter(proto.bias.data)
        height = out_ch
        width = proto.weight.data.view(height, -1).shape[1]
        u = torch.randn(height)
        v = torch.randn(width)
        m.weight_u = nn.Parameter(u / (u.norm() + 1e-12), requires_grad=False)
        m.weight_v = nn.Parameter(v / (v.norm() + 1e-12), requires_grad=False)
        m.weight_bar = nn.Parameter(proto.weight.data)
        self.module = m

    def forward(self, x, with_bias=True):
        m = self.module
        fresh, self._fresh = getattr(self, "_fresh", None), None
        if fresh is not None:      # the network ran this forward pass's power iterations for all its layers at once (DiscriminatorAP._power_iterations)
            w = ops.spectral_scale(m.weight_bar, fresh)
        else:
            w = ops.spectral_normalize(m.weight_bar, m.weight_u.data, m.weight_v.data)
        return ops.conv2d(x, w, m.bias if with_bias else None, 1, self.padding)


class DiscriminatorAP(nn.Module):
    _masks = None

    def __init__(self, dim=64, use_low=False, use_med=True, small=False):
        super().__init__()
        if small:
            raise NotImplementedError("'small' discriminator is not used by any shipped config")
        self.use_low, self.use_med = use_low, use_med
        self.leak = 0.1
        self.in_conv = nn.Sequential(Conv2d(1, dim, 7, stride=1, padding=(0, 3)), GroupNorm(group_count(dim), dim), Marker("lrelu"))
        self.convs1 = nn.Sequential(
            SpectralConv2d(dim, dim, 3, (0, 1)), Marker("lrelu"), Marker("avgpool 2"),
            SpectralConv2d(dim, 2 * dim, 3, (0, 1)), Dropout2d(0.05), Marker("lrelu"))
        self.convs2 = nn.Sequential(SpectralConv2d(2 * dim, 2 * dim, 3, (0, 1)), Marker("lrelu"), Marker("avgpool 2"))
        self.convs3 = nn.Sequential(
            Conv2d(2 * dim, 2 * dim, 3, stride=1, padding=(0, 1)), GroupNorm(group_count(2 * dim), 2 * dim), Marker("lrelu"), Marker("avgpool 2"),
            SpectralConv2d(2 * dim, 4 * dim, 3, (0, 1)), Dropout2d(0.05), Marker("lrelu"))
        if use_med:
            self.finalMed = nn.Sequential(SpectralConv2d(4 * dim, 1, 3, (0, 1)))
        if use_low:
            self.convs4 = nn.Sequential(
                SpectralConv2d(4 * dim, 2 * dim, 3, (0, 1)), Dropout2d(0.025), Marker("lrelu"), Marker("avgpool (1,2)"),
                SpectralConv2d(2 * dim, 4 * dim, (1, 3), (0, 1)), Dropout2d(0.025), Marker("lrelu"),
                SpectralConv2d(4 * dim, 4 * dim, (1, 3), (0, 1)), Dropout2d(0.025), Marker("lrelu"), Marker("avgpool (1,2)"),
                SpectralConv2d(4 * dim, 4 * dim, (1, 3), (0, 1)), Dropout2d(0.025), Marker("lrelu"),
                SpectralConv2d(4 * dim, 1, 1, (0, 0)))

    def _sn_act(self, conv, x, drop=None, pool=None):
        """SN conv (+bias in its epilogue, so the bias gradient comes out of the weight-gradient kernel) -> Dropout2d mask, LeakyReLU
        (-> AvgPool2d `pool`, fused into the activation pass: the full-resolution activation is never written)"""
        h = conv(x, with_bias=True)
        mask = drop.mask_for(h, self._masks) if drop is not None else None
        if pool is not None:
            return ops.act_avg_pool2d(h, pool, mask, ops.ACT_LRELU, self.leak)
        return ops.bias_act(h, None, mask, ops.ACT_LRELU, self.leak)

    def _low_head(self, mL):
        c = self.convs4
        h = self._sn_act(c[0], mL, c[1], pool=(1, 2))
        h = self._sn_act(c[4], h, c[5])
        h = self._sn_act(c[7], h, c[8], pool=(1, 2))
        h = self._sn_act(c[11], h, c[12])
        return c[14](h)

    def forward(self, x, return_features=False):
        with ops.scope("D"):
            return self._forward(x, return_features)

    def _power_iterations(self, return_features):
        """one power iteration of every spectral-norm layer this forward pass is going to use (the reference runs them layer by layer inside
        the forward; they are independent of each other and of the activations), batched into four launches"""
        layers = [self.convs1[0], self.convs1[3], self.convs2[0], self.convs3[4]]
        if self.use_med and not return_features:
            layers.append(self.finalMed[0])
        if self.use_low or return_features:
            layers += [self.convs4[i] for i in (0, 4, 7, 11, 14)]
        key = bool(return_features)
        banks = self.__dict__.setdefault("_sn_banks", {})
        bank = banks.get(key)
        if bank is None or not bank.valid():
            bank = banks[key] = ops.SpectralBank([(l.module.weight_bar, l.module.weight_u, l.module.weight_v) for l in layers])
        for l, fresh in zip(layers, bank.update()):
            l._fresh = fresh

    def _forward(self, x, return_features=False):
        """x: NCHW [N,1,64,W] (as the reference passes it) -> list of [N, -1] patch predictions"""
        if x.is_cuda:
            self._power_iterations(return_features)
        batch = x.shape[0]
        # the Dropout2d masks of this pass (two in the trunk, four in the low head) from one Philox launch
        dim = self.in_conv[0].out_channels
        drops = [(self.convs1[4], 2 * dim), (self.convs3[5], 4 * dim)]
        if self.use_low or return_features:
            drops += [(self.convs4[1], 2 * dim), (self.convs4[5], 4 * dim), (self.convs4[8], 4 * dim), (self.convs4[12], 4 * dim)]
        specs = [d.spec(batch, c) for d, c in drops]
        self._masks = rng.MaskBlock([sp for sp in specs if sp is not None], x.device)
        h = ops.to_nhwc(x)
        h = self.in_conv[0](h)
        h = self.in_conv[1](h, "lrelu", self.leak)
        h = self._sn_act(self.convs1[0], h, pool=2)
        h = self._sn_act(self.convs1[3], h, self.convs1[4])
        h = self._sn_act(self.convs2[0], h, pool=2)
        h = self.convs3[0](h)
        h = self.convs3[1](h, "lrelu", self.leak)
        h = ops.avg_pool2d(h, 2)
        mL = self._sn_act(self.convs3[4], h, self.convs3[5])
        if return_features:
            return ops.to_nchw(mL), ops.to_nchw(self._low_head(mL))
        outs = []
        if self.use_med:
            outs.append(self.finalMed[0](mL).reshape(batch, -1))
        if self.use_low:
            outs.append(self._low_head(mL).reshape(batch, -1))
        return outs
