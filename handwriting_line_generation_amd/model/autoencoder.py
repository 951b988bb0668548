"""Style autoencoder on HIP kernels: the `2tight` family of the reference's model/autoencoder.py
(Encoder2 :341-410, DecoderNoSkip :302-339, E_HWR :596-628, Autoencoder :8-66). Encoder2(32) doubles as the
perceptual-loss network of the GAN trainer (trainer/hw_with_style_trainer.py:136-160,725-748).

Notes on behaviour carried over from the reference:
  * `conv1` starts with an in-place ReLU, so the residual that is added back is the ReLU'd tensor;
  * Dropout2d layers are live whenever the module is in train mode - the GAN trainer never calls .eval() on it;
  * the encoder returns (code, mid_features) and both are compared by the perceptual loss.
"""
import torch
from torch import nn

from .. import ops, rng
from .layers import Conv1d, Conv2d, ConvTranspose2d, Dropout2d, GroupNorm, Marker, group_count


def _gn(ch):
    return GroupNorm(group_count(ch), ch)


class Encoder2(nn.Module):
    def __init__(self, out_dim=256):
        super().__init__()
        self.down_conv1 = nn.Sequential(Conv2d(1, 32, 5, padding=2), _gn(32), Marker("relu"), Marker("avgpool 2"), Conv2d(32, 32, 1))
        self.conv1 = nn.Sequential(Marker("relu (in place)"), Conv2d(32, 32, 3, padding=1), _gn(32), Dropout2d(0.1), Marker("relu"),
                                   Conv2d(32, 32, 3, padding=1))
        self.down_conv2 = nn.Sequential(_gn(32), Marker("relu"), Marker("avgpool 2"), Conv2d(32, 64, 1))
        self.conv2 = nn.Sequential(_gn(64), Dropout2d(0.1), Marker("relu"), Conv2d(64, 64, 3, padding=1), _gn(64), Dropout2d(0.1), Marker("relu"),
                                   Conv2d(64, 64, 3, padding=1))
        self.down_conv3 = nn.Sequential(_gn(64), Marker("relu"), Marker("avgpool 2"), Conv2d(64, 128, 3), _gn(128), Dropout2d(0.1), Marker("relu"),
                                        Conv2d(128, out_dim, (6, 3)))

    def forward(self, x):
        with ops.scope("Encoder"):
            return self._forward(x)

    def _forward(self, x):
        """x NCHW [N,1,64,W] -> (code [N,out,1,W/8-4], mid [N,64,16,W/4]) both NCHW like the reference"""
        d1, c1, d2, c2, d3 = self.down_conv1, self.conv1, self.down_conv2, self.conv2, self.down_conv3
        # the four Dropout2d masks of this pass from one Philox launch
        B = x.shape[0]
        specs = [m.spec(B, c) for m, c in ((c1[3], 32), (c2[1], 64), (c2[5], 64), (d3[5], 128))]
        mb = rng.MaskBlock([sp for sp in specs if sp is not None], x.device)
        h = ops.to_nhwc(x)
        h = d1[1](d1[0](h), "relu")
        h = d1[4](ops.avg_pool2d(h, 2))
        r = ops.relu(h)
        h = c1[2](c1[1](r), "relu", 0.0, c1[3].mask_for_shape(r.shape[0], 32, r.device, mb))
        h = ops.add(c1[5](h), r)
        h = d2[0](h, "relu")
        h = d2[3](ops.avg_pool2d(h, 2))
        res = h
        h = c2[0](h, "relu", 0.0, c2[1].mask_for_shape(h.shape[0], 64, h.device, mb))
        h = c2[4](c2[3](h), "relu", 0.0, c2[5].mask_for_shape(h.shape[0], 64, h.device, mb))
        mid = ops.add(c2[7](h), res)
        h = d3[0](mid, "relu")
        h = d3[3](ops.avg_pool2d(h, 2))
        h = d3[4](h, "relu", 0.0, d3[5].mask_for_shape(h.shape[0], 128, h.device, mb))
        code = d3[7](h)
        return ops.to_nchw(code), ops.to_nchw(mid)


class DecoderNoSkip(nn.Module):
    SPEC = [  # (out_ch, kernel, stride, pad)
        (256, (6, 3), 1, 0), (256, 3, 1, 0), (128, 4, 2, 1), (128, 3, 1, 1), (64, 4, 2, 1), (64, 3, 1, 1), (32, 4, 2, 1)]

    def __init__(self, input_dim=512):
        super().__init__()
        layers = [Marker("relu")]
        cin = input_dim
        for (cout, k, s, p) in self.SPEC:
            layers += [ConvTranspose2d(cin, cout, k, stride=s, padding=p), _gn(cout), Marker("relu")]
            cin = cout
        layers += [ConvTranspose2d(cin, 1, 3, padding=1), Marker("tanh")]
        self.up_conv1 = nn.Sequential(*layers)

    def forward(self, x, mid_features=None):
        h = ops.relu(ops.to_nhwc(x))
        u = self.up_conv1
        for i in range(len(self.SPEC)):
            h = u[2 + 3 * i](u[1 + 3 * i](h), "relu")
        h = ops.tanh(u[1 + 3 * len(self.SPEC)](h))
        return ops.to_nchw(h)


class E_HWR(nn.Module):
    SPEC = [(3, 1, 1), (3, 2, 2), (3, 4, 4), (5, 2, 1)]  # (kernel, pad, dilation)

    def __init__(self, n_class, n_in):
        super().__init__()
        layers = []
        cin = n_in
        for (k, p, d) in self.SPEC:
            layers += [Conv1d(cin, 512, k, 1, p, d), _gn(512), Marker("dropout 0.1 (elementwise)"), Marker("relu")]
            cin = 512
        layers += [Conv1d(512, n_class, 1), Marker("log softmax")]
        self.classify = nn.Sequential(*layers)
        self.p_drop = 0.1

    def forward(self, x):
        """x NCHW [N,C,1,W] code -> [W,N,n_class] log-probs"""
        h = ops.to_nhwc(x)
        c = self.classify
        for i in range(len(self.SPEC)):
            h = c[4 * i + 1](c[4 * i](h))
            if self.training:
                h = ops.mul_const(h, rng.element_mask(h, self.p_drop))
            h = ops.relu(h)
        return ops.log_softmax_tbc(c[16](h))


class Autoencoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        kind = config.get("type")
        dims = {"2": 256, "2tight": 32, "2tighter": 16}
        if kind not in dims:
            raise NotImplementedError("Autoencoder type %r: only the Encoder2/DecoderNoSkip family used by the shipped configs is built" % kind)
        out = dims[kind]
        self.encoder = Encoder2(out)
        self.decoder = DecoderNoSkip(out)
        if "hwr_batch" in config:
            raise NotImplementedError("E_HWR_batch is not used by any shipped config")
        self.hwr = E_HWR(config["hwr"], out) if "hwr" in config else None

    def forward(self, x):
        enc, mid = self.encoder(x)
        if self.hwr is None:
            return self.decoder(enc, mid)
        return self.decoder(enc, mid), self.hwr(enc)
