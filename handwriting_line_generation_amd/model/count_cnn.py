"""Spacing / duplication predictor on HIP kernels (reference: model/count_cnn.py:7-44).

Input: one-hot text [L,B,n_class] and a style vector; output [L,B,n_out] = cnn(cat(text, style)) * std + mean.
The reference's nn.Dropout2d on a 3-D tensor acts as per-(sample, channel) dropout; it is fused with the GroupNorm
and ReLU that surround it.
"""
import torch
from torch import nn

from .. import ops, rng
from .layers import Conv1d, Dropout2d, GroupNorm, Marker, group_count


class CountCNN(nn.Module):
    def __init__(self, class_size, style_size, hidden_size=128, n_out=1, emb_style=0):
        super().__init__()
        h = hidden_size
        self.cnn = nn.Sequential(
            Conv1d(class_size + style_size, h, 3, 1, 1), GroupNorm(group_count(h), h), Dropout2d(0.1), Marker("relu"),
            Conv1d(h, h // 2, 3, 1, 1), GroupNorm(group_count(h // 2), h // 2), Dropout2d(0.1), Marker("relu"),
            Conv1d(h // 2, h // 4, 3, 1, 1), GroupNorm(group_count(h // 4), h // 4), Marker("relu"),
            Conv1d(h // 4, n_out, 1, 1, 0))
        self.n_out = n_out
        if n_out == 2:
            self.mean = nn.Parameter(torch.tensor([2.0, 0.0]))
            self.std = nn.Parameter(torch.tensor([1.5, 0.5]))
        else:
            self.mean = nn.Parameter(torch.full((1, n_out), 2.0))
            self.std = nn.Parameter(torch.full((1, n_out), 1.0))

    def forward(self, input, style):
        with ops.scope("Spacer"):
            return self._forward(input, style)

    def _forward(self, input, style):
        """input [L,B,C] (time major) or NHWC [B,1,L,C]; style [B,S] -> [L,B,n_out]"""
        if input.dim() == 3:
            Lr, B, C = input.shape
            twin = ops.nhwc_of(input)
            input = twin if twin is not None else ops.permute4(input.contiguous(), (B, 1, Lr, C), (C, 0, B * C, 1))
        B, _, Lr, _ = input.shape
        c = self.cnn
        x = ops.cat_channels([input, style.contiguous()], (B, 1, Lr))
        specs = [m.spec(B, n.num_channels) for m, n in ((c[2], c[1]), (c[6], c[5]))]
        mb = rng.MaskBlock([sp for sp in specs if sp is not None], x.device)        # both Dropout2d masks from one Philox launch
        x = c[1](c[0](x), "relu", 0.0, c[2].mask_for_shape(B, c[1].num_channels, x.device, mb))
        x = c[5](c[4](x), "relu", 0.0, c[6].mask_for_shape(B, c[5].num_channels, x.device, mb))
        x = c[9](c[8](x), "relu")
        x = c[11](x)                                   # [B,1,L,n_out]
        y = ops.channel_affine(x, self.std.view(-1), self.mean.view(-1))
        return ops.permute_bl_to_lb(y)                 # [L,B,n_out]
