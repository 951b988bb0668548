"""StyleGAN-like line generator on HIP kernels - behaviour of the reference's model/pure_gen.py:12-311.

Data flow (NHWC): content one-hot rows [B,1,T,n_class] (+ embedded style broadcast on the channel axis)
  -> ConvTranspose (4,3): 1 row -> 4 rows -> two vertical-only nearest-upsample blocks -> two fused stride-2
  transposed-conv blocks -> 1x1 equal-lr conv -> tanh, giving a 64 x 4T image.
Every block is conv -> [noise, LeakyReLU(0.2), InstanceNorm, style affine] twice; the bracketed chain is one
fused kernel pipeline (ops.adain_epilogue). State-dict keys/shapes are those of the reference.
"""
import math

import torch
from torch import nn

from .. import ops, rng
from .layers import BlurBuffers, Conv2d, ConvTranspose2d, Linear, Marker


class _NoiseWeight(nn.Module):
    """per-channel noise gain; equal-lr parametrisation: effective = weight_orig * sqrt(2 / C) (pure_gen.py:72-79,218-247)"""

    def __init__(self, channel):
        super().__init__()
        self.weight_orig = nn.Parameter(torch.full((1, channel, 1, 1), 0.01))
        self.scale = math.sqrt(2.0 / channel)


class _StyleAffine(nn.Module):
    """style -> (gamma, beta) per channel; bias starts at gamma=1, beta=0 (pure_gen.py:52-69)"""

    def __init__(self, channel, style_dim):
        super().__init__()
        self.channel = channel
        self.style = Linear(style_dim, 2 * channel)
        with torch.no_grad():
            self.style.bias[:channel] = 1
            self.style.bias[channel:] = 0

    def forward(self, style):
        gb = self.style(style)
        return ops.split_cols(gb, [self.channel, self.channel])


class FusedUpsample(nn.Module):
    """3x3 weight -> averaged 4x4 -> stride-2 transposed conv (pure_gen.py:250-279)"""

    def __init__(self, in_channel, out_channel, kernel_size=3, padding=1):
        super().__init__()
        assert kernel_size == 3
        self.weight = nn.Parameter(torch.randn(in_channel, out_channel, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.zeros(out_channel))
        self.multiplier = math.sqrt(2.0 / (in_channel * kernel_size * kernel_size))
        self.pad = padding

    def forward(self, x):
        w4 = ops.fused_upsample_weight(self.weight, self.multiplier)
        return ops.conv_transpose2d(x, w4, self.bias, stride=2, padding=self.pad)


class StyledConvBlock(nn.Module):
    def __init__(self, in_channel, out_channel, style_dim, initial=False, upsample=False, only_vertical=False, fused=False):
        super().__init__()
        self.kind = "initial" if initial else ("fused" if (upsample and fused) else ("up" if upsample else "plain"))
        if initial:
            self.conv1 = ConvTranspose2d(in_channel, out_channel, (4, 3), padding=(0, 1))
        elif upsample and fused:
            self.conv1 = nn.Sequential(FusedUpsample(in_channel, out_channel, 3, padding=1), BlurBuffers(out_channel))
        elif upsample:
            self.up_scale = (2, 1) if only_vertical else (2, 2)
            self.conv1 = nn.Sequential(Marker("nearest upsample"), Conv2d(in_channel, out_channel, 3, padding=1), BlurBuffers(out_channel))
        else:
            self.conv1 = Conv2d(in_channel, out_channel, 3, padding=1)
        self.noise1 = _NoiseWeight(out_channel)
        self.adain1 = _StyleAffine(out_channel, style_dim)
        self.conv2 = Conv2d(out_channel, out_channel, 3, padding=1)
        self.noise2 = _NoiseWeight(out_channel)
        self.adain2 = _StyleAffine(out_channel, style_dim)

    def _first(self, x):
        if self.kind == "initial" or self.kind == "plain":
            return self.conv1(x)
        if self.kind == "fused":
            return self.conv1[1](self.conv1[0](x))
        return self.conv1[2](self.conv1[1](ops.upsample_nearest(x, self.up_scale)))

    def out_shape(self, N, H, W):
        """(H, W, C) of this block's activations for an input of H x W pixels"""
        C = self.noise1.weight_orig.shape[1]
        if self.kind == "initial":
            return 4 * H, W, C                      # ConvTranspose (4,3), pad (0,1): one row -> four
        if self.kind == "fused":
            return 2 * H, 2 * W, C
        if self.kind == "up":
            return H * self.up_scale[0], W * self.up_scale[1], C
        return H, W, C

    def forward(self, x, style, affine=None, noise=None):
        """affine: ((gamma1, beta1), (gamma2, beta2)) when the generator evaluated all style affines in one launch; noise: the forward
        pass's rng.NoiseBlock (all noise tensors of the pass from one launch)"""
        draw = noise.next if noise is not None else rng.noise_like_nhwc
        h = self._first(x)
        g, b = affine[0] if affine is not None else self.adain1(style)
        h = ops.adain_epilogue(h, draw(h), self.noise1.weight_orig, g, b, self.noise1.scale, 0.2)
        h = self.conv2(h)
        g, b = affine[1] if affine is not None else self.adain2(style)
        return ops.adain_epilogue(h, draw(h), self.noise2.weight_orig, g, b, self.noise2.scale, 0.2)


class _EqualConv1x1(nn.Module):
    """1x1 conv with run-time weight scaling sqrt(2/fan_in) (pure_gen.py:281-291)"""

    def __init__(self, in_channel, out_channel):
        super().__init__()
        self.conv = nn.Module()
        # registration order of the reference: nn.Conv2d registers (weight, bias), the equal-lr hook then deletes `weight` and registers
        # `weight_orig` behind `bias` - parameter order matters: torch.optim.Adam's checkpointed state is keyed by parameter index
        self.conv.bias = nn.Parameter(torch.zeros(out_channel))
        self.conv.weight_orig = nn.Parameter(torch.randn(out_channel, in_channel, 1, 1))
        self.scale = math.sqrt(2.0 / in_channel)

    def forward(self, x):
        return ops.conv2d(x, ops.scale(self.conv.weight_orig, self.scale), self.conv.bias)


class GenTape:
    """one taped generator forward: `image` is the leaf the losses see, `style_src` the (autograd) style tensor the generator was fed"""

    def __init__(self, tape, image, style_in, style_src):
        self.tape, self.image, self.style_in, self.style_src = tape, image, style_in, style_src

    def backward_sets(self, grads, targets):
        """grads: S gradients of the image; targets[s]: None or (buffer, mask) = where set s's parameter gradients go (ops.grad_set).
        One pass through the generator for all sets -> the S gradients of the style input"""
        with ops.scope("G"):
            res = self.tape.backward_sets(self.image, grads, targets)
        return res.get(id(self.style_in))


class SpacedGenerator(nn.Module):
    def __init__(self, n_class, style_size, dim=256, output_dim=1, n_style_trans=6, emb_dropout=False, append_style=False, small=False):
        super().__init__()
        if emb_dropout:
            raise NotImplementedError("style_emb dropout is not used by any shipped config")
        self.append_style = append_style
        in_ch = n_class + style_size if append_style else n_class
        self.conv = nn.Sequential(
            StyledConvBlock(in_ch, dim, style_size, initial=True),
            StyledConvBlock(dim, dim // 2, style_size, upsample=True, only_vertical=True),
            StyledConvBlock(dim // 2, dim // 4, style_size, upsample=True, only_vertical=True),
            StyledConvBlock(dim // 4, dim // 8, style_size, upsample=True, fused=True),
            StyledConvBlock(dim // 8, dim // 16, style_size, upsample=not small, fused=True),
        )
        self.out = nn.Sequential(_EqualConv1x1(dim // 16, output_dim), Marker("tanh"))
        emb = [Marker("pixel norm")]
        for _ in range(n_style_trans):
            emb += [Linear(style_size, style_size), Marker("leaky relu 0.2")]
        self.style_emb = nn.Sequential(*emb)
        self.gen = self.conv  # alias present in the reference's state-dict
        self._affine_bank = None
        self._style_chain = None
        # tape mode (set by the GAN trainer around its balanced lessons): the forward is recorded on an ops.Tape instead of autograd's graph
        # and the result is a leaf; the trainer collects the two or three gradients the lesson's loss groups leave on that leaf and sends
        # them through the generator in ONE backward pass (GenTape.backward_sets)
        self.tape_mode = False
        self.open_tapes = []

    def embed_style(self, style):
        h = ops.pixel_norm(style.contiguous())
        lin = [m for m in self.style_emb if isinstance(m, Linear)]
        # (the chain's backward kernel holds the whole batch in one workgroup: 16 rows; forward-only calls - generation - take any batch)
        if (h.shape[0] <= 16 or (not torch.is_grad_enabled() and ops.TAPE is None)) and h.shape[1] in (64, 128) and len(lin) <= 8:
            # the six Linear(128,128)+LeakyReLU layers run as one single-workgroup launch per direction (36 launches -> 2 per pass)
            if self._style_chain is None:
                self._style_chain = ops.MLPChain(lin, 0.2)
            return self._style_chain(h)
        for m in lin:
            h = ops.bias_act(ops.linear(h, m.weight, m.bias), None, None, ops.ACT_LRELU, 0.2)
        return h

    def forward(self, content, style, return_intermediate=False):
        with ops.scope("G"):
            if self.tape_mode and torch.is_grad_enabled() and not return_intermediate:
                return self._forward_taped(content, style)
            return self._forward(content, style, return_intermediate)

    def _forward_taped(self, content, style):
        if content.requires_grad:
            raise ops.L.HwgError("taped generator forward: the content rows carry no gradient in any lesson")
        tape = ops.Tape()
        s_in = tape.watch(style.detach())
        with ops.taping(tape), torch.no_grad():
            y = self._forward(content, s_in)
        y.requires_grad_(True)          # a leaf: the losses' backward passes stop here and leave their gradient in y.grad
        self.open_tapes.append(GenTape(tape, y, s_in, style if style.requires_grad else None))
        return y

    def take_tapes(self):
        tapes, self.open_tapes = self.open_tapes, []
        return tapes

    def _forward(self, content, style, return_intermediate=False):
        """content [T,B,n_class] (time major, as in the reference) or NHWC [B,1,T,n_class]; style [B,style]; -> NCHW [B,1,64,4T]"""
        if content.dim() == 3:
            T, B, C = content.shape
            twin = ops.nhwc_of(content) if not content.requires_grad else None      # (a one-hot made by ops.onehot_both carries its NHWC rows)
            if twin is not None:
                content = twin
            else:
                content = ops.permute4(content.contiguous(), (B, 1, T, C), (C, 0, B * C, 1)) if not content.requires_grad else \
                    ops.to_nhwc(content.permute(1, 2, 0).unsqueeze(2))
        B, _, T, _ = content.shape
        emb = self.embed_style(style)
        x = ops.cat_channels([content, emb], (B, 1, T)) if self.append_style else content
        # the ten style -> (gamma, beta) affines of the five blocks share their input: one bank launch instead of ten tiny GEMMs
        # (and one gradient kernel instead of ten data-gradient GEMMs + nine accumulations of d(emb))
        pairs = None
        if B <= 16 or (not torch.is_grad_enabled() and ops.TAPE is None):       # (backward: 16 rows at most - a taped forward runs under no_grad and IS followed by one; forward-only calls take any batch, 16 rows per block)
            if self._affine_bank is None:
                self._affine_bank = ops.LinearBank([m for blk in self.conv for m in (blk.adain1.style, blk.adain2.style)], halves=2)
            pairs = self._affine_bank(emb)
        shapes, hw = [], (1, T)
        for blk in self.conv:
            h_, w_, c_ = blk.out_shape(B, *hw)
            shapes += [(B, h_, w_, c_)] * 2
            hw = (h_, w_)
        noise = rng.NoiseBlock(shapes, x.device)
        for i, blk in enumerate(self.conv):
            x = blk(x, emb, None if pairs is None else (pairs[2 * i], pairs[2 * i + 1]), noise)
        y = ops.tanh(self.out[0](x))
        return ops.to_nchw(y)
