"""ORACLE (test infrastructure only - never imported by the product path).

Plain PyTorch fp32 CPU restatement of the hot-path networks, written functionally over a state-dict so that
it shares no code with the product modules. The heavy arithmetic of the reference lives in PyTorch ATen
("third-party arithmetic"), so the restatement calls the same torch.nn.functional ops at the
same call sites the reference does; each function cites the reference lines it follows. It is pinned against
outputs of the reference itself: tests/golden/modules_*.npz produced by tools/gen_golden.py (imports
/root/reference in the build container) and checked by tests/test_oracle_golden.py.

Random draws (generator noise, dropout masks) come from torch's global CPU generator in the same order and shapes
as the reference's forward pass, so seeding the generator reproduces the reference's stochastic forward.
"""
import math
import zlib

import torch
import torch.nn.functional as F


def _groups(ch):  # utils/util.py:391-404
    return 8 if ch >= 32 else 4


class SD:
    """state-dict view with a key prefix"""

    def __init__(self, sd, prefix=""):
        self.sd, self.p = sd, prefix

    def __call__(self, key):
        return self.sd[self.p + key]

    def has(self, key):
        return (self.p + key) in self.sd

    def sub(self, prefix):
        return SD(self.sd, self.p + prefix)


# ---------------------------------------------------------------- generator (model/pure_gen.py)
def _adain(x, style, s, C):  # pure_gen.py:52-69
    gb = F.linear(style, s("style.weight"), s("style.bias"))
    gamma, beta = gb[:, :C, None, None], gb[:, C:, None, None]
    return gamma * F.instance_norm(x, eps=1e-5) + beta


def _noise(x, s):  # pure_gen.py:72-79 + EqualLR :218-247
    w = s("weight_orig")
    return x + (w * math.sqrt(2.0 / w.shape[1])) * torch.randn_like(x)


def _blur(x):  # pure_gen.py:80-137
    C = x.shape[1]
    k = torch.tensor([[1., 2., 1.], [2., 4., 2.], [1., 2., 1.]])
    k = (k / k.sum()).view(1, 1, 3, 3).repeat(C, 1, 1, 1).to(x.dtype)
    return F.conv2d(x, k, padding=1, groups=C)


def _styled_block(x, style, s, kind):  # pure_gen.py:140-216
    if kind == "initial":
        h = F.conv_transpose2d(x, s("conv1.weight"), s("conv1.bias"), padding=(0, 1))
    elif kind == "up":
        h = F.interpolate(x, scale_factor=(2, 1), mode="nearest")
        h = _blur(F.conv2d(h, s("conv1.1.weight"), s("conv1.1.bias"), padding=1))
    else:  # fused upsample, pure_gen.py:250-279
        w = s("conv1.0.weight")
        w = F.pad(w * math.sqrt(2.0 / (w.shape[0] * 9)), [1, 1, 1, 1])
        w = (w[:, :, 1:, 1:] + w[:, :, :-1, 1:] + w[:, :, 1:, :-1] + w[:, :, :-1, :-1]) / 4
        h = _blur(F.conv_transpose2d(x, w, s("conv1.0.bias"), stride=2, padding=1))
    C = h.shape[1]
    h = _adain(F.leaky_relu(_noise(h, s.sub("noise1.")), 0.2), style, s.sub("adain1."), C)
    h = F.conv2d(h, s("conv2.weight"), s("conv2.bias"), padding=1)
    return _adain(F.leaky_relu(_noise(h, s.sub("noise2.")), 0.2), style, s.sub("adain2."), C)


def generator(sd, content, style, prefix=""):
    """content [T,B,n_class], style [B,S] -> [B,1,64,4T]  (pure_gen.py:42-50)"""
    s = SD(sd, prefix)
    x = content.permute(1, 2, 0).unsqueeze(2)
    e = style / torch.sqrt(torch.mean(style ** 2, dim=1, keepdim=True) + 1e-8)
    for i in range(1, 12, 2):
        e = F.leaky_relu(F.linear(e, s("style_emb.%d.weight" % i), s("style_emb.%d.bias" % i)), 0.2)
    x = torch.cat((x, e[:, :, None, None].expand(-1, -1, 1, x.size(3))), dim=1)
    for i, kind in enumerate(("initial", "up", "up", "fused", "fused")):
        x = _styled_block(x, e, s.sub("conv.%d." % i), kind)
    w = s("out.0.conv.weight_orig")
    return torch.tanh(F.conv2d(x, w * math.sqrt(2.0 / w.shape[1]), s("out.0.conv.bias")))


# ---------------------------------------------------------------- discriminator (model/discriminator_ap.py)
def _sn_conv(x, s, padding, update=True):  # discriminator_ap.py:20-32
    u, v, w = s("module.weight_u"), s("module.weight_v"), s("module.weight_bar")
    wm = w.reshape(w.shape[0], -1)
    if update:
        with torch.no_grad():
            v.copy_(_l2(torch.mv(wm.t(), u)))
            u.copy_(_l2(torch.mv(wm, v)))
    sigma = u.dot(wm.mv(v))
    return F.conv2d(x, w / sigma, s("module.bias"), padding=padding)


def _l2(v, eps=1e-12):
    return v / (v.norm() + eps)


def _drop2d(x, p, training):
    return F.dropout2d(x, p, training)


def discriminator(sd, x, prefix="", training=True):
    """x [N,1,64,W] -> [pM [N,W/8], pL [N,W/32]]  (discriminator_ap.py:139-161)"""
    s = SD(sd, prefix)
    lk = 0.1
    h = F.conv2d(x, s("in_conv.0.weight"), s("in_conv.0.bias"), padding=(0, 3))
    h = F.leaky_relu(F.group_norm(h, _groups(h.shape[1]), s("in_conv.1.weight"), s("in_conv.1.bias")), lk)
    h = F.avg_pool2d(F.leaky_relu(_sn_conv(h, s.sub("convs1.0."), (0, 1)), lk), 2)
    h = F.leaky_relu(_drop2d(_sn_conv(h, s.sub("convs1.3."), (0, 1)), 0.05, training), lk)
    h = F.avg_pool2d(F.leaky_relu(_sn_conv(h, s.sub("convs2.0."), (0, 1)), lk), 2)
    h = F.conv2d(h, s("convs3.0.weight"), s("convs3.0.bias"), padding=(0, 1))
    h = F.avg_pool2d(F.leaky_relu(F.group_norm(h, _groups(h.shape[1]), s("convs3.1.weight"), s("convs3.1.bias")), lk), 2)
    mL = F.leaky_relu(_drop2d(_sn_conv(h, s.sub("convs3.4."), (0, 1)), 0.05, training), lk)
    pM = _sn_conv(mL, s.sub("finalMed.0."), (0, 1))
    h = F.leaky_relu(_drop2d(_sn_conv(mL, s.sub("convs4.0."), (0, 1)), 0.025, training), lk)
    h = F.avg_pool2d(h, (1, 2))
    h = F.leaky_relu(_drop2d(_sn_conv(h, s.sub("convs4.4."), (0, 1)), 0.025, training), lk)
    h = F.leaky_relu(_drop2d(_sn_conv(h, s.sub("convs4.7."), (0, 1)), 0.025, training), lk)
    h = F.avg_pool2d(h, (1, 2))
    h = F.leaky_relu(_drop2d(_sn_conv(h, s.sub("convs4.11."), (0, 1)), 0.025, training), lk)
    pL = _sn_conv(h, s.sub("convs4.14."), (0, 0))
    n = x.shape[0]
    return [pM.reshape(n, -1), pL.reshape(n, -1)]


# ---------------------------------------------------------------- recogniser (model/cnn_only_hwr.py)
def _bn(x, s, training):
    return F.batch_norm(x, s("running_mean"), s("running_var"), s("weight"), s("bias"), training, 0.1, 1e-5)


def hwr(sd, image, prefix="", training=True):
    """image [B,1,64,W] -> log-probs [T,B,n_class]  (cnn_only_hwr.py:96-107); BatchNorm in batch-statistics mode when training"""
    s = SD(sd, prefix)
    pads = [1, 1, 1, 1, 1, 0, 0]
    x = image
    for i in range(7):
        x = F.conv2d(x, s("cnn.conv%d.weight" % i), s("cnn.conv%d.bias" % i), padding=pads[i])
        if i in (2, 4, 6):
            x = _bn(x, s.sub("cnn.batchnorm%d." % i), training)
        x = F.relu(x)
        if i in (0, 1):
            x = F.max_pool2d(x, 2, 2)
        elif i in (3, 5):
            x = F.max_pool2d(x, (2, 2), (2, 1), (0, 1))
    b, c, h, w = x.shape
    x = x.reshape(b, -1, w)
    for k, (dil, pad) in zip((0, 3, 6, 9), ((2, 2), (4, 4), (1, 0), (8, 8))):
        x = F.conv1d(x, s("cnn1d.%d.weight" % k), s("cnn1d.%d.bias" % k), padding=pad, dilation=dil)
        x = F.relu(_bn(x, s.sub("cnn1d.%d." % (k + 1)), training))
    x = F.conv1d(x, s("cnn1d.12.weight"), s("cnn1d.12.bias"))
    return F.log_softmax(x, dim=1).permute(2, 0, 1)


# ---------------------------------------------------------------- spacer (model/count_cnn.py)
def spacer(sd, onehot, style, prefix="", training=True):
    """onehot [L,B,C], style [B,S] -> [L,B,2]  (count_cnn.py:34-44)"""
    s = SD(sd, prefix)
    x = torch.cat((onehot.permute(1, 2, 0), style[..., None].expand(-1, -1, onehot.size(0))), dim=1)
    for k, drop in ((0, True), (4, True), (8, False)):
        x = F.conv1d(x, s("cnn.%d.weight" % k), s("cnn.%d.bias" % k), padding=1)
        x = F.group_norm(x, _groups(x.shape[1]), s("cnn.%d.weight" % (k + 1)), s("cnn.%d.bias" % (k + 1)))
        if drop:
            x = F.dropout2d(x, 0.1, training)
        x = F.relu(x)
    x = F.conv1d(x, s("cnn.11.weight"), s("cnn.11.bias"))
    return x.permute(2, 0, 1) * s("std") + s("mean")


# ---------------------------------------------------------------- style extractor (model/char_style.py)
def _char_expert(x, s):  # char_style.py:84-124, window < 3 variant
    res = x
    h = F.conv1d(F.relu(x), s("conv1.1.weight"), s("conv1.1.bias"), padding=1)
    h = F.relu(F.group_norm(h, _groups(h.shape[1]), s("conv1.2.weight"), s("conv1.2.bias")))
    h = F.conv1d(h, s("conv1.4.weight"), s("conv1.4.bias"), padding=1)
    h = F.conv1d(F.relu(h + res), s("conv2.1.weight"), s("conv2.1.bias"))
    h = F.relu(F.group_norm(h, _groups(h.shape[1]), s("conv2.2.weight"), s("conv2.2.bias")))
    h = F.adaptive_avg_pool1d(h, 1).view(x.size(0), -1)
    return F.linear(F.relu(F.linear(h, s("fc.0.weight"), s("fc.0.bias"))), s("fc.2.weight"), s("fc.2.bias"))


def style_extractor(sd, x, recog, prefix="", n_class=80, window=2):
    """x [B,1,64,W] author image, recog [B,n_class,T] log-probs -> style [B,S]  (char_style.py:193-297, single style)"""
    s = SD(sd, prefix)
    pads = [(2, 2, 2, 2), (1, 1, 1, 1), (1, 1, 0, 0), (1, 1, 1, 1), (1, 1, 0, 0), (1, 1, 0, 0), (1, 1, 0, 0)]
    strides = [1, 2, 1, 2, 1, (2, 1), (2, 1)]
    for i in range(7):
        x = F.conv2d(F.pad(x, pads[i], mode="replicate"), s("down.%d.conv.weight" % i), s("down.%d.conv.bias" % i), stride=strides[i])
        if i < 6:
            x = F.relu(F.group_norm(x, _groups(x.shape[1]), s("down.%d.norm.weight" % i), s("down.%d.norm.bias" % i)))
    B = x.size(0)
    x = x.view(B, x.size(1), x.size(3))
    diff = x.size(2) - recog.size(2)
    if diff > 0:
        recog = F.pad(recog, (diff // 2, diff // 2 + diff % 2), mode="replicate")
    elif diff < 0:
        x = F.pad(x, ((-diff) // 2, (-diff) // 2 + (-diff) % 2), mode="replicate")
    pred = torch.argmax(recog, dim=1)
    sdim = s("final_g_spacing_style.2.weight").shape[0]
    total = torch.zeros(B, sdim)
    wsum = torch.zeros(B)
    for c in range(1, n_class):
        where = (pred == c).nonzero()
        if where.numel() == 0:
            continue
        patches, owners = [], []
        for b, pos in where.tolist():
            lo, hi = pos - window, pos + window
            w_ = x[b:b + 1, :, max(lo, 0):min(hi, x.size(2) - 1) + 1]
            w_ = F.pad(w_, (max(0, -lo), max(0, hi - (x.size(2) - 1))))
            patches.append(w_)
            owners.append((b, math.exp(recog[b, c, pos])))
        styles = _char_expert(torch.cat(patches, dim=0), s.sub("char_extractor.%d." % c))
        for i, (b, sc) in enumerate(owners):
            total[b] = total[b] + sc * styles[i]
            wsum[b] = wsum[b] + sc
    avg = torch.where(wsum[..., None] != 0, total / wsum[..., None], total)
    xr = torch.cat((F.relu(x), recog), dim=1)
    xr = F.max_pool1d(F.relu(F.conv1d(xr, s("prep.0.weight"), s("prep.0.bias"), padding=2)), 2, 2)
    xr = F.conv1d(xr, s("prep.3.weight"), s("prep.3.bias"), padding=1)
    xr = F.relu(F.group_norm(xr, _groups(xr.shape[1]), s("prep.4.weight"), s("prep.4.bias")))
    xr = F.relu(F.conv1d(xr, s("prep.6.weight"), s("prep.6.bias"), padding=1))
    xr = F.adaptive_avg_pool1d(xr, 1).view(B, -1)
    comb = torch.cat((xr, avg), dim=1)
    comb = F.relu(F.linear(comb, s("final_g_spacing_style.0.weight"), s("final_g_spacing_style.0.bias")))
    return F.linear(comb, s("final_g_spacing_style.2.weight"), s("final_g_spacing_style.2.bias"))


# ---------------------------------------------------------------- Encoder2 / decoder / E_HWR (model/autoencoder.py)
def _gn(x, s, key):
    return F.group_norm(x, _groups(x.shape[1]), s(key + ".weight"), s(key + ".bias"))


def encoder2(sd, x, prefix="", training=True):
    """x [N,1,64,W] -> (code, mid_features)  (autoencoder.py:341-410; the residual of conv1 is the in-place-ReLU'd tensor)"""
    s = SD(sd, prefix)
    h = F.relu(_gn(F.conv2d(x, s("down_conv1.0.weight"), s("down_conv1.0.bias"), padding=2), s, "down_conv1.1"))
    h = F.conv2d(F.avg_pool2d(h, 2), s("down_conv1.4.weight"), s("down_conv1.4.bias"))
    r = F.relu(h)
    h = _gn(F.conv2d(r, s("conv1.1.weight"), s("conv1.1.bias"), padding=1), s, "conv1.2")
    h = F.conv2d(F.relu(F.dropout2d(h, 0.1, training)), s("conv1.5.weight"), s("conv1.5.bias"), padding=1) + r
    h = F.conv2d(F.avg_pool2d(F.relu(_gn(h, s, "down_conv2.0")), 2), s("down_conv2.3.weight"), s("down_conv2.3.bias"))
    res = h
    h = F.relu(F.dropout2d(_gn(h, s, "conv2.0"), 0.1, training))
    h = _gn(F.conv2d(h, s("conv2.3.weight"), s("conv2.3.bias"), padding=1), s, "conv2.4")
    h = F.conv2d(F.relu(F.dropout2d(h, 0.1, training)), s("conv2.7.weight"), s("conv2.7.bias"), padding=1) + res
    mid = h
    h = F.conv2d(F.avg_pool2d(F.relu(_gn(h, s, "down_conv3.0")), 2), s("down_conv3.3.weight"), s("down_conv3.3.bias"))
    h = F.relu(F.dropout2d(_gn(h, s, "down_conv3.4"), 0.1, training))
    return F.conv2d(h, s("down_conv3.7.weight"), s("down_conv3.7.bias")), mid


def decoder_noskip(sd, x, prefix=""):
    """autoencoder.py:302-339"""
    s = SD(sd, prefix)
    spec = [(1, 1, 0), (4, 1, 0), (7, 2, 1), (10, 1, 1), (13, 2, 1), (16, 1, 1), (19, 2, 1)]
    h = F.relu(x)
    for k, st, pd in spec:
        h = F.conv_transpose2d(h, s("up_conv1.%d.weight" % k), s("up_conv1.%d.bias" % k), stride=st, padding=pd)
        h = F.relu(_gn(h, s, "up_conv1.%d" % (k + 1)))
    return torch.tanh(F.conv_transpose2d(h, s("up_conv1.22.weight"), s("up_conv1.22.bias"), padding=1))


def e_hwr(sd, x, prefix="", training=True):
    """autoencoder.py:596-628"""
    s = SD(sd, prefix)
    h = x.view(x.size(0), x.size(1), x.size(3))
    for k, (pad, dil) in zip((0, 4, 8, 12), ((1, 1), (2, 2), (4, 4), (2, 1))):
        h = _gn(F.conv1d(h, s("classify.%d.weight" % k), s("classify.%d.bias" % k), padding=pad, dilation=dil), s, "classify.%d" % (k + 1))
        h = F.relu(F.dropout(h, 0.1, training))
    h = F.conv1d(h, s("classify.16.weight"), s("classify.16.bias"))
    return F.log_softmax(h, dim=1).permute(2, 0, 1)


# ---------------------------------------------------------------- deterministic parameter fill shared by fixtures and tests
def alias_of(k):
    return k.replace("gen.", "conv.", 1) if (k.startswith("gen.") or ".gen." in k) else k


def seeded_state_dict(module, seed):
    """State-dict for `module` filled from per-entry seeded generators. Depends only on (key names, shapes, seed), so the
    reference module in the build container and the product / oracle on the GPU box get identical weights without shipping them."""
    params = dict(module.named_parameters())
    sd = module.state_dict()
    out = {}
    for i, (k, v) in enumerate(sd.items()):
        g = torch.Generator().manual_seed(seed * 100003 + zlib.crc32(alias_of(k).encode()) % 1000003)
        leaf = k.rsplit(".", 1)[-1]
        alias = alias_of(k)
        if alias != k and alias in out:
            out[k] = out[alias]  # generator.gen is the same Sequential as generator.conv
            continue
        if not v.dtype.is_floating_point:
            out[k] = v.clone()
        elif k not in params:  # buffers
            if leaf == "running_mean":
                out[k] = 0.1 * torch.randn(v.shape, generator=g)
            elif leaf == "running_var":
                out[k] = 0.5 + torch.rand(v.shape, generator=g)
            else:
                out[k] = v.clone()  # constant blur kernels
        elif leaf in ("weight_u", "weight_v"):
            t = torch.randn(v.shape, generator=g)
            out[k] = t / t.norm()
        elif leaf == "weight_orig" and "noise" in k:
            out[k] = 0.5 * torch.randn(v.shape, generator=g)
        elif leaf == "weight_orig" or k.endswith("conv1.0.weight"):
            out[k] = torch.randn(v.shape, generator=g)  # equal-lr / FusedUpsample weights are N(0,1) by construction
        elif leaf == "std":
            out[k] = 1.0 + 0.2 * torch.rand(v.shape, generator=g)
        elif v.dim() <= 1:
            out[k] = (1.0 + 0.2 * torch.randn(v.shape, generator=g)) if leaf == "weight" else 0.1 * torch.randn(v.shape, generator=g)
        else:
            out[k] = torch.randn(v.shape, generator=g) / math.sqrt(v[0].numel())
    return out
