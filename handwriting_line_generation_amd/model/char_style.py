"""Character-specific style extractor on HIP kernels (reference: model/char_style.py:9-311, single-style mode).

Trunk: seven replicate-padded conv blocks collapse an author's concatenated lines [B',1,64,Wc] to a feature row
[B',1,Wc/4-2,256]. For every non-blank class found by arg-max over the recogniser's log-probs a 5-column window
(zero padded at the row ends) goes through that class's expert net; the expert outputs are averaged per author
weighted by exp(log-prob). A second branch (1-D convs over relu(features) ++ log-probs) gives a global vector and
a 2-layer MLP fuses both into the style vector.

The only host round trip is the arg-max map (B' x W' int32) that decides which experts run - the reference does a
`.item()` per found character here. Windows are listed in (class, author, column) order, which is the
reference's accumulation order, so the weighted sums round identically.
"""
import numpy as np
import torch
from torch import nn

from .. import ops
from . import expert_bank
from .layers import Conv1d, Conv2d, GroupNorm, Linear, Marker, group_count


class Conv2dBlock(nn.Module):
    """pad (replicate) -> conv -> GroupNorm -> ReLU; child names `conv` / `norm` as in the reference"""

    def __init__(self, input_dim, output_dim, kernel_size, stride, padding=0, norm="none", activation="relu", pad_type="replicate"):
        super().__init__()
        if pad_type not in ("replicate", "zero"):
            raise NotImplementedError("pad_type %r is not used by any shipped config" % pad_type)
        if isinstance(padding, int):
            padding = (padding,) * 4
        self.padding = tuple(padding)  # (left, right, top, bottom)
        self.pad_mode = "replicate" if pad_type == "replicate" else "constant"
        if norm == "group":
            self.norm = GroupNorm(group_count(output_dim), output_dim)
        elif norm == "none":
            self.norm = None
        else:
            raise NotImplementedError("norm %r is not used by any shipped config" % norm)
        if activation not in ("relu", "none", "lrelu"):
            raise NotImplementedError("activation %r is not used by any shipped config" % activation)
        self.activation = activation
        self.conv = Conv2d(input_dim, output_dim, kernel_size, stride)

    def forward(self, x):
        l, r, t, b = self.padding
        x = ops.pad2d(x, l, r, t, b, self.pad_mode)
        x = self.conv(x)
        slope = 0.2 if self.activation == "lrelu" else 0.0
        if self.norm is not None:
            return self.norm(x, self.activation, slope)
        if self.activation != "none":
            return ops.bias_act(x, None, None, ops.ACT_RELU if self.activation == "relu" else ops.ACT_LRELU, slope)
        return x


class CharExtractor(nn.Module):
    def __init__(self, input_dim, dim, style_dim, num_fc=1, small=False):
        super().__init__()
        if not small or num_fc != 1:
            raise NotImplementedError("only the window<3 ('small') single-fc expert of the shipped configs is built")
        self.conv1 = nn.Sequential(Marker("relu"), Conv1d(input_dim, dim, 3, padding=1), GroupNorm(group_count(dim), dim), Marker("relu"),
                                   Conv1d(dim, input_dim, 3, padding=1))
        self.conv2 = nn.Sequential(Marker("relu"), Conv1d(input_dim, 2 * dim, 1), GroupNorm(group_count(2 * dim), 2 * dim), Marker("relu"))
        self.fc = nn.Sequential(Linear(2 * dim, 2 * dim), Marker("relu"), Linear(2 * dim, style_dim))

    def forward(self, x):
        """x [n,1,5,C] windows -> [n, style_dim]"""
        n, _, wlen, _ = x.shape
        h = ops.relu(x)
        h = self.conv1[2](self.conv1[1](h), "relu")
        h = self.conv1[4](h)
        h = ops.relu(ops.add(h, x))
        h = self.conv2[2](self.conv2[1](h), "relu")
        h = ops.avg_pool2d(h, (1, wlen)).reshape(n, -1)
        h = ops.bias_act(ops.linear(h, self.fc[0].weight, self.fc[0].bias), None, None, ops.ACT_RELU)
        return self.fc[2](h)


class CharStyleEncoder(nn.Module):
    def __init__(self, input_dim, dim, style_dim, char_dim, char_style_dim, norm, activ, pad_type, n_class, global_pool=False,
                 average_found_char_style=0, num_final_g_spacing_style=1, num_char_fc=1, vae=False, window=6, small=False):
        super().__init__()
        if vae or char_style_dim > 0 or small or num_final_g_spacing_style != 1:
            raise NotImplementedError("only the single-style char-spec extractor of the shipped GAN configs is built (char_style_dim=0, no VAE)")
        self.n_class = n_class
        self.char_style_dim = style_dim
        self.single_style = True
        self.window = window
        down = [Conv2dBlock(input_dim, dim, 5, 1, 2, norm=norm, activation=activ, pad_type=pad_type)]
        for _ in range(2):
            down.append(Conv2dBlock(dim, 2 * dim, 4, 2, 1, norm=norm, activation=activ, pad_type=pad_type))
            dim *= 2
            down.append(Conv2dBlock(dim, dim, 3, 1, (1, 1, 0, 0), norm=norm, activation=activ, pad_type=pad_type))
        down.append(Conv2dBlock(dim, dim, 4, (2, 1), (1, 1, 0, 0), norm=norm, activation=activ, pad_type=pad_type))
        down.append(Conv2dBlock(dim, dim, 4, (2, 1), (1, 1, 0, 0), norm="none", activation="none", pad_type=pad_type))
        self.down = nn.Sequential(*down)
        self.feat_dim = dim
        self.prep = nn.Sequential(
            Conv1d(dim + n_class, dim, 5, 1, 2), Marker("relu"), Marker("maxpool1d 2"),
            Conv1d(dim, dim, 3, 1, 1), GroupNorm(group_count(dim), dim), Marker("relu"),
            Conv1d(dim, dim, 3, 1, 1), Marker("relu"))
        self.final_g_spacing_style = nn.Sequential(Linear(dim + style_dim, dim), Marker("relu"), Linear(dim, style_dim))
        self.char_extractor = nn.ModuleList([CharExtractor(dim, char_dim, style_dim, num_char_fc, window < 3) for _ in range(n_class)])
        self._bank = None   # pointer tables over the experts' parameters, built on first use (after the module sits on its device)

    @staticmethod
    def _align(x, recog):
        """replicate-pad the shorter of the feature row / log-prob row so both have the same length (char_style.py:198-202)"""
        diff = x.shape[2] - recog.shape[2]
        if diff > 0:
            recog = ops.pad2d(recog, diff // 2, diff // 2 + diff % 2, 0, 0, "replicate")
        elif diff < 0:
            d = -diff
            x = ops.pad2d(x, d // 2, d // 2 + d % 2, 0, 0, "replicate")
        return x, recog

    # workload counters (bench.py reports them: how many character windows / distinct experts a style extraction ran)
    stats = {"calls": 0, "windows": 0, "experts": 0}
    # Parity instrumentation (tests/test_pipeline_gpu.py): which character sits in which column is a DISCRETE decision - the arg-max of the
    # recogniser's log-probs - and a near-tie resolves differently under another fp32 summation order, which changes the loss graph itself
    # (another expert, another window). `last_argmax` keeps the map of the last call [B', T]; `forced_argmax` (same shape) overrides it.
    last_argmax = None
    forced_argmax = None

    def forward(self, x, recog):
        with ops.scope("StyleEx"):
            return self._forward(x, recog)

    def _trunk_eager(self, feat):
        for blk in self.down:
            feat = blk(feat)
        return feat

    def _trunk(self, feat):
        """the seven replicate-padded conv blocks: a pure function of the image geometry (no random draws, no host decisions), so its passes
        can be replayed from a recorded launch list (replay.py) - the experts behind it depend on the arg-max map and stay eager"""
        from .. import replay
        if replay.ENABLED:
            out = replay.forward(self.down, self._trunk_eager, "StyleEx", feat)
            if out is not None:
                return out
        return self._trunk_eager(feat)

    def _forward(self, x, recog):
        """x: NCHW [B',1,64,Wc] author image; recog: [B',n_class,Tc] log-probs (channel major, as the reference passes) or NHWC [B',1,Tc,n_class]"""
        B = x.shape[0]
        if recog.dim() == 3:
            recog = ops.permute(recog, (0, 2, 1)).unsqueeze(1)  # [B,1,T,n_class]
        # arg-max map of the recogniser output: computed and sent to the host (side stream) before the trunk convolutions are
        # enqueued, so that the later wait does not drain the main stream
        T0 = recog.shape[2]
        pred_fetch = ops.AsyncFetch(ops.argmax_rows(recog.reshape(B * T0, self.n_class)))
        feat = ops.to_nhwc(x)
        feat = self._trunk(feat)
        if feat.shape[1] != 1:
            raise ValueError("style extractor expects 64-pixel-high lines (feature height %d != 1)" % feat.shape[1])
        feat, recog = self._align(feat, recog)
        Wf = feat.shape[2]
        C = feat.shape[3]
        dev = feat.device

        # the branch that does not depend on the found characters goes first: the GPU works on it while the host waits for the
        # arg-max map and builds the window lists
        xr = ops.cat_channels([ops.relu(feat), recog], (B, 1, Wf))
        p = self.prep
        # (ReLU after the pool, riding along in the pooling kernels: relu(max(w)) == max(relu(w)) exactly, forward and backward)
        xr = ops.max_pool2d(ops.conv1d(xr, p[0].weight, p[0].bias, 1, 2, 1), (1, 2), (1, 2), relu=True)
        xr = p[4](p[3](xr), "relu")
        xr = ops.bias_act(ops.conv1d(xr, p[6].weight, p[6].bias, 1, 1, 1), None, None, ops.ACT_RELU)
        xr = ops.avg_pool2d(xr, (1, xr.shape[2])).reshape(B, -1)

        # which classes were recognised where (one small D2H copy)
        pred = pred_fetch.get().numpy().reshape(B, T0)
        if self.forced_argmax is not None:
            assert tuple(self.forced_argmax.shape) == (B, T0), "forced arg-max map has shape %s, expected %s" % (self.forced_argmax.shape, (B, T0))
            pred = np.asarray(self.forced_argmax).astype(pred.dtype)
        self.last_argmax = pred.copy()
        if Wf > T0:   # the log-probs were replicate-padded to the feature length: the arg-max map pads the same way
            d = Wf - T0
            pred = np.pad(pred, ((0, 0), (d // 2, d // 2 + d % 2)), mode="edge")
        bb, pp = np.nonzero(pred > 0)                 # row-major: author, then column
        st = self.stats
        st["calls"] += 1
        st["windows"] += int(bb.size)
        st["experts"] += int(np.unique(pred[bb, pp]).size) if bb.size else 0
        feat_rows = feat.reshape(B, Wf, C)
        if bb.size:
            cls_all = pred[bb, pp]
            order = np.argsort(cls_all, kind="stable")   # class-major, then (author, column): the reference's loop order
            cls_np = cls_all[order].astype(np.int32); b_np = bb[order].astype(np.int32); pos_np = pp[order].astype(np.int32)
            # one upload: window coordinates, run tables and the work-tile lists of both window lengths the experts use
            plan = expert_bank.make_plan(cls_np, dev, window_rows=(2 * self.window + 1, 1), extra=(b_np, pos_np))
            idx_b, idx_pos = plan["extra"]
            idx_cls = plan["eid"]
            patches = ops.gather_windows(feat_rows, idx_b, idx_pos, self.window)        # [n,1,2w+1,C]
            scores = ops.gather_scores(recog.reshape(B, Wf, self.n_class), idx_b, idx_pos, idx_cls)
            if self._bank is None:
                self._bank = expert_bank.ExpertBank(list(self.char_extractor))
            ex0 = self.char_extractor[0]
            char_styles = expert_bank.run_experts(self._bank, patches, plan, ex0.conv1[2].num_groups, ex0.conv2[2].num_groups)
            avg_char_style = ops.segment_weighted_mean(char_styles, scores, idx_b, B)
        else:
            avg_char_style = torch.zeros((B, self.char_style_dim), dtype=torch.float32, device=dev)

        comb = ops.cat_channels([xr.view(B, 1, 1, -1), avg_char_style.view(B, 1, 1, -1)], (B, 1, 1)).reshape(B, -1)
        f = self.final_g_spacing_style
        comb = ops.bias_act(ops.linear(comb, f[0].weight, f[0].bias), None, None, ops.ACT_RELU)
        return f[2](comb)
