"""`HWWithStyle`: the model container the trainer drives (reference: model/hw_with_style.py:81-337).

Same constructor config keys (including the substring-matched free-text options), same attributes
(`generator, discriminator, style_extractor, hwr, spacer, pred, spaced_label, counts, ...`) and the same
method surface (`forward, autoencode, extract_style, insert_spaces, onehot`), with every sub-network running on
the HIP kernels. `correct_pred` is the DTW alignment kernel instead of the reference's host double loop.
"""
import math
import os

import numpy as np
import torch

from .. import ops, rng
from ..logger import load_checkpoint
from ..base.base_model import BaseModel
from .char_style import CharStyleEncoder
from .cnn_only_hwr import CNNOnlyHWR
from .count_cnn import CountCNN
from .discriminator_ap import DiscriminatorAP
from .pure_gen import SpacedGenerator


def correct_pred(pred, label):
    """Optimal (DTW) alignment of the label to the recogniser output: int64 [T' >= T, B] (hw_with_style.py:18-74)."""
    out, _ = ops.dtw_align(pred.detach().contiguous(), label)
    return out


def correct_pred_async(pred, label):
    """enqueue the alignment now, collect it later with `.result()[0]` (lets the style extractor run in between)"""
    return ops.dtw_align_async(pred.detach().contiguous(), label)


class StyleTape:
    """one taped (recogniser -> style extractor) forward: `style` is the leaf the rest of the lesson sees"""

    def __init__(self, tape, style):
        self.tape, self.style = tape, style

    def backward(self, grad, target):
        """one loss group's style gradient through the style extractor and the recogniser behind it; its parameter gradients accumulate
        into `target` (ops.grad_set: None = the parameters' own gradients, or a stashed set's (buffer, mask))"""
        self.tape.backward_sets(self.style, [grad], [target])


class HWWithStyle(BaseModel):
    def __init__(self, config):
        super().__init__(config)
        g = config.get
        self.count_std = g("count_std", 0.1)
        self.dup_std = g("dup_std", 0.03)
        self.image_height = 64
        style_dim = g("style_dim", 256)
        dim = config["style_dim"] // 4 if "style_dim" in config else 64
        self.style_dim = style_dim
        self.char_style_dim = g("char_style_dim", 0)
        norm = g("style_norm", "none")
        activ = g("style_activ", "lrelu")
        pad_type = g("pad_type", "replicate")
        self.max_gen_length = g("max_gen_length", 500)
        self.num_class = num_class = config["num_class"]
        self.vae = False

        style_type = g("style", "normal")
        if "char" in style_type:
            dim = g("style_extractor_dim", dim)
            self.style_extractor = CharStyleEncoder(
                1, dim, style_dim, g("char_style_extractor_dim", dim * 2), self.char_style_dim, norm, activ, pad_type, num_class,
                global_pool=g("style_global_pool", False), average_found_char_style=config["average_found_char_style"],
                num_final_g_spacing_style=1, num_char_fc=1, vae=False, window=g("char_style_window", 6), small=False)
        else:
            self.style_extractor = None

        hwr_type = g("hwr", "CRNN")
        if "CNNOnly" in hwr_type:
            pad = "pad" in hwr_type
            if pad and "pad less" in hwr_type:
                pad = "less"
            self.hwr = CNNOnlyHWR(num_class, norm="group" if "group" in hwr_type else "batch", small="small" in hwr_type, pad=pad)
        elif "none" in hwr_type:
            self.hwr = None
        else:
            raise NotImplementedError("recogniser %r: only the CNN-only CTC recogniser is on the accelerated path" % hwr_type)
        self.hwr_frozen = False
        pre = g("pretrained_hwr")
        if pre is not None:
            if os.path.exists(pre):
                snap = load_checkpoint(pre)
                sd = {k[4:]: v for k, v in snap["state_dict"].items() if k.startswith("hwr.")} or snap["state_dict"]
                self.hwr.load_state_dict(sd)
            elif not g("RUN"):
                raise FileNotFoundError("Could not open pretrained HWR weights at " + pre)

        gen = g("generator")
        if gen == "none":
            self.generator = None
        elif gen is not None and "Pure" in gen:
            self.generator = SpacedGenerator(num_class, style_dim, g("gen_dim", 256), n_style_trans=g("n_style_trans", 6),
                                             emb_dropout=g("style_emb_dropout", False), append_style=g("gen_append_style", False),
                                             small="small" in gen)
        else:
            raise NotImplementedError("unknown generator: %r" % gen)

        if g("discriminator") is not None:
            d = config["discriminator"]
            self.discriminator = DiscriminatorAP(g("disc_dim", 64), use_low="use low" in d, use_med="no med" not in d, small="small" in d)

        if g("spacer"):
            self.count_duplicates = isinstance(config["spacer"], str) and "duplicate" in config["spacer"]
            self.spacer = CountCNN(num_class, style_dim, g("spacer_dim", 128), 2 if self.count_duplicates else 1)
        else:
            self.spacer = None

        self.create_mask = None
        self.style_from_normal = None
        self.guide_hwr = None
        self.style_discriminator = None
        self.use_hwr_pred_for_style = g("use_hwr_pred_for_style", True)
        self.pred = None
        self.spaced_label = None
        self.spaced_label_index = None
        self._dtw_pending = None
        self.spacing_pred = None
        self.mask_pred = None
        self.gen_spaced = None
        self.spaced_style = None
        self.counts = None

    # ------------------------------------------------------------------------------------------
    def forward(self, label, label_lengths, style, spaced=None):
        if spaced is None:
            label_onehot = self.onehot(label)
            self.counts = self.spacer(label_onehot, style)
            if rng.mode() == "device" and self.counts.is_cuda:
                # device generator: the expansion never leaves the GPU (only the expanded lengths are read back to size the result)
                idx, padded = self.insert_spaces_device(label, label_lengths, self.counts)
                spaced = self.onehot(idx)
            else:
                idx, padded = self.insert_spaces_index(label, label_lengths, self.counts)
                spaced = self.onehot(ops.h2d(idx.astype(np.int32), label.device))     # the one-hot is built on the device from [T,B] indices
            self.gen_padded = padded
            spaced = self._clip_spaced(spaced)
            self.gen_spaced = spaced
        return self.generator(spaced, style)

    def _clip_spaced(self, spaced):
        """trim blank columns so the generator sees at most max_gen_length steps (hw_with_style.py:241-261)"""
        if spaced.size(0) > self.max_gen_length:
            diff = self.max_gen_length - spaced.size(0)
            chars = spaced.argmax(2).cpu().numpy()
            x = spaced.size(0) - 1
            for x in range(spaced.size(0) - 1, 0, -1):  # last non-blank column (1 when there is none, like the reference's loop)
                if (chars[x] > 0).any():
                    break
            to_remove = min(diff, spaced.size(0) - x + 2)
            if to_remove > 0:
                spaced = spaced[:-to_remove]
        if spaced.size(0) > self.max_gen_length:
            diff = self.max_gen_length - spaced.size(0)
            chars = spaced.argmax(2).cpu().numpy()
            x = 0
            while x < spaced.size(0) - 1 and not (chars[x] > 0).any():
                x += 1
            to_remove = max(min(diff, x - 2), 0)
            if to_remove > 0:
                spaced = spaced[to_remove:]
        return spaced

    def autoencode(self, image, label, a_batch_size=None, stop_grad_extractor=False):
        style = self.extract_style(image, label, a_batch_size)
        if stop_grad_extractor:
            style = style.detach()
        if self.spaced_label is None:
            self.spaced_label_index = self.take_alignment(label)
            self.spaced_label = self.onehot(self.spaced_label_index)
        recon = self.forward(label, None, style, self.spaced_label)
        return recon, style

    def take_alignment(self, label):
        """DTW alignment of `label` to self.pred; uses the kernel launched ahead of time by extract_style when there is one"""
        pend, self._dtw_pending = self._dtw_pending, None
        if pend is not None:
            return pend.result()[0]
        return correct_pred(self.pred, label)

    # tape mode (set by the GAN trainer around its training lessons): recogniser-on-real-lines + style extractor are recorded on an
    # ops.Tape instead of autograd's graph and the style comes back as a leaf; the trainer sends each loss group's style gradient through
    # the tape itself (one pass per group, the passes on streams of their own - see HWWithStyleTrainer._style_backward)
    style_tape_mode = False

    def take_style_tapes(self):
        tapes, self.open_style_tapes = getattr(self, "open_style_tapes", []), []
        return tapes

    def extract_style(self, image, label, a_batch_size=None):
        if self.style_tape_mode and torch.is_grad_enabled() and self.pred is None and not image.requires_grad:
            tape = ops.Tape()
            with ops.taping(tape), torch.no_grad():
                style = self._extract_style(image, label, a_batch_size)
            tape._adopt_views([style])          # (the last op's result may come back as a reshape of the taped tensor)
            if id(style) in tape.live:          # (nothing on the path requires a gradient otherwise: frozen everything)
                style.requires_grad_(True)
                if not hasattr(self, "open_style_tapes"):
                    self.open_style_tapes = []
                self.open_style_tapes.append(StyleTape(tape, style))
            return style
        return self._extract_style(image, label, a_batch_size)

    def _extract_style(self, image, label, a_batch_size=None):
        if self.pred is None:
            self.pred = self.hwr(image, None)
            if label is not None and self.spaced_label is None and self.use_hwr_pred_for_style:
                # the alignment only depends on the recogniser output: start it now, its (small) result is fetched on a side stream
                # while the style extractor runs
                self._dtw_pending = correct_pred_async(self.pred, label)
        batch_size, feats, h, w = image.shape
        if a_batch_size is None:
            a_batch_size = batch_size
        if self.use_hwr_pred_for_style:
            spaced = self.pred                                   # [T,B,C]
        else:
            if self.spaced_label is None:
                self.spaced_label_index = correct_pred(self.pred, label)
                self.spaced_label = self.onehot(self.spaced_label_index)
            spaced = self.spaced_label
        T = spaced.shape[0]
        n_auth = batch_size // a_batch_size
        # lay the lines of one author side by side (image columns and recogniser time steps alike)
        if a_batch_size == 1:
            collapsed_image = image
        else:
            assert feats == 1
            collapsed_image = ops.permute(image.reshape(n_auth, a_batch_size, h, w), (0, 2, 1, 3)).reshape(n_auth, 1, h, a_batch_size * w)
        # [T,B,C] -> [B,T,C] -> [n_auth, A*T, C] -> NHWC rows [n_auth,1,A*T,C]
        collapsed_recog = ops.permute(spaced, (1, 0, 2)).reshape(n_auth, 1, a_batch_size * T, spaced.shape[2])
        style = self.style_extractor(collapsed_image, collapsed_recog)
        if a_batch_size == 1:
            return style
        return ops.repeat_rows(style, a_batch_size)

    def insert_spaces_index(self, label, label_lengths, counts):
        """expand text to per-column class indices using the predicted blank / duplicate counts: numpy noise and banker's rounding exactly
        as hw_with_style.py:302-328, but vectorised. The reference draws, per character, count ~ N(c0, count_std) then duplicates ~
        N(c1, dup_std) from numpy's global generator; one array-valued `np.random.normal` with the (mean, std) pairs interleaved in that
        order consumes the identical stream. -> (idx int64 [T,B] numpy, padded list)"""
        cn = counts.detach().cpu().numpy().astype(np.float64)
        lab = np.asarray(label.cpu().numpy())
        batch_size = lab.shape[1]
        max_count = max(math.ceil(float(cn.max())), 3)
        lens = [int(label_lengths[b]) for b in range(batch_size)]
        per = 2 if self.count_duplicates else 1
        locs = np.concatenate([cn[:n, b, :per].reshape(-1) for b, n in enumerate(lens)]) if sum(lens) else np.zeros(0)
        scales = np.tile(np.array([self.count_std, self.dup_std][:per], dtype=np.float64), sum(lens))
        draws = np.rint(np.random.normal(locs, scales)).astype(np.int64) if locs.size else np.zeros(0, dtype=np.int64)
        draws = np.maximum(draws, 0)       # a negative repeat count is an empty list in the reference's `[x] * n`
        lines = []
        pos = 0
        for b, n in enumerate(lens):
            d = draws[pos: pos + per * n]
            pos += per * n
            reps = np.empty(2 * n, dtype=np.int64)
            reps[0::2] = d[0::per]
            reps[1::2] = d[1::per] if self.count_duplicates else 1
            vals = np.zeros(2 * n, dtype=np.int64)
            vals[1::2] = lab[:n, b]
            lines.append(np.repeat(vals, reps))
        T = max(len(l) for l in lines) + max_count
        idx = np.zeros((T, batch_size), dtype=np.int64)
        padded = []
        for b, line in enumerate(lines):
            idx[: len(line), b] = line
            padded.append((T - len(line)) / T)
        return idx, padded

    def insert_spaces_device(self, label, label_lengths, counts, begin_only=False):
        """`insert_spaces_index` with the device generator (rng mode "device"): same expansion rule, draws from the Philox stream instead of
        numpy's global generator -> (idx int32 [T,B] on the GPU, padded list); `begin_only` returns the plan for `ops.DeviceRNG.insert_spaces_finish`
        (generation stream: the lengths travel to the host while the previous request renders)"""
        dev = counts.device
        lab = label if label.is_cuda else ops.h2d(label, dev)
        lens = ops.h2d(torch.as_tensor([int(n) for n in label_lengths], dtype=torch.int32), dev)
        plan = rng.device_rng().insert_spaces_begin(counts.detach(), lab.to(torch.int32).contiguous(), lens, self.count_std, self.dup_std,
                                                    self.count_duplicates)
        return plan if begin_only else rng.device_rng().insert_spaces_finish(plan)

    def insert_spaces(self, label, label_lengths, counts):
        """reference signature: (one-hot content [T,B,num_class] on the host, padded fractions)"""
        idx, padded = self.insert_spaces_index(label, label_lengths, counts)
        T, batch_size = idx.shape
        spaced = torch.zeros(T, batch_size, self.num_class)
        spaced.view(-1, self.num_class)[torch.arange(T * batch_size), torch.from_numpy(idx).view(-1)] = 1
        return spaced, padded

    def onehot(self, label):
        """[L,B] int labels -> [L,B,num_class] float one-hot on the labels' device"""
        if label.is_cuda:
            return ops.onehot_both(label.to(torch.int32).contiguous(), self.num_class)      # [L,B,C] (+ its NHWC twin for the networks)
        out = torch.zeros(label.size(0), label.size(1), self.num_class)
        out.view(-1, self.num_class)[torch.arange(label.numel()), label.reshape(-1).long()] = 1
        return out
