"""Batched generation loop (reference: generate.py:48-83 `generate()` -> `model(label, label_lengths, style)`).

`HWWithStyle.forward` has one host round trip in the middle: the spacer's predicted blank / duplicate counts go to the host, where
`insert_spaces` expands the text with numpy noise, and the expanded content comes back for the generator. Called request by request
the GPU idles during that round trip. `generate_stream` overlaps it: the spacer of request i+1 is enqueued (and its counts start
travelling to the host on the copy stream) before request i is rendered, so the host never waits for the GPU and the GPU always has
the next generator pass queued. Per request the kernels, their order within the request and the RNG draws are those of
`model(label, label_lengths, style)`; with the reference's host RNG the outputs are identical to calling the model in sequence.
"""
import numpy as np
import torch

from . import ops, rng
from .utils import string_utils


def _begin(model, label, label_lengths, style):
    """enqueue the spacer of one request and start the device->host copy of its counts"""
    counts = model.spacer(model.onehot(label), style)
    if rng.mode() == "device":      # device generator: the expansion plan is drawn on the GPU, only the expanded lengths come back
        return model.insert_spaces_device(label, label_lengths, counts, begin_only=True)
    return ops.AsyncFetch(counts)


def _render(model, label_host, label_lengths, style, fetch, device):
    if isinstance(fetch, tuple):
        idx, padded = rng.device_rng().insert_spaces_finish(fetch)
        spaced = model.onehot(idx)
    else:
        counts = fetch.get()
        idx, padded = model.insert_spaces_index(label_host, label_lengths, counts)
        spaced = model.onehot(ops.h2d(idx.astype(np.int32), device))
    spaced = model._clip_spaced(spaced)
    return model.generator(spaced, style), padded


def generate_stream(model, requests, device=None):
    """requests: iterable of (label [L,B] int tensor (host or device), label_lengths, style [B,style_dim] device tensor).
    Yields (image NCHW [B,1,64,W], padded fractions) per request, in order. Inference only (call under torch.no_grad())."""
    pending = None
    for label, label_lengths, style in requests:
        device = device or style.device
        label_host = label.cpu() if label.is_cuda else label
        label_dev = label if label.is_cuda else ops.h2d(label, device)
        fetch = _begin(model, label_dev, label_lengths, style)
        if pending is not None:
            yield _render(model, *pending, device)
        pending = (label_host, label_lengths, style, fetch)
    if pending is not None:
        yield _render(model, *pending, device)


# ---------------------------------------------------------------------------------------------------------------------------------
# the reference's generation helpers (generate.py:48-83, 796-828), same names and arguments
def get_style(config, model, instance, gpu=None):
    """style vector(s) of the lines in `instance`. With `trainer.style_together` (how the shipped character-style models are used) all
    lines of one author are laid side by side - image columns and recogniser time steps alike - and go through the style extractor once:
    recogniser -> (its own log-probs | DTW alignment of the text, one-hot) -> collapse per author -> CharStyleEncoder; returns
    [authors, style_dim]. Without it the extractor sees the raw batch and the first line's style is returned (generate.py:66-67)."""
    if "lookup" in config["model"].get("style", "") or "Lookup" in config["model"].get("style", ""):
        raise NotImplementedError("author-lookup styles are not used by any shipped config")
    tr = config.get("trainer", {})
    image, label = instance["image"], instance["label"]
    if gpu is not None:
        image = ops.h2d(image, gpu) if not image.is_cuda else image
        label = ops.h2d(label, gpu) if not label.is_cuda else label
    if not tr.get("style_together", False):
        style = model.style_extractor(image)          # (a character-style extractor needs the recogniser output: TypeError, as in the reference)
        return style[0:1]
    old = model.use_hwr_pred_for_style
    model.use_hwr_pred_for_style = bool(tr.get("use_hwr_pred_for_style", False))
    try:
        model.pred = model.spaced_label = model.spaced_label_index = None
        a_batch_size = instance.get("a_batch_size", image.shape[0])
        style = model.extract_style(image, label, a_batch_size)      # [B, style_dim], every author's style repeated for its lines
        return style[::a_batch_size].contiguous()
    finally:
        model.use_hwr_pred_for_style = old
        model.pred = model.spaced_label = model.spaced_label_index = None


def _text_label(text, char_to_idx, batch_size, gpu):
    label = string_utils.str2label_single(text, char_to_idx)
    label = torch.from_numpy(label.astype(np.int32))[:, None].expand(-1, batch_size).contiguous()
    return ops.h2d(label, gpu)


def generate(model, style, text, char_to_idx, gpu):
    """one image of `text` in `style` ([1, style_dim]) -> NCHW [1,1,64,W]"""
    label = _text_label(text, char_to_idx, 1, gpu)
    return model(label, torch.IntTensor(1).fill_(label.size(0)), style)


def interpolate(model, style1, style2, text, char_to_idx, gpu, step=0.05):
    """images of `text` while the style moves from style1 to style2 in steps of `step` -> (list of images, list of styles on the host)"""
    batch_size = style1.size(0)
    label = _text_label(text, char_to_idx, batch_size, gpu)
    label_len = torch.IntTensor(batch_size).fill_(len(text))
    results, styles = [], []
    for alpha in np.arange(0, 1.0, step):
        style = (style2 * float(alpha) + float(1 - alpha) * style1).contiguous()
        results.append(model(label, label_len, style))
        styles.append(style.cpu().detach())
    return results, styles
