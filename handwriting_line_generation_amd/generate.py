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

from . import ops


def _begin(model, label, label_lengths, style):
    """enqueue the spacer of one request and start the device->host copy of its counts"""
    counts = model.spacer(model.onehot(label), style)
    return ops.AsyncFetch(counts)


def _render(model, label_host, label_lengths, style, fetch, device):
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
