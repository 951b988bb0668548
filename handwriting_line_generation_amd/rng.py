"""Randomness used inside the forward passes (generator noise, Dropout2d channel masks).

The reference draws these with torch.randn_like / F.dropout2d on whatever device the model lives on
(model/pure_gen.py:206,212; nn.Dropout2d sites). Two sources are provided:

* "device" (default): Philox kernels in libhwg_hip.so - nothing crosses PCIe.
* "host": draws from torch's CPU generator with exactly the shapes and order the reference uses on CPU
  (randn over the NCHW shape, bernoulli over [N,C,1,1]) and uploads; used by the parity tests so that a seeded
  reference run and a seeded run of this package see identical noise and masks.
"""
import torch

from . import ops

_state = {"mode": "device", "rng": None}


def set_mode(mode, seed=0):
    assert mode in ("device", "host")
    _state["mode"] = mode
    _state["rng"] = ops.DeviceRNG(seed) if mode == "device" else None


def mode():
    return _state["mode"]


def seed_process(base_seed, rank=0):
    """Entry points call this once: device Philox stream keyed (base_seed, rank) so that data-parallel ranks draw different
    generator noise / dropout masks, and torch / numpy / random host generators seeded per rank as well."""
    import random

    import numpy as np
    set_mode("device", seed=(int(base_seed) * 1000003 + int(rank)) & 0x7FFFFFFFFFFFFFFF)
    torch.manual_seed(int(base_seed) + rank)
    np.random.seed((int(base_seed) + rank) % (1 << 32))
    random.seed(int(base_seed) + rank)
    _state["base_seed"] = int(base_seed)


def get_state():
    """what a checkpoint keeps so that a resumed run continues the Philox stream instead of replaying it"""
    r = _state["rng"]
    return {"mode": _state["mode"], "base_seed": _state.get("base_seed"), "seed": r.seed if r is not None else None,
            "offset": r.offset if r is not None else 0, "text_offset": r.text_offset if r is not None else 0}


def set_state(st, rank=0):
    """restore get_state(); rank r re-derives its own seed from the base seed (the checkpoint is written by rank 0)"""
    if st.get("mode") != "device" or st.get("seed") is None:
        return
    base = st.get("base_seed")
    seed = st["seed"] if base is None else (int(base) * 1000003 + int(rank)) & 0x7FFFFFFFFFFFFFFF
    set_mode("device", seed=seed)
    _state["rng"].offset = int(st["offset"])
    _state["rng"].text_offset = int(st.get("text_offset", 0))
    _state["base_seed"] = base


def _dev_rng():
    if _state["rng"] is None:
        _state["rng"] = ops.DeviceRNG(0)
    return _state["rng"]


def device_rng():
    """the process's device generator (Philox stream; its (seed, offset) pair is part of a checkpoint)"""
    return _dev_rng()


def noise_like_nhwc(x):
    """standard normal tensor shaped like the NHWC tensor x"""
    N, H, W, C = x.shape
    if _state["mode"] == "host":
        z = torch.randn(N, C, H, W)  # same draw as torch.randn_like(out) on the reference's NCHW tensor
        return z.permute(0, 2, 3, 1).contiguous().to(x.device)
    return _dev_rng().randn((N, H, W, C), x.device)


class NoiseBlock:
    """the noise tensors of one generator forward pass. Device mode: ONE Philox launch over the total element count when the first tensor is
    asked for, later requests are views of that buffer (ten launches per forward pass were ten launches of 4-6 us). Host mode (parity
    tests): every request is its own draw from torch's CPU generator, in the reference's order and shapes."""

    def __init__(self, shapes, device):
        self.shapes, self.device, self.buf, self.k, self.off = list(shapes), device, None, 0, 0
        # forward-only pass with the device generator: the stream positions are reserved as for the one launch, but nothing is written -
        # requests return ops.VirtualNoise records and the consumer (ops.adain_epilogue) draws the same values inside its kernel
        # (a taped forward - ops.TAPE set - runs under no_grad too, but its backward pass reads the noise: not forward-only)
        self.virtual = _state["mode"] != "host" and not torch.is_grad_enabled() and ops.TAPE is None
        self.base = None

    def next(self, x):
        shape = tuple(x.shape)
        if _state["mode"] == "host" or self.k >= len(self.shapes) or tuple(self.shapes[self.k]) != shape:
            self.k = len(self.shapes)           # (a shape off the plan: fall back to single draws for the rest of the pass)
            return noise_like_nhwc(x)
        n = shape[0] * shape[1] * shape[2] * shape[3]
        if self.virtual:
            if self.base is None:
                total = sum(((a * b * c * d) + 3) // 4 * 4 for a, b, c, d in self.shapes)
                g = _dev_rng()
                self.base = (g.seed, g.offset)
                g.offset += total // 4
            out = ops.VirtualNoise(self.base[0], self.base[1] + self.off // 4, shape)
            self.off += (n + 3) // 4 * 4
            self.k += 1
            return out
        if self.buf is None:
            total = sum(((a * b * c * d) + 3) // 4 * 4 for a, b, c, d in self.shapes)
            self.buf = _dev_rng().randn((total,), self.device)
        out = self.buf[self.off: self.off + n].view(shape)
        self.off += (n + 3) // 4 * 4
        self.k += 1
        return out


def channel_mask(N, C, p, device):
    """feature-dropout mask [N,C] holding 0 or 1/(1-p) (what F.dropout2d multiplies by)"""
    if _state["mode"] == "host":
        m = torch.empty(N, C, 1, 1).bernoulli_(1 - p).div_(1 - p)
        return m.view(N, C).to(device)
    return _dev_rng().dropmask((N, C), p, device)


class MaskBlock:
    """the Dropout2d masks of one network pass. Device mode: ONE Philox launch for all of them when the first is asked for (they were one
    3 us launch each, up to six per discriminator pass), later requests are views; the values and the stream position afterwards are those
    of the single draws. Host mode (parity tests), or a request off the plan: single draws, in the reference's order."""

    def __init__(self, specs, device):
        self.specs, self.device, self.views, self.k = [(int(n), int(c), float(p)) for n, c, p in specs], device, None, 0

    def next(self, N, C, p):
        if _state["mode"] == "host" or self.k >= len(self.specs) or self.specs[self.k] != (int(N), int(C), float(p)) or any((n * c) % 4 for n, c, _ in self.specs):
            self.k = len(self.specs)
            return channel_mask(N, C, p, self.device)
        if self.views is None:
            self.views = _dev_rng().dropmask_multi([n * c for n, c, _ in self.specs], [q for _, _, q in self.specs], self.device)
        out = self.views[self.k].view(N, C)
        self.k += 1
        return out


def randn_host_shaped(shape, device):
    if _state["mode"] == "host":
        return torch.randn(*shape).to(device)
    return _dev_rng().randn(tuple(shape), device)


def element_mask(x, p):
    """elementwise dropout multiplier shaped like the NHWC tensor x (nn.Dropout sites, autoencoder.py:603-614)"""
    N, H, W, C = x.shape
    if _state["mode"] == "host":
        m = torch.empty(N, C, H, W).bernoulli_(1 - p).div_(1 - p)
        return m.permute(0, 2, 3, 1).contiguous().to(x.device)
    return _dev_rng().dropmask((N, H, W, C), p, x.device)
