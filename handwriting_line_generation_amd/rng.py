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


def _dev_rng():
    if _state["rng"] is None:
        _state["rng"] = ops.DeviceRNG(0)
    return _state["rng"]


def noise_like_nhwc(x):
    """standard normal tensor shaped like the NHWC tensor x"""
    N, H, W, C = x.shape
    if _state["mode"] == "host":
        z = torch.randn(N, C, H, W)  # same draw as torch.randn_like(out) on the reference's NCHW tensor
        return z.permute(0, 2, 3, 1).contiguous().to(x.device)
    return _dev_rng().randn((N, H, W, C), x.device)


def channel_mask(N, C, p, device):
    """feature-dropout mask [N,C] holding 0 or 1/(1-p) (what F.dropout2d multiplies by)"""
    if _state["mode"] == "host":
        m = torch.empty(N, C, 1, 1).bernoulli_(1 - p).div_(1 - p)
        return m.view(N, C).to(device)
    return _dev_rng().dropmask((N, C), p, device)


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
