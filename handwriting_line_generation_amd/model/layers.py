"""NHWC layer modules backed by the hwg HIP kernels.

Each class subclasses the torch.nn layer the reference uses so that parameter names, shapes, default
initialisation and state-dict entries are identical (released checkpoints load unchanged), but `forward`
consumes/produces NHWC tensors and calls into libhwg_hip.so. 1-D layers treat [N,1,L,C] as their layout.
"""
import torch
from torch import nn

from .. import ops, rng

ACT = {"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "lrelu": ops.ACT_LRELU, "tanh": ops.ACT_TANH}


class Conv2d(nn.Conv2d):
    def forward(self, x):
        return ops.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation)


class ConvTranspose2d(nn.ConvTranspose2d):
    def forward(self, x):
        return ops.conv_transpose2d(x, self.weight, self.bias, self.stride, self.padding, self.output_padding, self.dilation)


class Conv1d(nn.Conv1d):
    def forward(self, x):
        return ops.conv1d(x, self.weight, self.bias, self.stride[0], self.padding[0], self.dilation[0])


class Linear(nn.Linear):
    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)


class GroupNorm(nn.GroupNorm):
    """GroupNorm with the following Dropout2d mask and activation fused into the same kernels."""

    def forward(self, x, act="none", slope=0.0, mask=None):
        return ops.group_norm(x, self.num_groups, self.weight, self.bias, self.eps, mask, ACT[act], slope)


class _BatchNormBase:
    """`num_batches_tracked` (an int64 buffer nothing on the training path reads: momentum is fixed) is counted on the host and folded into the
    buffer when the state dict is taken or loaded - the reference's `+= 1` on the device was one launch per BatchNorm layer and forward pass
    (seven per recogniser forward)."""

    def _fwd(self, x, act, slope):
        if self.training:
            y = ops.batch_norm_train(x, self.weight, self.bias, self.running_mean, self.running_var, self.momentum, self.eps, ACT[act], slope)
            self._tracked_pending = getattr(self, "_tracked_pending", 0) + 1
            return y
        return ops.norm_apply_frozen(x, self.running_mean, self.running_var, self.eps, self.weight, self.bias, ACT[act], slope)

    def flush_tracked(self):
        n = getattr(self, "_tracked_pending", 0)
        if n and self.num_batches_tracked is not None:
            self.num_batches_tracked += n
        self._tracked_pending = 0

    def state_dict(self, *args, **kwargs):
        self.flush_tracked()
        return super().state_dict(*args, **kwargs)

    def _save_to_state_dict(self, destination, prefix, keep_vars):      # (a parent's state_dict() reaches the layer here)
        self.flush_tracked()
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, *args, **kwargs):
        self._tracked_pending = 0
        super()._load_from_state_dict(*args, **kwargs)


class BatchNorm2d(_BatchNormBase, nn.BatchNorm2d):
    def forward(self, x, act="none", slope=0.0):
        return self._fwd(x, act, slope)


class BatchNorm1d(_BatchNormBase, nn.BatchNorm1d):
    def forward(self, x, act="none", slope=0.0):
        return self._fwd(x, act, slope)


class Dropout2d(nn.Module):
    """Channel dropout marker: produces the [N,C] multiplier that the neighbouring fused kernel applies."""

    def __init__(self, p, inplace=False):
        super().__init__()
        self.p = p

    def mask_for(self, x, block=None):
        return self.mask_for_shape(x.shape[0], x.shape[-1], x.device, block)

    def mask_for_shape(self, N, C, device, block=None):
        """block: the pass's rng.MaskBlock (all Dropout2d masks of a network pass from one launch) or None (a draw of its own)"""
        if not self.training or self.p == 0:
            return None
        if block is not None:
            return block.next(N, C, self.p)
        return rng.channel_mask(N, C, self.p, device)

    def spec(self, N, C):
        """this layer's entry in a rng.MaskBlock plan, or None when it draws nothing (eval mode)"""
        return (N, C, self.p) if (self.training and self.p != 0) else None

    def forward(self, x):
        m = self.mask_for(x)
        return x if m is None else ops.bias_act(x, None, m)


class Marker(nn.Module):
    """Placeholder keeping nn.Sequential indices aligned with the reference (activations, pools, pads)."""

    def __init__(self, what=""):
        super().__init__()
        self.what = what

    def extra_repr(self):
        return self.what

    def forward(self, x):
        raise RuntimeError("Marker modules only reserve a Sequential slot; the owning module runs the fused op")


class BlurBuffers(nn.Module):
    """Holds the constant `weight` / `weight_flip` buffers of the reference's Blur module (model/pure_gen.py:120-134)
    for state-dict compatibility; the blur itself is the hwg_blur3 kernel."""

    def __init__(self, channel):
        super().__init__()
        k = torch.tensor([[1., 2., 1.], [2., 4., 2.], [1., 2., 1.]])
        k = (k / k.sum()).view(1, 1, 3, 3)
        self.register_buffer("weight", k.repeat(channel, 1, 1, 1))
        self.register_buffer("weight_flip", torch.flip(k, [2, 3]).repeat(channel, 1, 1, 1))

    def forward(self, x):
        return ops.blur3(x)


def group_count(channels):
    """utils/util.py:391-404 (`getGroupSize`): 8 groups from 32 channels up, else 4; channel counts must divide."""
    goal = 8 if channels >= 32 else 4
    if channels % goal != 0:
        raise ValueError("channel count %d not divisible by %d groups (the reference's fallback is undefined here too)" % (channels, goal))
    return goal
