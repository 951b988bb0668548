"""Loss functions the configs name (`"loss": {"auto": "L1Loss", ...}`), reference: model/loss.py:16-30.
All reductions run in libhwg_hip.so."""
from .. import ops


def L1Loss(input, target):
    return ops.l1_loss(input, target)


def MSELoss(y_input, y_target):
    return ops.mse_loss(y_input, y_target.float())


MSE = MSELoss


def CTCLoss(input, target, input_len, target_len):
    """input [T,B,C] log-probs, target [B,L]; mean reduction, an infinite loss is reported as 0"""
    return ops.ctc_loss(input, target, input_len, target_len)


def _unsupported(name):
    def fn(*a, **k):
        raise NotImplementedError("%s is not used by any shipped config and is not on the accelerated path" % name)
    fn.__name__ = name
    return fn


HingeLoss = _unsupported("HingeLoss")
AdaptiveHingeLoss = _unsupported("AdaptiveHingeLoss")
CrossEntropyLoss = _unsupported("CrossEntropyLoss")
sigmoid_BCE_loss = _unsupported("sigmoid_BCE_loss")
