"""Test infrastructure (never imported by the product): the TRAINING STATE that teacher-forced trainer parity runs start each unit from.

tools/gen_golden_tf.py (build container, unmodified reference trainer) and tests/test_trainer_teacher_forced_gpu.py (HIP trainer) both
put their trainer into the state described here before every unit of lessons, so that each lesson kind is compared from one and the
same state instead of from the end of a chain that has drifted (reference: trainer/hw_with_style_trainer.py:207-418 reads weights, BN
running statistics, spectral-norm u/v, both Adam states, `prev_styles` and `saved_grads`).

Everything is reproducible from seeds and a few small stored numbers:
  * weights / buffers: oracle.torch_ref.seeded_state_dict(model, seed) - or, for the trained regime, those seeded weights plus an int8
    delta per tensor recorded from a reference trajectory (tests/golden/tf_trained_state.npz);
  * Adam moments: drawn from seeded generators and scaled, per tensor, by the RMS of the gradient the REFERENCE produced in that very
    iteration (stored in the golden): exp_avg ~ 0.5 rms N(0,1), exp_avg_sq = rms^2 U(0.25, 1.25), step 150. With moments of the size of
    the gradient the update lr*m_hat/(sqrt(v_hat)+eps) is a smooth function of the gradient (the regime of a run in progress); with
    zero moments (first step after init) it is sign(g), which turns last-bit differences of near-zero gradient elements into +-lr;
  * the style bank `prev_styles`: seeded normal vectors.
"""
import numpy as np
import torch

ADAM_STEP = 150


def seeded_moments(shape, rms, key):
    """(exp_avg, exp_avg_sq) for one tensor; fp32 CPU, a pure function of (shape, rms, key)"""
    g = torch.Generator().manual_seed(int(key))
    rms = float(np.float32(rms))
    # explicit fp32 draws: the reference's fp64-widened run sets torch's default dtype to float64, and a float64 draw is a different stream
    m = torch.randn(shape, generator=g, dtype=torch.float32) * (0.5 * rms)
    v = (torch.rand(shape, generator=g, dtype=torch.float32) + 0.25) * (rms * rms)
    return m, v


def moment_key(unit_iteration, tensor_index):
    return 7919 * (unit_iteration + 1) + tensor_index


def seeded_prev_styles(n, dim, seed):
    g = torch.Generator().manual_seed(int(seed))
    return [torch.randn(dim, generator=g, dtype=torch.float32) for _ in range(n)]


def quantize_delta(w, w0):
    """int8 image of (w - w0) with one fp32 scale per tensor -> (q int8, scale float32)"""
    d = (w - w0).to(torch.float32)
    amax = float(d.abs().max()) if d.numel() else 0.0
    scale = np.float32(amax / 127.0) if amax > 0 else np.float32(0.0)
    q = torch.zeros(d.shape, dtype=torch.int8) if scale == 0 else torch.clamp(torch.round(d / float(scale)), -127, 127).to(torch.int8)
    return q, scale


def apply_delta(w0, q, scale):
    """the fp32 tensor both sides load: w0 + q * scale, evaluated in fp32 on the CPU (bit-identical wherever it runs)"""
    # (.to(torch.float32), not .float(): the reference's widened run patches Tensor.float to return float64)
    return (w0.to(torch.float32) + q.to(torch.float32) * torch.tensor(float(scale), dtype=torch.float32)).to(torch.float32)
