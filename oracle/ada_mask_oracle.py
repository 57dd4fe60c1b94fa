"""CPU restatement (numpy, fp32) of the reference's shN adaptive mask -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(gscodec_studio_amd/) never does.

Follows, line by line:
    gsplat/compression_simulation/ada_mask.py:19-30   AnnealingMask.get_temperature
    gsplat/compression_simulation/ada_mask.py:32-40   AnnealingMask.forward (training / eval)
    gsplat/compression_simulation/ada_mask.py:42-44   get_binary_mask
    gsplat/compression_simulation/ada_mask.py:46-58   get_sparsity_loss
    gsplat/compression_simulation/ada_mask.py:60-62   get_mask_ratio
    gsplat/compression_simulation/simulation.py:319-324  simulate_compression_shN (when the mask applies)
    gsplat/compression_simulation/simulation.py:327-348  shN_gradient_threshold
and the autograd of those expressions written out by hand (torch's formulas: mul backward, sum over the broadcast dims,
sigmoid_backward = g (1 - y) y, division by a scalar, mean backward, binary_cross_entropy backward).

Pinned by tests/golden/make_golden_ada_mask.py, which imports the reference's classes in the build container, writes
tests/golden/ada_mask.npz and asserts this module reproduces every output.
"""
import math

import numpy as np

F32 = np.float32


def temperature(current_step, total_iters=30_000, start_temp=5.0, end_temp=0.1, annealing_start_iter=10_000):
    """ada_mask.py:19-30 (python doubles)."""
    if current_step < annealing_start_iter:
        return start_temp
    progress = (current_step - annealing_start_iter) / (total_iters - annealing_start_iter)
    progress = min(max(progress, 0), 1)
    return start_temp * math.exp(math.log(end_temp / start_temp) * progress)


def _sigmoid(v):
    v = np.asarray(v, F32)
    with np.errstate(over="ignore"):
        return (F32(1) / (F32(1) + np.exp(-v, dtype=F32))).astype(F32)


def soft_mask(logits, temp):
    """ada_mask.py:37: sigmoid(mask_logits / temperature), fp32 true division."""
    return _sigmoid(np.asarray(logits, F32) / F32(temp))


def binary_mask(logits):
    """ada_mask.py:39, 44: (sigmoid(mask_logits) >= 0.5).float()."""
    return (_sigmoid(logits) >= F32(0.5)).astype(F32)


def mask_forward(x, logits, temp, training=True):
    """ada_mask.py:32-40.  x [N, K-1, 3], logits [N, 1, 1] (or [N])."""
    x = np.asarray(x, F32)
    m = soft_mask(logits, temp) if training else binary_mask(logits)
    return (x * m.reshape(-1, 1, 1)).astype(F32)


def mask_backward(x, logits, temp, v_out, training=True):
    """Autograd of mask_forward: returns (v_x, v_logits [N]); v_logits is None in eval mode (the comparison is not
    differentiable)."""
    x, v_out = np.asarray(x, F32), np.asarray(v_out, F32)
    m = (soft_mask(logits, temp) if training else binary_mask(logits)).reshape(-1)
    v_x = (v_out * m.reshape(-1, 1, 1)).astype(F32)
    if not training:
        return v_x, None
    v_mask = (v_out * x).reshape(len(x), -1).sum(axis=1, dtype=np.float64).astype(F32)
    v_sig = (v_mask * (F32(1) - m) * m).astype(F32)
    return v_x, (v_sig / F32(temp)).astype(F32)


def sparsity_loss(logits, temp, target_sparsity=0.2, lambda_l1=0.01, lambda_target=0.1):
    """ada_mask.py:46-58: lambda_l1 * mean(mask) + lambda_target * BCE(mean(mask), target).
    Returns (loss, v_logits) with v_logits the gradient of the loss w.r.t. the logits (flat)."""
    logits = np.asarray(logits, F32).reshape(-1)
    m = soft_mask(logits, temp)
    s = F32(m.sum(dtype=np.float64) / len(m))
    t = F32(target_sparsity)
    # F.binary_cross_entropy clamps its logs at -100
    log_s = max(math.log(float(s)), -100.0) if s > 0 else -100.0
    log_1s = max(math.log(1.0 - float(s)), -100.0) if s < 1 else -100.0
    bce = -(float(t) * log_s + (1.0 - float(t)) * log_1s)
    loss = F32(lambda_l1 * float(s) + lambda_target * bce)
    # d loss / d s ; binary_cross_entropy_backward: (s - t) / max((1 - s) s, 1e-12)
    d_s = lambda_l1 + lambda_target * (float(s) - float(t)) / max((1.0 - float(s)) * float(s), 1e-12)
    v_m = F32(d_s) / F32(len(m))
    v_logits = ((v_m * (F32(1) - m) * m) / F32(temp)).astype(F32)
    return loss, v_logits


def mask_ratio(logits):
    """ada_mask.py:60-62: sum(sigmoid(logits) >= 0.5) / shape[0] (int64 / int -> fp32)."""
    logits = np.asarray(logits, F32)
    return F32(F32((_sigmoid(logits) >= F32(0.5)).sum()) / F32(logits.shape[0]))


def shn_gradient_threshold(param, grad):
    """simulation.py:327-348: returns the modified gradient (a copy)."""
    param, grad = np.asarray(param, F32), np.array(grad, F32, copy=True)
    zero_mask = (param == 0).all(axis=-1).all(axis=-1)
    non_zero_ratio = F32(1) - F32(zero_mask.sum()) / F32(zero_mask.shape[0])
    thr = 2e-3 if non_zero_ratio < F32(0.10) else 100
    norm = np.sqrt((grad.astype(np.float64) ** 2).reshape(len(grad), -1).sum(axis=1)).astype(F32)
    low = norm < F32(thr)
    grad[np.logical_and(zero_mask, low)] = 0
    return grad
