"""CPU oracle for the factorized-prior bits estimator -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product (gscodec_studio_amd) never does.

Restates ``Entropy_factorized_optimized_refactor.forward`` of the reference
(gsplat/compression_simulation/entropy_model.py:195-254) plus the analytic backward that
torch.autograd derives from it, in numpy (float64 by default so that it can arbitrate
between two fp32 implementations; pass dtype=np.float32 to mimic the reference's precision).
Pinned by tests/golden/make_golden_entropy.py, which imports the reference module in the build
container, records its outputs and autograd gradients, and asserts this file reproduces them.

The algorithm, per element x[n, c] with quantization step Q:
  lower/upper = f_p(x -/+ Q/2) where f_p is a tiny MLP 1 -> w1 -> ... -> wL -> 1 with
      h <- softplus(M_i[p]) h + b_i[p];  h <- h + tanh(F_i[p]) * tanh(h)   (all but the last layer)
  (entropy_model.py:229-238), sign = -sign(lower + upper) (247),
  likelihood = |sigmoid(sign upper) - sigmoid(sign lower)| (248), clamped from below at 1e-6 by
  LowerBound (249; gradient passes if likelihood >= bound or the incoming gradient is negative,
  entropy_model.py:355-357), bits = -log2(likelihood) (251).

Reference quirk that is REPRODUCED (it defines the numbers a drop-in has to match): the
"times = 32" reshape (219-228) views the zero-padded [2C, 1, N'] input as [64 C, 1, N'/32] and
tiles the C parameter sets 64 times along the batch dimension (231, ``matrix.repeat(2*times,1,1)``),
so batch row r*32 + j -- the j-th of 32 equal chunks along N of stacked row r -- is evaluated with
parameter set (r*32 + j) % C, not r % C.  For an element (n, c) that is
    p(n, c) = (32 c + n // chunk) % C,   chunk = N' / 32,   N' = N + (32 - N % 32).
For C = 4 the parameter set therefore depends only on the position n, for C = 3 on (2c + j) % 3.
"""
from __future__ import annotations

import numpy as np

TIMES = 32


def chunk_len(n: int) -> int:
    """N'/32 of the reference's padding (entropy_model.py:221-228): always pads, 32 when N % 32 == 0."""
    return (n + (TIMES - n % TIMES)) // TIMES


def param_channel(n_idx: np.ndarray, c_idx: np.ndarray, n: int, channels: int) -> np.ndarray:
    return (TIMES * c_idx + n_idx // chunk_len(n)) % channels


def _softplus(x):
    # torch.nn.functional.softplus: beta = 1, threshold = 20
    return np.where(x > 20.0, x, np.log1p(np.exp(np.minimum(x, 20.0))))


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def _mlp(h0, mats, biases, factors, p):
    """h0 [E]; mats[i] [C, w_{i+1}, w_i]; p [E] parameter set per element.  Returns the logits [E] and
    the tape (inputs of every layer, tanh of every hidden pre-activation)."""
    h = h0[:, None]  # [E, 1]
    hs, ts = [], []
    L = len(factors)
    for i in range(len(mats)):
        A = _softplus(mats[i])[p]            # [E, wo, wi]
        hs.append(h)
        z = np.einsum("eoi,ei->eo", A, h) + biases[i][p][:, :, 0]
        if i < L:
            t = np.tanh(z)
            ts.append(t)
            h = z + np.tanh(factors[i])[p][:, :, 0] * t
        else:
            h = z
    return h[:, 0], hs, ts


def _mlp_bwd(g_out, mats, biases, factors, p, hs, ts, g_mats, g_biases, g_factors):
    """Accumulates parameter gradients (w.r.t. the RAW parameters) and returns d/d h0 [E]."""
    L = len(factors)
    g = g_out[:, None]  # [E, 1]
    C = mats[0].shape[0]
    for i in reversed(range(len(mats))):
        if i < L:
            f = np.tanh(factors[i])[p][:, :, 0]
            t = ts[i]
            gf_t = g * t                                   # grad wrt tanh(F)
            gz = g * (1.0 + f * (1.0 - t * t))
            d_raw = gf_t * (1.0 - f * f)                   # through tanh(F)
            for c in range(C):
                g_factors[i][c, :, 0] += d_raw[p == c].sum(0)
        else:
            gz = g
        A_raw = mats[i]
        A = _softplus(A_raw)[p]
        gA = gz[:, :, None] * hs[i][:, None, :]            # [E, wo, wi] grad wrt softplus(M)
        gA_raw = gA * _sigmoid(A_raw)[p]
        for c in range(C):
            sel = p == c
            g_mats[i][c] += gA_raw[sel].sum(0)
            g_biases[i][c, :, 0] += gz[sel].sum(0)
        g = np.einsum("eoi,eo->ei", A, gz)
    return g[:, 0]


def factorized_bits_fwd(x, q, mats, biases, factors, bound=1e-6, dtype=np.float64, return_tape=False):
    """x [N, C]; q scalar or [C]; returns bits [N, C]."""
    x = np.asarray(x, dtype=dtype)
    N, C = x.shape
    mats = [np.asarray(m, dtype=dtype) for m in mats]
    biases = [np.asarray(b, dtype=dtype) for b in biases]
    factors = [np.asarray(f, dtype=dtype) for f in factors]
    qv = np.broadcast_to(np.asarray(q, dtype=dtype).reshape(-1), (C,)) if np.ndim(q) else np.full((C,), q, dtype=dtype)
    n_idx, c_idx = np.meshgrid(np.arange(N), np.arange(C), indexing="ij")
    n_idx, c_idx = n_idx.reshape(-1), c_idx.reshape(-1)
    p = param_channel(n_idx, c_idx, N, C)
    xe = x.reshape(-1)
    half = (dtype(0.5) * qv)[c_idx]
    lower, hs_l, ts_l = _mlp(xe - half, mats, biases, factors, p)
    upper, hs_u, ts_u = _mlp(xe + half, mats, biases, factors, p)
    sign = -np.sign(lower + upper)
    su, sl = _sigmoid(sign * upper), _sigmoid(sign * lower)
    lik = np.abs(su - sl)
    lik_b = np.maximum(lik, dtype(bound))
    bits = (-np.log2(lik_b)).reshape(N, C)
    if return_tape:
        return bits, dict(p=p, sign=sign, su=su, sl=sl, lik=lik, lik_b=lik_b, hs_l=hs_l, ts_l=ts_l, hs_u=hs_u, ts_u=ts_u,
                          mats=mats, biases=biases, factors=factors)
    return bits


def factorized_bits_bwd(x, q, mats, biases, factors, v_bits, bound=1e-6, dtype=np.float64):
    """Returns (v_x [N, C], [v_mats], [v_biases], [v_factors]) for the upstream gradient v_bits [N, C]."""
    bits, tp = factorized_bits_fwd(x, q, mats, biases, factors, bound, dtype, return_tape=True)
    N, C = bits.shape
    vb = np.asarray(v_bits, dtype=dtype).reshape(-1)
    g_lik_b = -vb / (np.log(dtype(2.0)) * tp["lik_b"])
    passthrough = (tp["lik"] >= dtype(bound)) | (g_lik_b < 0)       # LowerBound gradient (entropy_model.py:355-357)
    g_lik = np.where(passthrough, g_lik_b, 0.0)
    g_diff = g_lik * np.sign(tp["su"] - tp["sl"])                   # abs
    g_upper = g_diff * tp["sign"] * tp["su"] * (1.0 - tp["su"])
    g_lower = -g_diff * tp["sign"] * tp["sl"] * (1.0 - tp["sl"])
    g_mats = [np.zeros_like(m) for m in tp["mats"]]
    g_biases = [np.zeros_like(b) for b in tp["biases"]]
    g_factors = [np.zeros_like(f) for f in tp["factors"]]
    gx = _mlp_bwd(g_lower, tp["mats"], tp["biases"], tp["factors"], tp["p"], tp["hs_l"], tp["ts_l"], g_mats, g_biases, g_factors)
    gx = gx + _mlp_bwd(g_upper, tp["mats"], tp["biases"], tp["factors"], tp["p"], tp["hs_u"], tp["ts_u"], g_mats, g_biases, g_factors)
    return gx.reshape(N, C), g_mats, g_biases, g_factors
