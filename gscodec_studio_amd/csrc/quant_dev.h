// quant_dev.h -- the quantizers' per-value arithmetic (reference gsplat/compression_simulation/ops.py:39-75), shared by quantize.hip and the
// kernels that quantize while they load (projection_dyn.hip).  IEEE operations in the reference's order, no fma contraction (GS_FP_STRICT: this toolchain's
// __fmul_rn / __fadd_rn are plain `*` / `+` and would be fused): the same bits in every kernel they are inlined into.  Device code only.
#pragma once
#include "gs_common.h"

namespace {

GS_DEV float q_clamp(float x, float lo, float hi) {
    // torch.clamp: NaN propagates; min(max(x, lo), hi)
    float y = x < lo ? lo : x;
    y = y > hi ? hi : y;
    return y; // NaN compares false twice -> stays NaN
}

// Opt-in fusion: the activation the trainer applies right after the hook -- torch.exp for the log-scales,
// torch.sigmoid for the opacity logits (reference examples/simple_trainer.py:779-786) -- evaluated in the quantizer's own
// pass; the backward multiplies by its derivative, read off the activated output.  ACT 0 = none (bit-exact reference path).
template <int ACT>
GS_DEV float q_act(float v) {
    GS_FP_STRICT;
    if (ACT == GS_ACT_EXP) return expf(v);
    if (ACT == GS_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
    return v;
}
template <int ACT>
GS_DEV float q_act_grad(float out, float v) { // d act / d pre-activation, from the activated value
    GS_FP_STRICT;
    if (ACT == GS_ACT_EXP) return v * out;
    if (ACT == GS_ACT_SIGMOID) return v * out * (1.f - out);
    return v;
}

GS_DEV float q_round(float xc, float lo, float range, float qn) {
    GS_FP_STRICT;
    float norm = ((xc - lo) / range);
    float lvl = rintf((norm / qn)); // round half to even, as torch.round
    return (((lvl * qn) * range) + lo);
}

} // namespace
