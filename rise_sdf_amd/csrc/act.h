// Activations of the VanillaMLP layers (models/network_utils.py:109-157, 160-175) and their derivatives, shared by the
// per-layer kernels (mlp.hip, mlp_layer_bwd.hip).
#pragma once
#include "common.h"

namespace {

__device__ __forceinline__ float act_fwd(float z, int act)
{
    switch (act) {
    case RSDF_ACT_RELU: return max0(z);
    case RSDF_ACT_SOFTPLUS100: {
        // log1p(e^t) / 100 = max(z, 0) + ln2 / 100 * log2(1 + 2^(-|t| log2 e)) on the hardware exp2 / log2 units (v_exp_f32,
        // v_log_f32: ~1 ulp), the form the fused x2 kernels use (mlp_x2.hip softplus_scaled).  The libm-style log1pf(expf(t))
        // was ~60 vector instructions per element -- most of a 64-wide layer's forward; for t > 20 the second term is
        // exactly 0, torch's threshold.  Absolute difference to the libm form <= 1e-9.
        const float e = __builtin_amdgcn_exp2f(fabsf(z) * -144.26950408889634f);
        return max0(z) + __builtin_amdgcn_logf(1.0f + e) * 0.0069314718055994531f;
    }
    case RSDF_ACT_SIGMOID: return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.4426950408889634f));
    default: return z;
    }
}

// derivative expressed through the OUTPUT y = act(z), so the forward only has to keep y
__device__ __forceinline__ float act_bwd_from_y(float y, int act)
{
    switch (act) {
    case RSDF_ACT_RELU: return y > 0.0f ? 1.0f : 0.0f;
    case RSDF_ACT_SOFTPLUS100:
        // y = log(1+e^{100 z})/100  =>  sigmoid(100 z) = 1 - e^{-100 y} = 1 - 2^(-100 log2(e) y)   (hardware exp2; absolute
        // error <= 6e-8 on a value in [0, 1]: mlp_x2.hip softplus_grad_scaled)
        return 1.0f - __builtin_amdgcn_exp2f(y * -144.26950408889634f);
    case RSDF_ACT_SIGMOID: return y * (1.0f - y);
    default: return 1.0f;
    }
}

}  // namespace
