// Activations of the VanillaMLP layers (models/network_utils.py:109-157, 160-175) and their derivatives, shared by the
// per-layer kernels (mlp.hip, mlp_layer_bwd.hip).
#pragma once
#include "common.h"

namespace {

__device__ __forceinline__ float act_fwd(float z, int act)
{
    switch (act) {
    case RSDF_ACT_RELU: return fmaxf(z, 0.0f);
    case RSDF_ACT_SOFTPLUS100: {
        const float t = z * 100.0f;
        return t > 20.0f ? z : log1pf(expf(t)) / 100.0f;
    }
    case RSDF_ACT_SIGMOID: return 1.0f / (1.0f + expf(-z));
    default: return z;
    }
}

// derivative expressed through the OUTPUT y = act(z), so the forward only has to keep y
__device__ __forceinline__ float act_bwd_from_y(float y, int act)
{
    switch (act) {
    case RSDF_ACT_RELU: return y > 0.0f ? 1.0f : 0.0f;
    case RSDF_ACT_SOFTPLUS100:
        // y = log(1+e^{100 z})/100  =>  sigmoid(100 z) = 1 - e^{-100 y}
        return -expm1f(-100.0f * y);
    case RSDF_ACT_SIGMOID: return y * (1.0f - y);
    default: return 1.0f;
    }
}

}  // namespace
