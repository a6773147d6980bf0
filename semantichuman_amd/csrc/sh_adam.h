// The Adam update of one element and the per-step coefficients, shared by the multi-tensor optimiser kernel (adam.hip) and the
// latent-FC weight-gradient kernel that applies the update to the tile it has just computed (linear.hip, sh_linear_bwd_wgt_adam):
// one definition, so the fused form is bit-identical to "write the gradient, then sh_adam_step".
#pragma once
#include <math.h>

struct ShAdamHyper {
    const float* lr;              // device scalar
    const float* step;            // device scalar: updates applied so far (this update is number step + 1)
    double beta1, beta2;          // for the bias corrections (double, like torch's fused kernel)
    float w1, b2, w2, eps, wd;    // (float)(1 - beta1), (float)beta2, (float)(1 - beta2)
};

// beta^n for an integer n by repeated squaring in double: <= 24 dependent multiplies for any step count a float can hold exactly,
// against ~8 us of dependent latency for two calls of the library's general pow(double, double) - which every workgroup of the
// update kernels would pay before it can touch a byte.  The rounding error (<= 48 ulp of a double) vanishes in the conversion of the
// coefficients to float; one definition for every kernel keeps the fused and the two-kernel form bit-identical.
__device__ __forceinline__ double sh_pow_int(double b, unsigned n) {
    double r = 1.0;
    while (n) {
        if (n & 1u) r *= b;
        b *= b;
        n >>= 1;
    }
    return r;
}

__device__ __forceinline__ void sh_adam_coeffs(double beta1, double beta2, float steps_done, float lr, float& step_size, float& bc2_sqrt) {
    const unsigned step = (unsigned)steps_done + 1u;              // this update's 1-based index
    const double bc1 = 1.0 - sh_pow_int(beta1, step), bc2 = 1.0 - sh_pow_int(beta2, step);
    step_size = (float)((double)lr / bc1);
    bc2_sqrt = (float)sqrt(bc2);
}

__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, float w, float beta2, float w2, float eps, float wd,
                                            float step_size, float bc2_sqrt) {
#pragma clang fp contract(off)      // same rounding on the 16-byte and the scalar path (no call-site dependent FMA fusion)
    if (wd != 0.f) g += wd * p;                                   // coupled L2 (torch.optim.Adam, not AdamW)
    const float d = g - m;
    m = w < 0.5f ? m + w * d : g - d * (1.f - w);                 // lerp(m, g, 1 - beta1)
    v = beta2 * v + w2 * g * g;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p -= step_size * m / denom;
}
