// Multi-tensor Adam with coupled L2 weight decay - the optimiser of the reference training loops
// (main.py:262 `torch.optim.Adam(lr=1e-3, weight_decay=5e-5)`; steps at train_funcs.py:391-392,
// 509-510).  HBM-bound: 4 reads + 3 writes of every parameter-sized array, 0.8 GB per step for
// the 28.55 M-parameter autoencoder, so the kernel is a plain float4 stream.  Everything that
// changes from step to step (learning rate, step counts) is read from DEVICE memory, which keeps
// the launch replayable inside a hipGraph.
#include "sh_bf16.h"
#include "sh_adam.h"

#include <math.h>

namespace {

constexpr int AT = 60;          // tensors per launch (kernel-argument table: 3.6 KiB of the 4 KiB a launch may carry)
constexpr int ACH = 4096;       // elements per workgroup
constexpr int ANT = 256;

struct AdamArgs {
    float* p[AT];
    const float* g[AT];
    float* m[AT];
    float* v[AT];
    float* step[AT];
    __bf16* shadow[AT];           // optional bf16 working copy of the parameter, rewritten with the update (may be null)
    long n[AT];
    int blk_start[AT + 1];
    int nt, nontemporal;
    const float* lr;
    double beta1, beta2;          // for the bias corrections (double, like torch's fused kernel)
    float w1, b2, w2, eps, wd;    // (float)(1 - beta1), (float)beta2, (float)(1 - beta2)
};

__global__ __launch_bounds__(ANT) void adam_kernel(const AdamArgs a) {
    __shared__ float sh[2];
    int t = 0;
    while (t + 1 < a.nt && (int)blockIdx.x >= a.blk_start[t + 1]) ++t;
    const long base = (long)((int)blockIdx.x - a.blk_start[t]) * ACH;
    const long n = a.n[t];
    if (threadIdx.x == 0) {
        sh_adam_coeffs(a.beta1, a.beta2, a.step[t][0], a.lr[0], sh[0], sh[1]);
    }
    __syncthreads();
    const float step_size = sh[0], bc2_sqrt = sh[1];
    float* __restrict__ p = a.p[t];
    const float* __restrict__ g = a.g[t];
    float* __restrict__ m = a.m[t];
    float* __restrict__ v = a.v[t];
    __bf16* __restrict__ sw = a.shadow[t];
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                       reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    if (vec && base + ACH <= n) {
#pragma unroll
        for (int i = 0; i < ACH / (4 * ANT); ++i) {
            const long o = base + 4L * (threadIdx.x + i * ANT);
            // every byte is touched exactly once per step: stream it past the caches (the conv kernels running
            // beside an overlapped update live on L2 hits of their gathered rows)
            f32x4 pp, mm, vv, gg;
            if (a.nontemporal) {
                pp = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + o));
                mm = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(m + o));
                vv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(v + o));
                gg = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g + o));
            } else {
                pp = *reinterpret_cast<f32x4*>(p + o); mm = *reinterpret_cast<f32x4*>(m + o); vv = *reinterpret_cast<f32x4*>(v + o);
                gg = *reinterpret_cast<const f32x4*>(g + o);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float p1 = pp[j], m1 = mm[j], v1 = vv[j];
                adam_update(p1, gg[j], m1, v1, a.w1, a.b2, a.w2, a.eps, a.wd, step_size, bc2_sqrt);
                pp[j] = p1; mm[j] = m1; vv[j] = v1;
            }
            if (sw) *reinterpret_cast<bf16x4*>(sw + o) = sh_to_bf16x4(pp);      // read by the next forward pass: keep it cached
            if (a.nontemporal) {
                __builtin_nontemporal_store(pp, reinterpret_cast<f32x4*>(p + o));
                __builtin_nontemporal_store(mm, reinterpret_cast<f32x4*>(m + o));
                __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(v + o));
            } else {
                *reinterpret_cast<f32x4*>(p + o) = pp;
                *reinterpret_cast<f32x4*>(m + o) = mm;
                *reinterpret_cast<f32x4*>(v + o) = vv;
            }
        }
    } else {
        // ragged or unaligned end of a tensor (and every tensor smaller than a workgroup's 4096 elements: biases, the thin convs):
        // all of a thread's elements are requested before the first is used - one memory round trip, not sixteen in sequence
        constexpr int PER = ACH / ANT;
        float pp[PER], mm[PER], vv[PER], gg[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const long o = base + threadIdx.x + (long)i * ANT;
            const bool ok = o < n;
            pp[i] = ok ? p[o] : 0.f; mm[i] = ok ? m[o] : 0.f; vv[i] = ok ? v[o] : 0.f; gg[i] = ok ? g[o] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const long o = base + threadIdx.x + (long)i * ANT;
            if (o < n) {
                adam_update(pp[i], gg[i], mm[i], vv[i], a.w1, a.b2, a.w2, a.eps, a.wd, step_size, bc2_sqrt);
                p[o] = pp[i]; m[o] = mm[i]; v[o] = vv[i];
                if (sw) sw[o] = (__bf16)pp[i];
            }
        }
    }
}

// step counts advance after the update kernel has read them (same stream -> ordered)
__global__ void adam_bump_kernel(const AdamArgs a) {
    if ((int)threadIdx.x < a.nt) a.step[threadIdx.x][0] += 1.f;
}

}  // namespace

extern "C" {

static int adam_impl(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                     float* const* steps, void* const* shadow, const int64_t* numel, const float* lr, double beta1, double beta2,
                     double eps, double weight_decay, sh_stream_t stream) {
    SH_REQUIRE(n_tensors >= 0, SH_ERR_INVALID_ARG, "sh_adam_step: negative tensor count");
    if (n_tensors == 0) return SH_OK;
    SH_REQUIRE(params && grads && exp_avg && exp_avg_sq && steps && numel && lr, SH_ERR_INVALID_ARG, "sh_adam_step: null pointer");
    SH_REQUIRE(beta1 >= 0 && beta1 < 1 && beta2 >= 0 && beta2 < 1 && eps >= 0 && weight_decay >= 0, SH_ERR_INVALID_ARG,
               "sh_adam_step: hyper-parameter out of range");
    hipStream_t st = static_cast<hipStream_t>(stream);
    static const int nontemporal = sh_env_int("SH_ADAM_NT", 1, 0, 1);
    for (int t0 = 0; t0 < n_tensors; t0 += AT) {
        AdamArgs a{};
        a.nt = n_tensors - t0 < AT ? n_tensors - t0 : AT;
        long blocks = 0;
        for (int i = 0; i < a.nt; ++i) {
            const int k = t0 + i;
            // numel == 0: a parameter whose update was applied elsewhere (sh_linear_bwd_wgt_adam) - only its step count advances
            SH_REQUIRE(steps[k] && numel[k] >= 0 && (numel[k] == 0 || (params[k] && grads[k] && exp_avg[k] && exp_avg_sq[k])), SH_ERR_INVALID_ARG,
                       "sh_adam_step: tensor %d has a null pointer or a negative element count", k);
            a.p[i] = params[k]; a.g[i] = grads[k]; a.m[i] = exp_avg[k]; a.v[i] = exp_avg_sq[k]; a.step[i] = steps[k];
            a.shadow[i] = shadow ? static_cast<__bf16*>(shadow[k]) : nullptr;
            SH_REQUIRE(!a.shadow[i] || (reinterpret_cast<uintptr_t>(a.shadow[i]) & 7) == 0, SH_ERR_INVALID_ARG, "sh_adam_step_bf16: shadow %d misaligned", k);
            a.n[i] = numel[k];
            a.blk_start[i] = (int)blocks;
            blocks += (numel[k] + ACH - 1) / ACH;
            SH_REQUIRE(blocks < (1L << 31), SH_ERR_UNSUPPORTED, "sh_adam_step: too many elements in one launch");
        }
        a.blk_start[a.nt] = (int)blocks;
        a.nontemporal = nontemporal;
        a.lr = lr; a.beta1 = beta1; a.beta2 = beta2;
        a.w1 = (float)(1.0 - beta1); a.b2 = (float)beta2; a.w2 = (float)(1.0 - beta2); a.eps = (float)eps; a.wd = (float)weight_decay;
        if (blocks > 0) {
            long numel_sum = 0;
            for (int i = 0; i < a.nt; ++i) numel_sum += a.n[i];
            ShProfScope ps(st, "adam_kernel|tensors=%d blocks=%ld numel=%ld", a.nt, blocks, numel_sum);
            SH_LAUNCH_PS(ps, adam_kernel, dim3((unsigned)blocks), dim3(ANT), 0, st, a);
        }
        hipLaunchKernelGGL(adam_bump_kernel, dim3(1), dim3(64), 0, st, a);
        SH_CHECK_LAUNCH("adam_step");
    }
    return SH_OK;
}

int sh_adam_bump(int n_tensors, float* const* steps, sh_stream_t stream) {
    SH_REQUIRE(n_tensors >= 0, SH_ERR_INVALID_ARG, "sh_adam_bump: negative tensor count");
    if (n_tensors == 0) return SH_OK;
    SH_REQUIRE(steps, SH_ERR_INVALID_ARG, "sh_adam_bump: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int t0 = 0; t0 < n_tensors; t0 += AT) {
        AdamArgs a{};
        a.nt = n_tensors - t0 < AT ? n_tensors - t0 : AT;
        for (int i = 0; i < a.nt; ++i) {
            SH_REQUIRE(steps[t0 + i], SH_ERR_INVALID_ARG, "sh_adam_bump: tensor %d has a null step pointer", t0 + i);
            a.step[i] = steps[t0 + i];
        }
        hipLaunchKernelGGL(adam_bump_kernel, dim3(1), dim3(64), 0, st, a);
        SH_CHECK_LAUNCH("adam_bump");
    }
    return SH_OK;
}

int sh_adam_step(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                 float* const* steps, const int64_t* numel, const float* lr, double beta1, double beta2, double eps,
                 double weight_decay, sh_stream_t stream) {
    return adam_impl(n_tensors, params, grads, exp_avg, exp_avg_sq, steps, nullptr, numel, lr, beta1, beta2, eps, weight_decay, stream);
}

int sh_adam_step_bf16(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                      float* const* steps, void* const* shadow_bf16, const int64_t* numel, const float* lr, double beta1, double beta2,
                      double eps, double weight_decay, sh_stream_t stream) {
    return adam_impl(n_tensors, params, grads, exp_avg, exp_avg_sq, steps, shadow_bf16, numel, lr, beta1, beta2, eps, weight_decay, stream);
}

}  // extern "C"
