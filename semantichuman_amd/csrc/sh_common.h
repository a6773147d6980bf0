// Shared device helpers for the gfx950 kernels of libsh_kernels.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/sh_kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------- error plumbing
void sh_set_error(const char* fmt, ...);

#define SH_REQUIRE(cond, code, ...)            \
    do {                                       \
        if (!(cond)) {                         \
            sh_set_error(__VA_ARGS__);         \
            return (code);                     \
        }                                      \
    } while (0)

#define SH_CHECK_LAUNCH(name)                                                       \
    do {                                                                            \
        hipError_t e_ = hipGetLastError();                                          \
        if (e_ != hipSuccess) {                                                     \
            sh_set_error("%s: launch failed: %s", (name), hipGetErrorString(e_));   \
            return SH_ERR_LAUNCH;                                                   \
        }                                                                           \
    } while (0)

// ---------------------------------------------------------------------------- kernel timing
// RAII: records a hipEvent pair on `st` around the launches made inside its scope (only when
// profiling was enabled through sh_profile_enable).
bool sh_profile_on();
void sh_profile_push(const char* name, hipEvent_t a, hipEvent_t b);
struct ShProfScope {
    hipStream_t st; hipEvent_t a, b; bool on; bool ext; char name[96];
    ShProfScope(hipStream_t s, const char* fmt, ...);
    ~ShProfScope();
    // first launch of the scope takes the start event; every launch re-targets the stop event (the last one wins)
    hipEvent_t take_start() { hipEvent_t e = ext ? nullptr : a; ext = true; return e; }
};
// Launch inside a ShProfScope.  When profiling is on, the events are attached to the kernel dispatch itself
// (hipExtLaunchKernelGGL: begin / end of the kernel's execution, what rocprofv3 --kernel-trace reports) instead of
// bracketing it with event records on the stream, which would add the dispatch latency to every measurement.
#define SH_LAUNCH_PS(ps, kernel, grid, block, smem, st, ...)                                                          \
    do {                                                                                                              \
        if ((ps).on) hipExtLaunchKernelGGL(kernel, grid, block, smem, st, (ps).take_start(), (ps).b, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, grid, block, smem, st, __VA_ARGS__);                                          \
    } while (0)

// ---------------------------------------------------------------------------- activations
// Forward activation (reference models.py:19-32) ...
__device__ __forceinline__ float sh_act_fwd(float v, int act) {
    switch (act) {
        case SH_ACT_RELU: return v > 0.f ? v : 0.f;
        case SH_ACT_ELU: return v > 0.f ? v : expm1f(v);
        case SH_ACT_LEAKY_RELU: return v > 0.f ? v : 0.02f * v;
        case SH_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case SH_ACT_TANH: return tanhf(v);
        default: return v;
    }
}
// A quad at once: ONE wave-uniform branch per four elements instead of the switch per element (the conv epilogues ran a chain
// of scalar branches for each of up to 32 elements per lane and item), the model's own activation (ELU) tested first.
__device__ __forceinline__ f32x4 sh_act_fwd4(f32x4 a, int act) {
    if (act == SH_ACT_ELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = a[j] > 0.f ? a[j] : expm1f(a[j]);
    } else if (act != SH_ACT_IDENTITY) {
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = sh_act_fwd(a[j], act);
    }
    return a;
}
// ... and its derivative expressed through the activation OUTPUT y, so the backward pass needs
// nothing but the tensor the next layer consumed anyway (no saved pre-activation).
__device__ __forceinline__ float sh_act_grad_from_out(float y, int act) {
    // SELECT form, no switch: `act` is wave-uniform, and a switch becomes a chain of scalar branches that the epilogues and the
    // role-swapped weight-gradient kernels would run once per element (measured in wgrad_thin: ~40 taken branches per item).  Every
    // candidate is a couple of VALU operations; the expressions are the ones the switch had (same bits).
    const float neg = act == SH_ACT_ELU ? y + 1.f : act == SH_ACT_LEAKY_RELU ? 0.02f : 0.f;          // exp(x) = elu(x) + 1 for x <= 0
    const float piece = y > 0.f ? 1.f : neg;
    const float smooth = act == SH_ACT_SIGMOID ? y * (1.f - y) : 1.f - y * y;
    const bool is_piece = act == SH_ACT_RELU || act == SH_ACT_ELU || act == SH_ACT_LEAKY_RELU;
    const bool is_smooth = act == SH_ACT_SIGMOID || act == SH_ACT_TANH;
    return is_piece ? piece : is_smooth ? smooth : 1.f;
}

// XCD-aware work-item order.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8
// share an XCD and its 4 MiB L2); remapping so that each XCD owns a CONTIGUOUS range of work items
// lets the S-fold reuse of gathered neighbour rows hit in that XCD's L2 instead of going to the
// fabric.  Bijective for any grid size (cdna_hip_programming.md 5, 'XCD swizzle must be bijective').
// Placement only changes speed, never results.
__device__ __forceinline__ int sh_xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, local = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

// The index of a wave in its workgroup as a SCALAR (readfirstlane of threadIdx.x >> 6, which is uniform over a wave by
// construction - hipcc does not know).  Work-item decoding, loop bounds and branches derived from it are then uniform values:
// real branches and SALU arithmetic instead of predicated VALU code.  Round 5, measured kernel by kernel (profiles/
// r05_kernel_experiments.txt); -DSH_SCALAR_WAVE=0 builds the old form.
#ifndef SH_SCALAR_WAVE
#define SH_SCALAR_WAVE 1
#endif
__device__ __forceinline__ int sh_wave_id() {
#if SH_SCALAR_WAVE
    return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#else
    return (int)(threadIdx.x >> 6);
#endif
}

__device__ __forceinline__ float sh_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static inline int sh_ilog2_floor(int v) {
    int l = 0;
    while ((1 << (l + 1)) <= v) ++l;
    return l;
}
static inline int sh_cdiv(int a, int b) { return (a + b - 1) / b; }
// Arithmetic form of the fp32 matrix products (enum sh_mma_mode).  It is an ARGUMENT of every entry point whose kernel choice
// depends on it; the entry point pins it for the duration of the call on the calling thread (ShMmaScope) and the dispatch
// code reads it back with sh_f32_mma_mode().  No process-wide state: a forward pass on one thread and the backward pass of
// another node on autograd's thread each run in the form their own call names.
int sh_f32_mma_mode();
bool sh_mma_mode_valid(int mode);
struct ShMmaScope {
    int was;
    explicit ShMmaScope(int mode);
    ~ShMmaScope();
};
// tuning knob from the environment (read once by the caller through a function-local static)
int sh_env_int(const char* name, int dflt, int lo, int hi);
