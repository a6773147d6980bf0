// HBM-bound companions of the bf16 spiral convolution (BASELINE config 3): sparse mesh re-sampling (U, U^T, the list
// pre-sums of backward-data) and the activation backward, on bf16 tensors with fp32 arithmetic per element.
// Same structure as their fp32 twins in spmm_loss.hip / spiral_conv.hip: one work item = a 256-piece part of an output
// row, 16-byte accesses (8 channels), CSR entries are wave-uniform scalars, fixed summation order.
#include "sh_bf16.h"

namespace {

__device__ __forceinline__ void acc8(float (&a)[8], const u32x4& raw, float w) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(&raw);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = fmaf(w, (float)v[k], a[k]);
}
__device__ __forceinline__ u32x4 pack8(const float (&a)[8]) {
    const bf16x8 o = {(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3], (__bf16)a[4], (__bf16)a[5], (__bf16)a[6], (__bf16)a[7]};
    return *reinterpret_cast<const u32x4*>(&o);
}

// y[r,b,:] = sum_e val[e] * x[col[e],b,:]   (+ optional act'(yprev) epilogue, zero_row); strides in elements
__global__ __launch_bounds__(256) void spmm_bf16_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                        const float* __restrict__ val, const __bf16* __restrict__ x, long x_sv, long x_sb,
                                                        __bf16* __restrict__ y, long y_sv, long y_sb, const __bf16* __restrict__ yprev,
                                                        long yp_sv, long yp_sb, int act, int zero_row, int B, int rows, int C) {
    const int CW = C >> 3;
    const int per_row = B * CW;
    const int parts = (per_row + 255) >> 8;
    const long items = (long)rows * parts;
    for (long it = blockIdx.x; it < items; it += gridDim.x) {
        const int r = (int)(it / parts), part = (int)(it - (long)r * parts);
        const int e0 = rowptr[r], e1 = rowptr[r + 1];
        const bool zero = r == zero_row;
        const int j = part * 256 + threadIdx.x;
        if (j >= per_row) continue;
        const int b = j / CW, co = 8 * (j - b * CW);
        const long xo = (long)b * x_sb + co;
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int e = e0;
        for (; e + 1 < e1; e += 2) {                 // two independent 16-byte loads in flight (rows of U have <= 3 entries)
            const u32x4 x0 = *reinterpret_cast<const u32x4*>(x + (long)col[e] * x_sv + xo);
            const u32x4 x1 = *reinterpret_cast<const u32x4*>(x + (long)col[e + 1] * x_sv + xo);
            acc8(a, x0, val[e]);
            acc8(a, x1, val[e + 1]);
        }
        if (e < e1) acc8(a, *reinterpret_cast<const u32x4*>(x + (long)col[e] * x_sv + xo), val[e]);
        if (yprev) {
            const u32x4 raw = *reinterpret_cast<const u32x4*>(yprev + (long)r * yp_sv + (long)b * yp_sb + co);
            const bf16x8 yv = *reinterpret_cast<const bf16x8*>(&raw);
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] *= sh_act_grad_from_out((float)yv[k], act);
        }
        if (zero) {
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = 0.f;
        }
        *reinterpret_cast<u32x4*>(y + (long)r * y_sv + (long)b * y_sb + co) = pack8(a);
    }
}

// dpre = dy * act'(y), row zero_row forced to 0; bf16 in, bf16 out
__global__ __launch_bounds__(256) void act_backward_bf16_kernel(const __bf16* __restrict__ dy, long dy_sv, long dy_sb,
                                                                const __bf16* __restrict__ y, long y_sv, long y_sb,
                                                                __bf16* __restrict__ dp, long dp_sv, long dp_sb, int B, int R, int C, int act,
                                                                int zero_row) {
    const int CW = C >> 3;
    const int per_row = B * CW;
    const int parts = (per_row + 255) >> 8;
    const long items = (long)R * parts;
    for (long it = blockIdx.x; it < items; it += gridDim.x) {
        const int r = (int)(it / parts), part = (int)(it - (long)r * parts);
        const int j = part * 256 + threadIdx.x;
        if (j >= per_row) continue;
        const int b = j / CW, co = 8 * (j - b * CW);
        const u32x4 graw = *reinterpret_cast<const u32x4*>(dy + (long)r * dy_sv + (long)b * dy_sb + co);
        const u32x4 yraw = *reinterpret_cast<const u32x4*>(y + (long)r * y_sv + (long)b * y_sb + co);
        const bf16x8 g = *reinterpret_cast<const bf16x8*>(&graw), yv = *reinterpret_cast<const bf16x8*>(&yraw);
        float a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = r == zero_row ? 0.f : (float)g[k] * sh_act_grad_from_out((float)yv[k], act);
        *reinterpret_cast<u32x4*>(dp + (long)r * dp_sv + (long)b * dp_sb + co) = pack8(a);
    }
}

inline bool al16s(const void* p, long a, long b) { return ((reinterpret_cast<uintptr_t>(p) | (uintptr_t)(2 * a) | (uintptr_t)(2 * b)) & 15) == 0; }

}  // namespace

extern "C" {

int sh_spmm_bf16(const int32_t* rowptr, const int32_t* col, const float* val, const void* x, int64_t x_sv, int64_t x_sb, void* y,
                 int64_t y_sv, int64_t y_sb, const void* yprev, int64_t yp_sv, int64_t yp_sb, int act_prev, int zero_row, int B,
                 int rows, int C, sh_stream_t stream) {
    SH_REQUIRE(rowptr && col && val && x && y && B > 0 && rows > 0 && C > 0, SH_ERR_INVALID_ARG, "sh_spmm_bf16: bad argument");
    SH_REQUIRE(act_prev >= SH_ACT_IDENTITY && act_prev <= SH_ACT_TANH, SH_ERR_INVALID_ARG, "sh_spmm_bf16: unknown activation %d", act_prev);
    SH_REQUIRE(C % 8 == 0 && al16s(x, x_sv, x_sb) && al16s(y, y_sv, y_sb) && (!yprev || al16s(yprev, yp_sv, yp_sb)), SH_ERR_UNSUPPORTED,
               "sh_spmm_bf16: channels %% 8 == 0 and 16-byte aligned rows required (C = %d)", C);
    const long items = (long)rows * (((long)B * (C / 8) + 255) / 256);
    const int blocks = (int)(items < 8192 ? items : 8192);
    hipStream_t st = static_cast<hipStream_t>(stream);
    ShProfScope ps(st, "spmm_bf16_kernel|rows=%d B=%d C=%d", rows, B, C);
    SH_LAUNCH_PS(ps, spmm_bf16_kernel, dim3(blocks), dim3(256), 0, st, rowptr, col, val, static_cast<const __bf16*>(x), (long)x_sv, (long)x_sb,
                 static_cast<__bf16*>(y), (long)y_sv, (long)y_sb, static_cast<const __bf16*>(yprev), (long)yp_sv, (long)yp_sb, act_prev, zero_row,
                 B, rows, C);
    SH_CHECK_LAUNCH("spmm_bf16");
    return SH_OK;
}

int sh_act_backward_bf16(const void* dy, int64_t dy_sv, int64_t dy_sb, const void* y, int64_t y_sv, int64_t y_sb, void* dpre,
                         int64_t dp_sv, int64_t dp_sb, int B, int R, int C, int act, int zero_row, sh_stream_t stream) {
    SH_REQUIRE(dy && y && dpre && B > 0 && R > 0 && C > 0, SH_ERR_INVALID_ARG, "sh_act_backward_bf16: bad argument");
    SH_REQUIRE(act >= SH_ACT_IDENTITY && act <= SH_ACT_TANH, SH_ERR_INVALID_ARG, "sh_act_backward_bf16: unknown activation %d", act);
    SH_REQUIRE(C % 8 == 0 && al16s(dy, dy_sv, dy_sb) && al16s(y, y_sv, y_sb) && al16s(dpre, dp_sv, dp_sb), SH_ERR_UNSUPPORTED,
               "sh_act_backward_bf16: channels %% 8 == 0 and 16-byte aligned rows required (C = %d)", C);
    const long items = (long)R * (((long)B * (C / 8) + 255) / 256);
    const int blocks = (int)(items < 8192 ? items : 8192);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(act_backward_bf16_kernel, dim3(blocks), dim3(256), 0, st, static_cast<const __bf16*>(dy), (long)dy_sv, (long)dy_sb,
                       static_cast<const __bf16*>(y), (long)y_sv, (long)y_sb, static_cast<__bf16*>(dpre), (long)dp_sv, (long)dp_sb, B, R, C, act,
                       zero_row);
    SH_CHECK_LAUNCH("act_backward_bf16");
    return SH_OK;
}

}  // extern "C"
