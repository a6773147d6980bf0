// GPU-resident dataset path (reference autoencoder_dataset.py:26-58): the per-sample numpy
// normalisation + dummy-row padding done ONCE for the whole packed split on device, and the
// per-iteration batch assembly as a row gather from the resident tensor.
#include "sh_common.h"

namespace {

constexpr int NB = 256;

// fixed-order block reduction of three doubles (sum, or min/max through `op`)
template <int OP>   // 0 sum, 1 min, 2 max
__device__ __forceinline__ void block_reduce3(double* v, double (*red)[3]) {
    const int t = threadIdx.x;
    red[t][0] = v[0]; red[t][1] = v[1]; red[t][2] = v[2];
    __syncthreads();
    for (int s = NB / 2; s > 0; s >>= 1) {
        if (t < s) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const double a = red[t][d], b = red[t + s][d];
                red[t][d] = OP == 0 ? a + b : (OP == 1 ? ((b < a || b != b) ? b : a) : ((b > a || b != b) ? b : a));   // NaN propagates like np.max
            }
        }
        __syncthreads();
    }
    v[0] = red[0][0]; v[1] = red[0][1]; v[2] = red[0][2];
    __syncthreads();
}

// One workgroup per mesh; the mesh is normalised in place in its output slot (83 KB at 6890
// vertices: L2-resident between the passes).  Steps and their order follow autoencoder_dataset.py:29-44.
__global__ void normalize_pack_kernel(const float* __restrict__ raw, float* __restrict__ out, int N, int dummy_rows,
                                      unsigned flags, const float* __restrict__ j_root, const float* __restrict__ mean,
                                      const float* __restrict__ stdv, const float* __restrict__ center,
                                      const float* __restrict__ scale) {
    __shared__ double red[NB][3];
    const int m = blockIdx.x, t = threadIdx.x;
    const float* src = raw + (long)m * N * 3;
    float* dst = out + (long)m * (N + dummy_rows) * 3;
    for (int i = t; i < N * 3; i += NB) dst[i] = src[i];
    for (int i = N * 3 + t; i < (N + dummy_rows) * 3; i += NB) dst[i] = 0.f;            // :45-48 dummy node
    __syncthreads();
    if (flags & SH_NORM_ZEROMEAN) {                                                       // :29-30
        double s[3] = {0, 0, 0};
        for (int v = t; v < N; v += NB) { s[0] += dst[3 * v]; s[1] += dst[3 * v + 1]; s[2] += dst[3 * v + 2]; }
        block_reduce3<0>(s, red);
        const float mx = (float)(s[0] / N), my = (float)(s[1] / N), mz = (float)(s[2] / N);
        for (int v = t; v < N; v += NB) { dst[3 * v] -= mx; dst[3 * v + 1] -= my; dst[3 * v + 2] -= mz; }
        __syncthreads();
    }
    if (flags & SH_NORM_ZEROROOT) {                                                       // :31-32 root = J_regressor[0] . verts
        double s[3] = {0, 0, 0};
        for (int v = t; v < N; v += NB) {
            const double w = j_root[v];
            s[0] += w * dst[3 * v]; s[1] += w * dst[3 * v + 1]; s[2] += w * dst[3 * v + 2];
        }
        block_reduce3<0>(s, red);
        const float rx = (float)s[0], ry = (float)s[1], rz = (float)s[2];
        for (int v = t; v < N; v += NB) { dst[3 * v] -= rx; dst[3 * v + 1] -= ry; dst[3 * v + 2] -= rz; }
        __syncthreads();
    }
    if (flags & SH_NORM_ONELENGTH) {                                                      // :33-34 height (axis 1) -> 1.5
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (int v = t; v < N; v += NB) {
            const double y = dst[3 * v + 1];
            lo[1] = (y < lo[1] || y != y) ? y : lo[1]; hi[1] = (y > hi[1] || y != y) ? y : hi[1];
        }
        block_reduce3<1>(lo, red);
        block_reduce3<2>(hi, red);
        const float ext = (float)hi[1] - (float)lo[1];
        for (int i = t; i < N * 3; i += NB) dst[i] = dst[i] / ext * 1.5f;
        __syncthreads();
    }
    if (flags & SH_NORM_SMALL) {                                                          // :35-36
        for (int i = t; i < N * 3; i += NB) dst[i] = dst[i] / 1.5f;
        __syncthreads();
    }
    if (flags & SH_NORM_GASS) {                                                           // :37-39
        for (int i = t; i < N * 3; i += NB) dst[i] = (dst[i] - mean[i]) / stdv[i];
        __syncthreads();
    }
    if (flags & SH_NORM_NORMAL) {                                                         // :40-42
        const float* c = center + 3L * m;
        const float* sc = scale + 3L * m;
        for (int i = t; i < N * 3; i += NB) { const int d = i % 3; dst[i] = (dst[i] - c[d]) * sc[d]; }
        __syncthreads();
    }
    for (int i = t; i < N * 3; i += NB) { const float x = dst[i]; dst[i] = (x != x) ? 0.f : x; }   // :43 NaN -> 0
}

// out[j] = src[idx[j]] for rows of row_elems floats; float4 path when rows are 16-byte multiples
template <bool VEC4>
__global__ void gather_rows_kernel(const float* __restrict__ src, long row_elems, const int64_t* __restrict__ idx, int b,
                                   float* __restrict__ out) {
    const long per = VEC4 ? row_elems / 4 : row_elems;
    const long total = per * b;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long j = i / per, e = i - j * per;
        const long s = idx[j];
        if (VEC4)
            reinterpret_cast<f32x4*>(out)[j * per + e] = reinterpret_cast<const f32x4*>(src)[s * per + e];
        else
            out[j * per + e] = src[s * per + e];
    }
}

}  // namespace

extern "C" {

int sh_dataset_normalize(const float* raw, float* out, int n, int N, int dummy_rows, unsigned flags, const float* j_root,
                         const float* mean, const float* stdv, const float* center, const float* scale, sh_stream_t stream) {
    SH_REQUIRE(raw && out, SH_ERR_INVALID_ARG, "sh_dataset_normalize: null pointer");
    SH_REQUIRE(n > 0 && N > 0 && dummy_rows >= 0, SH_ERR_INVALID_ARG, "sh_dataset_normalize: bad size");
    SH_REQUIRE(!(flags & SH_NORM_ZEROROOT) || j_root, SH_ERR_INVALID_ARG, "sh_dataset_normalize: zeroroot needs J_regressor row 0");
    SH_REQUIRE(!(flags & SH_NORM_GASS) || (mean && stdv), SH_ERR_INVALID_ARG, "sh_dataset_normalize: gass needs mean and std");
    SH_REQUIRE(!(flags & SH_NORM_NORMAL) || (center && scale), SH_ERR_INVALID_ARG, "sh_dataset_normalize: normal needs center and scale");
    SH_REQUIRE((flags & ~0x3Fu) == 0, SH_ERR_INVALID_ARG, "sh_dataset_normalize: unknown flag bits 0x%x", flags);
    hipStream_t st = static_cast<hipStream_t>(stream);
    ShProfScope ps(st, "normalize_pack_kernel|n=%d N=%d flags=%u", n, N, flags);
    hipLaunchKernelGGL(normalize_pack_kernel, dim3(n), dim3(NB), 0, st, raw, out, N, dummy_rows, flags, j_root, mean, stdv, center,
                       scale);
    SH_CHECK_LAUNCH("dataset_normalize");
    return SH_OK;
}

int sh_gather_meshes(const float* src, int64_t row_elems, const int64_t* idx, int b, float* out, sh_stream_t stream) {
    SH_REQUIRE(src && idx && out, SH_ERR_INVALID_ARG, "sh_gather_meshes: null pointer");
    SH_REQUIRE(row_elems > 0 && b > 0, SH_ERR_INVALID_ARG, "sh_gather_meshes: bad size");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool vec = row_elems % 4 == 0 && (reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(out)) % 16 == 0;
    const long total = (vec ? row_elems / 4 : row_elems) * b;
    long grid = (total + 255) / 256;
    if (grid > 4096) grid = 4096;
    ShProfScope ps(st, "gather_rows_kernel<%s>|b=%d row=%ld", vec ? "true" : "false", b, (long)row_elems);
    if (vec)
        hipLaunchKernelGGL(gather_rows_kernel<true>, dim3((int)grid), dim3(256), 0, st, src, (long)row_elems, idx, b, out);
    else
        hipLaunchKernelGGL(gather_rows_kernel<false>, dim3((int)grid), dim3(256), 0, st, src, (long)row_elems, idx, b, out);
    SH_CHECK_LAUNCH("gather_meshes");
    return SH_OK;
}

}  // extern "C"
