// Body measurements on device: closed-polyline girths and bone lengths, batched over meshes
// (reference utils_SH.py:86-98 cal_length, :144-161 measure_body_quick).  Tiny, latency-bound
// work: one wavefront per (mesh, ring), one thread per (mesh, bone).
#include "sh_common.h"

namespace {

__device__ __forceinline__ void ring_point(const float* __restrict__ v, const int32_t* __restrict__ ra,
                                           const int32_t* __restrict__ rb, const float* __restrict__ rf, int i, float* q) {
    const float f = rf[i];
    const float* a = v + 3L * ra[i];
    const float* b = v + 3L * rb[i];
#pragma unroll
    for (int d = 0; d < 3; ++d) q[d] = a[d] * (1.f - f) + b[d] * f;          // utils_SH.py:155
}

// grid = B * P wavefronts (4 per block)
__global__ void girth_kernel(const float* __restrict__ v, long v_sb, const int32_t* __restrict__ ring_ptr,
                             const int32_t* __restrict__ ra, const int32_t* __restrict__ rb, const float* __restrict__ rf,
                             int B, int P, float* __restrict__ girth) {
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (wave >= B * P) return;
    const int b = wave / P, p = wave - b * P;
    const int beg = ring_ptr[p], n = ring_ptr[p + 1] - beg;
    const float* vb = v + (long)b * v_sb;
    float s = 0.f;
    // segment i joins point i and point (i+1) mod n; for n == 1 both are the same point (length 0),
    // for n == 2 the closing segment is counted as well, exactly like utils_SH.py:156-158
    for (int i = lane; i < n; i += 64) {
        const int j = (i + 1 == n) ? 0 : i + 1;
        float q0[3], q1[3];
        ring_point(vb, ra, rb, rf, beg + i, q0);
        ring_point(vb, ra, rb, rf, beg + j, q1);
        const float dx = q0[0] - q1[0], dy = q0[1] - q1[1], dz = q0[2] - q1[2];
        s += sqrtf(dx * dx + dy * dy + dz * dz);
    }
    s = sh_wave_sum(s);
    if (lane == 0) girth[wave] = s;
}

__global__ void bone_length_kernel(const float* __restrict__ kps, const int32_t* __restrict__ bones, int B, int K, int P,
                                   float* __restrict__ length) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * P) return;
    const int b = i / P, p = i - b * P;
    const float* k = kps + (long)b * K * 3;
    const int i0 = bones[3 * p], i1 = bones[3 * p + 1], i2 = bones[3 * p + 2];
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float tail = i2 >= 0 ? (k[3 * i1 + d] + k[3 * i2 + d]) / 2.f : k[3 * i1 + d];
        const float e = k[3 * i0 + d] - tail;
        s += e * e;
    }
    length[i] = sqrtf(s);
}

}  // namespace

extern "C" {

int sh_measure_girth(const float* v, int64_t v_sb, const int32_t* ring_ptr, const int32_t* ring_a, const int32_t* ring_b,
                     const float* ring_f, int B, int P, float* girth, sh_stream_t stream) {
    SH_REQUIRE(v && ring_ptr && ring_a && ring_b && ring_f && girth, SH_ERR_INVALID_ARG, "sh_measure_girth: null pointer");
    SH_REQUIRE(B > 0 && P > 0 && v_sb > 0, SH_ERR_INVALID_ARG, "sh_measure_girth: non-positive size");
    SH_REQUIRE((long)B * P < (1L << 30), SH_ERR_UNSUPPORTED, "sh_measure_girth: B*P too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
    ShProfScope ps(st, "girth_kernel|B=%d P=%d", B, P);
    hipLaunchKernelGGL(girth_kernel, dim3(sh_cdiv(B * P, 4)), dim3(256), 0, st, v, (long)v_sb, ring_ptr, ring_a, ring_b, ring_f,
                       B, P, girth);
    SH_CHECK_LAUNCH("measure_girth");
    return SH_OK;
}

int sh_bone_length(const float* kps, const int32_t* bones, int B, int K, int P, float* length, sh_stream_t stream) {
    SH_REQUIRE(kps && bones && length, SH_ERR_INVALID_ARG, "sh_bone_length: null pointer");
    SH_REQUIRE(B > 0 && K > 0 && P > 0, SH_ERR_INVALID_ARG, "sh_bone_length: non-positive size");
    SH_REQUIRE((long)B * P < (1L << 30), SH_ERR_UNSUPPORTED, "sh_bone_length: B*P too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
    ShProfScope ps(st, "bone_length_kernel|B=%d P=%d", B, P);
    hipLaunchKernelGGL(bone_length_kernel, dim3(sh_cdiv(B * P, 256)), dim3(256), 0, st, kps, bones, B, K, P, length);
    SH_CHECK_LAUNCH("bone_length");
    return SH_OK;
}

}  // extern "C"
