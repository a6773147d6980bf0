// Part-wise pairwise-distance loss of the semantic training loop (reference
// train_funcs.py:243-284 / :353-389 with utils_distance.calc_euclidean_dist_matrix (:366-376) and
// utils_SH.angle_skl (:442-478)).
//
// For every body part p, batch entry b and ordered vertex pair (i, j), i != j, of the part:
//     De   = |g_i - g_j| * scale[b,p]            ground-truth distance (scaled when the part is edited)
//     De_r = |r_i - r_j|                         reconstructed distance
//     w    = weight from the angle (degrees) between (g_i - g_j) and the part's bone direction:
//            all_one | angle/90 | sin(angle) | angle/90 thresholded;  parts flagged "leaf" use w = 1
//     pairs with w * De == 0 are dropped (reference `nozero_index`)
//     term = |w * De_r / De - w|   (relative form)     or   |w * De_r - w * De|   (absolute form)
//     loss = sum_p w_part[p] * mean_{kept pairs of p, all b} term
// The reference materialises [B, n, n, 3] direction tensors and [B, n, n] distance / weight matrices
// per part; here a workgroup stages the part's coordinates in LDS once and every thread sweeps one
// row of pairs in registers.  Partial sums are reduced in a fixed order (no atomics).
#include "sh_common.h"

namespace {

constexpr int PT = 32;      // rows (i) per workgroup
constexpr int JS = 4;       // threads per row: each sweeps every JS-th partner j (a part has ~400 vertices and a batch only
                            // ~2000 row tiles of 128: one thread per row left the chip at < 2 waves per SIMD)
constexpr int PNT = PT * JS;   // threads per workgroup

struct PLParams {
    const float* xr; const float* xg;        // [B][N1][3]
    const float* bone;                       // [B][P][3]
    const float* scale;                      // [B][P] or null
    const int* part_ptr; const int* part_vert;
    const int* tile_ptr;                     // [P+1] cumulative row tiles per part
    const int* flags;                        // [P] bit0: weights are all one (leaf parts)
    const float* w_part;                     // [P]
    int B, N1, P, T;                         // T = tile_ptr[P]
    int w_mode; float thr; int relat;
};

__device__ __forceinline__ float pair_weight(float vx, float vy, float vz, float d, float kx, float ky, float kz, float kn, int mode,
                                             float thr, bool all_one) {
    if (all_one || mode == 0) return 1.f;
    float c = fabsf((vx * kx + vy * ky + vz * kz) / (d * kn));
    c = (c != c) ? 1.f : c;                  // NaN -> 1   (utils_SH.py:459)
    c = fminf(fmaxf(c, 0.f), 1.f);
    const float ang = acosf(c) * 57.29577951308232f;
    if (mode == 1) return ang / 90.f;                                   // 'linear'
    if (mode == 2) return sinf(ang / 180.f * 3.14159265358979f);        // 'sin'
    const float w = ang / 90.f;                                         // 'threshold'
    return w < thr ? 0.f : w;
}

__device__ __forceinline__ int find_part(const int* tile_ptr, int P, int t) {
    int p = 0;
    while (p + 1 < P && tile_ptr[p + 1] <= t) ++p;
    return p;
}

// forward: partial[(b*T + t)*2 + {0,1}] = (sum of terms, number of kept pairs) of the block's rows.
// GRAD: the sweep also leaves the backward pass's row sums  graw[b][v][:] = sum_j coef_ij (r_i - r_j) / |r_i - r_j|  - everything
// the gradient needs except the factor 2 gscale w_p / count_p, which exists only once every row of the part is done: the
// backward pass is then one scaling launch instead of a second sweep over the 44.6 M pairs (acosf, two square roots and
// three divisions each).  The sums are formed exactly as pairdist_bwd_kernel forms them (same expressions, same order), so
// scale(graw) is that kernel's output bit for bit.
// (One kernel whether graw is wanted or not: two instantiations summed the loss to different last bits.)
__global__ __launch_bounds__(PNT) void pairdist_fwd_kernel(const PLParams q, float* __restrict__ partial, float* __restrict__ graw) {
    const bool GRAD = graw != nullptr;                              // wave-uniform: validation / no-grad calls skip the gradient arithmetic
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x / q.T, t = blockIdx.x - b * q.T;
    const int p = find_part(q.tile_ptr, q.P, t);
    const int v0 = q.part_ptr[p], n = q.part_ptr[p + 1] - v0;
    float* G = sm;            // [n][3]
    float* R = sm + 3 * n;    // [n][3]
    for (int i = threadIdx.x; i < n; i += PNT) {
        const long o = ((long)b * q.N1 + q.part_vert[v0 + i]) * 3;
        G[3 * i] = q.xg[o]; G[3 * i + 1] = q.xg[o + 1]; G[3 * i + 2] = q.xg[o + 2];
        R[3 * i] = q.xr[o]; R[3 * i + 1] = q.xr[o + 1]; R[3 * i + 2] = q.xr[o + 2];
    }
    __syncthreads();
    const float kx = q.bone[((long)b * q.P + p) * 3], ky = q.bone[((long)b * q.P + p) * 3 + 1], kz = q.bone[((long)b * q.P + p) * 3 + 2];
    const float kn = sqrtf(kx * kx + ky * ky + kz * kz);
    const float sc = q.scale ? q.scale[(long)b * q.P + p] : 1.f;
    const bool all_one = q.flags[p] & 1;
    const int i = (t - q.tile_ptr[p]) * PT + (threadIdx.x / JS), seg = threadIdx.x % JS;
    float s = 0.f, cnt = 0.f;
    float ax = 0.f, ay = 0.f, az = 0.f;
    if (i < n) {
        const float gx = G[3 * i], gy = G[3 * i + 1], gz = G[3 * i + 2];
        const float rx = R[3 * i], ry = R[3 * i + 1], rz = R[3 * i + 2];
        for (int j = seg; j < n; j += JS) {
            if (j == i) continue;
            const float vx = gx - G[3 * j], vy = gy - G[3 * j + 1], vz = gz - G[3 * j + 2];
            const float d = sqrtf(vx * vx + vy * vy + vz * vz);
            const float w = pair_weight(vx, vy, vz, d, kx, ky, kz, kn, q.w_mode, q.thr, all_one);
            const float De = d * sc;
            if (w * De == 0.f) continue;
            const float ux = rx - R[3 * j], uy = ry - R[3 * j + 1], uz = rz - R[3 * j + 2];
            const float Dr = sqrtf(ux * ux + uy * uy + uz * uz);
            const float e = q.relat ? (w * Dr / De - w) : (w * Dr - w * De);      // one evaluation for the term and its sign
            s += fabsf(e);
            cnt += 1.f;
            if (GRAD) {                                              // pairdist_bwd_kernel's statements
                if (Dr == 0.f) continue;
                const float sg = e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f);
                const float c = sg * (q.relat ? w / De : w) / Dr;
                ax += c * ux; ay += c * uy; az += c * uz;
            }
        }
    }
    if (GRAD) {
#pragma unroll
        for (int m = 1; m < JS; m <<= 1) {                           // the row's JS partial sums (adjacent lanes), fixed order
            ax += __shfl_xor(ax, m, 64); ay += __shfl_xor(ay, m, 64); az += __shfl_xor(az, m, 64);
        }
        if (graw != nullptr && i < n && seg == 0) {
            const long o = ((long)b * q.N1 + q.part_vert[v0 + i]) * 3;
            graw[o] = ax; graw[o + 1] = ay; graw[o + 2] = az;
        }
    }
    __shared__ float red[2][PNT / 64];
    s = sh_wave_sum(s); cnt = sh_wave_sum(cnt);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, c = 0.f;
        for (int w = 0; w < PNT / 64; ++w) { a += red[0][w]; c += red[1][w]; }
        partial[((long)b * q.T + t) * 2] = a;
        partial[((long)b * q.T + t) * 2 + 1] = c;
    }
}

// part_sum[p], part_cnt[p] = fixed-order sums over the part's (b, tile) partials; loss = sum_p w_p * sum/cnt
// (The partials are staged in LDS by the whole workgroup first when they fit: one thread per part walking them in global
// memory was a chain of ~200 dependent round trips, 34 us for 28 KB.  Same sums in the same order.)
constexpr int PD_STAGE_FLOATS = 15 * 1024;
__global__ __launch_bounds__(256) void pairdist_final_kernel(const float* __restrict__ partial, const int* __restrict__ tile_ptr,
                                                             const float* __restrict__ w_part, int B, int P, int T,
                                                             float* __restrict__ part_sum, float* __restrict__ part_cnt,
                                                             float* __restrict__ loss) {
    __shared__ float ls[64];
    __shared__ float stage[PD_STAGE_FLOATS];
    const long n = (long)B * T * 2;
    const bool staged = n <= PD_STAGE_FLOATS;
    if (staged) {
        for (int i = threadIdx.x; i < n; i += 256) stage[i] = partial[i];
        __syncthreads();
    }
    const float* src = staged ? stage : partial;
    const int p = threadIdx.x;
    float contrib = 0.f;
    if (p < P) {
        double s = 0.0, c = 0.0;
        for (int b = 0; b < B; ++b)
            for (int t = tile_ptr[p]; t < tile_ptr[p + 1]; ++t) {
                s += (double)src[((long)b * T + t) * 2];
                c += (double)src[((long)b * T + t) * 2 + 1];
            }
        part_sum[p] = (float)s;
        part_cnt[p] = (float)c;
        contrib = c > 0.0 ? (float)(w_part[p] * s / c) : 0.f;
    }
    if (threadIdx.x < 64) ls[threadIdx.x] = contrib;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int k = 0; k < P; ++k) tot += ls[k];
        loss[0] = tot;
    }
}

// backward w.r.t. the reconstruction: both (i,j) and (j,i) are terms of the loss and are equal, so
// d loss / d r_i = 2 * sum_j coef_ij * (r_i - r_j) / |r_i - r_j|
__global__ __launch_bounds__(PNT) void pairdist_bwd_kernel(const PLParams q, const float* __restrict__ part_cnt,
                                                          const float* __restrict__ gscale, float* __restrict__ grad) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x / q.T, t = blockIdx.x - b * q.T;
    const int p = find_part(q.tile_ptr, q.P, t);
    const int v0 = q.part_ptr[p], n = q.part_ptr[p + 1] - v0;
    float* G = sm;
    float* R = sm + 3 * n;
    for (int i = threadIdx.x; i < n; i += PNT) {
        const long o = ((long)b * q.N1 + q.part_vert[v0 + i]) * 3;
        G[3 * i] = q.xg[o]; G[3 * i + 1] = q.xg[o + 1]; G[3 * i + 2] = q.xg[o + 2];
        R[3 * i] = q.xr[o]; R[3 * i + 1] = q.xr[o + 1]; R[3 * i + 2] = q.xr[o + 2];
    }
    __syncthreads();
    const float kx = q.bone[((long)b * q.P + p) * 3], ky = q.bone[((long)b * q.P + p) * 3 + 1], kz = q.bone[((long)b * q.P + p) * 3 + 2];
    const float kn = sqrtf(kx * kx + ky * ky + kz * kz);
    const float sc = q.scale ? q.scale[(long)b * q.P + p] : 1.f;
    const bool all_one = q.flags[p] & 1;
    const float cnt = part_cnt[p];
    const int i = (t - q.tile_ptr[p]) * PT + (threadIdx.x / JS), seg = threadIdx.x % JS;
    if (i >= n) return;                                             // the JS threads of a row leave together
    const float norm = cnt > 0.f ? 2.f * gscale[0] * q.w_part[p] / cnt : 0.f;
    const float gx = G[3 * i], gy = G[3 * i + 1], gz = G[3 * i + 2];
    const float rx = R[3 * i], ry = R[3 * i + 1], rz = R[3 * i + 2];
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (int j = seg; j < n; j += JS) {
        if (j == i) continue;
        const float vx = gx - G[3 * j], vy = gy - G[3 * j + 1], vz = gz - G[3 * j + 2];
        const float d = sqrtf(vx * vx + vy * vy + vz * vz);
        const float w = pair_weight(vx, vy, vz, d, kx, ky, kz, kn, q.w_mode, q.thr, all_one);
        const float De = d * sc;
        if (w * De == 0.f) continue;
        const float ux = rx - R[3 * j], uy = ry - R[3 * j + 1], uz = rz - R[3 * j + 2];
        const float Dr = sqrtf(ux * ux + uy * uy + uz * uz);
        if (Dr == 0.f) continue;                                   // |.|' undefined at 0: no contribution
        const float e = q.relat ? (w * Dr / De - w) : (w * Dr - w * De);
        const float sg = e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f);
        const float c = sg * (q.relat ? w / De : w) / Dr;
        ax += c * ux; ay += c * uy; az += c * uz;
    }
#pragma unroll
    for (int m = 1; m < JS; m <<= 1) {                               // the row's JS partial sums (adjacent lanes), fixed order
        ax += __shfl_xor(ax, m, 64); ay += __shfl_xor(ay, m, 64); az += __shfl_xor(az, m, 64);
    }
    if (seg != 0) return;
    const long o = ((long)b * q.N1 + q.part_vert[v0 + i]) * 3;
    grad[o] = norm * ax; grad[o + 1] = norm * ay; grad[o + 2] = norm * az;
}

// backward from the forward pass's row sums: grad[b][v][:] = (2 gscale w_p / count_p) * graw[b][v][:] for the vertices of part p
// (rows outside every part stay zero: the caller cleared grad)
__global__ __launch_bounds__(256) void pairdist_scale_kernel(const float* __restrict__ graw, const int* __restrict__ part_ptr,
                                                            const int* __restrict__ part_vert, const float* __restrict__ w_part,
                                                            const float* __restrict__ part_cnt, const float* __restrict__ gscale, int B,
                                                            int N1, int P, float* __restrict__ grad) {
    const int nv = part_ptr[P];
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)B * nv) return;
    const int b = (int)(t / nv), k = (int)(t - (long)b * nv);
    int p = 0;
    while (p + 1 < P && part_ptr[p + 1] <= k) ++p;
    const float cnt = part_cnt[p];
    const float norm = cnt > 0.f ? 2.f * gscale[0] * w_part[p] / cnt : 0.f;
    const long o = ((long)b * N1 + part_vert[k]) * 3;
    grad[o] = norm * graw[o]; grad[o + 1] = norm * graw[o + 1]; grad[o + 2] = norm * graw[o + 2];
}

int fill(PLParams& q, const float* x_rec, const float* x_gt, const float* bone, const float* scale, const int32_t* part_ptr,
         const int32_t* part_vert, const int32_t* tile_ptr, const int32_t* flags, const float* w_part, int B, int N1, int P,
         int T, int w_mode, float thr, int relat) {
    SH_REQUIRE(x_rec && x_gt && bone && part_ptr && part_vert && tile_ptr && flags && w_part, SH_ERR_INVALID_ARG,
               "sh_part_pairdist_loss: null pointer");
    SH_REQUIRE(B > 0 && N1 > 0 && P > 0 && P <= 64 && T > 0, SH_ERR_INVALID_ARG, "sh_part_pairdist_loss: bad size (P must be <= 64)");
    SH_REQUIRE(w_mode >= 0 && w_mode <= 3, SH_ERR_INVALID_ARG, "sh_part_pairdist_loss: unknown weight mode %d", w_mode);
    q.xr = x_rec; q.xg = x_gt; q.bone = bone; q.scale = scale; q.part_ptr = part_ptr; q.part_vert = part_vert;
    q.tile_ptr = tile_ptr; q.flags = flags; q.w_part = w_part; q.B = B; q.N1 = N1; q.P = P; q.T = T;
    q.w_mode = w_mode; q.thr = thr; q.relat = relat;
    return SH_OK;
}

}  // namespace

extern "C" {

int sh_part_pairdist_tile_rows(void) { return PT; }

int sh_part_pairdist_loss_fwd(const float* x_rec, const float* x_gt, const float* bone, const float* scale, const int32_t* part_ptr,
                              const int32_t* part_vert, const int32_t* tile_ptr, const int32_t* flags, const float* w_part, int B,
                              int N1, int P, int T, int max_part, int w_mode, float w_threshold, int relat, float* loss,
                              float* part_sum, float* part_cnt, void* workspace, size_t workspace_bytes, sh_stream_t stream) {
    return sh_part_pairdist_loss_fwd_grad(x_rec, x_gt, bone, scale, part_ptr, part_vert, tile_ptr, flags, w_part, B, N1, P, T, max_part, w_mode,
                                          w_threshold, relat, loss, part_sum, part_cnt, nullptr, workspace, workspace_bytes, stream);
}

int sh_part_pairdist_loss_fwd_grad(const float* x_rec, const float* x_gt, const float* bone, const float* scale, const int32_t* part_ptr,
                                   const int32_t* part_vert, const int32_t* tile_ptr, const int32_t* flags, const float* w_part, int B,
                                   int N1, int P, int T, int max_part, int w_mode, float w_threshold, int relat, float* loss,
                                   float* part_sum, float* part_cnt, float* grad_raw, void* workspace, size_t workspace_bytes,
                                   sh_stream_t stream) {
    PLParams q{};
    const int rc = fill(q, x_rec, x_gt, bone, scale, part_ptr, part_vert, tile_ptr, flags, w_part, B, N1, P, T, w_mode, w_threshold, relat);
    if (rc != SH_OK) return rc;
    SH_REQUIRE(loss && part_sum && part_cnt && workspace, SH_ERR_INVALID_ARG, "sh_part_pairdist_loss_fwd: null output");
    SH_REQUIRE(workspace_bytes >= (size_t)B * T * 2 * sizeof(float), SH_ERR_WORKSPACE, "sh_part_pairdist_loss_fwd: workspace too small");
    SH_REQUIRE(max_part > 0 && (size_t)max_part * 24 <= 160 * 1024, SH_ERR_UNSUPPORTED, "sh_part_pairdist_loss: part of %d vertices exceeds LDS", max_part);
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* partial = static_cast<float*>(workspace);
    {
        ShProfScope ps(st, "pairdist_fwd_kernel|B=%d T=%d grad=%d", B, T, grad_raw ? 1 : 0);
        hipLaunchKernelGGL(pairdist_fwd_kernel, dim3((unsigned)(B * T)), dim3(PNT), (size_t)max_part * 24, st, q, partial, grad_raw);
    }
    hipLaunchKernelGGL(pairdist_final_kernel, dim3(1), dim3(256), 0, st, partial, tile_ptr, w_part, B, P, T, part_sum, part_cnt, loss);
    SH_CHECK_LAUNCH("part_pairdist_loss_fwd");
    return SH_OK;
}

int sh_part_pairdist_loss_bwd(const float* x_rec, const float* x_gt, const float* bone, const float* scale, const int32_t* part_ptr,
                              const int32_t* part_vert, const int32_t* tile_ptr, const int32_t* flags, const float* w_part, int B,
                              int N1, int P, int T, int max_part, int w_mode, float w_threshold, int relat, const float* part_cnt,
                              const float* gscale, float* grad, sh_stream_t stream) {
    PLParams q{};
    const int rc = fill(q, x_rec, x_gt, bone, scale, part_ptr, part_vert, tile_ptr, flags, w_part, B, N1, P, T, w_mode, w_threshold, relat);
    if (rc != SH_OK) return rc;
    SH_REQUIRE(part_cnt && gscale && grad, SH_ERR_INVALID_ARG, "sh_part_pairdist_loss_bwd: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(grad, 0, (size_t)B * N1 * 3 * sizeof(float), st) != hipSuccess) {
        sh_set_error("sh_part_pairdist_loss_bwd: memset failed");
        return SH_ERR_LAUNCH;
    }
    {
        ShProfScope ps(st, "pairdist_bwd_kernel|B=%d T=%d", B, T);
        hipLaunchKernelGGL(pairdist_bwd_kernel, dim3((unsigned)(B * T)), dim3(PNT), (size_t)max_part * 24, st, q, part_cnt, gscale, grad);
    }
    SH_CHECK_LAUNCH("part_pairdist_loss_bwd");
    return SH_OK;
}

int sh_part_pairdist_loss_bwd_scale(const float* grad_raw, const int32_t* part_ptr, const int32_t* part_vert, const float* w_part,
                                    const float* part_cnt, const float* gscale, int B, int N1, int P, int n_part_verts, float* grad,
                                    sh_stream_t stream) {
    SH_REQUIRE(grad_raw && part_ptr && part_vert && w_part && part_cnt && gscale && grad && B > 0 && N1 > 0 && P > 0 && n_part_verts > 0,
               SH_ERR_INVALID_ARG, "sh_part_pairdist_loss_bwd_scale: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(grad, 0, (size_t)B * N1 * 3 * sizeof(float), st) != hipSuccess) {
        sh_set_error("sh_part_pairdist_loss_bwd_scale: memset failed");
        return SH_ERR_LAUNCH;
    }
    const long n = (long)B * n_part_verts;
    hipLaunchKernelGGL(pairdist_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, grad_raw, part_ptr, part_vert, w_part, part_cnt,
                       gscale, B, N1, P, grad);
    SH_CHECK_LAUNCH("part_pairdist_loss_bwd_scale");
    return SH_OK;
}

}  // extern "C"
