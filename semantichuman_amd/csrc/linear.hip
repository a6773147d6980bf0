// Skinny dense layers (the latent FCs of the autoencoder, models.py:85-86,130,144) on CDNA4.
//
// The two latent nn.Linear layers are [B,55296]x[55296,256] and [B,256]x[256,55296] with B = 64:
// one dimension is the batch (tiny), one is huge, and every one of the six GEMMs of a training
// step (forward, dgrad, wgrad of each layer) streams a 56.6 MB weight-shaped matrix exactly once.
// They are HBM-bound (2 x 64 FLOP per weight byte/4); the vendor GEMM runs them at ~8 TFLOP/s
// (230 us for the split-K-less K = 55296 case).  One generic kernel covers all six:
//
//     C(m,n) = sum_r A(m,r) * B(n,r)            A, B, C addressed through explicit strides
//
// 64x64 output tile per workgroup (4 waves x [16 rows x 64 cols]), reduction chunks of 32 staged
// through LDS in the same swizzled [row][32] image as the spiral-conv kernel (16-B loads along
// whichever of the two dimensions is contiguous, transposing on the way into LDS if needed),
// v_mfma_f32_16x16x4_f32, optional split of the reduction over workgroups with partial slabs
// summed in a fixed order by a second kernel (deterministic, no atomics).
#include "sh_common.h"

namespace {

constexpr int LT = 64;      // tile rows (both operands)
constexpr int LK = 32;      // reduction chunk
constexpr int LTHREADS = 256;

struct SGParams {
    const float* a; long a_sm, a_sr;     // A(m,r) = a[m*a_sm + r*a_sr]
    const float* b; long b_sn, b_sr;     // B(n,r) = b[n*b_sn + r*b_sr]
    float* c; long c_sm, c_sn;           // C(m,n)
    const float* bias;                   // [N] added to C (only when nsplit == 1)
    float* slab;                         // [nsplit][M][N] partials (nsplit > 1)
    int M, N, R;                         // R = reduction length
    int rchunk, nsplit, n_mtiles, n_ntiles;
    int a_mode, b_mode;                  // 0: reduction index contiguous, 1: row index contiguous, 2: scalar
};

// stage a [64][32] tile of X(row, r) into registers: 2 float4 per thread
__device__ __forceinline__ void sg_load(const float* base, long s_row, long s_r, int mode, int row0, int nrows, int r0, int rend,
                                        int tid, f32x4 (&reg)[2], unsigned& mask) {
    mask = 0;
    if (mode == 0) {             // r contiguous: thread -> (row = tid>>3 (+32), quad = tid&7)
        const int q = tid & 7;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (tid >> 3) + 32 * i;
            const bool ok = row0 + row < nrows && r0 + 4 * q < rend;
            const long off = ok ? (long)(row0 + row) * s_row + (r0 + 4 * q) : 0;
            reg[i] = *reinterpret_cast<const f32x4*>(base + off);
            mask |= (ok ? 1u : 0u) << i;
        }
    } else if (mode == 1) {      // row contiguous: thread -> (r = tid>>4 (+16), row quad = tid&15)
        const int rq = tid & 15;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = (tid >> 4) + 16 * i;
            const bool ok = row0 + 4 * rq < nrows && r0 + r < rend;
            const long off = ok ? (long)(r0 + r) * s_r + (row0 + 4 * rq) : 0;
            reg[i] = *reinterpret_cast<const f32x4*>(base + off);
            mask |= (ok ? 1u : 0u) << i;
        }
    } else {                     // generic strides / ragged sizes: scalar, same mapping as mode 0
        const int q = tid & 7;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (tid >> 3) + 32 * i;
            reg[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (row0 + row < nrows && r0 + 4 * q + j < rend) reg[i][j] = base[(long)(row0 + row) * s_row + (long)(r0 + 4 * q + j) * s_r];
            mask |= 1u << i;
        }
    }
}

// write the staged registers into the swizzled LDS image  T[row][32]: quad q of row r at q ^ (r & 7)
__device__ __forceinline__ void sg_store(float* T, int mode, int tid, const f32x4 (&reg)[2], unsigned mask) {
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    if (mode != 1) {
        const int q = tid & 7;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (tid >> 3) + 32 * i;
            *reinterpret_cast<f32x4*>(T + row * LK + ((q ^ (row & 7)) << 2)) = ((mask >> i) & 1u) ? reg[i] : zero4;
        }
    } else {
        const int rq = tid & 15;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = (tid >> 4) + 16 * i;
            const f32x4 v = ((mask >> i) & 1u) ? reg[i] : zero4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = 4 * rq + j;
                T[row * LK + ((((r >> 2) ^ (row & 7)) << 2) | (r & 3))] = v[j];
            }
        }
    }
}

__global__ __launch_bounds__(LTHREADS) void skinny_gemm_kernel(const SGParams p) {
    __shared__ __attribute__((aligned(16))) float As[2][LT * LK];
    __shared__ __attribute__((aligned(16))) float Bs[2][LT * LK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // tile order: m fastest, then n, then split - workgroups sharing a B column tile are adjacent
    const int lin = sh_xcd_remap(blockIdx.x, gridDim.x);
    const int mt = lin % p.n_mtiles;
    const int nt = (lin / p.n_mtiles) % p.n_ntiles;
    const int sp = lin / (p.n_mtiles * p.n_ntiles);
    const int m0 = mt * LT, n0 = nt * LT;
    const int r_begin = sp * p.rchunk;
    const int r_end = min(p.R, r_begin + p.rchunk);
    const int nchunks = (r_end - r_begin + LK - 1) / LK;

    f32x4 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lrow = lane & 15, lq = lane >> 4;

    f32x4 ra[2], rb[2];
    unsigned ma, mb;
    if (nchunks > 0) {
        sg_load(p.a, p.a_sm, p.a_sr, p.a_mode, m0, p.M, r_begin, r_end, tid, ra, ma);
        sg_load(p.b, p.b_sn, p.b_sr, p.b_mode, n0, p.N, r_begin, r_end, tid, rb, mb);
        sg_store(As[0], p.a_mode, tid, ra, ma);
        sg_store(Bs[0], p.b_mode, tid, rb, mb);
    }
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        const int rn = r_begin + (c + 1 < nchunks ? c + 1 : c) * LK;       // clamped prefetch
        sg_load(p.a, p.a_sm, p.a_sr, p.a_mode, m0, p.M, rn, r_end, tid, ra, ma);
        sg_load(p.b, p.b_sn, p.b_sr, p.b_mode, n0, p.N, rn, r_end, tid, rb, mb);
        __builtin_amdgcn_sched_barrier(0);
        const float* Ab = As[buf] + (16 * wave + lrow) * LK;
        const float* Bb = Bs[buf] + lrow * LK;
#pragma unroll
        for (int ks = 0; ks < LK / 16; ++ks) {
            const int pq = ((lq + 4 * ks) ^ (lane & 7)) << 2;
            const f32x4 g = *reinterpret_cast<const f32x4*>(Ab + pq);
            f32x4 wq[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) wq[n] = *reinterpret_cast<const f32x4*>(Bb + n * 16 * LK + pq);
            // t outer, n inner: consecutive MFMAs go to different accumulators (the 16x16x4 form has a
            // 40-cycle dependent latency against a 32-cycle issue interval)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[n][t], g[t], acc[n], 0, 0, 0);
        }
        sg_store(As[buf ^ 1], p.a_mode, tid, ra, ma);
        sg_store(Bs[buf ^ 1], p.b_mode, tid, rb, mb);
        __syncthreads();
    }

    // lane holds C(m, n..n+3): m = m0 + 16*wave + lrow, n = n0 + 16*nn + 4*lq
    const int m = m0 + 16 * wave + lrow;
    if (m >= p.M) return;
    if (p.nsplit > 1) {
        float* dst = p.slab + ((long)sp * p.M + m) * p.N;
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) {
            const int n = n0 + 16 * nn + 4 * lq;
            if ((p.N & 3) == 0) {
                if (n < p.N) *reinterpret_cast<f32x4*>(dst + n) = acc[nn];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (n + j < p.N) dst[n + j] = acc[nn][j];
            }
        }
        return;
    }
    const bool vec = p.c_sn == 1 && (p.c_sm & 3) == 0 && (p.N & 3) == 0 && (reinterpret_cast<uintptr_t>(p.c) & 15) == 0;
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) {
        const int n = n0 + 16 * nn + 4 * lq;
        f32x4 v = acc[nn];
        if (p.bias) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (n + j < p.N) v[j] += p.bias[n + j];
        }
        if (vec) {
            if (n < p.N) *reinterpret_cast<f32x4*>(p.c + (long)m * p.c_sm + n) = v;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (n + j < p.N) p.c[(long)m * p.c_sm + (long)(n + j) * p.c_sn] = v[j];
        }
    }
}

// c[m*N + n] = sum_s slab[s][m][n] (+ bias[n]); fixed order -> deterministic
// Block = 64 outputs x 16 slab lanes (lane j sums slabs j, j+16, ... in order; the 16 lane sums are
// then combined in order): the slab loop is 16x shorter than one-thread-per-output.
__global__ __launch_bounds__(1024) void split_reduce_kernel(const float* __restrict__ slab, int nsplit, long mn, int N,
                                                            const float* __restrict__ bias, float* __restrict__ c) {
    __shared__ float red[16][64];
    const int ox = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + ox;
    float s = 0.f;
    if (i < mn) {
        int k = ry;
        for (; k + 48 < nsplit; k += 64) {
            const float a = slab[(long)k * mn + i], b = slab[(long)(k + 16) * mn + i];
            const float d = slab[(long)(k + 32) * mn + i], e = slab[(long)(k + 48) * mn + i];
            s += (a + b) + (d + e);
        }
        for (; k < nsplit; k += 16) s += slab[(long)k * mn + i];
    }
    red[ry][ox] = s;
    __syncthreads();
    if (ry == 0 && i < mn) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][ox];
        c[i] = t + (bias ? bias[i % N] : 0.f);
    }
}

// out[n] = sum_m x[m*N + n]   (bias gradient)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int M, int N, float* __restrict__ out) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += x[(long)m * N + n];
    out[n] = s;
}

int mode_of(const float* base, long s_row, long s_r, int rows, int R) {
    const bool aligned = (reinterpret_cast<uintptr_t>(base) & 15) == 0;
    if (s_r == 1 && (s_row & 3) == 0 && (R & 3) == 0 && aligned) return 0;
    if (s_row == 1 && (s_r & 3) == 0 && (rows & 3) == 0 && aligned) return 1;
    return 2;
}

struct SGPlan { int nsplit, rchunk; };
SGPlan plan_split(int M, int N, int R) {
    const int tiles = sh_cdiv(M, LT) * sh_cdiv(N, LT);
    SGPlan pl{1, R};
    if (tiles < 512 && R >= 2048) {
        int ns = sh_cdiv(1024, tiles);
        int rc = sh_cdiv(sh_cdiv(R, ns), LK) * LK;
        if (rc < 256) rc = 256;
        pl.rchunk = rc;
        pl.nsplit = sh_cdiv(R, rc);
    }
    return pl;
}

int run_gemm(SGParams& p, void* ws, size_t ws_bytes, hipStream_t st, const char* what) {
    const SGPlan pl = plan_split(p.M, p.N, p.R);
    p.nsplit = pl.nsplit; p.rchunk = pl.rchunk;
    p.n_mtiles = sh_cdiv(p.M, LT); p.n_ntiles = sh_cdiv(p.N, LT);
    p.a_mode = mode_of(p.a, p.a_sm, p.a_sr, p.M, p.R);
    p.b_mode = mode_of(p.b, p.b_sn, p.b_sr, p.N, p.R);
    const float* bias = p.bias;
    if (p.nsplit > 1) {
        SH_REQUIRE(p.c_sn == 1 && p.c_sm == p.N, SH_ERR_UNSUPPORTED, "%s: split reduction needs a contiguous output", what);
        SH_REQUIRE(ws && ws_bytes >= (size_t)p.nsplit * p.M * p.N * sizeof(float), SH_ERR_WORKSPACE, "%s: workspace too small", what);
        p.slab = static_cast<float*>(ws);
        p.bias = nullptr;
    }
    const long grid = (long)p.n_mtiles * p.n_ntiles * p.nsplit;
    {
        ShProfScope ps(st, "skinny_gemm_kernel|%s M=%d N=%d R=%d split=%d modes=%d%d", what, p.M, p.N, p.R, p.nsplit, p.a_mode, p.b_mode);
        hipLaunchKernelGGL(skinny_gemm_kernel, dim3((unsigned)grid), dim3(LTHREADS), 0, st, p);
    }
    if (p.nsplit > 1) {
        const long mn = (long)p.M * p.N;
        ShProfScope ps(st, "split_reduce_kernel");
        hipLaunchKernelGGL(split_reduce_kernel, dim3((unsigned)((mn + 63) / 64)), dim3(1024), 0, st, p.slab, p.nsplit, mn, p.N,
                           bias, p.c);
    }
    SH_CHECK_LAUNCH(what);
    return SH_OK;
}

}  // namespace

extern "C" {

size_t sh_linear_workspace(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    size_t need = 0;
    const int dims[3][3] = {{M, N, K}, {M, K, N}, {N, K, M}};      // fwd, bwd_data, bwd_wgt as (rows, cols, reduction)
    for (auto& d : dims) {
        const SGPlan pl = plan_split(d[0], d[1], d[2]);
        if (pl.nsplit > 1) {
            const size_t b = (size_t)pl.nsplit * d[0] * d[1] * sizeof(float);
            if (b > need) need = b;
        }
    }
    return need;
}

int sh_linear_fwd(const float* x, const float* weight, const float* bias, float* y, int M, int N, int K, void* workspace,
                  size_t workspace_bytes, sh_stream_t stream) {
    SH_REQUIRE(x && weight && y && M > 0 && N > 0 && K > 0, SH_ERR_INVALID_ARG, "sh_linear_fwd: bad argument");
    SGParams p{};
    p.a = x; p.a_sm = K; p.a_sr = 1;
    p.b = weight; p.b_sn = K; p.b_sr = 1;
    p.c = y; p.c_sm = N; p.c_sn = 1; p.bias = bias;
    p.M = M; p.N = N; p.R = K;
    return run_gemm(p, workspace, workspace_bytes, static_cast<hipStream_t>(stream), "linear_fwd");
}

int sh_linear_bwd_data(const float* dy, const float* weight, float* dx, int M, int N, int K, void* workspace,
                       size_t workspace_bytes, sh_stream_t stream) {
    SH_REQUIRE(dy && weight && dx && M > 0 && N > 0 && K > 0, SH_ERR_INVALID_ARG, "sh_linear_bwd_data: bad argument");
    SGParams p{};                                   // dx(m,k) = sum_n dy(m,n) W(n,k)
    p.a = dy; p.a_sm = N; p.a_sr = 1;
    p.b = weight; p.b_sn = 1; p.b_sr = K;           // B(k, n) = W[n*K + k]
    p.c = dx; p.c_sm = K; p.c_sn = 1; p.bias = nullptr;
    p.M = M; p.N = K; p.R = N;
    return run_gemm(p, workspace, workspace_bytes, static_cast<hipStream_t>(stream), "linear_bwd_data");
}

int sh_linear_bwd_wgt(const float* dy, const float* x, float* dW, float* dbias, int M, int N, int K, void* workspace,
                      size_t workspace_bytes, sh_stream_t stream) {
    SH_REQUIRE(dy && x && dW && M > 0 && N > 0 && K > 0, SH_ERR_INVALID_ARG, "sh_linear_bwd_wgt: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    SGParams p{};                                   // dW(n,k) = sum_m dy(m,n) x(m,k)
    p.a = dy; p.a_sm = 1; p.a_sr = N;               // A(n, m) = dy[m*N + n]
    p.b = x; p.b_sn = 1; p.b_sr = K;                // B(k, m) = x[m*K + k]
    p.c = dW; p.c_sm = K; p.c_sn = 1; p.bias = nullptr;
    p.M = N; p.N = K; p.R = M;
    const int rc = run_gemm(p, workspace, workspace_bytes, st, "linear_bwd_wgt");
    if (rc != SH_OK) return rc;
    if (dbias) {
        hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)sh_cdiv(N, 256)), dim3(256), 0, st, dy, M, N, dbias);
        SH_CHECK_LAUNCH("colsum");
    }
    return SH_OK;
}

}  // extern "C"
