// Dense layers with one tiny and one huge dimension in bf16: the latent nn.Linear pair fc_latent_enc / fc_latent_dec
// (reference models.py:85-86, applied at :130 and :144) and their autograd, for the bf16 compute path (BASELINE config 3).
//
// All six GEMMs of a training step stream one 28 MB bf16 working copy of a weight (or write one 56.6 MB fp32 weight
// gradient) and do 1.8 GFLOP: HBM-bound by two orders of magnitude at the bf16 MFMA rate.  One LDS-staged kernel serves
// them all:
//
//     Out[p][q] = sum_r A(q, r) * B(p, r)          q = the contiguous index of Out
//
// with each operand either "natural" (memory [index][r]) or "transposed" (memory [r][index]) - the layouts nn.Linear's
// forward, input-gradient and weight-gradient passes present - staged as 64 x 64 tiles through the LDS images of
// sh_bf16_tiles.h (transposed operands come back through ds_read_b64_tr_b16), fp32 operands (the latent code z and its
// gradient) converted on the way in.  256 threads = 4 waves x (32 q x 32 p), v_mfma_f32_16x16x32_bf16, fp32 accumulate;
// a long reduction is split over workgroups into fp32 partial slabs summed in a fixed order (deterministic, no atomics).
#include "sh_bf16_tiles.h"

namespace {

enum { TG_OUT_PARTIAL = 0, TG_OUT_F32 = 1, TG_OUT_BF16 = 2 };

struct TGParams {
    const void* a; long a_ld;          // natural: a[q * a_ld + r]; transposed: a[r * a_ld + q]
    const void* b; long b_ld;          // natural: b[p * b_ld + r]; transposed: b[r * b_ld + p]
    void* out; long out_ld;            // out[p * out_ld + q]; partial slabs: [split][P][Q] fp32
    const float* bias;                 // [Q] or null (direct outputs only)
    float* colsum;                     // [P]: sum over r of B(p, r) (transposed B only) or null
    int Q, P, R;
    int n_qt, n_pt, nsplit, stages_per_split;
};

template <bool TR, bool F32>
__device__ __forceinline__ void tg_stage_load(const void* base, long ld, int i0, int ni, int r0, int r_end, int tid, u32x4 (&reg)[2]) {
    // tile = 64 slow-index rows x 64 fast-index columns; thread -> pieces (row = id >> 3, c8 = id & 7), id = tid, tid + 256
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int id = tid + 256 * k, row = id >> 3, c8 = id & 7;
        const int slow = (TR ? r0 : i0) + row, fast = (TR ? i0 : r0) + 8 * c8;
        const bool ok = slow < (TR ? r_end : ni) && fast < (TR ? ni : r_end);
        reg[k] = ok ? tg_load8<F32>(base, (long)slow * ld + fast) : (u32x4){0u, 0u, 0u, 0u};
    }
}
template <bool TR>
__device__ __forceinline__ void tg_stage_store(char* img, int tid, const u32x4 (&reg)[2]) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int id = tid + 256 * k, row = id >> 3, c8 = id & 7;
        *reinterpret_cast<u32x4*>(img + (TR ? tg_tr_piece(row, c8) : tg_nat_piece(row, c8))) = reg[k];
    }
}

template <bool AT, bool BT, bool AF32, bool BF32, int OUT>
__global__ __launch_bounds__(256) void tgemm_bf16_kernel(const TGParams p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TG_IMG_BYTES];       // [buffer][A | B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bid = blockIdx.x;
    const int qt = bid % p.n_qt; bid /= p.n_qt;
    const int pt = bid % p.n_pt; const int split = bid / p.n_pt;
    const int q0 = qt * 64, p0 = pt * 64;
    const int st0 = split * p.stages_per_split;
    const int nst_all = (p.R + 63) >> 6;
    const int nst = min(p.stages_per_split, nst_all - st0);
    const int qh = wave & 1, ph = wave >> 1;

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float csum = 0.f;

    u32x4 ra[2], rb[2];
    if (nst > 0) {
        tg_stage_load<AT, AF32>(p.a, p.a_ld, q0, p.Q, st0 * 64, p.R, tid, ra);
        tg_stage_load<BT, BF32>(p.b, p.b_ld, p0, p.P, st0 * 64, p.R, tid, rb);
        tg_stage_store<AT>(smem, tid, ra);
        tg_stage_store<BT>(smem + TG_IMG_BYTES, tid, rb);
    }
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
        const int cur = st & 1;
        if (st + 1 < nst) {
            tg_stage_load<AT, AF32>(p.a, p.a_ld, q0, p.Q, (st0 + st + 1) * 64, p.R, tid, ra);
            tg_stage_load<BT, BF32>(p.b, p.b_ld, p0, p.P, (st0 + st + 1) * 64, p.R, tid, rb);
        }
        const char* ia = smem + cur * 2 * TG_IMG_BYTES;
        const char* ib = ia + TG_IMG_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = AT ? tg_tr_frag(ia, 2 * qh + i, kk, lane) : tg_nat_frag(ia, 2 * qh + i, kk, lane);
                fb[i] = BT ? tg_tr_frag(ib, 2 * ph + i, kk, lane) : tg_nat_frag(ib, 2 * ph + i, kk, lane);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (BT && p.colsum && qt == 0 && tid < 64) {          // column sums of B (bias gradient): thread = p index
            const char* col = ib + (tid >> 4) * TG_TR_BLK + (tid & 15) * 2;
#pragma unroll 8
            for (int r = 0; r < 64; ++r)
                csum += (float)*reinterpret_cast<const __bf16*>(col + (r >> 5) * TG_TR_KS + (r & 31) * 32);
        }
        if (st + 1 < nst) {
            char* na = smem + (cur ^ 1) * 2 * TG_IMG_BYTES;
            tg_stage_store<AT>(na, tid, ra);
            tg_stage_store<BT>(na + TG_IMG_BYTES, tid, rb);
        }
        __syncthreads();
    }

    // lane holds q = qb + 4 (lane >> 4) .. +3 of row p = pb + (lane & 15)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = q0 + 32 * qh + 16 * i + 4 * (lane >> 4), pp = p0 + 32 * ph + 16 * j + (lane & 15);
            if (q >= p.Q || pp >= p.P) continue;
            f32x4 v = acc[i][j];
            if (OUT == TG_OUT_PARTIAL) {
                *reinterpret_cast<f32x4*>(static_cast<float*>(p.out) + ((long)split * p.P + pp) * p.Q + q) = v;
            } else {
                if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + q);
                if (OUT == TG_OUT_F32) *reinterpret_cast<f32x4*>(static_cast<float*>(p.out) + (long)pp * p.out_ld + q) = v;
                else *reinterpret_cast<bf16x4*>(static_cast<__bf16*>(p.out) + (long)pp * p.out_ld + q) = sh_to_bf16x4(v);
            }
        }
    if (BT && p.colsum && qt == 0 && tid < 64 && p0 + tid < p.P) p.colsum[p0 + tid] = csum;
}

// out[p][q] = sum_s slab[s][p][q] (+ bias[q]): a wave owns 64 consecutive quads... a workgroup owns 16 quads x 16 slab lanes; lane l
// sums slabs l, l+16, ... (8 independent loads in flight), the 16 lanes are combined in a fixed order: deterministic
template <bool OUTBF16>
__global__ __launch_bounds__(256) void tg_reduce_kernel(const float* __restrict__ slab, int nsplit, long n, int Q, const float* __restrict__ bias,
                                                        void* __restrict__ out) {
    __shared__ f32x4 red[16][16];
    const int qx = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const long i = ((long)blockIdx.x * 16 + qx) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < n) {
        int k = sl;
        for (; k + 112 < nsplit; k += 128) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(slab + (long)(k + 16 * u) * n + i);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < nsplit; k += 16) s += *reinterpret_cast<const f32x4*>(slab + (long)k * n + i);
    }
    red[sl][qx] = s;
    __syncthreads();
    if (sl == 0 && i < n) {
        f32x4 t = red[0][qx];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += red[k][qx];
        if (bias) t += *reinterpret_cast<const f32x4*>(bias + (int)(i % Q));
        if (OUTBF16) *reinterpret_cast<bf16x4*>(static_cast<__bf16*>(out) + i) = sh_to_bf16x4(t);
        else *reinterpret_cast<f32x4*>(static_cast<float*>(out) + i) = t;
    }
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, long n) {
    const long n8 = n >> 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256)
        *reinterpret_cast<u32x4*>(dst + 8 * i) = tg_load8<true>(src, 8 * i);
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) dst[(n8 << 3) + threadIdx.x] = (__bf16)src[(n8 << 3) + threadIdx.x];
}

struct TGPlan { int n_qt, n_pt, nsplit, sps; };
TGPlan plan_tg(int Q, int P, int R) {
    TGPlan t;
    t.n_qt = sh_cdiv(Q, 64); t.n_pt = sh_cdiv(P, 64);
    const int nst = sh_cdiv(R, 64);
    const long tiles = (long)t.n_qt * t.n_pt;
    // split a long reduction until ~512 workgroups exist, keeping >= 8 stages (32 KB of the streamed operand) per split
    int ns = 1;
    if (tiles < 512 && nst >= 16) {
        ns = (int)(512 / tiles);
        if (ns > nst / 8) ns = nst / 8;
        if (ns < 1) ns = 1;
    }
    t.sps = sh_cdiv(nst, ns);
    t.nsplit = sh_cdiv(nst, t.sps);
    return t;
}

template <bool AT, bool BT, bool AF32, bool BF32>
int launch_tg(TGParams& p, const TGPlan& t, int out_dtype, void* ws, hipStream_t st, const char* what) {
    p.n_qt = t.n_qt; p.n_pt = t.n_pt; p.nsplit = t.nsplit; p.stages_per_split = t.sps;
    const int grid = t.n_qt * t.n_pt * t.nsplit;
    ShProfScope ps(st, "tgemm_bf16_kernel<%d,%d,%d,%d>|%s Q=%d P=%d R=%d split=%d", (int)AT, (int)BT, (int)AF32, (int)BF32, what, p.Q, p.P,
                   p.R, t.nsplit);
    if (t.nsplit > 1) {
        void* out = p.out; const float* bias = p.bias;
        p.out = ws; p.bias = nullptr;
        SH_LAUNCH_PS(ps, (tgemm_bf16_kernel<AT, BT, AF32, BF32, TG_OUT_PARTIAL>), dim3(grid), dim3(256), 0, st, p);
        const long n = (long)p.P * p.Q;
        const int rb = (int)((n / 4 + 15) / 16);
        if (out_dtype == SH_DTYPE_BF16)
            hipLaunchKernelGGL(tg_reduce_kernel<true>, dim3(rb), dim3(256), 0, st, static_cast<const float*>(ws), t.nsplit, n, p.Q, bias, out);
        else
            hipLaunchKernelGGL(tg_reduce_kernel<false>, dim3(rb), dim3(256), 0, st, static_cast<const float*>(ws), t.nsplit, n, p.Q, bias, out);
    } else if (out_dtype == SH_DTYPE_BF16) {
        SH_LAUNCH_PS(ps, (tgemm_bf16_kernel<AT, BT, AF32, BF32, TG_OUT_BF16>), dim3(grid), dim3(256), 0, st, p);
    } else {
        SH_LAUNCH_PS(ps, (tgemm_bf16_kernel<AT, BT, AF32, BF32, TG_OUT_F32>), dim3(grid), dim3(256), 0, st, p);
    }
    SH_CHECK_LAUNCH(what);
    return SH_OK;
}

inline bool dt_ok(int d) { return d == SH_DTYPE_F32 || d == SH_DTYPE_BF16; }
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

size_t sh_linear_workspace_bf16(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const TGPlan f = plan_tg(N, M, K), d = plan_tg(K, M, N);
    const size_t a = f.nsplit > 1 ? (size_t)f.nsplit * M * N * 4 : 0, b = d.nsplit > 1 ? (size_t)d.nsplit * M * K * 4 : 0;
    return (a > b ? a : b) + 16;
}

int sh_cast_f32_to_bf16(const float* src, void* dst, int64_t n, sh_stream_t stream) {
    SH_REQUIRE(src && dst && n > 0, SH_ERR_INVALID_ARG, "sh_cast_f32_to_bf16: bad argument");
    SH_REQUIRE(al16(src) && al16(dst), SH_ERR_INVALID_ARG, "sh_cast_f32_to_bf16: pointers must be 16-byte aligned");
    long blocks = (n / 8 + 255) / 256;
    blocks = blocks < 1 ? 1 : blocks > 4096 ? 4096 : blocks;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), src,
                       static_cast<__bf16*>(dst), (long)n);
    SH_CHECK_LAUNCH("cast_bf16");
    return SH_OK;
}

#define SH_LIN_COMMON(name)                                                                                                     \
    SH_REQUIRE(M > 0 && N > 0 && K > 0, SH_ERR_INVALID_ARG, name ": non-positive size");                                        \
    SH_REQUIRE(N % 8 == 0 && K % 8 == 0, SH_ERR_UNSUPPORTED, name ": N and K must be multiples of 8 (got %d, %d)", N, K)

int sh_linear_fwd_bf16(const void* x, int x_dtype, const void* weight_bf16, const float* bias, void* y, int y_dtype, int M, int N,
                       int K, void* workspace, size_t workspace_bytes, sh_stream_t stream) {
    SH_REQUIRE(x && weight_bf16 && y, SH_ERR_INVALID_ARG, "sh_linear_fwd_bf16: null pointer");
    SH_LIN_COMMON("sh_linear_fwd_bf16");
    SH_REQUIRE(dt_ok(x_dtype) && dt_ok(y_dtype) && al16(x) && al16(weight_bf16) && al16(y) && (!bias || al16(bias)), SH_ERR_INVALID_ARG,
               "sh_linear_fwd_bf16: bad dtype or misaligned pointer");
    const TGPlan t = plan_tg(N, M, K);
    SH_REQUIRE(t.nsplit == 1 || (workspace && al16(workspace) && workspace_bytes >= (size_t)t.nsplit * M * N * 4), SH_ERR_WORKSPACE,
               "sh_linear_fwd_bf16: workspace too small");
    TGParams p{};
    p.a = weight_bf16; p.a_ld = K; p.b = x; p.b_ld = K; p.out = y; p.out_ld = N; p.bias = bias; p.Q = N; p.P = M; p.R = K;
    hipStream_t st = static_cast<hipStream_t>(stream);
    return x_dtype == SH_DTYPE_F32 ? launch_tg<false, false, false, true>(p, t, y_dtype, workspace, st, "linear_fwd")
                                   : launch_tg<false, false, false, false>(p, t, y_dtype, workspace, st, "linear_fwd");
}

int sh_linear_bwd_data_bf16(const void* dy, int dy_dtype, const void* weight_bf16, void* dx, int dx_dtype, int M, int N, int K,
                            void* workspace, size_t workspace_bytes, sh_stream_t stream) {
    SH_REQUIRE(dy && weight_bf16 && dx, SH_ERR_INVALID_ARG, "sh_linear_bwd_data_bf16: null pointer");
    SH_LIN_COMMON("sh_linear_bwd_data_bf16");
    SH_REQUIRE(dt_ok(dy_dtype) && dt_ok(dx_dtype) && al16(dy) && al16(weight_bf16) && al16(dx), SH_ERR_INVALID_ARG,
               "sh_linear_bwd_data_bf16: bad dtype or misaligned pointer");
    const TGPlan t = plan_tg(K, M, N);
    SH_REQUIRE(t.nsplit == 1 || (workspace && al16(workspace) && workspace_bytes >= (size_t)t.nsplit * M * K * 4), SH_ERR_WORKSPACE,
               "sh_linear_bwd_data_bf16: workspace too small");
    TGParams p{};
    p.a = weight_bf16; p.a_ld = K; p.b = dy; p.b_ld = N; p.out = dx; p.out_ld = K; p.Q = K; p.P = M; p.R = N;
    hipStream_t st = static_cast<hipStream_t>(stream);
    return dy_dtype == SH_DTYPE_F32 ? launch_tg<true, false, false, true>(p, t, dx_dtype, workspace, st, "linear_bwd_data")
                                    : launch_tg<true, false, false, false>(p, t, dx_dtype, workspace, st, "linear_bwd_data");
}

int sh_linear_bwd_wgt_bf16(const void* dy, int dy_dtype, const void* x, int x_dtype, float* dW, float* dbias, int M, int N, int K,
                           sh_stream_t stream) {
    SH_REQUIRE(dy && x && dW, SH_ERR_INVALID_ARG, "sh_linear_bwd_wgt_bf16: null pointer");
    SH_LIN_COMMON("sh_linear_bwd_wgt_bf16");
    SH_REQUIRE(dt_ok(dy_dtype) && dt_ok(x_dtype) && al16(dy) && al16(x) && al16(dW), SH_ERR_INVALID_ARG,
               "sh_linear_bwd_wgt_bf16: bad dtype or misaligned pointer");
    TGPlan t = plan_tg(K, N, M);
    t.nsplit = 1; t.sps = sh_cdiv(M, 64);                    // the batch is the reduction: never split (the output is the big side)
    TGParams p{};
    p.a = x; p.a_ld = K; p.b = dy; p.b_ld = N; p.out = dW; p.out_ld = K; p.colsum = dbias; p.Q = K; p.P = N; p.R = M;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool xf = x_dtype == SH_DTYPE_F32, df = dy_dtype == SH_DTYPE_F32;
    if (xf) return df ? launch_tg<true, true, true, true>(p, t, SH_DTYPE_F32, nullptr, st, "linear_bwd_wgt")
                      : launch_tg<true, true, true, false>(p, t, SH_DTYPE_F32, nullptr, st, "linear_bwd_wgt");
    return df ? launch_tg<true, true, false, true>(p, t, SH_DTYPE_F32, nullptr, st, "linear_bwd_wgt")
              : launch_tg<true, true, false, false>(p, t, SH_DTYPE_F32, nullptr, st, "linear_bwd_wgt");
}

}  // extern "C"
