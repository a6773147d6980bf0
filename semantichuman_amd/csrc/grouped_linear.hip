// Grouped (ragged) small dense layers: the 3 x 17 per-part nn.Linear layers of SpiralAutoencoder_multiz_partkps
// (reference models.py:200-204, applied one by one at :236,:252,:269).  Each is tiny (16 x 8 x 3200 MACs), so what they
// cost as 17 separate calls - times forward / backward-data / backward-weight, times three passes per iteration - is
// ~450 kernel launches.  Here all groups of one family run in ONE launch per pass direction.
//
// Every operation is phrased as, for group g:   out_g(i, j) = sum_r A_g(i, r) * B_g(j, r)  [+ bias_g(j)]
// with arbitrary element strides, so forward, backward-data and backward-weight are the same two kernels:
//   thread form  one thread per output element, j fastest            (many outputs, short sums)
//   wave form    one wavefront per output element, lanes stride r    (few outputs, sums >= 512 long)
// Fixed summation order in both (deterministic); fp32 FMA chain.  No MFMA: the whole family is < 0.1 GFLOP per step.
#include "sh_common.h"

namespace {

constexpr int GL_MAX = 24;      // groups per launch (the descriptor table travels as kernel arguments, < 4 KiB)

struct GLParams {
    const float* a[GL_MAX]; long a_si[GL_MAX], a_sr[GL_MAX];
    const float* b[GL_MAX]; long b_sj[GL_MAX], b_sr[GL_MAX];
    float* out[GL_MAX]; long o_si[GL_MAX], o_sj[GL_MAX];
    const float* bias[GL_MAX];          // added to out(i, j) by index j, or null
    float* colsum[GL_MAX];              // if non-null: colsum[i] = sum_r A(i, r)   (bias gradient), written by the j == 0 item
    int ni[GL_MAX], nj[GL_MAX], nr[GL_MAX];
    int blk0[GL_MAX + 1];               // first workgroup of each group: a workgroup never straddles groups, so the group
                                        // lookup is a scalar loop per workgroup, not a search per thread
    int ng;
};

__device__ __forceinline__ int gl_group(const GLParams& p, int blk) {
    int g = 0;
    while (g + 1 < p.ng && blk >= p.blk0[g + 1]) ++g;
    return g;
}

// Every form keeps the summation order of the plain definition - s = fma(A(i, r), B(j, r), s) for r = 0, 1, ... from s = 0
// (thread forms), or lane l taking r = l, l + 64, ... and the 64 partial sums added by the same butterfly (wave forms) - so
// the forms below differ in speed only, never in bits.

// ---- thread forms: one thread per output element ------------------------------------------------------------------------
// general: eight products' operands are loaded before their FMAs (the loop is a chain of L2 round trips otherwise: 48 steps
// were 20 us for 1080 outputs), pointers advance by addition
__device__ __forceinline__ void gl_thread_general(const GLParams& p, int g, int local) {
    const int i = local / p.nj[g], j = local - i * p.nj[g];
    const float* a = p.a[g] + (long)i * p.a_si[g];
    const float* b = p.b[g] + (long)j * p.b_sj[g];
    const long asr = p.a_sr[g], bsr = p.b_sr[g];
    const int nr = p.nr[g];
    float s = 0.f, cs = 0.f;
    const bool want_cs = p.colsum[g] != nullptr && j == 0;
    int r = 0;
    for (; r + 8 <= nr; r += 8) {
        float av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { av[u] = a[u * asr]; bv[u] = b[u * bsr]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { s = fmaf(av[u], bv[u], s); if (want_cs) cs += av[u]; }
        a += 8 * asr; b += 8 * bsr;
    }
    for (; r < nr; ++r) {
        const float av = *a;
        s = fmaf(av, *b, s);
        if (want_cs) cs += av;
        a += asr; b += bsr;
    }
    if (p.bias[g]) s += p.bias[g][j];
    p.out[g][(long)i * p.o_si[g] + (long)j * p.o_sj[g]] = s;
    if (want_cs) p.colsum[g][i] = cs;
}

// short contiguous rows on both sides (the decoders' 16-wide layers forward: 2.6 M outputs of 16 products): a thread keeps ITS
// row of B in registers and walks every i - B is read once instead of ni times, A's rows are the same address for the whole
// wave (scalar loads)
constexpr int GL_ROW_MAX = 16;
__device__ __forceinline__ bool gl_rows_form(const GLParams& p, int g) {
    return p.a_sr[g] == 1 && p.b_sr[g] == 1 && p.nr[g] <= GL_ROW_MAX && p.nr[g] % 4 == 0 && p.colsum[g] == nullptr && p.o_sj[g] == 1 &&
           ((reinterpret_cast<uintptr_t>(p.a[g]) | reinterpret_cast<uintptr_t>(p.b[g]) | (uintptr_t)(p.a_si[g] * 4) | (uintptr_t)(p.b_sj[g] * 4)) & 15) == 0;
}
__device__ __forceinline__ void gl_thread_rows(const GLParams& p, int g, int j) {
    const int nr = p.nr[g], ni = p.ni[g];
    float bv[GL_ROW_MAX];
    const float* b = p.b[g] + (long)j * p.b_sj[g];
#pragma unroll
    for (int q = 0; q < GL_ROW_MAX / 4; ++q)
        if (4 * q < nr) {
            const float4 v = *reinterpret_cast<const float4*>(b + 4 * q);
            bv[4 * q] = v.x; bv[4 * q + 1] = v.y; bv[4 * q + 2] = v.z; bv[4 * q + 3] = v.w;
        }
    const float bias = p.bias[g] ? p.bias[g][j] : 0.f;
    const bool has_bias = p.bias[g] != nullptr;
    for (int i = 0; i < ni; ++i) {
        const float* a = p.a[g] + (long)i * p.a_si[g];
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < GL_ROW_MAX / 4; ++q)
            if (4 * q < nr) {
                const float4 v = *reinterpret_cast<const float4*>(a + 4 * q);
                s = fmaf(v.x, bv[4 * q], s); s = fmaf(v.y, bv[4 * q + 1], s); s = fmaf(v.z, bv[4 * q + 2], s); s = fmaf(v.w, bv[4 * q + 3], s);
            }
        if (has_bias) s += bias;
        p.out[g][(long)i * p.o_si[g] + j] = s;
    }
}

// items of a group in the thread kernel: outputs, or columns j in the rows form
__host__ __device__ __forceinline__ long gl_thread_items(const GLParams& p, int g, bool rows) { return rows ? p.nj[g] : (long)p.ni[g] * p.nj[g]; }

__global__ __launch_bounds__(256) void grouped_thread_kernel(const GLParams p) {
    const int g = gl_group(p, (int)blockIdx.x);
    const int local = ((int)blockIdx.x - p.blk0[g]) * 256 + (int)threadIdx.x;
    const bool rows = gl_rows_form(p, g);
    if (local >= gl_thread_items(p, g, rows)) return;
    if (rows) gl_thread_rows(p, g, local);
    else gl_thread_general(p, g, local);
}

// ---- wave forms: one wavefront per output element (or per tile of them), lanes stride r ------------------------------------
// (a) B(j, r) = b[j + r * nj] (backward-data of a 16-wide layer: the weight's rows are the reduction index): ONE wave forms all nj
//     <= 16 outputs of its i - a lane reads its rows of B whole (16-byte pieces, consecutive lanes consecutive rows) instead of
//     nj waves each picking one 4-byte column out of every 64-byte row
// (b) both operands contiguous in r (forward of the encoders' 8-wide layers): a wave forms all nj <= 8 columns of its row i -
//     9 loads per 8 outputs instead of 16 (taller tiles leave too few waves: the kernel is a chain of L2 round trips)
// (c) general: one output per wave
constexpr int GL_WJ = 16, GL_TI = 1, GL_TJ = 8;
__device__ __forceinline__ int gl_wave_form(const GLParams& p, int g) {
    if (p.colsum[g] != nullptr) return 2;
    if (p.b_sj[g] == 1 && p.b_sr[g] == p.nj[g] && p.nj[g] <= GL_WJ && p.nj[g] % 4 == 0 && p.o_sj[g] == 1 &&
        ((reinterpret_cast<uintptr_t>(p.b[g]) & 15) == 0))
        return 0;
    if (p.a_sr[g] == 1 && p.b_sr[g] == 1 && p.nj[g] <= GL_TJ) return 1;
    return 2;
}
__host__ __device__ __forceinline__ long gl_wave_items(int form, int ni, int nj) {
    return form == 0 ? ni : form == 1 ? (ni + GL_TI - 1) / GL_TI : (long)ni * nj;
}

__global__ __launch_bounds__(256) void grouped_wave_kernel(const GLParams p) {
    const int g = gl_group(p, (int)blockIdx.x);
    const int item = ((int)blockIdx.x - p.blk0[g]) * 4 + (int)(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int form = gl_wave_form(p, g);
    const int ni = p.ni[g], nj = p.nj[g], nr = p.nr[g];
    if (item >= gl_wave_items(form, ni, nj)) return;
    if (form == 0) {
        const int i = item;
        const float* a = p.a[g] + (long)i * p.a_si[g];
        const float* b = p.b[g];
        const long asr = p.a_sr[g];
        float acc[GL_WJ];
#pragma unroll
        for (int j = 0; j < GL_WJ; ++j) acc[j] = 0.f;
        // eight of the lane's reduction steps have their operands requested before the first is used (the loop is a chain of L2
        // round trips otherwise); the FMAs stay in r order
        constexpr int U = 8;
        int r = lane;
        for (; r + 64 * (U - 1) < nr; r += 64 * U) {
            float av[U]; float4 bv[U][GL_WJ / 4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                av[u] = a[(long)(r + 64 * u) * asr];
                const float* br = b + (long)(r + 64 * u) * nj;
#pragma unroll
                for (int q = 0; q < GL_WJ / 4; ++q)
                    if (4 * q < nj) bv[u][q] = *reinterpret_cast<const float4*>(br + 4 * q);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int q = 0; q < GL_WJ / 4; ++q)
                    if (4 * q < nj) {
                        acc[4 * q] = fmaf(av[u], bv[u][q].x, acc[4 * q]); acc[4 * q + 1] = fmaf(av[u], bv[u][q].y, acc[4 * q + 1]);
                        acc[4 * q + 2] = fmaf(av[u], bv[u][q].z, acc[4 * q + 2]); acc[4 * q + 3] = fmaf(av[u], bv[u][q].w, acc[4 * q + 3]);
                    }
        }
        for (; r < nr; r += 64) {
            const float av = a[(long)r * asr];
            const float* br = b + (long)r * nj;
#pragma unroll
            for (int q = 0; q < GL_WJ / 4; ++q)
                if (4 * q < nj) {
                    const float4 v = *reinterpret_cast<const float4*>(br + 4 * q);
                    acc[4 * q] = fmaf(av, v.x, acc[4 * q]); acc[4 * q + 1] = fmaf(av, v.y, acc[4 * q + 1]);
                    acc[4 * q + 2] = fmaf(av, v.z, acc[4 * q + 2]); acc[4 * q + 3] = fmaf(av, v.w, acc[4 * q + 3]);
                }
        }
#pragma unroll
        for (int j = 0; j < GL_WJ; ++j)
            if (j < nj) {
                float s = sh_wave_sum(acc[j]);
                if (lane == 0) {
                    if (p.bias[g]) s += p.bias[g][j];
                    p.out[g][(long)i * p.o_si[g] + j] = s;
                }
            }
        return;
    }
    if (form == 1) {
        const int i0 = item * GL_TI;
        float acc[GL_TI][GL_TJ];
#pragma unroll
        for (int u = 0; u < GL_TI; ++u)
#pragma unroll
            for (int j = 0; j < GL_TJ; ++j) acc[u][j] = 0.f;
        const float* a[GL_TI];
#pragma unroll
        for (int u = 0; u < GL_TI; ++u) a[u] = p.a[g] + (long)min(i0 + u, ni - 1) * p.a_si[g];
        const float* b = p.b[g];
        const long bsj = p.b_sj[g];
        constexpr int U = 8;
        int r = lane;
        for (; r + 64 * (U - 1) < nr; r += 64 * U) {
            float av[U][GL_TI], bv[U][GL_TJ];
#pragma unroll
            for (int w = 0; w < U; ++w) {
#pragma unroll
                for (int u = 0; u < GL_TI; ++u) av[w][u] = a[u][r + 64 * w];
#pragma unroll
                for (int j = 0; j < GL_TJ; ++j) bv[w][j] = j < nj ? b[(long)j * bsj + r + 64 * w] : 0.f;
            }
#pragma unroll
            for (int w = 0; w < U; ++w)
#pragma unroll
                for (int u = 0; u < GL_TI; ++u)
#pragma unroll
                    for (int j = 0; j < GL_TJ; ++j) acc[u][j] = fmaf(av[w][u], bv[w][j], acc[u][j]);
        }
        for (; r < nr; r += 64) {
            float av[GL_TI], bv[GL_TJ];
#pragma unroll
            for (int u = 0; u < GL_TI; ++u) av[u] = a[u][r];
#pragma unroll
            for (int j = 0; j < GL_TJ; ++j) bv[j] = j < nj ? b[(long)j * bsj + r] : 0.f;
#pragma unroll
            for (int u = 0; u < GL_TI; ++u)
#pragma unroll
                for (int j = 0; j < GL_TJ; ++j) acc[u][j] = fmaf(av[u], bv[j], acc[u][j]);
        }
#pragma unroll
        for (int u = 0; u < GL_TI; ++u)
#pragma unroll
            for (int j = 0; j < GL_TJ; ++j)
                if (j < nj && i0 + u < ni) {
                    float s = sh_wave_sum(acc[u][j]);
                    if (lane == 0) {
                        if (p.bias[g]) s += p.bias[g][j];
                        p.out[g][(long)(i0 + u) * p.o_si[g] + (long)j * p.o_sj[g]] = s;
                    }
                }
        return;
    }
    const int i = item / nj, j = item - i * nj;
    const float* a = p.a[g] + (long)i * p.a_si[g];
    const float* b = p.b[g] + (long)j * p.b_sj[g];
    const long asr = p.a_sr[g], bsr = p.b_sr[g];
    float s = 0.f, cs = 0.f;
    int r = lane;
    for (; r + 64 * 7 < nr; r += 64 * 8) {
        float av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { av[u] = a[(long)(r + 64 * u) * asr]; bv[u] = b[(long)(r + 64 * u) * bsr]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { s = fmaf(av[u], bv[u], s); cs += av[u]; }
    }
    for (; r < nr; r += 64) {
        const float av = a[(long)r * asr];
        s = fmaf(av, b[(long)r * bsr], s);
        cs += av;
    }
    s = sh_wave_sum(s);
    if (lane == 0) {
        if (p.bias[g]) s += p.bias[g][j];
        p.out[g][(long)i * p.o_si[g] + (long)j * p.o_sj[g]] = s;
    }
    if (p.colsum[g] != nullptr && j == 0) {
        cs = sh_wave_sum(cs);
        if (lane == 0) p.colsum[g][i] = cs;
    }
}

// host mirrors of the device-side form choices (same predicates on the same fields)
bool gl_rows_form_host(const GLParams& p, int g) {
    return p.a_sr[g] == 1 && p.b_sr[g] == 1 && p.nr[g] <= GL_ROW_MAX && p.nr[g] % 4 == 0 && p.colsum[g] == nullptr && p.o_sj[g] == 1 &&
           ((reinterpret_cast<uintptr_t>(p.a[g]) | reinterpret_cast<uintptr_t>(p.b[g]) | (uintptr_t)(p.a_si[g] * 4) | (uintptr_t)(p.b_sj[g] * 4)) & 15) == 0;
}
int gl_wave_form_host(const GLParams& p, int g) {
    if (p.colsum[g] != nullptr) return 2;
    if (p.b_sj[g] == 1 && p.b_sr[g] == p.nj[g] && p.nj[g] <= GL_WJ && p.nj[g] % 4 == 0 && p.o_sj[g] == 1 && ((reinterpret_cast<uintptr_t>(p.b[g]) & 15) == 0))
        return 0;
    if (p.a_sr[g] == 1 && p.b_sr[g] == 1 && p.nj[g] <= GL_TJ) return 1;
    return 2;
}

// launches the groups [g0, g0 + p.ng) already filled into p
int gl_launch(GLParams& p, hipStream_t st, const char* what) {
    long outputs = 0, max_r = 0;
    for (int g = 0; g < p.ng; ++g) {
        outputs += (long)p.ni[g] * p.nj[g];
        if (p.nr[g] > max_r) max_r = p.nr[g];
    }
    if (outputs == 0) return SH_OK;
    const bool wave = max_r >= 512 && outputs <= 65536;
    long blocks = 0;
    for (int g = 0; g < p.ng; ++g) {
        p.blk0[g] = (int)blocks;
        const long items = wave ? gl_wave_items(gl_wave_form_host(p, g), p.ni[g], p.nj[g]) : gl_thread_items(p, g, gl_rows_form_host(p, g));
        blocks += wave ? (items + 3) / 4 : (items + 255) / 256;
        SH_REQUIRE(blocks < (1L << 30), SH_ERR_UNSUPPORTED, "%s: too many outputs", what);
    }
    p.blk0[p.ng] = (int)blocks;
    ShProfScope ps(st, "%s|%s groups=%d outputs=%ld blocks=%ld", wave ? "grouped_wave_kernel" : "grouped_thread_kernel", what, p.ng, outputs, blocks);
    if (wave) SH_LAUNCH_PS(ps, grouped_wave_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p);
    else SH_LAUNCH_PS(ps, grouped_thread_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p);
    SH_CHECK_LAUNCH(what);
    return SH_OK;
}

}  // namespace

extern "C" {

int sh_grouped_linear_fwd(int G, const float* x, int64_t x_rs, const int64_t* x_off, const float* const* w,
                          const float* const* bias, float* y, int64_t y_rs, const int64_t* y_off, int M, const int* N, const int* K,
                          sh_stream_t stream) {
    SH_REQUIRE(G >= 0 && M > 0, SH_ERR_INVALID_ARG, "sh_grouped_linear_fwd: bad sizes");
    if (G == 0) return SH_OK;
    SH_REQUIRE(x && x_off && w && y && y_off && N && K, SH_ERR_INVALID_ARG, "sh_grouped_linear_fwd: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int g0 = 0; g0 < G; g0 += GL_MAX) {
        GLParams p{};
        p.ng = G - g0 < GL_MAX ? G - g0 : GL_MAX;
        for (int t = 0; t < p.ng; ++t) {
            const int g = g0 + t;
            SH_REQUIRE(w[g] && N[g] > 0 && K[g] > 0, SH_ERR_INVALID_ARG, "sh_grouped_linear_fwd: bad group %d", g);
            p.a[t] = x + x_off[g]; p.a_si[t] = x_rs; p.a_sr[t] = 1;                 // A(i = m, r = k) = x[m][x_off + k]
            p.b[t] = w[g]; p.b_sj[t] = K[g]; p.b_sr[t] = 1;                         // B(j = n, r = k) = W[n][k]
            p.out[t] = y + y_off[g]; p.o_si[t] = y_rs; p.o_sj[t] = 1;
            p.bias[t] = bias ? bias[g] : nullptr; p.colsum[t] = nullptr;
            p.ni[t] = M; p.nj[t] = N[g]; p.nr[t] = K[g];
        }
        const int rc = gl_launch(p, st, "grouped_linear_fwd");
        if (rc != SH_OK) return rc;
    }
    return SH_OK;
}

int sh_grouped_linear_bwd_data(int G, const float* dy, int64_t y_rs, const int64_t* y_off, const float* const* w, float* dx,
                               int64_t x_rs, const int64_t* x_off, int M, const int* N, const int* K, sh_stream_t stream) {
    SH_REQUIRE(G >= 0 && M > 0, SH_ERR_INVALID_ARG, "sh_grouped_linear_bwd_data: bad sizes");
    if (G == 0) return SH_OK;
    SH_REQUIRE(dy && y_off && w && dx && x_off && N && K, SH_ERR_INVALID_ARG, "sh_grouped_linear_bwd_data: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int g0 = 0; g0 < G; g0 += GL_MAX) {
        GLParams p{};
        p.ng = G - g0 < GL_MAX ? G - g0 : GL_MAX;
        for (int t = 0; t < p.ng; ++t) {
            const int g = g0 + t;
            SH_REQUIRE(w[g] && N[g] > 0 && K[g] > 0, SH_ERR_INVALID_ARG, "sh_grouped_linear_bwd_data: bad group %d", g);
            p.a[t] = dy + y_off[g]; p.a_si[t] = y_rs; p.a_sr[t] = 1;                // A(i = m, r = n) = dy[m][y_off + n]
            p.b[t] = w[g]; p.b_sj[t] = 1; p.b_sr[t] = K[g];                         // B(j = k, r = n) = W[n][k]
            p.out[t] = dx + x_off[g]; p.o_si[t] = x_rs; p.o_sj[t] = 1;
            p.bias[t] = nullptr; p.colsum[t] = nullptr;
            p.ni[t] = M; p.nj[t] = K[g]; p.nr[t] = N[g];
        }
        const int rc = gl_launch(p, st, "grouped_linear_bwd_data");
        if (rc != SH_OK) return rc;
    }
    return SH_OK;
}

int sh_grouped_linear_bwd_wgt(int G, const float* dy, int64_t y_rs, const int64_t* y_off, const float* x, int64_t x_rs,
                              const int64_t* x_off, float* const* dW, float* const* dbias, int M, const int* N, const int* K,
                              sh_stream_t stream) {
    SH_REQUIRE(G >= 0 && M > 0, SH_ERR_INVALID_ARG, "sh_grouped_linear_bwd_wgt: bad sizes");
    if (G == 0) return SH_OK;
    SH_REQUIRE(dy && y_off && x && x_off && dW && N && K, SH_ERR_INVALID_ARG, "sh_grouped_linear_bwd_wgt: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int g0 = 0; g0 < G; g0 += GL_MAX) {
        GLParams p{};
        p.ng = G - g0 < GL_MAX ? G - g0 : GL_MAX;
        for (int t = 0; t < p.ng; ++t) {
            const int g = g0 + t;
            SH_REQUIRE(dW[g] && N[g] > 0 && K[g] > 0, SH_ERR_INVALID_ARG, "sh_grouped_linear_bwd_wgt: bad group %d", g);
            p.a[t] = dy + y_off[g]; p.a_si[t] = 1; p.a_sr[t] = y_rs;                // A(i = n, r = m) = dy[m][y_off + n]
            p.b[t] = x + x_off[g]; p.b_sj[t] = 1; p.b_sr[t] = x_rs;                 // B(j = k, r = m) = x[m][x_off + k]
            p.out[t] = dW[g]; p.o_si[t] = K[g]; p.o_sj[t] = 1;
            p.bias[t] = nullptr; p.colsum[t] = dbias ? dbias[g] : nullptr;          // db[n] = sum_m dy[m][y_off + n]
            p.ni[t] = N[g]; p.nj[t] = K[g]; p.nr[t] = M;
        }
        const int rc = gl_launch(p, st, "grouped_linear_bwd_wgt");
        if (rc != SH_OK) return rc;
    }
    return SH_OK;
}

}  // extern "C"
