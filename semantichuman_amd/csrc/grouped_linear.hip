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
    int item0[GL_MAX + 1];              // first output item of each group
    int ng;
};

__device__ __forceinline__ int gl_group(const GLParams& p, int item) {
    int g = 0;
    while (g + 1 < p.ng && item >= p.item0[g + 1]) ++g;
    return g;
}

__global__ __launch_bounds__(256) void grouped_thread_kernel(const GLParams p) {
    const int item = blockIdx.x * 256 + threadIdx.x;
    if (item >= p.item0[p.ng]) return;
    const int g = gl_group(p, item);
    const int local = item - p.item0[g];
    const int i = local / p.nj[g], j = local - i * p.nj[g];
    const float* a = p.a[g] + (long)i * p.a_si[g];
    const float* b = p.b[g] + (long)j * p.b_sj[g];
    const long asr = p.a_sr[g], bsr = p.b_sr[g];
    const int nr = p.nr[g];
    float s = 0.f, cs = 0.f;
    const bool want_cs = p.colsum[g] != nullptr && j == 0;
    for (int r = 0; r < nr; ++r) {
        const float av = a[r * asr];
        s = fmaf(av, b[r * bsr], s);
        if (want_cs) cs += av;
    }
    if (p.bias[g]) s += p.bias[g][j];
    p.out[g][(long)i * p.o_si[g] + (long)j * p.o_sj[g]] = s;
    if (want_cs) p.colsum[g][i] = cs;
}

__global__ __launch_bounds__(256) void grouped_wave_kernel(const GLParams p) {
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (item >= p.item0[p.ng]) return;
    const int g = gl_group(p, item);
    const int local = item - p.item0[g];
    const int i = local / p.nj[g], j = local - i * p.nj[g];
    const float* a = p.a[g] + (long)i * p.a_si[g];
    const float* b = p.b[g] + (long)j * p.b_sj[g];
    const long asr = p.a_sr[g], bsr = p.b_sr[g];
    const int nr = p.nr[g];
    float s = 0.f, cs = 0.f;
    for (int r = lane; r < nr; r += 64) {
        const float av = a[r * asr];
        s = fmaf(av, b[r * bsr], s);
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

// launches the groups [g0, g0 + p.ng) already filled into p
int gl_launch(GLParams& p, hipStream_t st, const char* what) {
    long items = 0, max_r = 0;
    for (int g = 0; g < p.ng; ++g) {
        p.item0[g] = (int)items;
        items += (long)p.ni[g] * p.nj[g];
        if (p.nr[g] > max_r) max_r = p.nr[g];
        SH_REQUIRE(items < (1L << 30), SH_ERR_UNSUPPORTED, "%s: too many outputs", what);
    }
    p.item0[p.ng] = (int)items;
    if (items == 0) return SH_OK;
    const bool wave = max_r >= 512 && items <= 65536;
    ShProfScope ps(st, "%s|%s groups=%d items=%ld", wave ? "grouped_wave_kernel" : "grouped_thread_kernel", what, p.ng, items);
    if (wave) SH_LAUNCH_PS(ps, grouped_wave_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, p);
    else SH_LAUNCH_PS(ps, grouped_thread_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, p);
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
