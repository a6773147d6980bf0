// Spiral convolution on CDNA4 (gfx950): fused neighbour-gather + fp32 MFMA GEMM.
//
//   forward        y[r,b,:]  = act( sum_s x[table[r,s],b,:] . W_s^T + bias )       (models.py:40-51)
//   backward-data  dx[u,b,:] = sum_s ( sum_{e in list(u,s)} dpre[src[e],b,:] ) . W_s (autograd of :42,:45)
//   backward-wgt   dW[:,s,:] = sum_{r,b} dpre[r,b,:]^T x[table[r,s],b,:]           (autograd of :45)
//
// forward and backward-data are the SAME kernel (gather_gemm_kernel): a tile of 128 (row,batch)
// pairs, the gathered operand staged chunk-by-chunk (32 floats of K) through LDS, the weight
// chunk staged beside it, v_mfma_f32_16x16x4_f32 accumulating in registers, epilogue fused.
// The gathered [M, S*C] matrix of the reference never exists in memory.
//
// Tiling (one workgroup = 4 waves = 256 threads):
//   rows   TM = 128 = TV vertices x TB batch entries (TB = power of two <= 128); wave w owns rows
//          32w..32w+31 (two 16-row MFMA tiles) and ALL output channels (NT tiles of 16)
//   K      chunks of KC = 32 floats, double-buffered in LDS, next chunk's global loads are issued
//          before the current chunk's MFMAs (register staging), one barrier per chunk
//   LDS    rows of 32 floats, 16-B quads XOR-swizzled: quad q of row r lives at q ^ (r & 7), which
//          makes both the ds_write_b128 staging and the ds_read_b128 operand reads conflict-free
//   MFMA   operands swapped (A = weight tile, B = gathered tile) so that each lane ends up with
//          4 CONSECUTIVE output channels of one row -> 16-byte epilogue loads/stores
//
// The MFMA sums K in the order (q, t): k = 16*ks + 4*(lane>>4) + t, which is a fixed permutation
// of the natural order; fp32 results therefore differ from a sequential dot product only by
// rounding (tolerance stated in tests/), and are bitwise reproducible run to run.
#include "sh_common.h"

namespace {

constexpr int TM = 128;
constexpr int KC = 32;
constexpr int NTHREADS = 256;

struct GGParams {
    const float* x; long x_sv, x_sb;
    const int* table;   // forward: [R][S] gather rows; MULTI: lptr [R*S+1]
    const int* lsrc;    // MULTI: list entries
    const float* w;     // [Nout][K] row-major
    const float* bias;  // [Nout] or null
    float* y; long y_sv, y_sb;
    const float* yprev; long yp_sv, yp_sb;
    int B, R, S, Cg, Nout, K;
    int act;            // forward: activation of this layer; backward: activation that produced x
    int zero_row;
    int log2TB, n_btiles, nchunks;
    int vec_out;        // Nout % 4 == 0 and output strides 16-B aligned
};

template <int NT, bool VEC4, bool MULTI, bool BWD_EPI>
__global__ __launch_bounds__(NTHREADS) void gather_gemm_kernel(const GGParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);            // [2][TM][KC]
    float* Ws = As + 2 * TM * KC;                          // [2][NT*16][KC]
    int* Ts = reinterpret_cast<int*>(Ws + 2 * NT * 16 * KC);   // table / list-pointer tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int TB = 1 << p.log2TB, TV = TM >> p.log2TB;
    const int bt = blockIdx.x % p.n_btiles, vt = blockIdx.x / p.n_btiles;
    const int v0 = vt * TV, b0 = bt * TB;
    const int S = p.S;

    {   // table tile: rows v0 .. v0+TV-1 are contiguous in the table
        const int nT = TV * S + (MULTI ? 1 : 0);
        const long lim = (long)p.R * S + (MULTI ? 1 : 0);
        for (int i = tid; i < nT; i += NTHREADS) {
            const long g = (long)v0 * S + i;
            Ts[i] = g < lim ? p.table[g] : (MULTI ? p.table[lim - 1] : 0);
        }
    }
    __syncthreads();

    // ---- staging assignment: thread -> quad q of rows rbase + 32*i
    const int q = tid & 7, rbase = tid >> 3;
    int a_vl[4];
    long a_boff[4];
    bool a_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = rbase + 32 * i;
        const int vl = row >> p.log2TB, bl = row & (TB - 1);
        a_vl[i] = vl;
        a_ok[i] = (v0 + vl) < p.R && (b0 + bl) < p.B;
        a_boff[i] = (long)(b0 + bl) * p.x_sb;
    }
    constexpr int WQ = NT >= 2 ? NT / 2 : 1;
    const bool w_thread = (NT >= 2) || tid < 128;

    f32x4 ra[4];
    f32x4 rw[WQ];

    auto gather_quad = [&](int i, int s, int ch) -> f32x4 {
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        if (!MULTI) {
            const int u = Ts[a_vl[i] * S + s];
            r = *reinterpret_cast<const f32x4*>(p.x + (long)u * p.x_sv + a_boff[i] + ch);
        } else {
            const int e0 = Ts[a_vl[i] * S + s], e1 = Ts[a_vl[i] * S + s + 1];
            for (int e = e0; e < e1; ++e) {
                const int u = p.lsrc[e];
                r += *reinterpret_cast<const f32x4*>(p.x + (long)u * p.x_sv + a_boff[i] + ch);
            }
        }
        return r;
    };
    auto gather_scalar = [&](int i, int s, int ch) -> float {
        float r = 0.f;
        if (!MULTI) {
            const int u = Ts[a_vl[i] * S + s];
            r = p.x[(long)u * p.x_sv + a_boff[i] + ch];
        } else {
            const int e0 = Ts[a_vl[i] * S + s], e1 = Ts[a_vl[i] * S + s + 1];
            for (int e = e0; e < e1; ++e) r += p.x[(long)p.lsrc[e] * p.x_sv + a_boff[i] + ch];
        }
        return r;
    };

    auto load_chunk = [&](int c) {
        const int k = c * KC + 4 * q;
        if (VEC4) {
            const bool kok = k < p.K;
            int s = 0, ch = 0;
            if (kok) { s = k / p.Cg; ch = k - s * p.Cg; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (kok && a_ok[i]) ra[i] = gather_quad(i, s, ch);
            }
#pragma unroll
            for (int i = 0; i < WQ; ++i) {
                const int n = rbase + 32 * i;
                rw[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (w_thread && kok && n < p.Nout)
                    rw[i] = *reinterpret_cast<const f32x4*>(p.w + (long)n * p.K + k);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < WQ; ++i) rw[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kk = k + j;
                if (kk < p.K) {
                    const int s = kk / p.Cg, ch = kk - s * p.Cg;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (a_ok[i]) ra[i][j] = gather_scalar(i, s, ch);
#pragma unroll
                    for (int i = 0; i < WQ; ++i) {
                        const int n = rbase + 32 * i;
                        if (w_thread && n < p.Nout) rw[i][j] = p.w[(long)n * p.K + kk];
                    }
                }
            }
        }
    };
    auto store_chunk = [&](int buf) {
        float* Ab = As + buf * TM * KC;
        float* Wb = Ws + buf * NT * 16 * KC;
        const int pq = (q ^ (rbase & 7)) << 2;     // (rbase + 32 i) & 7 == rbase & 7
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(Ab + (rbase + 32 * i) * KC + pq) = ra[i];
        if (w_thread) {
#pragma unroll
            for (int i = 0; i < WQ; ++i) *reinterpret_cast<f32x4*>(Wb + (rbase + 32 * i) * KC + pq) = rw[i];
        }
    };

    f32x4 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int lrow = lane & 15, lq = lane >> 4;
    auto compute = [&](int buf) {
        const float* Ab = As + buf * TM * KC + (32 * wave + lrow) * KC;
        const float* Wb = Ws + buf * NT * 16 * KC + lrow * KC;
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
            const int pq = ((lq + 4 * ks) ^ (lane & 7)) << 2;
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(Ab + pq);
            const f32x4 g1 = *reinterpret_cast<const f32x4*>(Ab + 16 * KC + pq);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const f32x4 wq = *reinterpret_cast<const f32x4*>(Wb + n * 16 * KC + pq);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[t], g0[t], acc[0][n], 0, 0, 0);
                    acc[1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[t], g1[t], acc[1][n], 0, 0, 0);
                }
            }
        }
    };

    // ---- main loop: one barrier per K-chunk, next chunk's global loads in flight during MFMAs
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int c = 0; c < p.nchunks; ++c) {
        const bool more = c + 1 < p.nchunks;
        if (more) load_chunk(c + 1);
        compute(c & 1);
        if (more) store_chunk((c + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds channels n0..n0+3 of tile row (32*wave + 16*m + lrow)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int row = 32 * wave + 16 * m + lrow;
        const int vl = row >> p.log2TB, bl = row & (TB - 1);
        const int v = v0 + vl, b = b0 + bl;
        if (v >= p.R || b >= p.B) continue;
        float* yrow = p.y + (long)v * p.y_sv + (long)b * p.y_sb;
        const float* yp = (BWD_EPI && p.yprev) ? p.yprev + (long)v * p.yp_sv + (long)b * p.yp_sb : nullptr;
        const bool zero = v == p.zero_row;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int n0 = n * 16 + lq * 4;
            if (n0 >= p.Nout) continue;
            f32x4 a = acc[m][n];
            if (p.vec_out) {
                if (!BWD_EPI) {
                    if (p.bias) a += *reinterpret_cast<const f32x4*>(p.bias + n0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[j] = sh_act_fwd(a[j], p.act);
                } else if (yp) {
                    const f32x4 yv = *reinterpret_cast<const f32x4*>(yp + n0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[j] *= sh_act_grad_from_out(yv[j], p.act);
                }
                if (zero) a = (f32x4){0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(yrow + n0) = a;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (n0 + j >= p.Nout) continue;
                    float vv = a[j];
                    if (!BWD_EPI) {
                        if (p.bias) vv += p.bias[n0 + j];
                        vv = sh_act_fwd(vv, p.act);
                    } else if (yp) {
                        vv *= sh_act_grad_from_out(yp[n0 + j], p.act);
                    }
                    yrow[n0 + j] = zero ? 0.f : vv;
                }
            }
        }
    }
}

template <int NT, bool VEC4, bool MULTI, bool BWD_EPI>
int launch_gg(const GGParams& p, int nblocks, hipStream_t st) {
    const int TV = TM >> p.log2TB;
    const size_t smem = (size_t)(2 * TM * KC + 2 * NT * 16 * KC) * sizeof(float) + (size_t)(TV * p.S + 1) * sizeof(int);
    {
        ShProfScope ps(st, "gather_gemm_kernel<%d, %s, %s, %s>", NT, VEC4 ? "true" : "false", MULTI ? "true" : "false",
                       BWD_EPI ? "true" : "false");
        hipLaunchKernelGGL((gather_gemm_kernel<NT, VEC4, MULTI, BWD_EPI>), dim3(nblocks), dim3(NTHREADS), smem, st, p);
    }
    SH_CHECK_LAUNCH("gather_gemm");
    return SH_OK;
}

template <bool MULTI, bool BWD_EPI>
int dispatch_gg(GGParams& p, hipStream_t st) {
    // batch tile: largest power of two <= min(B rounded up, TM)
    int tb = 1;
    while (tb < p.B && tb < TM) tb <<= 1;
    p.log2TB = sh_ilog2_floor(tb);
    const int TV = TM >> p.log2TB;
    p.n_btiles = sh_cdiv(p.B, tb);
    p.nchunks = sh_cdiv(p.K, KC);
    const long nblocks = (long)sh_cdiv(p.R, TV) * p.n_btiles;
    SH_REQUIRE(nblocks > 0 && nblocks < (1L << 31), SH_ERR_UNSUPPORTED, "gather_gemm: grid %ld out of range", nblocks);
    SH_REQUIRE(p.Nout <= 128, SH_ERR_UNSUPPORTED, "gather_gemm: more than 128 output channels (%d) not built", p.Nout);
    SH_REQUIRE(p.S <= 64, SH_ERR_UNSUPPORTED, "gather_gemm: spiral length %d > 64", p.S);
    const bool vec4 = (p.Cg % 4 == 0) && (p.x_sv % 4 == 0) && (p.x_sb % 4 == 0) &&
                      ((reinterpret_cast<uintptr_t>(p.x) | reinterpret_cast<uintptr_t>(p.w)) % 16 == 0);
    p.vec_out = (p.Nout % 4 == 0) && (p.y_sv % 4 == 0) && (p.y_sb % 4 == 0) &&
                (reinterpret_cast<uintptr_t>(p.y) % 16 == 0) &&
                (!p.bias || reinterpret_cast<uintptr_t>(p.bias) % 16 == 0) &&
                (!p.yprev || ((p.yp_sv % 4 == 0) && (p.yp_sb % 4 == 0) && reinterpret_cast<uintptr_t>(p.yprev) % 16 == 0));
    const int nt = sh_cdiv(p.Nout, 16);
#define SH_GG_CASE(NTV)                                                                  \
    return vec4 ? launch_gg<NTV, true, MULTI, BWD_EPI>(p, (int)nblocks, st)              \
                : launch_gg<NTV, false, MULTI, BWD_EPI>(p, (int)nblocks, st)
    if (nt <= 1) { SH_GG_CASE(1); }
    if (nt <= 2) { SH_GG_CASE(2); }
    if (nt <= 4) { SH_GG_CASE(4); }
    SH_GG_CASE(8);
#undef SH_GG_CASE
}

// ------------------------------------------------------------------------------------------
// weight gradient.  A workgroup owns a column group of the weight (KCW = 64*CTW of the K = S*Cin
// columns, all Cout rows) and a contiguous range of 32-row steps of the (vertex,batch) row space;
// per step it stages the gathered tile G [32][KCW] and the dpre tile P [32][Cout] in LDS and
// accumulates  acc[col][co] += sum_rows G[row][col] * P[row][co]  with 16x16x4 MFMAs whose K
// dimension is the ROW index.  Each block ends by writing its partial to a slab; a second kernel
// sums the slabs in a fixed order (no atomics -> bitwise reproducible).
constexpr int TMW = 32;

struct WGParams {
    const float* dpre; long dp_sv, dp_sb;
    const float* x; long x_sv, x_sb;
    const int* table;
    float* slab;        // [nrc][Cout*K] then [nrc][Cout] bias partials
    long slab_stride;   // Cout*K
    long bias_off;      // nrc*Cout*K
    int B, R, S, Cin, Cout, K;
    int log2TB, n_btiles, nsteps, steps_per_block;
};

template <int COT, int CTW, bool VEC4>
__global__ __launch_bounds__(NTHREADS) void wgrad_kernel(const WGParams p) {
    constexpr int KCW = 64 * CTW;
    constexpr int LDG = KCW + 16;                       // == 16 (mod 32): conflict-free ds_read_b32
    constexpr int LDP = COT == 1 ? 16 : COT * 16 + 16;
    constexpr int GQ = KCW / 4;                         // quads per G row
    constexpr int GROWS = NTHREADS / GQ;                // G rows covered per pass
    constexpr int GP = TMW / GROWS;                     // passes
    constexpr int PQ = COT * 4;                         // quads per P row
    constexpr int PTOT = TMW * PQ;                      // total P quads
    constexpr int PP = (PTOT + NTHREADS - 1) / NTHREADS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Gs = reinterpret_cast<float*>(smem);          // [2][TMW][LDG]
    float* Ps = Gs + 2 * TMW * LDG;                      // [2][TMW][LDP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cg = blockIdx.x, rc = blockIdx.y;
    const int TB = 1 << p.log2TB;
    const int step0 = rc * p.steps_per_block;
    const int step1 = min(step0 + p.steps_per_block, p.nsteps);

    // fixed (s, channel) of this thread's G quad: the column group never changes
    const int gq = tid % GQ, grow0 = tid / GQ;
    const int k = cg * KCW + 4 * gq;
    int s4[4], c4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int kk = k + j;
        s4[j] = kk < p.K ? kk / p.Cin : -1;
        c4[j] = kk < p.K ? kk - s4[j] * p.Cin : 0;
    }

    f32x4 rg[GP];
    f32x4 rp[PP];

    auto load_step = [&](int step) {
        const int vt = step / p.n_btiles, bt = step - vt * p.n_btiles;
        const int v0 = vt * (TMW >> p.log2TB), b0 = bt * TB;
#pragma unroll
        for (int i = 0; i < GP; ++i) {
            const int row = grow0 + GROWS * i;
            const int v = v0 + (row >> p.log2TB), b = b0 + (row & (TB - 1));
            rg[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (v < p.R && b < p.B) {
                if (VEC4) {
                    if (s4[0] >= 0) {
                        const int u = p.table[(long)v * p.S + s4[0]];
                        rg[i] = *reinterpret_cast<const f32x4*>(p.x + (long)u * p.x_sv + (long)b * p.x_sb + c4[0]);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (s4[j] >= 0) {
                            const int u = p.table[(long)v * p.S + s4[j]];
                            rg[i][j] = p.x[(long)u * p.x_sv + (long)b * p.x_sb + c4[j]];
                        }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < PP; ++i) {
            const int idx = tid + NTHREADS * i;
            rp[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (idx < PTOT) {
                const int row = idx / PQ, pq = idx - row * PQ;
                const int v = v0 + (row >> p.log2TB), b = b0 + (row & (TB - 1));
                if (v < p.R && b < p.B) {
                    const float* src = p.dpre + (long)v * p.dp_sv + (long)b * p.dp_sb + 4 * pq;
                    if ((p.Cout & 3) == 0 && ((p.dp_sv | p.dp_sb) & 3) == 0) {
                        if (4 * pq < p.Cout) rp[i] = *reinterpret_cast<const f32x4*>(src);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (4 * pq + j < p.Cout) rp[i][j] = src[j];
                    }
                }
            }
        }
    };
    auto store_step = [&](int buf) {
        float* Gb = Gs + buf * TMW * LDG;
        float* Pb = Ps + buf * TMW * LDP;
#pragma unroll
        for (int i = 0; i < GP; ++i) *reinterpret_cast<f32x4*>(Gb + (grow0 + GROWS * i) * LDG + 4 * gq) = rg[i];
#pragma unroll
        for (int i = 0; i < PP; ++i) {
            const int idx = tid + NTHREADS * i;
            if (idx < PTOT) {
                const int row = idx / PQ, pq = idx - row * PQ;
                *reinterpret_cast<f32x4*>(Pb + row * LDP + 4 * pq) = rp[i];
            }
        }
    };

    f32x4 acc[CTW][COT];
#pragma unroll
    for (int a = 0; a < CTW; ++a)
#pragma unroll
        for (int b = 0; b < COT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;

    const int lcol = lane & 15, lk = lane >> 4;
    auto compute = [&](int buf) {
        const float* Gb = Gs + buf * TMW * LDG + (wave * CTW) * 16 + lcol;
        const float* Pb = Ps + buf * TMW * LDP + lcol;
#pragma unroll
        for (int kk = 0; kk < TMW / 4; ++kk) {
            const int row = 4 * kk + lk;
            float gf[CTW], pf[COT];
#pragma unroll
            for (int a = 0; a < CTW; ++a) gf[a] = Gb[row * LDG + a * 16];
#pragma unroll
            for (int b = 0; b < COT; ++b) pf[b] = Pb[row * LDP + b * 16];
#pragma unroll
            for (int a = 0; a < CTW; ++a)
#pragma unroll
                for (int b = 0; b < COT; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(gf[a], pf[b], acc[a][b], 0, 0, 0);
        }
        if (cg == 0 && tid < COT * 16) {
            const float* Pc = Ps + buf * TMW * LDP + tid;
#pragma unroll 8
            for (int r = 0; r < TMW; ++r) bsum += Pc[r * LDP];
        }
    };

    if (step0 < step1) {
        load_step(step0);
        store_step(0);
        __syncthreads();
        for (int st = step0; st < step1; ++st) {
            const bool more = st + 1 < step1;
            const int buf = (st - step0) & 1;
            if (more) load_step(st + 1);
            compute(buf);
            if (more) store_step(buf ^ 1);
            __syncthreads();
        }
    }

    // lane holds weight columns kcol..kcol+3 of output channel co
    float* slab = p.slab + (long)rc * p.slab_stride;
#pragma unroll
    for (int a = 0; a < CTW; ++a) {
        const int kcol = cg * KCW + (wave * CTW + a) * 16 + lk * 4;
#pragma unroll
        for (int b = 0; b < COT; ++b) {
            const int co = b * 16 + lcol;
            if (co >= p.Cout) continue;
            float* dst = slab + (long)co * p.K + kcol;
            if ((p.K & 3) == 0) {
                if (kcol < p.K) *reinterpret_cast<f32x4*>(dst) = acc[a][b];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (kcol + j < p.K) dst[j] = acc[a][b][j];
            }
        }
    }
    if (cg == 0 && tid < p.Cout) p.slab[p.bias_off + (long)rc * p.Cout + tid] = bsum;
}

__global__ void slab_reduce_kernel(const float* __restrict__ slab, long stride, int nslab, long n, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int r = 0; r < nslab; ++r) s += slab[(long)r * stride + i];
    out[i] = s;
}

struct WGPlan {
    int log2TB, n_btiles, nsteps, steps_per_block, nrc, ncg, ctw, cot;
};

WGPlan plan_wgrad(int B, int R, int S, int Cin, int Cout) {
    WGPlan w;
    const int K = S * Cin;
    int tb = 1;
    while (tb < B && tb < TMW) tb <<= 1;
    w.log2TB = sh_ilog2_floor(tb);
    const int tv = TMW >> w.log2TB;
    w.n_btiles = sh_cdiv(B, tb);
    w.nsteps = sh_cdiv(R, tv) * w.n_btiles;
    w.ctw = K > 64 ? 2 : 1;
    w.ncg = sh_cdiv(K, 64 * w.ctw);
    const int cot = sh_cdiv(Cout, 16);
    w.cot = cot <= 1 ? 1 : cot <= 2 ? 2 : cot <= 4 ? 4 : 8;
    int nrc = 512 / w.ncg;
    if (nrc < 1) nrc = 1;
    if (nrc > w.nsteps) nrc = w.nsteps;
    w.steps_per_block = sh_cdiv(w.nsteps, nrc);
    w.nrc = sh_cdiv(w.nsteps, w.steps_per_block);
    return w;
}

template <int COT, int CTW>
int launch_wg(const WGParams& p, const WGPlan& w, bool vec4, hipStream_t st) {
    constexpr int KCW = 64 * CTW;
    constexpr int LDG = KCW + 16;
    constexpr int LDP = COT == 1 ? 16 : COT * 16 + 16;
    const size_t smem = (size_t)2 * TMW * (LDG + LDP) * sizeof(float);
    dim3 grid(w.ncg, w.nrc);
    ShProfScope ps(st, "wgrad_kernel<%d, %d, %s>", COT, CTW, vec4 ? "true" : "false");
    if (vec4) hipLaunchKernelGGL((wgrad_kernel<COT, CTW, true>), grid, dim3(NTHREADS), smem, st, p);
    else hipLaunchKernelGGL((wgrad_kernel<COT, CTW, false>), grid, dim3(NTHREADS), smem, st, p);
    SH_CHECK_LAUNCH("wgrad");
    return SH_OK;
}

__global__ void weight_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int S, int Cin, int Cout) {
    // wt[ci][s*Cout + co] = w[co][s*Cin + ci]
    const long n = (long)S * Cin * Cout;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int co = (int)(i % Cout);
    const long t = i / Cout;
    const int s = (int)(t % S);
    const int ci = (int)(t / S);
    wt[i] = w[(long)co * S * Cin + (long)s * Cin + ci];
}

__global__ void act_backward_kernel(const float* __restrict__ dy, long dy_sv, long dy_sb,
                                    const float* __restrict__ y, long y_sv, long y_sb,
                                    float* __restrict__ dp, long dp_sv, long dp_sb,
                                    int B, int R, int C, int act, int zero_row) {
    const long n = (long)R * B * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long t = i / C;
        const int b = (int)(t % B);
        const int r = (int)(t / B);
        const float g = dy[r * dy_sv + b * dy_sb + c] * sh_act_grad_from_out(y[r * y_sv + b * y_sb + c], act);
        dp[r * dp_sv + b * dp_sb + c] = r == zero_row ? 0.f : g;
    }
}

}  // namespace

extern "C" {

int sh_spiral_conv_fwd(const float* x, int64_t x_sv, int64_t x_sb, const int32_t* table, const float* weight,
                       const float* bias, float* y, int64_t y_sv, int64_t y_sb, int B, int R, int S, int Cin,
                       int Cout, int act, int zero_row, sh_stream_t stream) {
    SH_REQUIRE(x && table && weight && y, SH_ERR_INVALID_ARG, "sh_spiral_conv_fwd: null pointer");
    SH_REQUIRE(B > 0 && R > 0 && S > 0 && Cin > 0 && Cout > 0, SH_ERR_INVALID_ARG,
               "sh_spiral_conv_fwd: non-positive size B=%d R=%d S=%d Cin=%d Cout=%d", B, R, S, Cin, Cout);
    SH_REQUIRE(act >= SH_ACT_IDENTITY && act <= SH_ACT_TANH, SH_ERR_INVALID_ARG, "sh_spiral_conv_fwd: unknown activation %d", act);
    GGParams p{};
    p.x = x; p.x_sv = x_sv; p.x_sb = x_sb;
    p.table = table; p.lsrc = nullptr; p.w = weight; p.bias = bias;
    p.y = y; p.y_sv = y_sv; p.y_sb = y_sb;
    p.yprev = nullptr;
    p.B = B; p.R = R; p.S = S; p.Cg = Cin; p.Nout = Cout; p.K = S * Cin;
    p.act = act; p.zero_row = zero_row;
    return dispatch_gg<false, false>(p, static_cast<hipStream_t>(stream));
}

int sh_spiral_conv_bwd_data(const float* dpre, int64_t dp_sv, int64_t dp_sb, const int32_t* lptr, const int32_t* lsrc,
                            const float* weight_t, float* dx, int64_t dx_sv, int64_t dx_sb, const float* yprev,
                            int64_t yp_sv, int64_t yp_sb, int act_prev, int zero_row, int B, int n_in, int S, int Cin,
                            int Cout, sh_stream_t stream) {
    SH_REQUIRE(dpre && lptr && lsrc && weight_t && dx, SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_data: null pointer");
    SH_REQUIRE(B > 0 && n_in > 0 && S > 0 && Cin > 0 && Cout > 0, SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_data: non-positive size");
    SH_REQUIRE(act_prev >= SH_ACT_IDENTITY && act_prev <= SH_ACT_TANH, SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_data: unknown activation %d", act_prev);
    GGParams p{};
    p.x = dpre; p.x_sv = dp_sv; p.x_sb = dp_sb;
    p.table = lptr; p.lsrc = lsrc; p.w = weight_t; p.bias = nullptr;
    p.y = dx; p.y_sv = dx_sv; p.y_sb = dx_sb;
    p.yprev = yprev; p.yp_sv = yp_sv; p.yp_sb = yp_sb;
    p.B = B; p.R = n_in; p.S = S; p.Cg = Cout; p.Nout = Cin; p.K = S * Cout;
    p.act = act_prev; p.zero_row = zero_row;
    return dispatch_gg<true, true>(p, static_cast<hipStream_t>(stream));
}

int sh_weight_transpose(const float* weight, float* weight_t, int S, int Cin, int Cout, sh_stream_t stream) {
    SH_REQUIRE(weight && weight_t && S > 0 && Cin > 0 && Cout > 0, SH_ERR_INVALID_ARG, "sh_weight_transpose: bad argument");
    const long n = (long)S * Cin * Cout;
    hipLaunchKernelGGL(weight_transpose_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), weight, weight_t, S, Cin, Cout);
    SH_CHECK_LAUNCH("weight_transpose");
    return SH_OK;
}

size_t sh_spiral_conv_bwd_wgt_workspace(int B, int R, int S, int Cin, int Cout) {
    if (B <= 0 || R <= 0 || S <= 0 || Cin <= 0 || Cout <= 0) return 0;
    const WGPlan w = plan_wgrad(B, R, S, Cin, Cout);
    return (size_t)w.nrc * ((size_t)Cout * S * Cin + Cout) * sizeof(float);
}

int sh_spiral_conv_bwd_wgt(const float* dpre, int64_t dp_sv, int64_t dp_sb, const float* x, int64_t x_sv, int64_t x_sb,
                           const int32_t* table, float* dW, float* dbias, void* workspace, size_t workspace_bytes, int B,
                           int R, int S, int Cin, int Cout, sh_stream_t stream) {
    SH_REQUIRE(dpre && x && table && dW && workspace, SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_wgt: null pointer");
    SH_REQUIRE(B > 0 && R > 0 && S > 0 && Cin > 0 && Cout > 0, SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_wgt: non-positive size");
    SH_REQUIRE(Cout <= 128, SH_ERR_UNSUPPORTED, "sh_spiral_conv_bwd_wgt: more than 128 output channels (%d) not built", Cout);
    SH_REQUIRE(workspace_bytes >= sh_spiral_conv_bwd_wgt_workspace(B, R, S, Cin, Cout), SH_ERR_WORKSPACE,
               "sh_spiral_conv_bwd_wgt: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const WGPlan w = plan_wgrad(B, R, S, Cin, Cout);
    WGParams p{};
    p.dpre = dpre; p.dp_sv = dp_sv; p.dp_sb = dp_sb;
    p.x = x; p.x_sv = x_sv; p.x_sb = x_sb; p.table = table;
    p.slab = static_cast<float*>(workspace);
    p.B = B; p.R = R; p.S = S; p.Cin = Cin; p.Cout = Cout; p.K = S * Cin;
    p.slab_stride = (long)Cout * p.K;
    p.bias_off = (long)w.nrc * p.slab_stride;
    p.log2TB = w.log2TB; p.n_btiles = w.n_btiles; p.nsteps = w.nsteps; p.steps_per_block = w.steps_per_block;
    const bool vec4 = (Cin % 4 == 0) && (x_sv % 4 == 0) && (x_sb % 4 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0) &&
                      (reinterpret_cast<uintptr_t>(dpre) % 16 == 0);
    int rc;
#define SH_WG_CASE(C, T) rc = launch_wg<C, T>(p, w, vec4, st)
    if (w.ctw == 1) {
        if (w.cot == 1) SH_WG_CASE(1, 1); else if (w.cot == 2) SH_WG_CASE(2, 1); else if (w.cot == 4) SH_WG_CASE(4, 1); else SH_WG_CASE(8, 1);
    } else {
        if (w.cot == 1) SH_WG_CASE(1, 2); else if (w.cot == 2) SH_WG_CASE(2, 2); else if (w.cot == 4) SH_WG_CASE(4, 2); else SH_WG_CASE(8, 2);
    }
#undef SH_WG_CASE
    if (rc != SH_OK) return rc;
    const long n = p.slab_stride;
    ShProfScope ps(st, "slab_reduce_kernel");
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p.slab, p.slab_stride, w.nrc, n, dW);
    if (dbias)
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((Cout + 255) / 256)), dim3(256), 0, st, p.slab + p.bias_off,
                           (long)Cout, w.nrc, (long)Cout, dbias);
    SH_CHECK_LAUNCH("slab_reduce");
    return SH_OK;
}

int sh_act_backward(const float* dy, int64_t dy_sv, int64_t dy_sb, const float* y, int64_t y_sv, int64_t y_sb, float* dpre,
                    int64_t dp_sv, int64_t dp_sb, int B, int R, int C, int act, int zero_row, sh_stream_t stream) {
    SH_REQUIRE(dy && y && dpre && B > 0 && R > 0 && C > 0, SH_ERR_INVALID_ARG, "sh_act_backward: bad argument");
    SH_REQUIRE(act >= SH_ACT_IDENTITY && act <= SH_ACT_TANH, SH_ERR_INVALID_ARG, "sh_act_backward: unknown activation %d", act);
    const long n = (long)R * B * C;
    const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(act_backward_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), dy, dy_sv, dy_sb, y,
                       y_sv, y_sb, dpre, dp_sv, dp_sb, B, R, C, act, zero_row);
    SH_CHECK_LAUNCH("act_backward");
    return SH_OK;
}

}  // extern "C"
