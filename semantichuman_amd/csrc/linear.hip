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
#include "sh_bf16.h"
#include "sh_adam.h"

#include <type_traits>

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

// ------------------------------------------------------------------------------------------
// Streaming forms for M <= 64 (the batch): wave-autonomous, no LDS, no barrier - the scheme of
// wgrad_stream_kernel (spiral_conv.hip).  Every wave owns an output tile and feeds the matrix pipe
// straight from 16-byte global loads; a few steps of loads are in flight while one multiplies.
// MFMA t of a step takes element t of each loaded quad, so one 16-byte load feeds four MFMAs.
#ifndef SH_LS_DEPTH
#define SH_LS_DEPTH 2      // steps of loads in flight besides the one multiplying; measured 2 <= 3 < 5 (register pressure)
#endif
constexpr int LS_DEPTH = SH_LS_DEPTH;

// runs f(s + J, J) for J = 0 .. D-1, stopping after the last valid step; the early exits are explicit branches so the
// compiler sees that a skipped step ends the loop (independent `if`s make it drain all loads where the paths merge)
template <int J, int D, class F>
__device__ __forceinline__ bool ls_run_steps(int s, int n, F&& f) {
    f(s + J, std::integral_constant<int, J>{});
    if constexpr (J + 1 < D) {
        if (s + J + 1 >= n) return true;
        return ls_run_steps<J + 1, D>(s, n, f);
    }
    return false;
}

struct LSParams {
    const float* a;      // small operand (x or dy), row-major [M][*]
    const float* w;      // weight-shaped operand
    const float* bias;
    float* out;          // result (nsplit == 1) ...
    float* slab;         // ... or [nsplit][M][cols] partials
    float* dbias;        // weight-gradient kernel: column sums of dy (bias gradient) or null
    int M, N, K;
    int range, nsplit, groups;     // reduction range per split (multiple of 16), #splits, #64-wide output groups
    // weight-gradient kernel with the Adam update applied to the tile (sh_linear_bwd_wgt_adam): `out` is the PARAMETER
    float* am;           // exp_avg
    float* av;           // exp_avg_sq
    __bf16* shadow;      // bf16 working copy of the parameter (the bf16 path's latent FCs), rewritten with the update, or null
    ShAdamHyper ad;
};

// forward: y[m][n] = sum_k x[m][k] W[n][k].  Item = (64 output columns, k range); both operands contiguous in k.
template <int MT>
__global__ __launch_bounds__(LTHREADS) void linear_fwd_stream_kernel(const LSParams p) {
    const int lane = threadIdx.x & 63, wave = sh_wave_id();
    const int item = sh_xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
    if (item >= p.groups * p.nsplit) return;
    const int ng = item % p.groups, sp = item / p.groups;
    const int n0 = ng * 64, k_begin = sp * p.range;
    const int nsteps = (min(p.K, k_begin + p.range) - k_begin) >> 4;
    const int lr = lane & 15, kq = lane >> 4;
    const float* xrow[MT];
    const float* wrow[4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xrow[mt] = p.a + (long)min(16 * mt + lr, p.M - 1) * p.K + k_begin + 4 * kq;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) wrow[nt] = p.w + (long)(n0 + 16 * nt + lr) * p.K + k_begin + 4 * kq;
    f32x4 xa[LS_DEPTH][MT], wb[LS_DEPTH][4], acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto load = [&](int s, f32x4 (&x4)[MT], f32x4 (&w4)[4]) {
        const int off = 16 * (s < nsteps ? s : nsteps - 1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) x4[mt] = *reinterpret_cast<const f32x4*>(xrow[mt] + off);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) w4[nt] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wrow[nt] + off));
    };
    auto mma = [&](const f32x4 (&x4)[MT], const f32x4 (&w4)[4]) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[nt][t], x4[mt][t], acc[mt][nt], 0, 0, 0);
    };
    auto step = [&](int s, auto J) {
        constexpr int j = decltype(J)::value;
        load(s + LS_DEPTH - 1, xa[(j + LS_DEPTH - 1) % LS_DEPTH], wb[(j + LS_DEPTH - 1) % LS_DEPTH]);
        __builtin_amdgcn_sched_barrier(0);
        mma(xa[j], wb[j]);
    };
    if (nsteps > 0) {
#pragma unroll
        for (int d = 0; d < LS_DEPTH - 1; ++d) load(d, xa[d], wb[d]);
        for (int s = 0; s < nsteps; s += LS_DEPTH)
            if (ls_run_steps<0, LS_DEPTH>(s, nsteps, step)) break;
    }
    // lane holds y[m = 16 mt + lr][n0 + 16 nt + 4 kq .. +3]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = 16 * mt + lr;
        if (m >= p.M) continue;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = n0 + 16 * nt + 4 * kq;
            f32x4 v = acc[mt][nt];
            if (p.nsplit > 1) {
                *reinterpret_cast<f32x4*>(p.slab + ((long)sp * p.M + m) * p.N + n) = v;
            } else {
                if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                *reinterpret_cast<f32x4*>(p.out + (long)m * p.N + n) = v;
            }
        }
    }
}

// forward, LDS-DMA form (round 4).  Measured on the kernel above (profiles/r04_kernel_experiments.txt): it is paced neither by HBM
// nor by the matrix pipe - every load instruction fetches 64-byte pieces of 16 rows, i.e. half of 16 different 128-byte lines,
// the batch operand is fetched once per wave, and one step of loads is all a wave has in flight.  Here a workgroup's four waves
// are the four 64-column groups of ONE k range (or four neighbouring groups when nothing is split): the batch operand's tile is
// fetched once per workgroup, every operand streams memory -> LDS with global_load_lds_dwordx4 in WHOLE lines (eight lanes take
// the eight 16-byte pieces of a row's 128 bytes = 32 k), three stages of 40 KiB (8 KiB shared x + 4 x 8 KiB weight) rotate with
// one barrier per stage and two stages in flight, and the MFMA fragments are read back with ds_read_b128 (pieces XOR-swizzled
// by the row so that the sixteen rows of a fragment hit different banks).  Same products in the same order as the kernel
// above: bit-identical results.
// X3 (round 5; mma_mode SH_MMA_SPLIT3 / SH_MMA_PLANES3): the products in the bf16x3 form of the conv kernels.  A stage is ONE
// k-step of v_mfma_f32_16x16x32_bf16: a lane's eight k are the two pieces it reads anyway (k-slots {4 kq ..+3} and {16 + 4 kq ..+3},
// the same for both operands - which k a slot holds is free as long as the operands agree), split exactly into three bf16 terms
// (sh_split3) in registers, six partial products per (row tile, column tile): 96 MFMAs of 16 cycles per stage and wave against
// 128 of 32 (round 4 measured these passes paced by the fp32 MFMA issue, not by the 56.6 MB stream).
constexpr int LFD_STAGE = 40 * 1024, LFD_LDS = 3 * LFD_STAGE;
// the six leading partial products of (ah + am + al)(bh + bm + bl), smallest first, on one accumulator
__device__ __forceinline__ f32x4 lin_x3_mma(const bf16x8 ah, const bf16x8 am, const bf16x8 al, const bf16x8 bh, const bf16x8 bm, const bf16x8 bl,
                                            f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
}
struct LinX3 { bf16x8 h, m, l; };
__device__ __forceinline__ LinX3 lin_split(const f32x4 a, const f32x4 b) {
    u32x4 h, m, l;
    sh_split3(a, b, h, m, l);
    return LinX3{__builtin_bit_cast(bf16x8, h), __builtin_bit_cast(bf16x8, m), __builtin_bit_cast(bf16x8, l)};
}

template <int MT, bool X3 = false>
__global__ __launch_bounds__(LTHREADS) void linear_fwd_dma_kernel(const LSParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int item = sh_xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;       // groups % 4 == 0: the four items share sp
    const int ng = item % p.groups, sp = item / p.groups;
    const int n0 = ng * 64, k_begin = sp * p.range;
    const int nst = (min(p.K, k_begin + p.range) - k_begin) >> 5;          // stages of 32 k
    // batches of more than 64 rows (round 5: config 5's decode runs the decoder FC at 1024): blockIdx.y = the 64-row chunk - the
    // weight is streamed once per chunk, from the Infinity Cache after the first (56.6 MB of 256)
    const int m_lo = 64 * (int)blockIdx.y, m_n = min(64, p.M - m_lo);
    const float* pa = p.a + (long)m_lo * p.K;
    const int lr = lane & 15, kq = lane >> 4;
    typedef __attribute__((address_space(3))) char* lptr_t;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lptr_t)smem);
    auto dma16 = [](const void* gsrc, unsigned lds_dst) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory", "m0");
    };
    // DMA instruction q of a 64-row tile covers rows 8 q .. 8 q + 7: lane -> row 8 q + (lane >> 3), slot lane & 7 holds piece
    // slot ^ ((row >> 1) & 7) of the row's eight.  A wave fetches its own weight tile (8 instructions) and a quarter of the
    // shared x tile (instructions 2 wave, 2 wave + 1).
    const int drow = lane >> 3;
    const float* wsrc[8];
    const float* xsrc[2];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int r = 8 * q + drow, piece = (lane & 7) ^ ((r >> 1) & 7);
        wsrc[q] = p.w + (long)(n0 + r) * p.K + k_begin + 4 * piece;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 8 * (2 * wave + j) + drow, piece = (lane & 7) ^ ((r >> 1) & 7);
        xsrc[j] = pa + (long)min(r, m_n - 1) * p.K + k_begin + 4 * piece;
    }
    auto issue = [&](int st) {
        const unsigned slot = lds0 + (unsigned)((st % 3) * LFD_STAGE);
        const int off = 32 * (st < nst ? st : nst - 1);
#pragma unroll
        for (int j = 0; j < 2; ++j) dma16(xsrc[j] + off, slot + (unsigned)((2 * wave + j) * 1024));
#pragma unroll
        for (int q = 0; q < 8; ++q) dma16(wsrc[q] + off, slot + (unsigned)(8192 + wave * 8192 + q * 1024));
    };
    f32x4 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // fragment (tile t, step h of the stage): row 16 t + lr, piece 4 h + kq -> byte (16 t + lr) * 128 + ((4 h + kq) ^ ((lr >> 1) & 7)) * 16
    const int sw = (lr >> 1) & 7;
    const int foff0 = lr * 128 + ((kq ^ sw) << 4), foff1 = lr * 128 + (((4 + kq) ^ sw) << 4);
    if (nst > 0) {                                                          // uniform over the workgroup (same sp)
        issue(0);
        issue(1);
        for (int st = 0; st < nst; ++st) {
            asm volatile("s_waitcnt vmcnt(10)" ::: "memory");               // this wave's part of stage st has landed (st + 1 may be in flight)
            __syncthreads();                                                // ... everybody's has; and everybody is done with stage st - 1
            issue(st + 2);                                                  // into the slot of stage st - 1
            const char* sl = smem + (st % 3) * LFD_STAGE;
            const char* wl = sl + 8192 + wave * 8192;
            if constexpr (X3) {
                LinX3 xs[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    xs[mt] = lin_split(*reinterpret_cast<const f32x4*>(sl + mt * 2048 + foff0), *reinterpret_cast<const f32x4*>(sl + mt * 2048 + foff1));
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const LinX3 ws = lin_split(*reinterpret_cast<const f32x4*>(wl + nt * 2048 + foff0), *reinterpret_cast<const f32x4*>(wl + nt * 2048 + foff1));
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = lin_x3_mma(ws.h, ws.m, ws.l, xs[mt].h, xs[mt].m, xs[mt].l, acc[mt][nt]);
                }
                continue;
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int fo = h == 0 ? foff0 : foff1;
                f32x4 w4[4], x4[MT];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) w4[nt] = *reinterpret_cast<const f32x4*>(wl + nt * 2048 + fo);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) x4[mt] = *reinterpret_cast<const f32x4*>(sl + mt * 2048 + fo);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[nt][t], x4[mt][t], acc[mt][nt], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // lane holds y[m = m_lo + 16 mt + lr][n0 + 16 nt + 4 kq .. +3]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (16 * mt + lr >= m_n) continue;
        const int m = m_lo + 16 * mt + lr;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = n0 + 16 * nt + 4 * kq;
            f32x4 v = acc[mt][nt];
            if (p.nsplit > 1) {
                *reinterpret_cast<f32x4*>(p.slab + ((long)sp * p.M + m) * p.N + n) = v;
            } else {
                if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                *reinterpret_cast<f32x4*>(p.out + (long)m * p.N + n) = v;
            }
        }
    }
}

// Forward for LARGE batches and a short reduction (round 5: BASELINE config 5 decodes 1024 latents through fc_latent_dec, 55 296 x
// 256), bf16x3 form.  The chunked DMA kernel above re-splits every weight tile once per 64-row chunk of the batch (16 times at
// 1024) and ran at 94 TF.  Here a workgroup owns 64 output columns for the WHOLE batch: its weight tile [64][K] is split ONCE
// into three-plane MFMA fragments resident in LDS (K / 32 x 12 KiB: K <= 384), then its eight waves walk the batch in passes
// of 128 rows - a wave's 16 rows arrive straight from global memory (two 16-byte loads per lane and k-step: whole 128-byte
// lines; the activations are L2-resident), are split in registers (one split per 24 MFMAs) and multiplied against the four
// column tiles' fragments.
constexpr int LFW_WAVES = 8;
__global__ __launch_bounds__(LFW_WAVES * 64) void linear_fwd_wide_x3_kernel(const LSParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* Wl = reinterpret_cast<u32x4*>(smem);                       // [K / 32][4 tiles][3 planes][64 lanes]
    const int lane = threadIdx.x & 63, wave = sh_wave_id();
    const int n0 = sh_xcd_remap(blockIdx.x, gridDim.x) * 64;
    const int nks = p.K >> 5;
    const int lr = lane & 15, kq = lane >> 4;
    for (int f = wave; f < nks * 4; f += LFW_WAVES) {                  // fragment (k-step ks, column tile nt): split once
        const int ks = f >> 2, nt = f & 3;
        const float* src = p.w + (long)(n0 + 16 * nt + lr) * p.K + 32 * ks + 8 * kq;
        u32x4 h, m, l;
        sh_split3(*reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 4), h, m, l);
        u32x4* d = Wl + (long)f * 192 + lane;
        d[0] = h; d[64] = m; d[128] = l;
    }
    __syncthreads();
    // (measured and not kept, gpurun_out/r05e22: a whole pass of activation loads issued a pass ahead - 255 us; two row tiles per
    // wave so that a fragment read feeds 12 MFMAs - 281 us, it spills; non-temporal stores - slower: the 64-byte pieces of a
    // row's four column tiles merge in L2 only with ordinary stores.  This form: 238 us against 312 for the chunked kernel.)
    for (int m0 = 16 * wave; m0 < p.M; m0 += 16 * LFW_WAVES) {
        const int mrow = m0 + lr;
        const float* xr = p.a + (long)min(mrow, p.M - 1) * p.K + 8 * kq;
        f32x4 acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 xa = *reinterpret_cast<const f32x4*>(xr), xb = *reinterpret_cast<const f32x4*>(xr + 4);
        for (int ks = 0; ks < nks; ++ks) {
            const int kn = ks + 1 < nks ? ks + 1 : ks;
            const f32x4 na = *reinterpret_cast<const f32x4*>(xr + 32 * kn), nb = *reinterpret_cast<const f32x4*>(xr + 32 * kn + 4);
            const LinX3 xs = lin_split(xa, xb);
            const u32x4* wk = Wl + (long)ks * 4 * 192 + lane;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const u32x4 r0 = wk[nt * 192], r1 = wk[nt * 192 + 64], r2 = wk[nt * 192 + 128];
                acc[nt] = lin_x3_mma(__builtin_bit_cast(bf16x8, r0), __builtin_bit_cast(bf16x8, r1), __builtin_bit_cast(bf16x8, r2), xs.h, xs.m,
                                     xs.l, acc[nt]);
            }
            xa = na; xb = nb;
        }
        if (mrow < p.M) {                                               // lane holds y[mrow][n0 + 16 nt + 4 kq .. +3]
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int n = n0 + 16 * nt + 4 * kq;
                f32x4 v = acc[nt];
                if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                *reinterpret_cast<f32x4*>(p.out + (long)mrow * p.N + n) = v;
            }
        }
    }
}

// backward-data: dx[m][k] = sum_n dy[m][n] W[n][k].  Item = (64 output columns k, n range).  The weight quad runs along
// the OUTPUT index: lane (a, rr) loads W[nb + 4 rr + e][k0 + 4 a ..+3] for e = 0..3; MFMA (e, t) reduces over the four rows
// {nb + 4 rr' + e} and produces the columns {k0 + 4 a' + t}.  dy quads run along n: element e pairs with weight row e.
template <int MT>
__global__ __launch_bounds__(LTHREADS) void linear_bwd_data_stream_kernel(const LSParams p) {
    const int lane = threadIdx.x & 63, wave = sh_wave_id();
    const int item = sh_xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
    if (item >= p.groups * p.nsplit) return;
    const int kg = item % p.groups, sp = item / p.groups;
    const int k0 = kg * 64, n_begin = sp * p.range;
    const int nsteps = (min(p.N, n_begin + p.range) - n_begin) >> 4;
    const int la = lane & 15, rr = lane >> 4;
    const float* drow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) drow[mt] = p.a + (long)min(16 * mt + la, p.M - 1) * p.N + n_begin + 4 * rr;
    const float* wbase = p.w + (long)(n_begin + 4 * rr) * p.K + k0 + 4 * la;
    f32x4 dq[LS_DEPTH][MT], wq[LS_DEPTH][4], acc[4][MT];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[t][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto load = [&](int s, f32x4 (&d4)[MT], f32x4 (&w4)[4]) {
        const int nb = 16 * (s < nsteps ? s : nsteps - 1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) d4[mt] = *reinterpret_cast<const f32x4*>(drow[mt] + nb);
#pragma unroll
        for (int e = 0; e < 4; ++e) w4[e] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wbase + (long)(nb + e) * p.K));
    };
    auto mma = [&](const f32x4 (&d4)[MT], const f32x4 (&w4)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[e][t], d4[mt][e], acc[t][mt], 0, 0, 0);
    };
    auto step = [&](int s, auto J) {
        constexpr int j = decltype(J)::value;
        load(s + LS_DEPTH - 1, dq[(j + LS_DEPTH - 1) % LS_DEPTH], wq[(j + LS_DEPTH - 1) % LS_DEPTH]);
        __builtin_amdgcn_sched_barrier(0);
        mma(dq[j], wq[j]);
    };
    if (nsteps > 0) {
#pragma unroll
        for (int d = 0; d < LS_DEPTH - 1; ++d) load(d, dq[d], wq[d]);
        for (int s = 0; s < nsteps; s += LS_DEPTH)
            if (ls_run_steps<0, LS_DEPTH>(s, nsteps, step)) break;
    }
    // acc[t][mt][jj] = dx[m = 16 mt + la][k0 + 16 rr + 4 jj + t]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = 16 * mt + la;
        if (m >= p.M) continue;
        float* dst = (p.nsplit > 1 ? p.slab + ((long)sp * p.M + m) * p.K : p.out + (long)m * p.K) + k0 + 16 * rr;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
            *reinterpret_cast<f32x4*>(dst + 4 * jj) = (f32x4){acc[0][mt][jj], acc[1][mt][jj], acc[2][mt][jj], acc[3][mt][jj]};
    }
}

// backward-data in the bf16x3 form (round 5).  Steps of 32 rows of W: lane (a, rr) loads the quads W[n(j)][k0 + 4 a ..+3] of ITS
// eight reduction rows n(j) = nb + 4 rr + j (j < 4), nb + 16 + 4 rr + (j - 4) - every load instruction covers four whole 256-byte
// row segments - and dy[m][n(0..3)], dy[m][n(4..7)] (two quads); element t of the eight weight quads, split exactly into three
// bf16 terms, is the A operand for the output columns {k0 + 4 a' + t}, the dy quads the B operand: 4 MT x 6 MFMAs of 16 cycles
// per 32 rows against 16 MT x 2 of 32 in the fp32 form, behind 4 + MT splits.  Same items, slabs and output mapping.
template <int MT>
__global__ __launch_bounds__(LTHREADS) void linear_bwd_data_x3_kernel(const LSParams p) {
    const int lane = threadIdx.x & 63, wave = sh_wave_id();
    const int item = sh_xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
    if (item >= p.groups * p.nsplit) return;
    const int kg = item % p.groups, sp = item / p.groups;
    const int k0 = kg * 64, n_begin = sp * p.range;
    const int nsteps = (min(p.N, n_begin + p.range) - n_begin) >> 5;          // range % 32 == 0
    const int la = lane & 15, rr = lane >> 4;
    const float* drow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) drow[mt] = p.a + (long)min(16 * mt + la, p.M - 1) * p.N + n_begin + 4 * rr;
    const float* wbase = p.w + (long)(n_begin + 4 * rr) * p.K + k0 + 4 * la;
    f32x4 dq[2][MT][2], wq[2][8], acc[4][MT];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[t][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto load = [&](int s, f32x4 (&d4)[MT][2], f32x4 (&w4)[8]) {
        const int nb = 32 * (s < nsteps ? s : nsteps - 1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            d4[mt][0] = *reinterpret_cast<const f32x4*>(drow[mt] + nb);
            d4[mt][1] = *reinterpret_cast<const f32x4*>(drow[mt] + nb + 16);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            w4[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wbase + (long)(nb + (j & 3) + 16 * (j >> 2)) * p.K));
    };
    auto mma = [&](const f32x4 (&d4)[MT][2], const f32x4 (&w4)[8]) {
        LinX3 ds[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) ds[mt] = lin_split(d4[mt][0], d4[mt][1]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const LinX3 ws = lin_split((f32x4){w4[0][t], w4[1][t], w4[2][t], w4[3][t]}, (f32x4){w4[4][t], w4[5][t], w4[6][t], w4[7][t]});
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[t][mt] = lin_x3_mma(ws.h, ws.m, ws.l, ds[mt].h, ds[mt].m, ds[mt].l, acc[t][mt]);
        }
    };
    if (nsteps > 0) {
        load(0, dq[0], wq[0]);
        for (int s = 0; s < nsteps; s += 2) {
            load(s + 1, dq[1], wq[1]);
            __builtin_amdgcn_sched_barrier(0);
            mma(dq[0], wq[0]);
            if (s + 1 >= nsteps) break;
            load(s + 2, dq[0], wq[0]);
            __builtin_amdgcn_sched_barrier(0);
            mma(dq[1], wq[1]);
        }
    }
    // acc[t][mt][jj] = dx[m = 16 mt + la][k0 + 16 rr + 4 jj + t]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = 16 * mt + la;
        if (m >= p.M) continue;
        float* dst = (p.nsplit > 1 ? p.slab + ((long)sp * p.M + m) * p.K : p.out + (long)m * p.K) + k0 + 16 * rr;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
            *reinterpret_cast<f32x4*>(dst + 4 * jj) = (f32x4){acc[0][mt][jj], acc[1][mt][jj], acc[2][mt][jj], acc[3][mt][jj]};
    }
}

// weight gradient: dW[n][k] = sum_m dy[m][n] x[m][k].  Item = 64 x 64 output tile; the reduction index m (<= 64) is the
// MFMA K dimension, both operands are loaded as quads along their OUTPUT index: lane (a, rr) holds dy[m0+rr][n0+4a..] and
// x[m0+rr][k0+4a..]; MFMA (t', t) produces dW[n0 + 4 i + t'][k0 + 4 j + t] - one pair of loads feeds 16 MFMAs.
// (A 16 x 256 tile with 1 KiB-contiguous writes was measured slower: 39 / 48 us against 38 / 40 us.)
__global__ __launch_bounds__(LTHREADS) void linear_bwd_wgt_stream_kernel(const LSParams p) {
    const int lane = threadIdx.x & 63, wave = sh_wave_id();
    const int item = sh_xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
    const int ktiles = p.K >> 6;
    if (item >= (p.N >> 6) * ktiles) return;
    const int kt = item % ktiles, ntile = item / ktiles;
    const int n0 = ntile * 64, k0 = kt * 64;
    const int la = lane & 15, rr = lane >> 4;
    const int nsteps = (p.M + 3) >> 2;
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 dq[4], xq[4];
    auto load = [&](int s, f32x4& d4, f32x4& x4) {
        s = s < nsteps ? s : nsteps - 1;
        const int m = 4 * s + rr;
        const int mc = m < p.M ? m : p.M - 1;
        d4 = *reinterpret_cast<const f32x4*>(p.a + (long)mc * p.N + n0 + 4 * la);
        x4 = *reinterpret_cast<const f32x4*>(p.w + (long)mc * p.K + k0 + 4 * la);
        if (m >= p.M) d4 = (f32x4){0.f, 0.f, 0.f, 0.f};            // rows are the reduction index
    };
    // the tiles of the first column block also own the bias gradient of their 64 rows: db[n] = sum_m dy[m][n]
    const bool own_bias = kt == 0 && p.dbias != nullptr;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    auto mma = [&](const f32x4& d4, const f32x4& x4) {
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int tk = 0; tk < 4; ++tk) acc[tn][tk] = __builtin_amdgcn_mfma_f32_16x16x4f32(d4[tn], x4[tk], acc[tn][tk], 0, 0, 0);
        if (own_bias) bsum += d4;
    };
    // M <= 64 -> at most 16 steps: four register sets rotate
    load(0, dq[0], xq[0]); load(1, dq[1], xq[1]); load(2, dq[2], xq[2]);
    for (int s = 0; s < nsteps; s += 4) {
        load(s + 3, dq[3], xq[3]); __builtin_amdgcn_sched_barrier(0); mma(dq[0], xq[0]);
        if (s + 1 >= nsteps) break;
        load(s + 4, dq[0], xq[0]); __builtin_amdgcn_sched_barrier(0); mma(dq[1], xq[1]);
        if (s + 2 >= nsteps) break;
        load(s + 5, dq[1], xq[1]); __builtin_amdgcn_sched_barrier(0); mma(dq[2], xq[2]);
        if (s + 3 >= nsteps) break;
        load(s + 6, dq[2], xq[2]); __builtin_amdgcn_sched_barrier(0); mma(dq[3], xq[3]);
    }
    if (own_bias) {                                                   // fixed order: steps in sequence, then the four row groups
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = bsum[j];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            bsum[j] = v;
        }
        if (rr == 0) *reinterpret_cast<f32x4*>(p.dbias + n0 + 4 * la) = bsum;
    }
    // acc[tn][tk][jj] = dW[n0 + 16 rr + 4 jj + tn][k0 + 4 la + tk]
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            float* dst = p.out + (long)(n0 + 16 * rr + 4 * jj + tn) * p.K + k0 + 4 * la;
            __builtin_nontemporal_store((f32x4){acc[tn][0][jj], acc[tn][1][jj], acc[tn][2][jj], acc[tn][3][jj]}, reinterpret_cast<f32x4*>(dst));
        }
}

// LDS-DMA form of the weight gradient (round 2).  The streaming kernel above keeps three 2-KiB steps of loads in flight per
// wave against 16 MFMAs (0.2 us) per step: a tile is a chain of ~5 round trips.  Here a workgroup takes one 64-row block of
// dy and FOUR 64-column blocks of x: the whole dy tile [M <= 64][64] (shared) and each wave's x tile [M][64] go memory -> LDS
// in one volley of global_load_lds_dwordx4 (256-byte rows, four per instruction, their 16-byte pieces XOR-swizzled by the row
// so that the four row groups of a fragment read hit different banks), one wait, one barrier, then the same 16 x 16 MFMAs
// per 4 reduction rows as above with both quads read from LDS.  80 KiB per workgroup: two per CU, one loading while the
// other multiplies.  Same products in the same order as the streaming kernel: bit-identical results.
// X3 (round 5): steps of 32 reduction rows on v_mfma_f32_16x16x32_bf16; a lane's eight rows are m = 32 s + 4 j + rr (j = 0..7; rows
// 4 apart, so that the four row groups of a read keep their different swizzles: no bank conflicts), element tn / tk of its eight
// dy / x quads split exactly into three bf16 terms: 16 x 6 MFMAs of 16 cycles per 32 rows against 8 x 16 of 32.
// ADAM (round 5): the tile is not written as a gradient; it is the `g` of torch.optim.Adam's update of the same tile of the parameter
// (p.out), exp_avg (p.am) and exp_avg_sq (p.av), applied here with adam.hip's own update function: the same bits as "store dW, then
// sh_adam_step", without the 2 x 4 bytes per weight of writing and re-reading the gradient.  The step's coefficients (two double
// pow) are computed by every wave while its tile loads are in flight.
// A16 / W16: dy / x is a bf16 tensor (the bf16 path's layer): its tile is read with ordinary 16-byte loads, widened to fp32 (exact) in
// registers and written into the same LDS layout; the products are the fp32 MFMA's (a product of two bf16 values is exact in fp32:
// what the bf16 MFMA computes, in another summation order).
constexpr int LWD_TILE = 16 * 1024, LWD_LDS = 5 * LWD_TILE;
template <bool X3 = false, bool ADAM = false, bool A16 = false, bool W16 = false>
// (80 KiB of LDS: two workgroups per CU.  The Adam / bf16x3 instances are pinned to that occupancy - 256 registers each, measured
// with the fused update, profiles/r05_fc_adam_probe.txt; the plain fp32 instance keeps the compiler's own choice: ADVICE r5)
__global__ __launch_bounds__(LTHREADS) __attribute__((amdgpu_waves_per_eu((X3 || ADAM) ? 2 : 1, (X3 || ADAM) ? 2 : 8)))
void linear_bwd_wgt_dma_kernel(const LSParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ktiles = p.K >> 6, kgroups = (ktiles + 3) >> 2;
    const int item = sh_xcd_remap(blockIdx.x, gridDim.x);
    const int kg = item % kgroups, ntile = item / kgroups;
    const int kt = kg * 4 + wave;
    const bool active = kt < ktiles;
    const int n0 = ntile * 64, k0 = (active ? kt : 0) * 64;
    const int la = lane & 15, rr = lane >> 4;
    const int nsteps = (p.M + 3) >> 2;
    typedef __attribute__((address_space(3))) char* lptr_t;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lptr_t)smem);
    auto dma16 = [](const void* gsrc, unsigned lds_dst) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory", "m0");
    };
    // instruction i of a tile: rows 4 i .. 4 i + 3; lane -> row 4 i + (lane >> 4), slot lane & 15 holds piece slot ^ ((row & 3) << 2)
    const int piece = (lane & 15) ^ ((lane >> 4) << 2);
    // a bf16 tile row is 8 pieces of 8 elements: lane -> (row, piece j); elements 8 j .. 8 j + 7 are fp32 pieces 2 j and 2 j + 1 of the row
    auto stage16 = [&](const __bf16* src, long ld, int col0, int row, int j, char* tile) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + (long)min(row, p.M - 1) * ld + col0 + 8 * j);
        char* d = tile + (row >> 2) * 1024 + (row & 3) * 256;
        const int sw = (row & 3) << 2;
        *reinterpret_cast<f32x4*>(d + (((2 * j) ^ sw) << 4)) = (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
        *reinterpret_cast<f32x4*>(d + (((2 * j + 1) ^ sw) << 4)) = (f32x4){(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
    };
    if constexpr (A16) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {                                 // this wave's quarter (16 rows x 8 pieces) of the shared dy tile
            const int q = 64 * i + lane;
            stage16(reinterpret_cast<const __bf16*>(p.a), p.N, n0, 16 * wave + (q >> 3), q & 7, smem);
        }
    } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {                                     // this wave's quarter of the shared dy tile
        const int ins = 4 * wave + i, row = 4 * ins + (lane >> 4);
        dma16(p.a + (long)min(row, p.M - 1) * p.N + n0 + 4 * piece, lds0 + (unsigned)(ins * 1024));
    }
    }
    if constexpr (W16) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {                                 // its own x tile: 64 rows x 8 pieces
            const int q = 64 * i + lane;
            stage16(reinterpret_cast<const __bf16*>(p.w), p.K, k0, q >> 3, q & 7, smem + (1 + wave) * LWD_TILE);
        }
    } else {
#pragma unroll
    for (int ins = 0; ins < 16; ++ins) {                              // its own x tile
        const int row = 4 * ins + (lane >> 4);
        dma16(p.w + (long)min(row, p.M - 1) * p.K + k0 + 4 * piece, lds0 + (unsigned)((1 + wave) * LWD_TILE + ins * 1024));
    }
    }
    float step_size = 0.f, bc2_sqrt = 1.f;
    if constexpr (ADAM) sh_adam_coeffs(p.ad.beta1, p.ad.beta2, p.ad.step[0], p.ad.lr[0], step_size, bc2_sqrt);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!active) return;
    const char* ds = smem;
    const char* xs = smem + (1 + wave) * LWD_TILE;
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool own_bias = kt == 0 && p.dbias != nullptr;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    const int roff = rr * 256 + ((la ^ (rr << 2)) << 4);               // row (4 s + rr): (row & 3) == rr
    // ADAM: weight / exp_avg / exp_avg_sq of tile rows (tn, jj = 0..3), two sets: the first is requested here, underneath the products
    f32x4 apf[ADAM ? 2 : 1][4][3];
    auto adam_load = [&](int tn, int buf) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const long o = (long)(n0 + 16 * rr + 4 * jj + tn) * p.K + k0 + 4 * la;
            apf[buf][jj][0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.out + o));
            apf[buf][jj][1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.am + o));
            apf[buf][jj][2] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.av + o));
        }
    };
    if constexpr (ADAM) adam_load(0, 0);
    if constexpr (X3) {
        const int nst32 = (p.M + 31) >> 5;
        for (int s = 0; s < nst32; ++s) {
            f32x4 d[8], x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {                               // row 32 s + 4 j + rr = step (8 s + j) of the layout above
                d[j] = *reinterpret_cast<const f32x4*>(ds + (8 * s + j) * 1024 + roff);
                x[j] = *reinterpret_cast<const f32x4*>(xs + (8 * s + j) * 1024 + roff);
                if (32 * s + 4 * j + rr >= p.M) d[j] = (f32x4){0.f, 0.f, 0.f, 0.f};      // rows are the reduction index
                if (own_bias) bsum += d[j];
            }
            LinX3 xk[4];
#pragma unroll
            for (int tk = 0; tk < 4; ++tk)
                xk[tk] = lin_split((f32x4){x[0][tk], x[1][tk], x[2][tk], x[3][tk]}, (f32x4){x[4][tk], x[5][tk], x[6][tk], x[7][tk]});
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) {
                const LinX3 dn = lin_split((f32x4){d[0][tn], d[1][tn], d[2][tn], d[3][tn]}, (f32x4){d[4][tn], d[5][tn], d[6][tn], d[7][tn]});
#pragma unroll
                for (int tk = 0; tk < 4; ++tk) acc[tn][tk] = lin_x3_mma(dn.h, dn.m, dn.l, xk[tk].h, xk[tk].m, xk[tk].l, acc[tn][tk]);
            }
        }
    } else
    for (int s = 0; s < nsteps; ++s) {
        f32x4 d4 = *reinterpret_cast<const f32x4*>(ds + s * 1024 + roff);
        const f32x4 x4 = *reinterpret_cast<const f32x4*>(xs + s * 1024 + roff);
        if (4 * s + rr >= p.M) d4 = (f32x4){0.f, 0.f, 0.f, 0.f};       // rows are the reduction index
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int tk = 0; tk < 4; ++tk) acc[tn][tk] = __builtin_amdgcn_mfma_f32_16x16x4f32(d4[tn], x4[tk], acc[tn][tk], 0, 0, 0);
        if (own_bias) bsum += d4;
    }
    if (own_bias) {                                                   // fixed order: steps in sequence, then the four row groups
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = bsum[j];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            bsum[j] = v;
        }
        if (rr == 0) *reinterpret_cast<f32x4*>(p.dbias + n0 + 4 * la) = bsum;
    }
    if constexpr (ADAM) {
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {                                // four rows of the tile at a time, the next four in flight
            if (tn + 1 < 4) adam_load(tn + 1, (tn + 1) & 1);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const long o = (long)(n0 + 16 * rr + 4 * jj + tn) * p.K + k0 + 4 * la;
                f32x4 pp = apf[tn & 1][jj][0], mm = apf[tn & 1][jj][1], vv = apf[tn & 1][jj][2];
#pragma unroll
                for (int tk = 0; tk < 4; ++tk) {
                    float p1 = pp[tk], m1 = mm[tk], v1 = vv[tk];
                    adam_update(p1, acc[tn][tk][jj], m1, v1, p.ad.w1, p.ad.b2, p.ad.w2, p.ad.eps, p.ad.wd, step_size, bc2_sqrt);
                    pp[tk] = p1; mm[tk] = m1; vv[tk] = v1;
                }
                *reinterpret_cast<f32x4*>(p.out + o) = pp;               // the parameter is read by the next forward pass: keep it cached
                if (p.shadow) *reinterpret_cast<bf16x4*>(p.shadow + o) = sh_to_bf16x4(pp);
                __builtin_nontemporal_store(mm, reinterpret_cast<f32x4*>(p.am + o));
                __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(p.av + o));
            }
        }
    } else {
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            float* dst = p.out + (long)(n0 + 16 * rr + 4 * jj + tn) * p.K + k0 + 4 * la;
            __builtin_nontemporal_store((f32x4){acc[tn][0][jj], acc[tn][1][jj], acc[tn][2][jj], acc[tn][3][jj]}, reinterpret_cast<f32x4*>(dst));
        }
    }
}

// plan of the streaming forms: enough items for one wave per SIMD (1024), reduction ranges multiples of 16
struct LSPlan { bool ok; int range, nsplit, groups; };
LSPlan plan_stream(int M, int out_cols, int red_len, int gran = 16, bool chunked = false) {   // gran: reduction steps of the kernel (bf16x3 forms: 32)
    LSPlan pl{false, red_len, 1, out_cols / 64};
    static const int on = sh_env_int("SH_LIN_STREAM", 1, 0, 1);
    // chunked: the forward LDS-DMA kernel takes any batch as 64-row chunks (gridDim.y) when the reduction is not split
    // (only with the LDS-DMA forward kernel enabled - under SH_LIN_DMA=0 a batch above 64 rows keeps the skinny GEMM: ADVICE r5)
    static const int dma_on = sh_env_int("SH_LIN_DMA", 1, 0, 1);
    const bool big_ok = chunked && dma_on && red_len < 1024 && red_len % 32 == 0 && (out_cols / 64) % 4 == 0;
    if (!on || (M > 64 && !big_ok) || out_cols % 64 != 0 || red_len % gran != 0) return pl;
    pl.ok = true;
    static const int items = sh_env_int("SH_LIN_ITEMS", 1024, 64, 1 << 20);
    if (pl.groups < items / 2 && red_len >= 1024) {
        int ns = items / pl.groups;
        int rg = sh_cdiv(sh_cdiv(red_len, ns), gran) * gran;
        if (rg < 64) rg = 64;
        pl.range = rg;
        pl.nsplit = sh_cdiv(red_len, rg);
    }
    return pl;
}
bool aligned16(const void* a, const void* b, const void* c) {
    return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0;
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
        SH_LAUNCH_PS(ps, skinny_gemm_kernel, dim3((unsigned)grid), dim3(LTHREADS), 0, st, p);
    }
    if (p.nsplit > 1) {
        const long mn = (long)p.M * p.N;
        ShProfScope ps(st, "split_reduce_kernel");
        SH_LAUNCH_PS(ps, split_reduce_kernel, dim3((unsigned)((mn + 63) / 64)), dim3(1024), 0, st, p.slab, p.nsplit, mn, p.N,
                           bias, p.c);
    }
    SH_CHECK_LAUNCH(what);
    return SH_OK;
}

// launches the forward / backward-data streaming kernel (+ the split reduction); `cols` = output columns
// does this call run the bf16x3 kernels?  (mma_mode SH_MMA_SPLIT3 / SH_MMA_PLANES3; SH_LIN_X3=0 keeps the fp32 MFMA kernels)
inline bool lin_x3(int mma_mode) {
    static const int on = sh_env_int("SH_LIN_X3", 1, 0, 1);
    return on && (mma_mode == SH_MMA_SPLIT3 || mma_mode == SH_MMA_PLANES3);
}

template <bool FWD>
int run_stream(LSParams& p, const LSPlan& pl, int cols, void* ws, size_t ws_bytes, hipStream_t st, const char* what, bool x3 = false) {
    p.range = pl.range; p.nsplit = pl.nsplit; p.groups = pl.groups;
    const float* bias = p.bias;
    if (p.nsplit > 1) {
        SH_REQUIRE(ws && ws_bytes >= (size_t)p.nsplit * p.M * cols * sizeof(float), SH_ERR_WORKSPACE, "%s: workspace too small", what);
        p.slab = static_cast<float*>(ws);
    }
    const int items = p.groups * p.nsplit, grid = sh_cdiv(items, 4), mchunks = sh_cdiv(p.M, 64), mt = p.M > 64 ? 4 : sh_cdiv(p.M, 16);
    SH_REQUIRE(mchunks == 1 || (FWD && p.nsplit == 1), SH_ERR_UNSUPPORTED, "%s: more than 64 rows need the unsplit forward form", what);
    {
        ShProfScope ps(st, "%s<%d>|M=%d N=%d K=%d split=%d", FWD ? "linear_fwd_stream_kernel" : "linear_bwd_data_stream_kernel", mt, p.M,
                       p.N, p.K, p.nsplit);
#define SH_LS_CASE(MTV)                                                                                                   \
    if (FWD) SH_LAUNCH_PS(ps, linear_fwd_stream_kernel<MTV>, dim3(grid), dim3(LTHREADS), 0, st, p);                     \
    else SH_LAUNCH_PS(ps, linear_bwd_data_stream_kernel<MTV>, dim3(grid), dim3(LTHREADS), 0, st, p)
        static const int dma_on = sh_env_int("SH_LIN_DMA", 1, 0, 1);
        const bool dma = FWD && dma_on && p.K % 32 == 0 && p.range % 32 == 0 && p.groups % 4 == 0;
        SH_REQUIRE(mchunks == 1 || dma, SH_ERR_UNSUPPORTED, "%s: more than 64 rows need the LDS-DMA forward form (SH_LIN_DMA=1)", what);
        if (x3 && !FWD) {
            snprintf(ps.name, sizeof ps.name, "linear_bwd_data_x3_kernel<%d>|M=%d N=%d K=%d split=%d", mt, p.M, p.N, p.K, p.nsplit);
            if (mt == 1) SH_LAUNCH_PS(ps, linear_bwd_data_x3_kernel<1>, dim3(grid), dim3(LTHREADS), 0, st, p);
            else if (mt == 2) SH_LAUNCH_PS(ps, linear_bwd_data_x3_kernel<2>, dim3(grid), dim3(LTHREADS), 0, st, p);
            else if (mt == 3) SH_LAUNCH_PS(ps, linear_bwd_data_x3_kernel<3>, dim3(grid), dim3(LTHREADS), 0, st, p);
            else SH_LAUNCH_PS(ps, linear_bwd_data_x3_kernel<4>, dim3(grid), dim3(LTHREADS), 0, st, p);
        } else if (x3 && dma) {
            static bool attr3_set = false;
            if (!attr3_set) {
                const void* ks[4] = {reinterpret_cast<const void*>(linear_fwd_dma_kernel<1, true>), reinterpret_cast<const void*>(linear_fwd_dma_kernel<2, true>),
                                     reinterpret_cast<const void*>(linear_fwd_dma_kernel<3, true>), reinterpret_cast<const void*>(linear_fwd_dma_kernel<4, true>)};
                for (const void* k : ks)
                    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                        (void)hipGetLastError();
                        sh_set_error("%s: cannot raise the dynamic LDS limit to %d bytes", what, LFD_LDS);
                        return SH_ERR_LAUNCH;
                    }
                attr3_set = true;
            }
            snprintf(ps.name, sizeof ps.name, "linear_fwd_x3_kernel<%d>|M=%d N=%d K=%d split=%d", mt, p.M, p.N, p.K, p.nsplit);
            if (mt == 1) SH_LAUNCH_PS(ps, (linear_fwd_dma_kernel<1, true>), dim3(grid, mchunks), dim3(LTHREADS), LFD_LDS, st, p);
            else if (mt == 2) SH_LAUNCH_PS(ps, (linear_fwd_dma_kernel<2, true>), dim3(grid, mchunks), dim3(LTHREADS), LFD_LDS, st, p);
            else if (mt == 3) SH_LAUNCH_PS(ps, (linear_fwd_dma_kernel<3, true>), dim3(grid, mchunks), dim3(LTHREADS), LFD_LDS, st, p);
            else SH_LAUNCH_PS(ps, (linear_fwd_dma_kernel<4, true>), dim3(grid, mchunks), dim3(LTHREADS), LFD_LDS, st, p);
        } else if (dma) {
            static bool attr_set = false;
            if (!attr_set) {
                const void* ks[4] = {reinterpret_cast<const void*>(linear_fwd_dma_kernel<1>), reinterpret_cast<const void*>(linear_fwd_dma_kernel<2>),
                                     reinterpret_cast<const void*>(linear_fwd_dma_kernel<3>), reinterpret_cast<const void*>(linear_fwd_dma_kernel<4>)};
                for (const void* k : ks)
                    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                        (void)hipGetLastError();
                        sh_set_error("%s: cannot raise the dynamic LDS limit to %d bytes", what, LFD_LDS);
                        return SH_ERR_LAUNCH;
                    }
                attr_set = true;
            }
            snprintf(ps.name, sizeof ps.name, "linear_fwd_dma_kernel<%d>|M=%d N=%d K=%d split=%d", mt, p.M, p.N, p.K, p.nsplit);
            if (mt == 1) SH_LAUNCH_PS(ps, linear_fwd_dma_kernel<1>, dim3(grid, mchunks), dim3(LTHREADS), LFD_LDS, st, p);
            else if (mt == 2) SH_LAUNCH_PS(ps, linear_fwd_dma_kernel<2>, dim3(grid, mchunks), dim3(LTHREADS), LFD_LDS, st, p);
            else if (mt == 3) SH_LAUNCH_PS(ps, linear_fwd_dma_kernel<3>, dim3(grid, mchunks), dim3(LTHREADS), LFD_LDS, st, p);
            else SH_LAUNCH_PS(ps, linear_fwd_dma_kernel<4>, dim3(grid, mchunks), dim3(LTHREADS), LFD_LDS, st, p);
        } else if (mt == 1) { SH_LS_CASE(1); } else if (mt == 2) { SH_LS_CASE(2); } else if (mt == 3) { SH_LS_CASE(3); } else { SH_LS_CASE(4); }
#undef SH_LS_CASE
    }
    if (p.nsplit > 1) {
        const long mn = (long)p.M * cols;
        ShProfScope ps(st, "split_reduce_kernel");
        SH_LAUNCH_PS(ps, split_reduce_kernel, dim3((unsigned)((mn + 63) / 64)), dim3(1024), 0, st, p.slab, p.nsplit, mn, cols, bias,
                           p.out);
    }
    SH_CHECK_LAUNCH(what);
    return SH_OK;
}

}  // namespace

// Measured and not kept (round 3, MI355X, batch 64, 256 x 55 296): the six latent GEMMs in the bf16x3 form of the conv kernels
// (both fp32 operands split while staged, six bf16 MFMAs per product) took 22.8-32.7 us each against 25.5-33.5 us here - these
// passes are not bound by the matrix pipe but by how many bytes of the 56.6 MB weight stream a CU keeps in flight.

extern "C" {

size_t sh_linear_workspace(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    size_t need = 0;
    {   // streaming forms: forward splits K (output [M][N]), backward-data splits N (output [M][K])
        for (int gran : {16, 32}) {                                  // fp32 MFMA forms / bf16x3 forms
            const LSPlan f = plan_stream(M, N, K, gran), d = plan_stream(M, K, N, gran);
            if (f.ok && f.nsplit > 1 && (size_t)f.nsplit * M * N * sizeof(float) > need) need = (size_t)f.nsplit * M * N * sizeof(float);
            if (d.ok && d.nsplit > 1 && (size_t)d.nsplit * M * K * sizeof(float) > need) need = (size_t)d.nsplit * M * K * sizeof(float);
        }
    }
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
                  size_t workspace_bytes, int mma_mode, sh_stream_t stream) {
    SH_REQUIRE(x && weight && y && M > 0 && N > 0 && K > 0, SH_ERR_INVALID_ARG, "sh_linear_fwd: bad argument");
    SH_REQUIRE(sh_mma_mode_valid(mma_mode), SH_ERR_INVALID_ARG, "sh_linear_fwd: unknown mma_mode %d", mma_mode);
    if (lin_x3(mma_mode) && M > 64 && K % 32 == 0 && K <= 384 && N % 64 == 0 && aligned16(x, weight, y) &&
        (!bias || (reinterpret_cast<uintptr_t>(bias) & 15) == 0)) {
        // large batch, short reduction: the weight tile split once per workgroup (linear_fwd_wide_x3_kernel)
        static const int wide_on = sh_env_int("SH_LIN_WIDE", 1, 0, 1);
        if (wide_on) {
            hipStream_t st = static_cast<hipStream_t>(stream);
            const size_t smem = (size_t)(K / 32) * 4 * 3072;
            static bool attr_set = false;
            if (!attr_set) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(linear_fwd_wide_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        160 * 1024) != hipSuccess) {
                    (void)hipGetLastError();
                    sh_set_error("linear_fwd: cannot raise the dynamic LDS limit to %zu bytes", smem);
                    return SH_ERR_LAUNCH;
                }
                attr_set = true;
            }
            LSParams s{};
            s.a = x; s.w = weight; s.bias = bias; s.out = y; s.M = M; s.N = N; s.K = K;
            ShProfScope ps(st, "linear_fwd_wide_x3_kernel|M=%d N=%d K=%d", M, N, K);
            SH_LAUNCH_PS(ps, linear_fwd_wide_x3_kernel, dim3(N / 64), dim3(LFW_WAVES * 64), smem, st, s);
            SH_CHECK_LAUNCH("linear_fwd_wide");
            return SH_OK;
        }
    }
    {
        // bf16x3 form: the LDS-DMA kernel's shapes (32-wide stages, four column groups per workgroup); anything else keeps fp32 MFMA
        const bool x3 = lin_x3(mma_mode) && K % 32 == 0 && (N / 64) % 4 == 0;
        static const int chunk_on = sh_env_int("SH_LIN_CHUNKED", 1, 0, 1);
        const LSPlan pl = plan_stream(M, N, K, x3 ? 32 : 16, chunk_on && M > 64);
        if (pl.ok && aligned16(x, weight, y) && (!bias || (reinterpret_cast<uintptr_t>(bias) & 15) == 0)) {
            LSParams s{};
            s.a = x; s.w = weight; s.bias = bias; s.out = y; s.M = M; s.N = N; s.K = K;
            return run_stream<true>(s, pl, N, workspace, workspace_bytes, static_cast<hipStream_t>(stream), "linear_fwd", x3);
        }
    }
    SGParams p{};
    p.a = x; p.a_sm = K; p.a_sr = 1;
    p.b = weight; p.b_sn = K; p.b_sr = 1;
    p.c = y; p.c_sm = N; p.c_sn = 1; p.bias = bias;
    p.M = M; p.N = N; p.R = K;
    return run_gemm(p, workspace, workspace_bytes, static_cast<hipStream_t>(stream), "linear_fwd");
}

int sh_linear_bwd_data(const float* dy, const float* weight, float* dx, int M, int N, int K, void* workspace,
                       size_t workspace_bytes, int mma_mode, sh_stream_t stream) {
    SH_REQUIRE(dy && weight && dx && M > 0 && N > 0 && K > 0, SH_ERR_INVALID_ARG, "sh_linear_bwd_data: bad argument");
    SH_REQUIRE(sh_mma_mode_valid(mma_mode), SH_ERR_INVALID_ARG, "sh_linear_bwd_data: unknown mma_mode %d", mma_mode);
    {
        const bool x3 = lin_x3(mma_mode) && N % 32 == 0;
        const LSPlan pl = plan_stream(M, K, N, x3 ? 32 : 16);
        if (pl.ok && aligned16(dy, weight, dx)) {
            LSParams s{};
            s.a = dy; s.w = weight; s.bias = nullptr; s.out = dx; s.M = M; s.N = N; s.K = K;
            return run_stream<false>(s, pl, K, workspace, workspace_bytes, static_cast<hipStream_t>(stream), "linear_bwd_data", x3);
        }
    }
    SGParams p{};                                   // dx(m,k) = sum_n dy(m,n) W(n,k)
    p.a = dy; p.a_sm = N; p.a_sr = 1;
    p.b = weight; p.b_sn = 1; p.b_sr = K;           // B(k, n) = W[n*K + k]
    p.c = dx; p.c_sm = K; p.c_sn = 1; p.bias = nullptr;
    p.M = M; p.N = K; p.R = N;
    return run_gemm(p, workspace, workspace_bytes, static_cast<hipStream_t>(stream), "linear_bwd_data");
}

// the LDS-DMA weight-gradient kernels need 80 KiB of dynamic LDS: raise the limit once
static bool wgt_dma_attr() {
    static bool attr_set = false;
    if (!attr_set) {
        for (const void* k : {reinterpret_cast<const void*>(linear_bwd_wgt_dma_kernel<false>), reinterpret_cast<const void*>(linear_bwd_wgt_dma_kernel<true>),
                              reinterpret_cast<const void*>(linear_bwd_wgt_dma_kernel<false, true>), reinterpret_cast<const void*>(linear_bwd_wgt_dma_kernel<true, true>),
                              reinterpret_cast<const void*>(linear_bwd_wgt_dma_kernel<false, true, true, false>),
                              reinterpret_cast<const void*>(linear_bwd_wgt_dma_kernel<false, true, false, true>),
                              reinterpret_cast<const void*>(linear_bwd_wgt_dma_kernel<false, true, true, true>)})
            if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                (void)hipGetLastError();
                sh_set_error("linear_bwd_wgt: cannot raise the dynamic LDS limit to %d bytes", LWD_LDS);
                return false;
            }
        attr_set = true;
    }
    return true;
}

int sh_linear_bwd_wgt(const float* dy, const float* x, float* dW, float* dbias, int M, int N, int K, void* workspace,
                      size_t workspace_bytes, int mma_mode, sh_stream_t stream) {
    SH_REQUIRE(dy && x && dW && M > 0 && N > 0 && K > 0, SH_ERR_INVALID_ARG, "sh_linear_bwd_wgt: bad argument");
    SH_REQUIRE(sh_mma_mode_valid(mma_mode), SH_ERR_INVALID_ARG, "sh_linear_bwd_wgt: unknown mma_mode %d", mma_mode);
    hipStream_t st = static_cast<hipStream_t>(stream);
    static const int stream_on = sh_env_int("SH_LIN_STREAM", 1, 0, 1);
    if (stream_on && M <= 64 && N % 64 == 0 && K % 64 == 0 && aligned16(dy, x, dW)) {
        LSParams s{};
        s.a = dy; s.w = x; s.out = dW; s.M = M; s.N = N; s.K = K;
        s.dbias = (dbias && (reinterpret_cast<uintptr_t>(dbias) & 15) == 0) ? dbias : nullptr;      // fused into the tile kernel
        const int items = (N / 64) * (K / 64);
        static const int dma_on = sh_env_int("SH_LIN_WGT_DMA", 1, 0, 1);
        if (dma_on) {
            if (!wgt_dma_attr()) return SH_ERR_LAUNCH;
            const int wgs = (N / 64) * sh_cdiv(K / 64, 4);
            if (lin_x3(mma_mode)) {
                ShProfScope ps(st, "linear_bwd_wgt_x3_kernel|M=%d N=%d K=%d", M, N, K);
                SH_LAUNCH_PS(ps, linear_bwd_wgt_dma_kernel<true>, dim3(wgs), dim3(LTHREADS), LWD_LDS, st, s);
            } else {
                ShProfScope ps(st, "linear_bwd_wgt_dma_kernel|M=%d N=%d K=%d", M, N, K);
                SH_LAUNCH_PS(ps, linear_bwd_wgt_dma_kernel<false>, dim3(wgs), dim3(LTHREADS), LWD_LDS, st, s);
            }
        } else {
            ShProfScope ps(st, "linear_bwd_wgt_stream_kernel|M=%d N=%d K=%d", M, N, K);
            SH_LAUNCH_PS(ps, linear_bwd_wgt_stream_kernel, dim3(sh_cdiv(items, 4)), dim3(LTHREADS), 0, st, s);
        }
        if (dbias && !s.dbias) hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)sh_cdiv(N, 256)), dim3(256), 0, st, dy, M, N, dbias);
        SH_CHECK_LAUNCH("linear_bwd_wgt");
        return SH_OK;
    }
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

int sh_linear_bwd_wgt_adam_ok(int M, int N, int K) {
    static const int on = sh_env_int("SH_LIN_WGT_ADAM", 1, 0, 1);
    return on && M > 0 && M <= 64 && N > 0 && K > 0 && N % 64 == 0 && K % 64 == 0;
}

int sh_linear_bwd_wgt_adam(const void* dy, int dy_dtype, const void* x, int x_dtype, float* weight, void* weight_bf16, float* exp_avg,
                           float* exp_avg_sq, const float* step, const float* lr, double beta1, double beta2, double eps, double weight_decay,
                           float* dbias, int M, int N, int K, int mma_mode, sh_stream_t stream) {
    SH_REQUIRE(dy && x && weight && exp_avg && exp_avg_sq && step && lr && M > 0 && N > 0 && K > 0, SH_ERR_INVALID_ARG,
               "sh_linear_bwd_wgt_adam: bad argument");
    SH_REQUIRE(sh_mma_mode_valid(mma_mode), SH_ERR_INVALID_ARG, "sh_linear_bwd_wgt_adam: unknown mma_mode %d", mma_mode);
    SH_REQUIRE((dy_dtype == SH_DTYPE_F32 || dy_dtype == SH_DTYPE_BF16) && (x_dtype == SH_DTYPE_F32 || x_dtype == SH_DTYPE_BF16), SH_ERR_INVALID_ARG,
               "sh_linear_bwd_wgt_adam: unknown element type");
    SH_REQUIRE(beta1 >= 0 && beta1 < 1 && beta2 >= 0 && beta2 < 1 && eps >= 0 && weight_decay >= 0, SH_ERR_INVALID_ARG,
               "sh_linear_bwd_wgt_adam: hyper-parameter out of range");
    SH_REQUIRE(sh_linear_bwd_wgt_adam_ok(M, N, K) && aligned16(dy, x, weight) && aligned16(exp_avg, exp_avg_sq, weight) &&
               (reinterpret_cast<uintptr_t>(weight_bf16) & 7) == 0, SH_ERR_UNSUPPORTED,
               "sh_linear_bwd_wgt_adam: M=%d N=%d K=%d is not served by the tile kernel (M <= 64, N and K multiples of 64, 16-byte aligned "
               "tensors): use sh_linear_bwd_wgt + sh_adam_step", M, N, K);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!wgt_dma_attr()) return SH_ERR_LAUNCH;
    LSParams s{};
    s.a = static_cast<const float*>(dy); s.w = static_cast<const float*>(x); s.out = weight; s.M = M; s.N = N; s.K = K;
    s.am = exp_avg; s.av = exp_avg_sq; s.shadow = static_cast<__bf16*>(weight_bf16);
    s.ad.lr = lr; s.ad.step = step; s.ad.beta1 = beta1; s.ad.beta2 = beta2;
    s.ad.w1 = (float)(1.0 - beta1); s.ad.b2 = (float)beta2; s.ad.w2 = (float)(1.0 - beta2); s.ad.eps = (float)eps; s.ad.wd = (float)weight_decay;
    const bool a16 = dy_dtype == SH_DTYPE_BF16, w16 = x_dtype == SH_DTYPE_BF16;
    // the bias gradient is summed from the dy tile in LDS, whatever it was loaded from
    s.dbias = (dbias && (reinterpret_cast<uintptr_t>(dbias) & 15) == 0) ? dbias : nullptr;
    SH_REQUIRE(!dbias || s.dbias || !a16, SH_ERR_UNSUPPORTED, "sh_linear_bwd_wgt_adam: a bf16 dy needs a 16-byte aligned dbias");
    const int wgs = (N / 64) * sh_cdiv(K / 64, 4);
    if (a16 || w16) {                                  // bf16 operands: their products are exact on the fp32 MFMA
        ShProfScope ps(st, "linear_bwd_wgt_adam_kernel|M=%d N=%d K=%d dy=%s x=%s", M, N, K, a16 ? "bf16" : "f32", w16 ? "bf16" : "f32");
        if (a16 && w16) SH_LAUNCH_PS(ps, (linear_bwd_wgt_dma_kernel<false, true, true, true>), dim3(wgs), dim3(LTHREADS), LWD_LDS, st, s);
        else if (a16)   SH_LAUNCH_PS(ps, (linear_bwd_wgt_dma_kernel<false, true, true, false>), dim3(wgs), dim3(LTHREADS), LWD_LDS, st, s);
        else            SH_LAUNCH_PS(ps, (linear_bwd_wgt_dma_kernel<false, true, false, true>), dim3(wgs), dim3(LTHREADS), LWD_LDS, st, s);
    } else if (lin_x3(mma_mode)) {
        ShProfScope ps(st, "linear_bwd_wgt_adam_x3_kernel|M=%d N=%d K=%d", M, N, K);
        SH_LAUNCH_PS(ps, (linear_bwd_wgt_dma_kernel<true, true>), dim3(wgs), dim3(LTHREADS), LWD_LDS, st, s);
    } else {
        ShProfScope ps(st, "linear_bwd_wgt_adam_kernel|M=%d N=%d K=%d", M, N, K);
        SH_LAUNCH_PS(ps, (linear_bwd_wgt_dma_kernel<false, true>), dim3(wgs), dim3(LTHREADS), LWD_LDS, st, s);
    }
    if (dbias && !s.dbias) hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)sh_cdiv(N, 256)), dim3(256), 0, st, static_cast<const float*>(dy), M, N, dbias);
    SH_CHECK_LAUNCH("linear_bwd_wgt_adam");
    return SH_OK;
}

}  // extern "C"
