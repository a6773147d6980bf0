// Spiral convolution in bf16 on CDNA4 (gfx950) - the reduced-precision path of BASELINE config 3.
//
//   forward        y[r,b,:]  = act( sum_s x[table[r,s],b,:] . W_s^T + bias )       (reference models.py:40-51)
//   backward-data  dx[u,b,:] = sum_s dpre[table_t[u,s],b,:] . W_s                   (autograd of :42,:45; the same kernel
//                                                                                   over the transposed table)
// bf16 activations and working weights, fp32 accumulation (v_mfma_f32_16x16x32_bf16), bias / activation / mask applied in
// fp32 before the result is rounded to bf16 once.  At 16x the fp32 MFMA rate the layer is no longer arithmetic-bound: what
// is left is the S-fold neighbour gather (L2 -> CU) and the HBM traffic of the activations, so the kernel is built as a
// gather stream:
//   * the whole weight of the layer (or of this workgroup's slice of the output channels) sits in LDS for the lifetime of
//     the workgroup, pre-ordered into MFMA fragments by sh_conv_wfrag_prep_multi: a fragment read is one conflict-free
//     ds_read_b128 at base + 16 * lane.  Nothing else uses LDS and the main loop has NO barrier;
//   * a WAVE owns work items of RT vertices x 16 batch entries.  Lane (r = lane & 15, kq = lane >> 4) loads the 16-byte
//     piece x[nbr(v, s)][b0 + r][c .. c+7] it feeds to the matrix pipe itself (B operand); the gather table line of a
//     vertex lives in one VGPR across the wave (lane l holds table[v][l]) and the entry of the current k-step is read with
//     v_readlane into an SGPR -> the gathered row's base address is scalar, the per-lane part is a loop constant;
//   * workgroups are persistent: a wave walks its share of the work items of its XCD's contiguous chunk (neighbouring
//     vertices share gathered rows -> L2 hits), D k-steps of loads in flight ahead of the MFMAs;
//   * operands are swapped (A = weight fragment, B = gathered rows) so a lane ends up with 4 CONSECUTIVE output channels
//     of one (vertex, batch) row -> 8-byte bf16 stores.
// 3-channel fp32 tensors (the xyz input of the first encoder layer, the xyz gradient entering the last decoder layer) are
// read as fp32 triples and padded to quads in registers; a 3-channel fp32 OUTPUT (x_hat, dL/dx) is written as fp32.
#include "sh_bf16.h"

#include <type_traits>

namespace {

// gathered channels % 32 == 0 | == 16 | == 3 (fp32 triples) | % 64 == 0 in the FULL-LINE form (BC_C64):
// with 128- or 256-byte rows a k-step of the C32 form touches 16 rows x 64 bytes - sixteen half lines per load instruction -
// and those layers gathered at half the bytes/us of the 64-byte-row layers (where an instruction is 1 KiB contiguous).  Here a
// load instruction covers EIGHT rows x one whole 128-byte line (64 channels): lane (i, kb) fetches piece kb + 4 (i & 1) of row
// i >> 1.  For the MFMA the two lanes of a row pose as two logical rows holding the lower / upper 32 channels; one product with
// the lower-half weight fragments (right for the even logical rows) and one with the upper-half fragments (right for the odd
// ones) go to two accumulators whose valid halves are added across the lane pair in the epilogue.  Twice the MFMAs (the
// matrix pipes idle ~90 % of these launches), half as many, twice as efficient load instructions.
// BC_C32C (round 3): the C32 form with LINE-WISE loads.  In the MFMA operand layout the 16 rows of a tile sit on 16 consecutive lanes,
// so every quad of lanes of a load instruction touches four different cache lines and the vector L1 serves the instruction as
// 64 separate 16-byte accesses - measured (tools/exp/ta_probe.hip, rows resident in the XCD's L2): 6.8-8.7 TB/s of gathered
// bytes chip-wide in that layout against 16.6-21.9 TB/s when four consecutive lanes read 64 contiguous bytes of ONE row.  Here
// lane l loads piece l & 3 of row l >> 2 (16 accesses of 64 bytes per instruction) and the pieces are turned into the MFMA
// layout - lane (r, kq) takes the piece of lane 4 r + kq - by four ds_bpermute_b32 when the k-step is multiplied (the LDS
// crossbar, no LDS memory: all of it holds the weight).
enum { BC_C32 = 0, BC_C16 = 1, BC_C3F = 2, BC_C64 = 3, BC_C32C = 4 };

struct BCParams {
    const char* x; long x_rb, x_bb;            // byte strides of (row, batch entry)
    const int* table;                          // [R][S]
    const u32x4* wfrag;                        // [nks][nt_tot][64] 16-byte fragments
    const float* bias;
    char* y; long y_rb, y_bb;
    const char* yprev; long yp_rb, yp_bb;      // bf16 output of the layer that produced x (backward epilogue)
    int B, R, S, Cg, Nout, nks, nt_tot;
    int act, zero_row;
    int n_vg, n_tiles, nsplit;
    // ragged backward-data (conv_bf16r_kernel): per output row its sources as a LIST - rag_rows [R][rag_L] rows of x, rag_pos [R][rag_L]
    // the spiral position whose weight multiplies each (-1 behind the last source); ncg = Cg / 32
    const int* rag_rows; const int* rag_pos; int rag_L, ncg;
};

struct __attribute__((packed, aligned(4))) bc_f3 { float a, b, c; };

constexpr int bc_depth(int NT, int RT, int MODE) {
    const int per = RT * (MODE == BC_C3F ? 6 : 4);         // registers of one ring slot
    const int room = 88 - (MODE == BC_C64 ? 2 : 1) * NT * RT * 4 - (MODE == BC_C32C ? 3 : 1) * RT * 4;     // minus accumulators (two sets in the full-line form) and the converted / permuted operands
    const int d = room / per;
    const int cap = MODE == BC_C3F ? 4 : 8;
    const int lo = 3;
    return d < lo ? lo : d > cap ? cap : d;
}

template <int J, int D, class F>
__device__ __forceinline__ bool bc_ring_steps(int ks, int nks, F&& f) {
    f(std::integral_constant<int, J>{}, ks + J);
    if constexpr (J + 1 < D) {
        if (ks + J + 1 >= nks) return false;               // explicit early exit: the compiler must see that a skipped step ends the loop
        return bc_ring_steps<J + 1, D>(ks, nks, f);
    } else {
        return true;
    }
}

template <int NT, int RT, int MODE, bool BWD, bool OUTF32>
__global__ __launch_bounds__(1024) void conv_bf16_kernel(const BCParams p) {
    // k-steps of gathered loads in flight per wave: what bounds these layers is bytes in flight per CU (a wave's k-step is
    // RT KiB, the round trip ~2 us), so as deep as the 128-VGPR budget allows next to the accumulators
    constexpr int D = bc_depth(NT, RT, MODE);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* Wl = reinterpret_cast<u32x4*>(smem);            // [nks][NT][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;  // blocks b and b+8 share an XCD (speed only, never correctness)
    const int nwg_x = ((int)gridDim.x - xcd + 7) >> 3;
    const int slice = li % p.nsplit, lj = li / p.nsplit;   // the host launches a multiple of 8 * nsplit workgroups
    const int ngrp = nwg_x / p.nsplit;
    const int q8 = p.n_tiles >> 3, r8 = p.n_tiles & 7;
    const int t_begin = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int t_end = t_begin + (xcd < r8 ? q8 + 1 : q8);
    const int stride = ngrp * nw;
    const int r16 = lane & 15, kq = lane >> 4;
    const int S = p.S, sl = lane < S ? lane : S - 1;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    auto load_table = [&](int t, int (&tv)[RT]) {
        const int tt = t < t_end ? t : (t_end > 0 ? t_end - 1 : 0);      // clamped: a valid (possibly unused) table line
        const int vg = tt % p.n_vg;
#pragma unroll
        for (int m = 0; m < RT; ++m) {
            const int v = vg * RT + m < p.R ? vg * RT + m : p.R - 1;
            tv[m] = p.table[(long)v * S + sl];
        }
    };

    int t = t_begin + lj * nw + wave;
    int tv[RT], tvn[RT];
    load_table(t, tv);                                     // first table line: in flight under the weight copy
    {
        // weight fragments -> LDS by LDS-DMA: one 1-KiB fragment (k-step, channel tile) per wave instruction, all of a wave's
        // instructions in flight at once, no register round trip (the load -> ds_write loop this replaces cost ~7 us of
        // every launch of a 128-KiB layer)
        typedef __attribute__((address_space(3))) char* lptr_t;
        const unsigned wl_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lptr_t)smem);
        const int nfrag = p.nks * NT;
        for (int f = __builtin_amdgcn_readfirstlane(wave); f < nfrag; f += nw) {
            const int n = f % NT, ks = f / NT;
            const char* src = reinterpret_cast<const char*>(p.wfrag + ((long)ks * p.nt_tot + slice * NT + n) * 64 + lane);
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(__builtin_amdgcn_readfirstlane(wl_lds + (unsigned)f * 1024u)) : "memory", "m0");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    for (; t < t_end; t += stride) {
        load_table(t + stride, tvn);                       // next item's table line: in flight under this item's k loop
        const int bs = t / p.n_vg, vg = t - bs * p.n_vg;
        constexpr int TROWS = MODE == BC_C64 ? 8 : 16;      // batch entries per tile
        const int b0 = bs * TROWS, v0 = vg * RT;
        const int rl = MODE == BC_C64 ? r16 >> 1 : r16;     // this lane's row inside the tile
        const int bl = b0 + rl < p.B ? b0 + rl : p.B - 1;               // rows past B read the last entry; never stored
        const int bl_co = b0 + (lane >> 2) < p.B ? b0 + (lane >> 2) : p.B - 1;      // line-wise loads: lane -> (row lane >> 2, piece lane & 3)
        const char* xl = MODE == BC_C32C ? p.x + (long)bl_co * p.x_bb + (lane & 3) * 16
                                         : p.x + (long)bl * p.x_bb +
                         (MODE == BC_C32 ? kq * 16 : MODE == BC_C64 ? (kq + 4 * (r16 & 1)) * 16 : MODE == BC_C16 ? (kq & 1) * 16 : 0);

        // running load position (uniform): spiral position and channel offset of the next k-step to load
        int ls = 0, lc = 0;
        constexpr int KPS = MODE == BC_C64 ? 2 : 1;        // k-steps per ring slot
        using raw_t = typename std::conditional<MODE == BC_C3F, bc_f3[2], bf16x8>::type;
        raw_t ring[D][RT];
        auto issue = [&](raw_t (&a)[RT]) {
            if constexpr (MODE == BC_C32 || MODE == BC_C32C) {
                const int s = ls < S ? ls : S - 1;
#pragma unroll
                for (int m = 0; m < RT; ++m) {
                    const int row = __builtin_amdgcn_readlane(tv[m], s);
                    *reinterpret_cast<bf16x8*>(&a[m]) = *reinterpret_cast<const bf16x8*>(xl + (long)row * p.x_rb + 2 * lc);
                }
                lc += 32;
                if (lc >= p.Cg) { lc = 0; ++ls; }
            } else if constexpr (MODE == BC_C64) {
                const int s = ls < S ? ls : S - 1;
#pragma unroll
                for (int m = 0; m < RT; ++m) {
                    const int row = __builtin_amdgcn_readlane(tv[m], s);
                    *reinterpret_cast<bf16x8*>(&a[m]) = *reinterpret_cast<const bf16x8*>(xl + (long)row * p.x_rb + 2 * lc);
                }
                lc += 64;
                if (lc >= p.Cg) { lc = 0; ++ls; }
            } else if constexpr (MODE == BC_C16) {
                const int s0 = ls < S ? ls : S - 1, s1 = ls + 1 < S ? ls + 1 : S - 1;
#pragma unroll
                for (int m = 0; m < RT; ++m) {
                    const int r0 = __builtin_amdgcn_readlane(tv[m], s0), r1 = __builtin_amdgcn_readlane(tv[m], s1);
                    const int row = (kq & 2) ? r1 : r0;
                    *reinterpret_cast<bf16x8*>(&a[m]) = *reinterpret_cast<const bf16x8*>(xl + (long)row * p.x_rb);
                }
                ls += 2;
            } else {
                const int sa = ls + 2 * kq, sb = sa + 1;
                const int ca = sa < S ? sa : S - 1, cb = sb < S ? sb : S - 1;
#pragma unroll
                for (int m = 0; m < RT; ++m) {
                    const int ra = __shfl(tv[m], ca, 64), rb = __shfl(tv[m], cb, 64);
                    bc_f3* dst = reinterpret_cast<bc_f3*>(&a[m]);
                    dst[0] = *reinterpret_cast<const bc_f3*>(xl + (long)ra * p.x_rb);
                    dst[1] = *reinterpret_cast<const bc_f3*>(xl + (long)rb * p.x_rb);
                }
                ls += 8;
            }
        };
        f32x4 acc[RT][NT], acc2[MODE == BC_C64 ? RT : 1][MODE == BC_C64 ? NT : 1];
#pragma unroll
        for (int m = 0; m < RT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                acc[m][n] = zero4;
                if constexpr (MODE == BC_C64) acc2[m][n] = zero4;
            }
        auto compute = [&](int slot, const raw_t (&a)[RT]) {
#pragma unroll
            for (int h = 0; h < KPS; ++h) {
                bf16x8 g[RT];
#pragma unroll
                for (int m = 0; m < RT; ++m) {
                    if constexpr (MODE == BC_C3F) {
                        const bc_f3* s2 = reinterpret_cast<const bc_f3*>(&a[m]);
                        g[m] = (bf16x8){(__bf16)s2[0].a, (__bf16)s2[0].b, (__bf16)s2[0].c, (__bf16)0.f,
                                        (__bf16)s2[1].a, (__bf16)s2[1].b, (__bf16)s2[1].c, (__bf16)0.f};
                    } else if constexpr (MODE == BC_C32C) {                  // loaded line-wise: fetch this lane's operand from lane 4 r + kq
                        const u32x4 raw = *reinterpret_cast<const u32x4*>(&a[m]);
                        const int src = (4 * r16 + kq) << 2;
                        u32x4 t;
#pragma unroll
                        for (int j = 0; j < 4; ++j) t[j] = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)raw[j]);
                        g[m] = *reinterpret_cast<const bf16x8*>(&t);
                    } else {
                        g[m] = *reinterpret_cast<const bf16x8*>(&a[m]);      // full-line form: the same operand for both halves
                    }
                }
                const u32x4* wk = Wl + ((long)(slot * KPS + h) * NT) * 64 + lane;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const u32x4 wraw = wk[n * 64];
                    const bf16x8 w = *reinterpret_cast<const bf16x8*>(&wraw);
#pragma unroll
                    for (int m = 0; m < RT; ++m) {
                        if (MODE == BC_C64 && h == 1) acc2[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, g[m], acc2[m][n], 0, 0, 0);
                        else acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, g[m], acc[m][n], 0, 0, 0);
                    }
                }
            }
        };

        // ---- k loop: ring of D register sets, statically named (runtime-indexed vector arrays would go to scratch)
#pragma unroll
        for (int d = 0; d < D - 1; ++d) issue(ring[d]);
        auto step = [&](auto J, int ks) {
            constexpr int j = decltype(J)::value;
            issue(ring[(j + D - 1) % D]);
            __builtin_amdgcn_sched_barrier(0);             // the prefetch loads stay ahead of the MFMAs
            compute(ks, ring[j]);
        };
        const int nslots = p.nks / KPS;
        for (int ks = 0; ks < nslots; ks += D)
            if (!bc_ring_steps<0, D>(ks, nslots, step)) break;

        // ---- epilogue: lane holds channels c0..c0+3 (c0 = 16 n + 4 kq) of row (v0 + m, b0 + r16)
        const int b = b0 + rl;
        if constexpr (MODE == BC_C64) {                    // even logical row: lower-half products; its odd neighbour: upper-half products
#pragma unroll
            for (int m = 0; m < RT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[m][n][j] += __shfl_xor(acc2[m][n][j], 1, 64);
        }
        const bool owner = MODE != BC_C64 || (r16 & 1) == 0;           // full-line form: the even lane of a pair stores the row
#pragma unroll
        for (int m = 0; m < RT; ++m) {
            const int v = v0 + m;
            if (v >= p.R || b >= p.B || !owner) continue;
            char* yrow = p.y + (long)v * p.y_rb + (long)b * p.y_bb;
            const char* yp = (BWD && p.yprev) ? p.yprev + (long)v * p.yp_rb + (long)b * p.yp_bb : nullptr;
            const bool zero = v == p.zero_row;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int c0 = (slice * NT + n) * 16 + kq * 4;
                if (c0 >= p.Nout) continue;
                f32x4 a = acc[m][n];
                if (!BWD) {
                    if (p.bias) {
                        if (OUTF32) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) a[j] += c0 + j < p.Nout ? p.bias[c0 + j] : 0.f;
                        } else {
                            a += *reinterpret_cast<const f32x4*>(p.bias + c0);
                        }
                    }
                    a = sh_act_fwd4(a, p.act);
                } else if (yp) {
                    const f32x4 yv = sh_from_bf16x4(*reinterpret_cast<const bf16x4*>(yp + 2 * c0));
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[j] *= sh_act_grad_from_out(yv[j], p.act);
                }
                if (zero) a = zero4;
                if (OUTF32) {
                    float* dst = reinterpret_cast<float*>(yrow) + c0;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (c0 + j < p.Nout) dst[j] = a[j];
                } else {
                    *reinterpret_cast<bf16x4*>(yrow + 2 * c0) = sh_to_bf16x4(a);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < RT; ++m) tv[m] = tvn[m];
    }
}

// ------------------------------------------------------------------------------------------
// Backward-data over RAGGED source lists (round 6; the bf16 sibling of p3_conv.hip's conv_p3r_kernel).  The dense transposed
// table has one slot per (input row, spiral position): a slot nobody reads from points at the zero row (more than half of the
// slots on the down-sampling levels - their loads are issued all the same), a slot several output rows read through points at an
// extra row that a pre-sum launch (spmm_bf16: eight launches of a backward pass here, ~55 us of a 0.9-ms step) has to fill first.
// A list of (source row, position) pairs per input row has neither: every source is a real row of dpre and the sums are formed by
// the matrix pipe (linearity) - in fp32, where the pre-summed rows were rounded to bf16 once more.  One vertex x 16 batch entries
// per wave item (two vertices could not share weight-fragment reads: their positions differ step by step), line-wise loads as in
// BC_C32C (lane l: piece l & 3 of batch row l >> 2), gathered channels % 32 == 0, bf16 on both sides.
template <int NT>
__global__ __launch_bounds__(1024) void conv_bf16r_kernel(const BCParams p) {
    constexpr int D = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const u32x4* Wl = reinterpret_cast<const u32x4*>(smem);       // [nks][NT][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
    const int nwg_x = ((int)gridDim.x - xcd + 7) >> 3;
    const int slice = li % p.nsplit, lj = li / p.nsplit;
    const int ngrp = nwg_x / p.nsplit;
    const int q8 = p.n_tiles >> 3, r8 = p.n_tiles & 7;
    const int t_begin = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int t_end = t_begin + (xcd < r8 ? q8 + 1 : q8);
    const int stride = ngrp * nw;
    const int r16 = lane & 15, kq = lane >> 4;
    const int Lp = p.rag_L, ll = lane < Lp ? lane : Lp - 1;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const unsigned rowmul = (unsigned)(p.x_rb >> 4);
    // the list of an item's vertex: lane j holds source j (row pre-multiplied by the row stride in 16-byte units) and its position
    auto load_list = [&](int t, int& rows, int& pos) {
        const int tt = t < t_end ? t : (t_end > 0 ? t_end - 1 : 0);
        const int v = tt % p.n_vg;
        rows = (int)((unsigned)p.rag_rows[(long)v * Lp + ll] * rowmul);
        pos = lane < Lp ? p.rag_pos[(long)v * Lp + ll] : -1;
    };
    int t = t_begin + lj * nw + wave;
    int tv, tp, tvn, tpn;
    load_list(t, tv, tp);
    {
        typedef __attribute__((address_space(3))) char* lptr_t;
        const unsigned wl_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lptr_t)smem);
        const int nfrag = p.nks * NT;
        for (int f = __builtin_amdgcn_readfirstlane(wave); f < nfrag; f += nw) {
            const int n = f % NT, ks = f / NT;
            const char* src = reinterpret_cast<const char*>(p.wfrag + ((long)ks * p.nt_tot + slice * NT + n) * 64 + lane);
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(__builtin_amdgcn_readfirstlane(wl_lds + (unsigned)f * 1024u)) : "memory", "m0");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const int perm = (4 * r16 + kq) << 2;                  // the MFMA operand of lane (r, kq) is the piece lane 4 r + kq loaded
    for (; t < t_end; t += stride) {
        load_list(t + stride, tvn, tpn);
        const int bs = t / p.n_vg, v = t - bs * p.n_vg;
        const int b0 = bs * 16;
        const int bl_co = b0 + (lane >> 2) < p.B ? b0 + (lane >> 2) : p.B - 1;
        const char* xl = p.x + (long)bl_co * p.x_bb + (lane & 3) * 16;
        const int L = __builtin_popcountll(__builtin_amdgcn_ballot_w64(tp >= 0));      // sources of this vertex (uniform)
        const int nks = L * p.ncg;
        int lj2 = 0, lc = 0;                               // running load position: list entry, channel group (uniform)
        u32x4 ring[D];
        auto issue = [&](u32x4& a) {
            const int j = lj2 < L ? lj2 : (L > 0 ? L - 1 : 0);      // past the end: the last entry again (never multiplied)
            const int row = __builtin_amdgcn_readlane(tv, j);
            a = *reinterpret_cast<const u32x4*>(xl + (long)((unsigned long)(unsigned)row << 4) + (long)lc * 64);
            if (++lc >= p.ncg) { lc = 0; ++lj2; }
        };
        f32x4 acc[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = zero4;
        int cj = 0, cc = 0;                                // entry / channel group of the k-step being multiplied (uniform)
        auto compute = [&](const u32x4& a) {
            const int s = __builtin_amdgcn_readlane(tp, cj);
            const u32x4* wk = Wl + ((long)(s * p.ncg + cc) * NT) * 64 + lane;
            if (++cc >= p.ncg) { cc = 0; ++cj; }
            u32x4 g4;
#pragma unroll
            for (int j = 0; j < 4; ++j) g4[j] = (unsigned)__builtin_amdgcn_ds_bpermute(perm, (int)a[j]);
            const bf16x8 g = *reinterpret_cast<const bf16x8*>(&g4);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const u32x4 wraw = wk[n * 64];
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&wraw), g, acc[n], 0, 0, 0);
            }
        };
#pragma unroll
        for (int d = 0; d < D - 1; ++d) issue(ring[d]);
        auto step = [&](auto J, int ks) {
            constexpr int j = decltype(J)::value;
            issue(ring[(j + D - 1) % D]);
            __builtin_amdgcn_sched_barrier(0);
            compute(ring[j]);
        };
        for (int ks = 0; ks < nks; ks += D)
            if (!bc_ring_steps<0, D>(ks, nks, step)) break;

        // ---- epilogue: lane holds channels c0..c0+3 (c0 = 16 n + 4 kq) of row (v, b0 + r16)
        const int b = b0 + r16;
        if (v < p.R && b < p.B) {
            char* yrow = p.y + (long)v * p.y_rb + (long)b * p.y_bb;
            const char* yp = p.yprev ? p.yprev + (long)v * p.yp_rb + (long)b * p.yp_bb : nullptr;
            const bool zero = v == p.zero_row;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int c0 = (slice * NT + n) * 16 + kq * 4;
                if (c0 >= p.Nout) continue;
                f32x4 a = acc[n];
                if (yp) {
                    const f32x4 yv = sh_from_bf16x4(*reinterpret_cast<const bf16x4*>(yp + 2 * c0));
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[j] *= sh_act_grad_from_out(yv[j], p.act);
                }
                if (zero) a = zero4;
                *reinterpret_cast<bf16x4*>(yrow + 2 * c0) = sh_to_bf16x4(a);
            }
        }
        tv = tvn; tp = tpn;
    }
}

// ------------------------------------------------------------------------------------------
// fp32 master weight -> fragment-ordered bf16 working copy (one launch for all layers of a stack).
//   transpose == 0 (forward):        W'[row = co][k = (s, ci)] = W[co][s*Cin + ci]            Cg = Cin,  Nout = Cout
//   transpose == 1 (backward-data):  W'[row = ci][k = (s, co)] = W[co][s*Cin + ci]            Cg = Cout, Nout = Cin
constexpr int WF_MAX = 24;
struct WFragArgs {
    const float* w[WF_MAX]; u32x4* out[WF_MAX];
    int S[WF_MAX], Cin[WF_MAX], Cout[WF_MAX], tr[WF_MAX], nks[WF_MAX], nt_tot[WF_MAX], block0[WF_MAX + 1];
    int nd;
};
__global__ __launch_bounds__(256) void wfrag_prep_kernel(const WFragArgs a) {
    int d = 0;
    while (d + 1 < a.nd && a.block0[d + 1] <= (int)blockIdx.x) ++d;
    const long i = (long)((int)blockIdx.x - a.block0[d]) * 256 + threadIdx.x;      // 16-byte piece index
    const long total = (long)a.nks[d] * a.nt_tot[d] * 64;
    if (i >= total) return;
    const int lane = (int)(i & 63), f = (int)(i >> 6), nt = f % a.nt_tot[d], ks = f / a.nt_tot[d];
    const int S = a.S[d], Cin = a.Cin[d], Cout = a.Cout[d], tr = a.tr[d];
    const int Cg = tr ? Cout : Cin, Nout = tr ? Cin : Cout;
    const int row = nt * 16 + (lane & 15), k0 = 32 * ks + 8 * (lane >> 4);
    const float* w = a.w[d];
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        int s, c;
        if (Cg == 3) { s = k >> 2; c = k & 3; } else { s = k / Cg; c = k - s * Cg; }
        float v = 0.f;
        if (row < Nout && s < S && c < Cg) {
            const int co = tr ? c : row, ci = tr ? row : c;
            v = w[(long)co * S * Cin + (long)s * Cin + ci];
        }
        o[j] = (__bf16)v;
    }
    a.out[d][i] = *reinterpret_cast<const u32x4*>(&o);
}

int g_num_cus = 0;
int num_cus() {
    if (!g_num_cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) g_num_cus = prop.multiProcessorCount;
        if (g_num_cus <= 0) g_num_cus = 256;
    }
    return g_num_cus;
}

template <int NT, int RT, int MODE, bool BWD, bool OUTF32>
int launch_bc(BCParams& p, hipStream_t st) {
    auto kern = conv_bf16_kernel<NT, RT, MODE, BWD, OUTF32>;
    const size_t smem = (size_t)p.nks * NT * 1024;
    static size_t attr_set = 0;                            // dynamic LDS above 64 KiB needs the attribute raised once
    if (smem > 65536 && smem > attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            sh_set_error("conv_bf16: cannot raise the dynamic LDS limit to %zu bytes", smem);
            return SH_ERR_LAUNCH;
        }
        attr_set = 160 * 1024;
    }
    p.n_vg = sh_cdiv(p.R, RT);
    const long tiles = (long)p.n_vg * sh_cdiv(p.B, MODE == BC_C64 ? 8 : 16);
    SH_REQUIRE(tiles < (1L << 30), SH_ERR_UNSUPPORTED, "conv_bf16: %ld work items", tiles);
    p.n_tiles = (int)tiles;
    // waves per workgroup and workgroups per CU from the LDS footprint of the resident weight (<= 16 waves per CU: the
    // kernel is built for <= 128 VGPRs)
    const int per_cu = smem <= 36 * 1024 ? 4 : smem <= 76 * 1024 ? 2 : 1;
    int nw = 16 / per_cu;
    // few work items (coarse levels): as many resident waves as there are items (loads in flight per CU are what these
    // launches live on), fewer waves per workgroup only when even one item per wave leaves waves idle
    while (nw > 4 && (long)num_cus() * per_cu * (nw >> 1) >= tiles * p.nsplit) nw >>= 1;
    long groups = (tiles + nw - 1) / nw;                         // workgroups (per channel slice) that still get an item
    const long cap = (long)num_cus() * per_cu / p.nsplit;
    if (groups > cap) groups = cap;
    if (groups < 8) groups = 8;
    groups = (groups + 7) / 8 * 8;                               // a multiple of 8 per slice -> every XCD label has all slices
    const int grid = (int)groups * p.nsplit;
    ShProfScope ps(st, "conv_bf16_kernel<%d, %d, %d, %s, %s>|R=%d B=%d S=%d Cg=%d N=%d grid=%dx%d", NT, RT, MODE, BWD ? "true" : "false",
                   OUTF32 ? "true" : "false", p.R, p.B, p.S, p.Cg, p.Nout, grid, nw * 64);
    SH_LAUNCH_PS(ps, kern, dim3(grid), dim3(nw * 64), smem, st, p);
    SH_CHECK_LAUNCH("conv_bf16");
    return SH_OK;
}

template <int NT>
int launch_bcr(BCParams& p, hipStream_t st) {
    auto kern = conv_bf16r_kernel<NT>;
    const size_t smem = (size_t)p.nks * NT * 1024;
    static size_t attr_set = 0;
    if (smem > 65536 && smem > attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            sh_set_error("conv_bf16r: cannot raise the dynamic LDS limit to %zu bytes", smem);
            return SH_ERR_LAUNCH;
        }
        attr_set = 160 * 1024;
    }
    p.n_vg = p.R;
    const long tiles = (long)p.R * sh_cdiv(p.B, 16);
    SH_REQUIRE(tiles < (1L << 30), SH_ERR_UNSUPPORTED, "conv_bf16r: %ld work items", tiles);
    p.n_tiles = (int)tiles;
    const int per_cu = smem <= 36 * 1024 ? 4 : smem <= 76 * 1024 ? 2 : 1;
    int nw = 16 / per_cu;
    while (nw > 4 && (long)num_cus() * per_cu * (nw >> 1) >= tiles * p.nsplit) nw >>= 1;
    long groups = (tiles + nw - 1) / nw;
    const long cap = (long)num_cus() * per_cu / p.nsplit;
    if (groups > cap) groups = cap;
    if (groups < 8) groups = 8;
    groups = (groups + 7) / 8 * 8;
    const int grid = (int)groups * p.nsplit;
    ShProfScope ps(st, "conv_bf16r_kernel<%d>|R=%d B=%d S=%d Cg=%d N=%d grid=%dx%d L=%d", NT, p.R, p.B, p.S, p.Cg, p.Nout, grid, nw * 64, p.rag_L);
    SH_LAUNCH_PS(ps, kern, dim3(grid), dim3(nw * 64), smem, st, p);
    SH_CHECK_LAUNCH("conv_bf16r");
    return SH_OK;
}

// channel tiles per workgroup of the ragged form: the whole (slice of the) weight resident, as dispatch_bc_nt
inline int bcr_nt(int nks, int nt_tot, int* nsplit) {
    int nt = nt_tot > 8 ? 8 : nt_tot, ns = nt_tot / nt;
    while (nt > 1 && (long)nks * nt > 128) { nt >>= 1; ns <<= 1; }
    *nsplit = ns;
    return nt;
}

template <int MODE, bool BWD, bool OUTF32>
int dispatch_bc_nt(BCParams& p, hipStream_t st) {
    // channel tiles per workgroup: the resident weight must fit LDS (nks x NT KiB <= 128 KiB), else split the output channels
    int nt = p.nt_tot > 8 ? 8 : p.nt_tot;
    p.nsplit = p.nt_tot / nt;
    while (nt > 1 && (long)p.nks * nt > 128) { nt >>= 1; p.nsplit <<= 1; }
    SH_REQUIRE((long)p.nks * nt <= 150, SH_ERR_UNSUPPORTED, "conv_bf16: K = %d too long for an LDS-resident weight slice", p.nks * 32);
    const long tiles16 = (long)p.R * sh_cdiv(p.B, MODE == BC_C64 ? 8 : 16);      // work items if every wave took ONE vertex
    const long fill = 2L * num_cus() * 16;                       // aim for >= 2 items per resident wave
    if constexpr (OUTF32 || MODE == BC_C3F) {
        SH_REQUIRE(nt == 1 && p.nsplit == 1, SH_ERR_UNSUPPORTED, "conv_bf16: a 3-channel fp32 side needs <= 16 channels on the other (%d)",
                   p.Nout);
        return launch_bc<1, (MODE == BC_C3F ? 2 : 4), MODE, BWD, OUTF32>(p, st);
    } else {
        constexpr int R4 = MODE == BC_C64 ? 2 : 4;              // the full-line form keeps two accumulator sets: 4 vertices per wave would not fit 128 VGPRs
        if (nt == 1) return launch_bc<1, R4, MODE, BWD, false>(p, st);
        if (nt == 2) return tiles16 / 4 >= fill ? launch_bc<2, R4, MODE, BWD, false>(p, st) : launch_bc<2, 2, MODE, BWD, false>(p, st);
        if (nt == 4) return tiles16 / 2 >= fill ? launch_bc<4, 2, MODE, BWD, false>(p, st) : launch_bc<4, 1, MODE, BWD, false>(p, st);
        if constexpr (MODE != BC_C64) {                        // (two accumulator sets of 8 x 2 tiles would not fit the register budget)
            if (tiles16 / 2 >= fill) return launch_bc<8, 2, MODE, BWD, false>(p, st);
        }
        return launch_bc<8, 1, MODE, BWD, false>(p, st);
    }
}

template <bool BWD>
int dispatch_bc(BCParams& p, int in_f32, int out_f32, hipStream_t st) {
    const ShFragGeom g = sh_frag_geom(p.S, p.Cg, p.Nout);
    p.nks = g.nks; p.nt_tot = g.nt_tot;
    SH_REQUIRE(p.S <= 64, SH_ERR_UNSUPPORTED, "conv_bf16: spiral length %d > 64", p.S);
    if (in_f32) {
        SH_REQUIRE(p.Cg == 3, SH_ERR_UNSUPPORTED, "conv_bf16: an fp32 input must have 3 channels (got %d)", p.Cg);
        SH_REQUIRE(!out_f32, SH_ERR_UNSUPPORTED, "conv_bf16: 3 -> 3 channel layers are not built in bf16");
        return dispatch_bc_nt<BC_C3F, BWD, false>(p, st);
    }
    SH_REQUIRE(p.Cg == 16 || p.Cg % 32 == 0, SH_ERR_UNSUPPORTED,
               "conv_bf16: %d gathered channels (built for 16 and multiples of 32; the fp32 path takes any)", p.Cg);
    SH_REQUIRE(((reinterpret_cast<uintptr_t>(p.x) | (uintptr_t)p.x_rb | (uintptr_t)p.x_bb) & 15) == 0, SH_ERR_INVALID_ARG,
               "conv_bf16: gathered tensor must be 16-byte aligned with 16-byte-multiple strides");
    if (out_f32) {
        SH_REQUIRE(p.Nout <= 16, SH_ERR_UNSUPPORTED, "conv_bf16: an fp32 output has <= 16 channels (got %d)", p.Nout);
        return p.Cg == 16 ? dispatch_bc_nt<BC_C16, BWD, true>(p, st) : dispatch_bc_nt<BC_C32, BWD, true>(p, st);
    }
    static const int c64_on = sh_env_int("SH_BC_C64", 1, 0, 1);      // full-line form for 64 / 128 gathered channels
    SH_REQUIRE(p.Nout % 4 == 0 && ((reinterpret_cast<uintptr_t>(p.y) | (uintptr_t)p.y_rb | (uintptr_t)p.y_bb) & 7) == 0 &&
               (!p.yprev || ((reinterpret_cast<uintptr_t>(p.yprev) | (uintptr_t)p.yp_rb | (uintptr_t)p.yp_bb) & 7) == 0) &&
               (!p.bias || (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0),
               SH_ERR_UNSUPPORTED, "conv_bf16: bf16 output needs channels %% 4 == 0 and 8-byte aligned rows");
    // ... where the layer has <= 2 channel tiles per workgroup: with 4 or 8 the doubled weight-fragment reads from LDS cost more
    // than the better loads bring (measured: 128 -> 64 and 64 -> 128 channel layers 19-26 us -> 22-27 us; 64 -> 32: 28 -> 21 us)
    int nt_wg = p.nt_tot > 8 ? 8 : p.nt_tot;
    while (nt_wg > 1 && (long)p.nks * nt_wg > 128) nt_wg >>= 1;
    static const int co_on = sh_env_int("SH_BC_CO", 2, 0, 2);          // line-wise loads (BC_C32C); 2 (default): also instead of the full-line form (21.4 -> 17.0, 20.4 -> 14.8 us)
    if (p.Cg % 64 == 0 && c64_on && nt_wg <= 2 && co_on < 2) return dispatch_bc_nt<BC_C64, BWD, false>(p, st);
    if (p.Cg % 32 == 0 && co_on) return dispatch_bc_nt<BC_C32C, BWD, false>(p, st);
    return p.Cg == 16 ? dispatch_bc_nt<BC_C16, BWD, false>(p, st) : dispatch_bc_nt<BC_C32, BWD, false>(p, st);
}

inline bool dtype_ok(int d) { return d == SH_DTYPE_F32 || d == SH_DTYPE_BF16; }
inline long esz(int d) { return d == SH_DTYPE_BF16 ? 2 : 4; }

}  // namespace

extern "C" {

size_t sh_conv_wfrag_bytes(int S, int Cg, int Nout) {
    if (S <= 0 || Cg <= 0 || Nout <= 0) return 0;
    const ShFragGeom g = sh_frag_geom(S, Cg, Nout);
    return (size_t)g.nks * g.nt_tot * 1024;
}

int sh_conv_wfrag_prep_multi(int n_layers, const float* const* weight, void* const* wfrag, const int* S, const int* Cin,
                             const int* Cout, const int* transpose, sh_stream_t stream) {
    SH_REQUIRE(n_layers > 0 && weight && wfrag && S && Cin && Cout && transpose, SH_ERR_INVALID_ARG, "sh_conv_wfrag_prep_multi: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int l0 = 0; l0 < n_layers; l0 += WF_MAX) {
        WFragArgs a{};
        a.nd = n_layers - l0 < WF_MAX ? n_layers - l0 : WF_MAX;
        int blocks = 0;
        for (int i = 0; i < a.nd; ++i) {
            const int k = l0 + i;
            SH_REQUIRE(weight[k] && wfrag[k] && S[k] > 0 && Cin[k] > 0 && Cout[k] > 0, SH_ERR_INVALID_ARG,
                       "sh_conv_wfrag_prep_multi: bad layer %d", k);
            SH_REQUIRE((reinterpret_cast<uintptr_t>(wfrag[k]) & 15) == 0, SH_ERR_INVALID_ARG, "sh_conv_wfrag_prep_multi: wfrag %d misaligned", k);
            const ShFragGeom g = transpose[k] ? sh_frag_geom(S[k], Cout[k], Cin[k]) : sh_frag_geom(S[k], Cin[k], Cout[k]);
            a.w[i] = weight[k]; a.out[i] = static_cast<u32x4*>(wfrag[k]);
            a.S[i] = S[k]; a.Cin[i] = Cin[k]; a.Cout[i] = Cout[k]; a.tr[i] = transpose[k] ? 1 : 0; a.nks[i] = g.nks; a.nt_tot[i] = g.nt_tot;
            a.block0[i] = blocks;
            blocks += (g.nks * g.nt_tot * 64 + 255) / 256;
        }
        a.block0[a.nd] = blocks;
        ShProfScope ps(st, "wfrag_prep_kernel|layers=%d", a.nd);
        SH_LAUNCH_PS(ps, wfrag_prep_kernel, dim3(blocks), dim3(256), 0, st, a);
        SH_CHECK_LAUNCH("wfrag_prep");
    }
    return SH_OK;
}

int sh_spiral_conv_fwd_bf16(const void* x, int x_dtype, int64_t x_sv, int64_t x_sb, const int32_t* table, const void* wfrag,
                            const float* bias, void* y, int y_dtype, int64_t y_sv, int64_t y_sb, int B, int R, int S, int Cin,
                            int Cout, int act, int zero_row, sh_stream_t stream) {
    SH_REQUIRE(x && table && wfrag && y, SH_ERR_INVALID_ARG, "sh_spiral_conv_fwd_bf16: null pointer");
    SH_REQUIRE(B > 0 && R > 0 && S > 0 && Cin > 0 && Cout > 0, SH_ERR_INVALID_ARG, "sh_spiral_conv_fwd_bf16: non-positive size");
    SH_REQUIRE(act >= SH_ACT_IDENTITY && act <= SH_ACT_TANH, SH_ERR_INVALID_ARG, "sh_spiral_conv_fwd_bf16: unknown activation %d", act);
    SH_REQUIRE(dtype_ok(x_dtype) && dtype_ok(y_dtype), SH_ERR_INVALID_ARG, "sh_spiral_conv_fwd_bf16: unknown dtype");
    BCParams p{};
    p.x = static_cast<const char*>(x); p.x_rb = x_sv * esz(x_dtype); p.x_bb = x_sb * esz(x_dtype);
    p.table = table; p.wfrag = static_cast<const u32x4*>(wfrag); p.bias = bias;
    p.y = static_cast<char*>(y); p.y_rb = y_sv * esz(y_dtype); p.y_bb = y_sb * esz(y_dtype);
    p.B = B; p.R = R; p.S = S; p.Cg = Cin; p.Nout = Cout; p.act = act; p.zero_row = zero_row;
    return dispatch_bc<false>(p, x_dtype == SH_DTYPE_F32, y_dtype == SH_DTYPE_F32, static_cast<hipStream_t>(stream));
}

int sh_spiral_conv_bwd_data_bf16(const void* dpre, int dp_dtype, int64_t dp_sv, int64_t dp_sb, const int32_t* table_t,
                                 const void* wfrag_t, void* dx, int dx_dtype, int64_t dx_sv, int64_t dx_sb, const void* yprev,
                                 int64_t yp_sv, int64_t yp_sb, int act_prev, int zero_row, int B, int n_in, int S, int Cin,
                                 int Cout, sh_stream_t stream) {
    SH_REQUIRE(dpre && table_t && wfrag_t && dx, SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_data_bf16: null pointer");
    SH_REQUIRE(B > 0 && n_in > 0 && S > 0 && Cin > 0 && Cout > 0, SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_data_bf16: non-positive size");
    SH_REQUIRE(act_prev >= SH_ACT_IDENTITY && act_prev <= SH_ACT_TANH, SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_data_bf16: unknown activation");
    SH_REQUIRE(dtype_ok(dp_dtype) && dtype_ok(dx_dtype), SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_data_bf16: unknown dtype");
    BCParams p{};
    p.x = static_cast<const char*>(dpre); p.x_rb = dp_sv * esz(dp_dtype); p.x_bb = dp_sb * esz(dp_dtype);
    p.table = table_t; p.wfrag = static_cast<const u32x4*>(wfrag_t); p.bias = nullptr;
    p.y = static_cast<char*>(dx); p.y_rb = dx_sv * esz(dx_dtype); p.y_bb = dx_sb * esz(dx_dtype);
    p.yprev = static_cast<const char*>(yprev); p.yp_rb = yp_sv * 2; p.yp_bb = yp_sb * 2;
    p.B = B; p.R = n_in; p.S = S; p.Cg = Cout; p.Nout = Cin; p.act = act_prev; p.zero_row = zero_row;
    return dispatch_bc<true>(p, dp_dtype == SH_DTYPE_F32, dx_dtype == SH_DTYPE_F32, static_cast<hipStream_t>(stream));
}

int sh_spiral_conv_bf16_rag_ok(int B, int S, int Cg, int Nout, int rag_L) {
    if (B <= 0 || S <= 0 || S > 64 || Cg <= 0 || Cg % 32 != 0 || Nout <= 0 || Nout % 4 != 0 || rag_L <= 0 || rag_L > 64) return 0;
    const ShFragGeom g = sh_frag_geom(S, Cg, Nout);
    if (g.nt_tot > 8 && g.nt_tot % 8 != 0) return 0;
    int ns;
    const int nt = bcr_nt(g.nks, g.nt_tot, &ns);
    return (nt == 1 || nt == 2 || nt == 4 || nt == 8) && nt * ns == g.nt_tot && (long)g.nks * nt <= 150;
}

int sh_spiral_conv_bwd_data_bf16_rag(const void* dpre, int64_t dp_sv, int64_t dp_sb, const int32_t* rag_rows, const int32_t* rag_pos,
                                     int rag_L, const void* wfrag_t, void* dx, int64_t dx_sv, int64_t dx_sb, const void* yprev,
                                     int64_t yp_sv, int64_t yp_sb, int act_prev, int zero_row, int B, int n_in, int S, int Cin, int Cout,
                                     sh_stream_t stream) {
    SH_REQUIRE(dpre && rag_rows && rag_pos && wfrag_t && dx, SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_data_bf16_rag: null pointer");
    SH_REQUIRE(B > 0 && n_in > 0 && S > 0 && Cin > 0 && Cout > 0, SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_data_bf16_rag: non-positive size");
    SH_REQUIRE(act_prev >= SH_ACT_IDENTITY && act_prev <= SH_ACT_TANH, SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_data_bf16_rag: unknown activation");
    SH_REQUIRE(sh_spiral_conv_bf16_rag_ok(B, S, Cout, Cin, rag_L), SH_ERR_UNSUPPORTED,
               "sh_spiral_conv_bwd_data_bf16_rag: B=%d S=%d gathered channels=%d output channels=%d lists of %d is not taken (gathered "
               "channels %% 32 == 0, output channels %% 4 == 0, lists of at most 64 sources)", B, S, Cout, Cin, rag_L);
    BCParams p{};
    p.x = static_cast<const char*>(dpre); p.x_rb = dp_sv * 2; p.x_bb = dp_sb * 2;
    p.wfrag = static_cast<const u32x4*>(wfrag_t);
    p.y = static_cast<char*>(dx); p.y_rb = dx_sv * 2; p.y_bb = dx_sb * 2;
    p.yprev = static_cast<const char*>(yprev); p.yp_rb = yp_sv * 2; p.yp_bb = yp_sb * 2;
    p.B = B; p.R = n_in; p.S = S; p.Cg = Cout; p.Nout = Cin; p.act = act_prev; p.zero_row = zero_row;
    p.rag_rows = rag_rows; p.rag_pos = rag_pos; p.rag_L = rag_L; p.ncg = Cout / 32;
    const ShFragGeom g = sh_frag_geom(S, Cout, Cin);
    p.nks = g.nks; p.nt_tot = g.nt_tot;
    SH_REQUIRE(((reinterpret_cast<uintptr_t>(p.x) | (uintptr_t)p.x_rb | (uintptr_t)p.x_bb) & 15) == 0, SH_ERR_INVALID_ARG,
               "sh_spiral_conv_bwd_data_bf16_rag: gathered tensor must be 16-byte aligned with 16-byte-multiple strides");
    SH_REQUIRE(((reinterpret_cast<uintptr_t>(p.y) | (uintptr_t)p.y_rb | (uintptr_t)p.y_bb) & 7) == 0 &&
               (!p.yprev || ((reinterpret_cast<uintptr_t>(p.yprev) | (uintptr_t)p.yp_rb | (uintptr_t)p.yp_bb) & 7) == 0),
               SH_ERR_UNSUPPORTED, "sh_spiral_conv_bwd_data_bf16_rag: bf16 rows must be 8-byte aligned");
    SH_REQUIRE((unsigned long)n_in * (unsigned long)(p.x_rb >> 4) < (1UL << 32), SH_ERR_UNSUPPORTED,
               "sh_spiral_conv_bwd_data_bf16_rag: gathered tensor too large for 32-bit piece offsets");
    const int nt = bcr_nt(p.nks, p.nt_tot, &p.nsplit);
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (nt) {
        case 1: return launch_bcr<1>(p, st);
        case 2: return launch_bcr<2>(p, st);
        case 4: return launch_bcr<4>(p, st);
        default: return launch_bcr<8>(p, st);
    }
}

}  // extern "C"
