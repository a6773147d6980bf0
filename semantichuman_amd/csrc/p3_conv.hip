// Spiral convolution of the fp32 path in the THREE-PLANE form (SH_MMA_PLANES3) on CDNA4 (gfx950).
//
//   forward        y[r,b,:]  = act( sum_s x[table[r,s],b,:] . W_s^T + bias )       (reference models.py:40-51)
//   backward-data  dx[u,b,:] = sum_s dpre[table_t[u,s],b,:] . W_s                   (autograd of :42,:45)
//
// Arithmetic: every fp32 operand is an EXACT sum of three bf16 numbers, v = h + m + l (8 + 8 + 8 significand bits,
// sh_split3), and a product row is the six leading (or all nine) terms of (Wh + Wm + Wl)(Xh + Xm + Xl) on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation - the arithmetic of the bf16x3 kernels of spiral_conv.hip (fp32-level
// error; tests/test_p3.py gates it against a float64 evaluation next to the exact fp32 MFMA form).  What differs is WHERE
// the split happens: there every consumer splits what it gathered (~5.5 VALU operations per gathered element, S-fold per
// activation, which is what bounds those kernels); here the PRODUCER of an activation or gradient writes the three planes
// once, beside the fp32 tensor the element-wise consumers keep using, and the consumers only load and multiply.
//
// Plane image of a tensor [rows][B][C] (B % 16 == 0; C % 32 == 0 or C == 16): FRAGMENT-MAJOR, i.e. stored in exactly the
// order the matrix instruction wants its B operand, one 1-KiB fragment per (row, 16 batch entries, 32 channels, plane):
//     C % 32 == 0:  byte ((row * nbg + bg) * C/32 + cg) * 3072 + plane * 1024 + lane * 16,   lane = kb * 16 + (b & 15)
//                   holds the 8 bf16 of channels cg * 32 + kb * 8 .. + 7 of batch entry bg * 16 + (b & 15)
//     C == 16:      byte (row * nbg + bg) * 1536 + plane * 512 + (kb2 * 16 + (b & 15)) * 16,  kb2 = channel / 8
//                   (a k-step of 32 covers two spiral positions: lanes 0-31 read one neighbour, lanes 32-63 the next)
// so a wave's gather instruction is ONE contiguous 1-KiB (or two 512-byte) access with lane l at base + 16 l - the
// vector L1 serves it as eight full lines instead of 64 separate 16-byte accesses (6.8-8.7 TB/s of gathered
// bytes in the MFMA operand layout of a row-major tensor against 16.6-21.9 TB/s line-wise) - and nothing is permuted,
// converted or staged between the load and the MFMA.
//
// Kernel structure = the bf16 path's gather stream (bf16_conv.hip): the layer's weight (three planes of fragments) sits in
// LDS for the lifetime of the persistent workgroup, a WAVE owns items of RT vertices x 16 batch entries, D k-steps of loads
// in flight, no barrier in the loop, table line of a vertex in one VGPR across the wave (v_readlane -> scalar row base).
#include "sh_bf16.h"

#include <atomic>
#include <type_traits>

namespace {

std::atomic<long> g_p3_launches{0};                // sh_p3_launch_count(): diagnostics only

struct P3Params {
    const char* xp; long x_vb, x_bgb;          // plane image of the gathered tensor: bytes per row / per 16-batch group
    const int* table;                          // [R][S]
    const u32x4* wfrag;                        // [nks][nt_tot][3][64] 16-byte fragment pieces
    const float* bias;
    float* y; long y_sv, y_sb;                 // fp32 output, element strides of (row, batch entry); may be NULL
    char* yp; long yp_vb, yp_bgb;              // plane image of the output; may be NULL
    const float* yprev; long yv_sv, yv_sb;     // fp32 output of the layer that produced x (backward epilogue)
    // ... or its plane image (same geometry as the image of this launch's output): h + m + l IS the fp32 value, and the image is
    // what the layer's plane weight gradient has just streamed - the fp32 tensor was last touched in the forward pass (round 6:
    // the two encoder backward-data launches behind a plane weight gradient ran 35 / 32 us on cold fp32 rows, 28 / 28 behind the
    // exact kernels, which had gathered those very rows)
    const char* yprev_img; long yvi_vb, yvi_bgb;
    // ragged backward-data (conv_p3r_kernel): per output row its sources as a LIST - rag_rows [R][rag_L] rows of the gathered image,
    // rag_pos [R][rag_L] the spiral position whose weight multiplies each (-1: padding behind the row's last source)
    const int* rag_rows; const int* rag_pos; int rag_L;
    // grouped lists (conv_p3g_kernel): g_rows [n_grp][g_L] rows of the gathered image, g_pos [n_grp][g_L] one position byte per member of
    // the group (0xFF: this member does not read the row; 0xFFFFFFFF behind the group's last entry), g_out [n_grp][4] the members'
    // output rows (-1: none)
    const int* g_rows; const unsigned* g_pos; const int* g_out; int g_L, n_grp;
    int B, R, S, Cg, Nout, nks, nt_tot, ncg;
    int act, zero_row;
    int n_vg, n_tiles, nsplit;
    int skip_row;                              // backward-data: the all-zero row "no source" entries of the table point at, or -1
    // backward-data: rows >= n_img of the gathered tensor (the pre-summed rows behind the real ones) have NO image - the kernel
    // reads their fp32 values (element strides xf_sv, xf_sb) and splits them itself (~6 % of the gathered tiles: cheaper than
    // having their producers - riders inside the weight-gradient launches - stream 6 more bytes per element)
    const float* xf; long xf_sv, xf_sb; int n_img;
};

// k-steps of plane loads in flight per wave (ring slots): what the 128-VGPR budget leaves next to the accumulators and one
// set of weight fragments.  These launches live on bytes in flight per CU (matrix pipe, LDS and vector L1 all sit near 30 %).
#ifndef P3_BUDGET
#define P3_BUDGET 88
#endif
#ifndef P3_DCAP
#define P3_DCAP 4
#endif
constexpr int p3_depth(int NT, int RT) {
    const int d = (P3_BUDGET - NT * RT * 4 - 12) / (RT * 12);
    return d < 2 ? 2 : d > P3_DCAP ? P3_DCAP : d;
}

// the quad (channels c0 .. c0 + 3 of row v, batch entry bs * 16 + r16) of a tensor from its plane image: (h + m) + l, exact
__device__ __forceinline__ f32x4 p3_quad_from_image(const char* img, long vb, long bgb, int v, int bs, int r16, int c0, int nch) {
    const bool o16 = nch == 16;
    const char* src = img + (long)v * vb + (long)bs * bgb +
                      (o16 ? ((c0 >> 3) * 16 + r16) * 16 : (c0 >> 5) * 3072 + (((c0 & 31) >> 3) * 16 + r16) * 16) + ((c0 >> 2) & 1) * 8;
    const int opb = o16 ? 512 : 1024;
    const u32x2 h = *reinterpret_cast<const u32x2*>(src), m = *reinterpret_cast<const u32x2*>(src + opb), l = *reinterpret_cast<const u32x2*>(src + 2 * opb);
    f32x4 y;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        y[2 * i] = (__builtin_bit_cast(float, h[i] << 16) + __builtin_bit_cast(float, m[i] << 16)) + __builtin_bit_cast(float, l[i] << 16);
        y[2 * i + 1] = (__builtin_bit_cast(float, h[i] & 0xFFFF0000u) + __builtin_bit_cast(float, m[i] & 0xFFFF0000u)) +
                       __builtin_bit_cast(float, l[i] & 0xFFFF0000u);
    }
    return y;
}

template <int J, int D, class F>
__device__ __forceinline__ bool p3_ring_steps(int ks, int nks, F&& f) {
    f(std::integral_constant<int, J>{}, ks + J);
    if constexpr (J + 1 < D) {
        if (ks + J + 1 >= nks) return false;
        return p3_ring_steps<J + 1, D>(ks, nks, f);
    } else {
        return true;
    }
}

#ifndef P3_WAVES_PER_EU
#define P3_WAVES_PER_EU 4
#endif
// Diagnostic builds only (tools/exp/r06_p3_ablate.sh; never the shipped library): what of a k-step is left in conv_p3_kernel -
// 1: no gathered loads (operands from registers), 2: the loads alone (no weight reads, no MFMAs), 3: loads + weight-fragment reads from
// LDS, no MFMAs, 4: loads + MFMAs with weight fragments from registers (no LDS reads).  Results are garbage; times tell what bounds it.
#ifndef P3_ABLATE
#define P3_ABLATE 0
#endif
template <int NT, int RT, bool C16, bool BWD, int NP, bool F32R = false>      // F32R: rows >= n_img are read as fp32 and split here
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(P3_WAVES_PER_EU, 8))) void conv_p3_kernel(const P3Params p) {
    constexpr int D = p3_depth(NT, RT);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const u32x4* Wl = reinterpret_cast<const u32x4*>(smem);       // [nks][NT][3][64]
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
    const int S = p.S, sl = lane < S ? lane : S - 1;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    const unsigned rowmul = (unsigned)(p.x_vb >> 4);
    const int skip_key = F32R ? p.skip_row : (int)((unsigned)p.skip_row * rowmul);      // what a "no source" entry looks like in tv[]
    auto load_table = [&](int t, int (&tv)[RT]) {
        const int tt = t < t_end ? t : (t_end > 0 ? t_end - 1 : 0);
        const int vg = tt % p.n_vg;
#pragma unroll
        for (int m = 0; m < RT; ++m) {
            const int v = vg * RT + m < p.R ? vg * RT + m : p.R - 1;
            // (without fp32-only rows the entry is kept PRE-MULTIPLIED by the image's row stride in 16-byte units: a gather address
            // is a shift and an add away from the v_readlane instead of behind a 64-bit multiply)
            tv[m] = F32R ? p.table[(long)v * S + sl] : (int)((unsigned)p.table[(long)v * S + sl] * rowmul);
        }
    };

    int t = t_begin + lj * nw + wave;
    int tv[RT], tvn[RT];
    load_table(t, tv);
    {
        // weight fragments (three planes) -> LDS by LDS-DMA, one 1-KiB fragment per wave instruction
        typedef __attribute__((address_space(3))) char* lptr_t;
        const unsigned wl_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lptr_t)smem);
        const int nfrag = p.nks * NT * 3;
        for (int f = __builtin_amdgcn_readfirstlane(wave); f < nfrag; f += nw) {
            const int pl = f % 3, n = (f / 3) % NT, ks = f / (3 * NT);
            const char* src = reinterpret_cast<const char*>(p.wfrag + (((long)ks * p.nt_tot + slice * NT + n) * 3 + pl) * 64 + lane);
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(__builtin_amdgcn_readfirstlane(wl_lds + (unsigned)f * 1024u)) : "memory", "m0");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    constexpr int PB = C16 ? 512 : 1024;                   // bytes between the planes of a fragment
    for (; t < t_end; t += stride) {
        load_table(t + stride, tvn);
        const int bs = t / p.n_vg, vg = t - bs * p.n_vg;
        const int v0 = vg * RT;
        const char* xl = p.xp + (long)bs * p.x_bgb + (C16 ? ((kq & 1) * 16 + r16) * 16 : lane * 16);

        int ls = 0, lc = 0;                                // running load position (uniform)
        u32x4 ring[D][RT][3];
        auto issue = [&](u32x4 (&a)[RT][3]) {
            // Three loads per (vertex, k-step) whatever the row is - an image row: its three planes; an fp32-only row (BWD, row >=
            // n_img): the lane's 8 channels as two quads (+ a repeat), split into planes when the k-step is multiplied.  Only the
            // ADDRESSES depend on the kind of row, never whether a load is issued: the waits stay counted.
            if constexpr (!C16) {
                const int s = ls < S ? ls : S - 1;
#pragma unroll
                for (int m = 0; m < RT; ++m) {
                    const int row = __builtin_amdgcn_readlane(tv[m], s);
                    const bool f32row = F32R && row >= p.n_img;                   // wave-uniform
                    const char* src = f32row ? reinterpret_cast<const char*>(p.xf + (long)row * p.xf_sv + (long)(bs * 16 + r16) * p.xf_sb + lc * 32 + kq * 8)
                                             : xl + (F32R ? (long)row * p.x_vb : (long)((unsigned long)(unsigned)row << 4)) + (long)lc * 3072;
                    const int o1 = f32row ? 16 : PB, o2 = f32row ? 0 : 2 * PB;
                    if constexpr (P3_ABLATE == 1) {
                        a[m][0] = a[m][1] = a[m][2] = (u32x4){(unsigned)row, (unsigned)lane, (unsigned)lc, 0x3f803f80u};
                    } else {
                    a[m][0] = *reinterpret_cast<const u32x4*>(src);
                    a[m][1] = *reinterpret_cast<const u32x4*>(src + o1);
                    a[m][2] = *reinterpret_cast<const u32x4*>(src + o2);
                    }
                }
                if (++lc >= p.ncg) { lc = 0; ++ls; }
            } else {
                const int s0 = ls < S ? ls : S - 1, s1 = ls + 1 < S ? ls + 1 : S - 1;
#pragma unroll
                for (int m = 0; m < RT; ++m) {
                    const int r0 = __builtin_amdgcn_readlane(tv[m], s0), r1 = __builtin_amdgcn_readlane(tv[m], s1);
                    const int row = (kq & 2) ? r1 : r0;
                    const bool f32row = F32R && row >= p.n_img;                   // per half-wave
                    const char* src = f32row ? reinterpret_cast<const char*>(p.xf + (long)row * p.xf_sv + (long)(bs * 16 + r16) * p.xf_sb + (kq & 1) * 8)
                                             : xl + (F32R ? (long)row * p.x_vb : (long)((unsigned long)(unsigned)row << 4));
                    const int o1 = f32row ? 16 : PB, o2 = f32row ? 0 : 2 * PB;
                    a[m][0] = *reinterpret_cast<const u32x4*>(src);
                    a[m][1] = *reinterpret_cast<const u32x4*>(src + o1);
                    a[m][2] = *reinterpret_cast<const u32x4*>(src + o2);
                }
                ls += 2;
            }
        };
        f32x4 acc[RT][NT];
#pragma unroll
        for (int m = 0; m < RT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = zero4;
        int cs = 0, cc = 0;                                // position / channel group of the k-step being multiplied (uniform)
        auto compute = [&](int ks, u32x4 (&a)[RT][3]) {
            // "no source" entries (down-sampling levels: half of them) gather the zero row: exact zeros through the matrix pipe.
            // The entry is wave-uniform (one vertex per 16 batch rows), so the products are skipped by a scalar branch - bitwise
            // the same result.  (The loads are not skipped: a conditional load would make every wait a full drain, and the zero
            // row is L1-resident.)
            bool live[RT];
            bool any = false;
            const int cs_was = cs;
#pragma unroll
            for (int m = 0; m < RT; ++m) {
                live[m] = true;
                if constexpr (BWD) {
                    if (p.skip_row >= 0) {
                        if constexpr (!C16) {
                            live[m] = __builtin_amdgcn_readlane(tv[m], cs) != skip_key;
                        } else {
                            const int s1 = cs + 1 < S ? cs + 1 : cs;
                            live[m] = __builtin_amdgcn_readlane(tv[m], cs) != skip_key || __builtin_amdgcn_readlane(tv[m], s1) != skip_key;
                        }
                    }
                }
                any = any || live[m];
            }
            if constexpr (BWD) {
                if constexpr (C16) { cs += 2; if (cs >= S) cs = S - 1; }
                else if (++cc >= p.ncg) { cc = 0; if (cs + 1 < S) ++cs; }
            }
            if (!any) return;
            // planes of this k-step's gathered operand: the loaded image pieces, or the split of an fp32-only row
#pragma unroll
            for (int m = 0; m < RT; ++m) {
                if constexpr (F32R) {
                    bool f32row;
                    if constexpr (!C16) {
                        f32row = __builtin_amdgcn_readlane(tv[m], cs_was) >= p.n_img;
                    } else {
                        const int s1 = cs_was + 1 < S ? cs_was + 1 : cs_was;
                        const int r0 = __builtin_amdgcn_readlane(tv[m], cs_was), r1 = __builtin_amdgcn_readlane(tv[m], s1);
                        f32row = ((kq & 2) ? r1 : r0) >= p.n_img;
                    }
                    if (__builtin_amdgcn_ballot_w64(f32row)) {                    // uniform: any lane of the wave
                        u32x4 h, mm, l;
                        sh_split3(*reinterpret_cast<const f32x4*>(&a[m][0]), *reinterpret_cast<const f32x4*>(&a[m][1]), h, mm, l);
                        if (f32row) { a[m][0] = h; a[m][1] = mm; a[m][2] = l; }
                    }
                }
            }
            if constexpr (P3_ABLATE == 2) {
#pragma unroll
                for (int m = 0; m < RT; ++m) acc[m][0][0] += __builtin_bit_cast(float, a[m][0][0] ^ a[m][1][1] ^ a[m][2][2]);
                return;
            }
            const u32x4* wk = Wl + ((long)ks * NT) * 192 + lane;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                u32x4 r0, r1, r2;
                if constexpr (P3_ABLATE == 4) {
                    r0 = (u32x4){(unsigned)lane, 0x3f803f80u, (unsigned)n, 1u}; r1 = r0; r2 = r0;
                } else {
                    r0 = wk[n * 192]; r1 = wk[n * 192 + 64]; r2 = wk[n * 192 + 128];
                }
                if constexpr (P3_ABLATE == 3) {
#pragma unroll
                    for (int m = 0; m < RT; ++m) acc[m][n][0] += __builtin_bit_cast(float, r0[0] ^ r1[1] ^ r2[2] ^ a[m][0][0] ^ a[m][1][1] ^ a[m][2][2]);
                    continue;
                }
                const bf16x8 wh = *reinterpret_cast<const bf16x8*>(&r0), wm = *reinterpret_cast<const bf16x8*>(&r1),
                             wl = *reinterpret_cast<const bf16x8*>(&r2);
#pragma unroll
                for (int m = 0; m < RT; ++m) {
                    if (RT > 1 && !live[m]) continue;
                    const bf16x8 xh = *reinterpret_cast<const bf16x8*>(&a[m][0]), xm = *reinterpret_cast<const bf16x8*>(&a[m][1]),
                                 xl2 = *reinterpret_cast<const bf16x8*>(&a[m][2]);
                    f32x4 c = acc[m][n];
                    if constexpr (NP == 9) {
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xl2, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xm, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xl2, c, 0, 0, 0);
                    }
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl2, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xm, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xh, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xm, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, c, 0, 0, 0);
                    acc[m][n] = c;
                }
            }
        };

#pragma unroll
        for (int d = 0; d < D - 1; ++d) issue(ring[d]);
        auto step = [&](auto J, int ks) {
            constexpr int j = decltype(J)::value;
            issue(ring[(j + D - 1) % D]);
            __builtin_amdgcn_sched_barrier(0);
            compute(ks, ring[j]);
        };
        for (int ks = 0; ks < p.nks; ks += D)
            if (!p3_ring_steps<0, D>(ks, p.nks, step)) break;

        // ---- epilogue: lane holds channels c0..c0+3 (c0 = 16 n + 4 kq) of row (v0 + m, bs * 16 + r16)
        const int b = bs * 16 + r16;
#pragma unroll
        for (int m = 0; m < RT; ++m) {
            const int v = v0 + m;
            if (v >= p.R || b >= p.B) continue;
            const bool zero = v == p.zero_row;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int c0 = (slice * NT + n) * 16 + kq * 4;
                if (c0 >= p.Nout) continue;
                f32x4 a = acc[m][n];
                if (!BWD) {
                    if (p.bias) a += *reinterpret_cast<const f32x4*>(p.bias + c0);
                    a = sh_act_fwd4(a, p.act);
                } else if (p.yprev_img) {
                    const f32x4 yv = p3_quad_from_image(p.yprev_img, p.yvi_vb, p.yvi_bgb, v, bs, r16, c0, p.Nout);
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[j] *= sh_act_grad_from_out(yv[j], p.act);
                } else if (p.yprev) {
                    const f32x4 yv = *reinterpret_cast<const f32x4*>(p.yprev + (long)v * p.yv_sv + (long)b * p.yv_sb + c0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[j] *= sh_act_grad_from_out(yv[j], p.act);
                }
                if (zero) a = zero4;
                if (p.y) *reinterpret_cast<f32x4*>(p.y + (long)v * p.y_sv + (long)b * p.y_sb + c0) = a;
                if (p.yp) {
                    u32x2 h, mm, l;
                    sh_split3_quad(a, h, mm, l);
                    const bool o16 = p.Nout == 16;
                    char* dst = p.yp + (long)v * p.yp_vb + (long)bs * p.yp_bgb +
                                (o16 ? ((c0 >> 3) * 16 + r16) * 16 : (c0 >> 5) * 3072 + (((c0 & 31) >> 3) * 16 + r16) * 16) + (kq & 1) * 8;
                    const int opb = o16 ? 512 : 1024;
                    *reinterpret_cast<u32x2*>(dst) = h;
                    *reinterpret_cast<u32x2*>(dst + opb) = mm;
                    *reinterpret_cast<u32x2*>(dst + 2 * opb) = l;
                }
            }
        }
#pragma unroll
        for (int m = 0; m < RT; ++m) tv[m] = tvn[m];
    }
}

// ------------------------------------------------------------------------------------------
// Backward-data over RAGGED source lists (round 6).  The dense transposed table has one slot per (input row, spiral position):
// a slot nobody reads from points at the zero row (8 233 of 37 906 slots at 3446 rows x 11 positions, MORE THAN HALF on the
// down-sampling levels - their loads are issued all the same), a slot several output rows read through points at an extra row
// that a pre-sum launch has to fill first (4 495 rows there: an 18-us launch and, for the fp32-only form of those rows, a split
// inside the conv).  The number of SOURCES of an input row, however, is on average the spiral length or half of it and never
// much more (mean 10.1, max 15 there; mean 5.1, max 11 on a down-sampling level): a list of (source row, position) pairs per
// input row - every source an IMAGE row, the sums formed by the matrix pipe by linearity, W_s (a + b) = W_s a + W_s b - is
// shorter than the dense line, has no empty slots and needs no pre-summed rows.  One vertex x 16 batch entries per wave item
// (two vertices could not share weight-fragment reads: their positions differ step by step), resident weight, !C16.
template <int NT, int NP>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(P3_WAVES_PER_EU, 8))) void conv_p3r_kernel(const P3Params p) {
    constexpr int D = p3_depth(NT, 1);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const u32x4* Wl = reinterpret_cast<const u32x4*>(smem);       // [nks][NT][3][64]
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
    const unsigned rowmul = (unsigned)(p.x_vb >> 4);
    // the list of an item's vertex: lane j holds source j (row pre-multiplied by the image's row stride in 16-byte units) and its position
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
        const int nfrag = p.nks * NT * 3;
        for (int f = __builtin_amdgcn_readfirstlane(wave); f < nfrag; f += nw) {
            const int pl = f % 3, n = (f / 3) % NT, ks = f / (3 * NT);
            const char* src = reinterpret_cast<const char*>(p.wfrag + (((long)ks * p.nt_tot + slice * NT + n) * 3 + pl) * 64 + lane);
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(__builtin_amdgcn_readfirstlane(wl_lds + (unsigned)f * 1024u)) : "memory", "m0");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    for (; t < t_end; t += stride) {
        load_list(t + stride, tvn, tpn);
        const int bs = t / p.n_vg, v = t - bs * p.n_vg;
        const char* xl = p.xp + (long)bs * p.x_bgb + lane * 16;
        const int L = __builtin_popcountll(__builtin_amdgcn_ballot_w64(tp >= 0));      // sources of this vertex (uniform)
        const int nks = L * p.ncg;
        int lj2 = 0, lc = 0;                               // running load position: list entry, channel group (uniform)
        u32x4 ring[D][3];
        auto issue = [&](u32x4 (&a)[3]) {
            const int j = lj2 < L ? lj2 : (L > 0 ? L - 1 : 0);      // past the end: the last entry again (never multiplied)
            const int row = __builtin_amdgcn_readlane(tv, j);
            const char* src = xl + (long)((unsigned long)(unsigned)row << 4) + (long)lc * 3072;
            a[0] = *reinterpret_cast<const u32x4*>(src);
            a[1] = *reinterpret_cast<const u32x4*>(src + 1024);
            a[2] = *reinterpret_cast<const u32x4*>(src + 2048);
            if (++lc >= p.ncg) { lc = 0; ++lj2; }
        };
        f32x4 acc[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = zero4;
        int cj = 0, cc = 0;                                // entry / channel group of the k-step being multiplied (uniform)
        auto compute = [&](u32x4 (&a)[3]) {
            const int s = __builtin_amdgcn_readlane(tp, cj);
            const u32x4* wk = Wl + ((long)(s * p.ncg + cc) * NT) * 192 + lane;
            if (++cc >= p.ncg) { cc = 0; ++cj; }
            const bf16x8 xh = *reinterpret_cast<const bf16x8*>(&a[0]), xm = *reinterpret_cast<const bf16x8*>(&a[1]),
                         xl2 = *reinterpret_cast<const bf16x8*>(&a[2]);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const u32x4 r0 = wk[n * 192], r1 = wk[n * 192 + 64], r2 = wk[n * 192 + 128];
                const bf16x8 wh = *reinterpret_cast<const bf16x8*>(&r0), wm = *reinterpret_cast<const bf16x8*>(&r1),
                             wl = *reinterpret_cast<const bf16x8*>(&r2);
                f32x4 c = acc[n];
                if constexpr (NP == 9) {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xl2, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xm, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xl2, c, 0, 0, 0);
                }
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl2, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xm, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xh, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xm, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, c, 0, 0, 0);
                acc[n] = c;
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
            if (!p3_ring_steps<0, D>(ks, nks, step)) break;

        // ---- epilogue: lane holds channels c0..c0+3 (c0 = 16 n + 4 kq) of row (v, bs * 16 + r16)
        const int b = bs * 16 + r16;
        if (v < p.R && b < p.B) {
            const bool zero = v == p.zero_row;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int c0 = (slice * NT + n) * 16 + kq * 4;
                if (c0 >= p.Nout) continue;
                f32x4 a = acc[n];
                if (p.yprev_img) {
                    const f32x4 yv = p3_quad_from_image(p.yprev_img, p.yvi_vb, p.yvi_bgb, v, bs, r16, c0, p.Nout);
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[j] *= sh_act_grad_from_out(yv[j], p.act);
                } else if (p.yprev) {
                    const f32x4 yv = *reinterpret_cast<const f32x4*>(p.yprev + (long)v * p.yv_sv + (long)b * p.yv_sb + c0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[j] *= sh_act_grad_from_out(yv[j], p.act);
                }
                if (zero) a = zero4;
                if (p.y) *reinterpret_cast<f32x4*>(p.y + (long)v * p.y_sv + (long)b * p.y_sb + c0) = a;
                if (p.yp) {
                    u32x2 h, mm, l;
                    sh_split3_quad(a, h, mm, l);
                    const bool o16 = p.Nout == 16;
                    char* dst = p.yp + (long)v * p.yp_vb + (long)bs * p.yp_bgb +
                                (o16 ? ((c0 >> 3) * 16 + r16) * 16 : (c0 >> 5) * 3072 + (((c0 & 31) >> 3) * 16 + r16) * 16) + (kq & 1) * 8;
                    const int opb = o16 ? 512 : 1024;
                    *reinterpret_cast<u32x2*>(dst) = h;
                    *reinterpret_cast<u32x2*>(dst + opb) = mm;
                    *reinterpret_cast<u32x2*>(dst + 2 * opb) = l;
                }
            }
        }
        tv = tvn; tp = tpn;
    }
}

// ------------------------------------------------------------------------------------------
// GROUPED lists (round 6): the plane convs with a resident weight are paced by the L2 -> CU gather (14-17 TB/s of plane bytes; section 4f
// of DESIGN.md), and most of what they gather they gather again: neighbouring vertices' spirals overlap - a row is read once per spiral
// that contains it, ~S times per launch and batch group.  Here a wave item is a GROUP of up to four output rows chosen on the host for
// overlapping lists (mesh_ops.group_lists: greedy matching on shared rows, twice) and the list of the group is the UNION of its
// members' rows: every row is loaded once (three 1-KiB plane fragments per 32 channels, as before) and multiplied once per member
// that reads it, with that member's position's weight fragments, into that member's accumulators.  Matrix work and weight-fragment
// reads are those of the one-vertex kernels; the gathered bytes fall by the overlap (measured on the 6890-vertex template: lists
// of 10-11 per row -> unions of ~21-25 per four rows).  Forward (bias, activation) and backward-data over ragged sources (activation
// derivative from the image or the fp32 tensor of the producing layer) share the kernel.
// C16: the gathered tensor has 16 channels (image fragments of 1536 bytes: two 8-channel pieces per batch entry and plane) and a k-step
// spans TWO list entries - lanes kq < 2 hold the first entry's row, lanes kq >= 2 the second's.  A member reads the two entries at two
// unrelated positions, so its weight operand is put together per lane: lane (r, kq) takes piece r + 16 (2 (pos & 1) + (kq & 1)) of
// weight fragment pos >> 1, pos = the position of its half's entry; a member that reads only one of the two entries multiplies the
// other half of the operand as zeros.
template <int NT, int G, bool BWD, int NP, bool C16 = false>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(P3_WAVES_PER_EU, 8))) void conv_p3g_kernel(const P3Params p) {
    constexpr int D = p3_depth(NT * G, 1) < 3 ? 3 : p3_depth(NT * G, 1);
    constexpr int PB = C16 ? 512 : 1024;                   // bytes between the planes of a gathered fragment
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const u32x4* Wl = reinterpret_cast<const u32x4*>(smem);       // [nks][NT][3][64]
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
    const int Lp = p.g_L, ll = lane < Lp ? lane : Lp - 1;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const unsigned rowmul = (unsigned)(p.x_vb >> 4);
    // the list of an item's group: lane j holds entry j (row pre-multiplied by the image's row stride in 16-byte units) and its positions
    auto load_list = [&](int t, int& rows, unsigned& pos) {
        const int tt = t < t_end ? t : (t_end > 0 ? t_end - 1 : 0);
        const int g = tt % p.n_vg;
        rows = (int)((unsigned)p.g_rows[(long)g * Lp + ll] * rowmul);
        pos = lane < Lp ? p.g_pos[(long)g * Lp + ll] : 0xFFFFFFFFu;
    };
    int t = t_begin + lj * nw + wave;
    int tv, tvn;
    unsigned tp, tpn;
    load_list(t, tv, tp);
    {
        typedef __attribute__((address_space(3))) char* lptr_t;
        const unsigned wl_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lptr_t)smem);
        const int nfrag = p.nks * NT * 3;
        for (int f = __builtin_amdgcn_readfirstlane(wave); f < nfrag; f += nw) {
            const int pl = f % 3, n = (f / 3) % NT, ks = f / (3 * NT);
            const char* src = reinterpret_cast<const char*>(p.wfrag + (((long)ks * p.nt_tot + slice * NT + n) * 3 + pl) * 64 + lane);
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(__builtin_amdgcn_readfirstlane(wl_lds + (unsigned)f * 1024u)) : "memory", "m0");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    for (; t < t_end; t += stride) {
        load_list(t + stride, tvn, tpn);
        const int bs = t / p.n_vg, grp = t - bs * p.n_vg;
        const char* xl = p.xp + (long)bs * p.x_bgb + (C16 ? ((kq & 1) * 16 + r16) * 16 : lane * 16);
        const int L = __builtin_popcountll(__builtin_amdgcn_ballot_w64(tp != 0xFFFFFFFFu));      // entries of this group (uniform)
        const int nks = C16 ? (L + 1) >> 1 : L * p.ncg;
        int lj2 = 0, lc = 0;                               // running load position: list entry, channel group (uniform)
        u32x4 ring[D][3];
        auto issue = [&](u32x4 (&a)[3]) {
            const int j = lj2 < L ? lj2 : (L > 0 ? L - 1 : 0);      // past the end: the last entry again (never multiplied)
            int row = __builtin_amdgcn_readlane(tv, j);
            if constexpr (C16) {
                const int j1 = lj2 + 1 < L ? lj2 + 1 : j;
                const int row1 = __builtin_amdgcn_readlane(tv, j1);
                row = (kq & 2) ? row1 : row;
            }
            const char* src = xl + (long)((unsigned long)(unsigned)row << 4) + (C16 ? 0L : (long)lc * 3072);
            a[0] = *reinterpret_cast<const u32x4*>(src);
            a[1] = *reinterpret_cast<const u32x4*>(src + PB);
            a[2] = *reinterpret_cast<const u32x4*>(src + 2 * PB);
            if constexpr (C16) lj2 += 2;
            else if (++lc >= p.ncg) { lc = 0; ++lj2; }
        };
        f32x4 acc[G][NT];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[g][n] = zero4;
        int cj = 0, cc = 0;                                // entry / channel group of the k-step being multiplied (uniform)
        auto compute = [&](u32x4 (&a)[3]) {
            const unsigned pk = (unsigned)__builtin_amdgcn_readlane((int)tp, cj);
            unsigned pk1 = 0xFFFFFFFFu;                    // C16: the positions of the k-step's second entry
            const int ccw = cc;
            if constexpr (C16) {
                if (cj + 1 < L) pk1 = (unsigned)__builtin_amdgcn_readlane((int)tp, cj + 1);
                cj += 2;
            } else if (++cc >= p.ncg) { cc = 0; ++cj; }
            bf16x8 xh = *reinterpret_cast<const bf16x8*>(&a[0]), xm = *reinterpret_cast<const bf16x8*>(&a[1]),
                   xl2 = *reinterpret_cast<const bf16x8*>(&a[2]);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int s = (int)((pk >> (8 * g)) & 0xFFu);
                const u32x4* wk;
                if constexpr (C16) {
                    const int s1 = (int)((pk1 >> (8 * g)) & 0xFFu);
                    if (s == 0xFF && s1 == 0xFF) continue;             // (uniform) this member reads neither entry
                    const int q = (kq & 2) ? (s1 == 0xFF ? 0 : s1) : (s == 0xFF ? 0 : s);      // this lane's half: its entry's position
                    wk = Wl + ((long)(q >> 1) * NT) * 192 + (r16 + 16 * (2 * (q & 1) + (kq & 1)));
                    if (s == 0xFF || s1 == 0xFF) {                     // (uniform) one entry only: the other half of the operand is zero
                        const bool dead = (kq & 2) ? s1 == 0xFF : s == 0xFF;
                        const u32x4 z = {0u, 0u, 0u, 0u};
                        const u32x4 h4 = dead ? z : a[0], m4 = dead ? z : a[1], l4 = dead ? z : a[2];
                        xh = *reinterpret_cast<const bf16x8*>(&h4); xm = *reinterpret_cast<const bf16x8*>(&m4); xl2 = *reinterpret_cast<const bf16x8*>(&l4);
                    } else {
                        xh = *reinterpret_cast<const bf16x8*>(&a[0]); xm = *reinterpret_cast<const bf16x8*>(&a[1]); xl2 = *reinterpret_cast<const bf16x8*>(&a[2]);
                    }
                } else {
                    if (s == 0xFF) continue;               // (uniform) this member does not read the row
                    wk = Wl + ((long)(s * p.ncg + ccw) * NT) * 192 + lane;
                }
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const u32x4 r0 = wk[n * 192], r1 = wk[n * 192 + 64], r2 = wk[n * 192 + 128];
                    const bf16x8 wh = *reinterpret_cast<const bf16x8*>(&r0), wm = *reinterpret_cast<const bf16x8*>(&r1),
                                 wl = *reinterpret_cast<const bf16x8*>(&r2);
                    f32x4 c = acc[g][n];
                    if constexpr (NP == 9) {
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xl2, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xm, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xl2, c, 0, 0, 0);
                    }
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl2, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xm, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xh, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xm, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, c, 0, 0, 0);
                    acc[g][n] = c;
                }
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
            if (!p3_ring_steps<0, D>(ks, nks, step)) break;

        // ---- epilogue: per member, lane holds channels c0..c0+3 (c0 = 16 n + 4 kq) of row (v, bs * 16 + r16)
        const int b = bs * 16 + r16;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int v = p.g_out[(long)grp * 4 + g];      // (uniform)
            if (v < 0 || v >= p.R || b >= p.B) continue;
            const bool zero = v == p.zero_row;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int c0 = (slice * NT + n) * 16 + kq * 4;
                if (c0 >= p.Nout) continue;
                f32x4 a = acc[g][n];
                if (!BWD) {
                    if (p.bias) a += *reinterpret_cast<const f32x4*>(p.bias + c0);
                    a = sh_act_fwd4(a, p.act);
                } else if (p.yprev_img) {
                    const f32x4 yv = p3_quad_from_image(p.yprev_img, p.yvi_vb, p.yvi_bgb, v, bs, r16, c0, p.Nout);
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[j] *= sh_act_grad_from_out(yv[j], p.act);
                } else if (p.yprev) {
                    const f32x4 yv = *reinterpret_cast<const f32x4*>(p.yprev + (long)v * p.yv_sv + (long)b * p.yv_sb + c0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[j] *= sh_act_grad_from_out(yv[j], p.act);
                }
                if (zero) a = zero4;
                if (p.y) *reinterpret_cast<f32x4*>(p.y + (long)v * p.y_sv + (long)b * p.y_sb + c0) = a;
                if (p.yp) {
                    u32x2 h, mm, l;
                    sh_split3_quad(a, h, mm, l);
                    const bool o16 = p.Nout == 16;
                    char* dst = p.yp + (long)v * p.yp_vb + (long)bs * p.yp_bgb +
                                (o16 ? ((c0 >> 3) * 16 + r16) * 16 : (c0 >> 5) * 3072 + (((c0 & 31) >> 3) * 16 + r16) * 16) + (kq & 1) * 8;
                    const int opb = o16 ? 512 : 1024;
                    *reinterpret_cast<u32x2*>(dst) = h;
                    *reinterpret_cast<u32x2*>(dst + opb) = mm;
                    *reinterpret_cast<u32x2*>(dst + 2 * opb) = l;
                }
            }
        }
        tv = tvn; tp = tpn;
    }
}

// ------------------------------------------------------------------------------------------
// Weight-STREAMING form for layers whose three-plane weight does not fit LDS (the coarsest level: 512 x 128 and 1024 x 64
// weights = 384 KiB of planes).  Same gather stream per wave (D = 4 k-steps of plane loads in flight, continuous over the whole
// K loop), but the weight passes through LDS in chunks of KC = 4 k-steps x 4 channel tiles (48 KiB, double-buffered): the 16
// waves of a workgroup each fetch 3 of a chunk's 48 fragments into registers while the previous chunk is multiplied, write them
// to the idle buffer and meet at ONE barrier per chunk.  The weight loads are ordinary (compiler-visible) loads, so every wait is
// a counted s_waitcnt for exactly those loads - the gather ring is never drained.  A wave owns ONE vertex x 16 batch entries x
// 64 output channels per round; wider layers split their channels over workgroup slices (the gather repeats per slice: these
// layers are matrix-bound).  Every wave of a workgroup runs the same number of rounds (idle waves multiply a clamped item and
// store nothing): the barriers are uniform.
constexpr int P3S_NT = 4, P3S_KC = 4;

// RT vertices per wave share every weight fragment read: with one vertex a fragment (1 KiB of LDS) feeds 6 MFMAs = 96 cycles
// of one SIMD, i.e. the four SIMDs ask for exactly the 128 bytes / clock LDS delivers - matrix pipe and LDS co-limited; with
// two, half of that.  RT = 2 runs 8 waves per workgroup (two per SIMD, up to 256 VGPRs each), RT = 1 sixteen.
template <int RT, bool BWD, int NP, int NT = P3S_NT>
__global__ __launch_bounds__(1024 / RT) void conv_p3s_kernel(const P3Params p) {
    constexpr int D = P3S_KC, WAVES = 16 / RT, WF = P3S_KC * NT * 3 / WAVES;      // WF: fragments of a chunk this wave fetches
    static_assert(WF * WAVES == P3S_KC * NT * 3, "a chunk's fragments must divide evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* Wl = reinterpret_cast<u32x4*>(smem);                   // [2][KC][NT][3][64]
    constexpr int CHUNK_PIECES = P3S_KC * NT * 192;               // 16-byte pieces per chunk buffer
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
    const int nwg_x = ((int)gridDim.x - xcd + 7) >> 3;
    const int slice = li % p.nsplit, lj = li / p.nsplit;
    const int ngrp = nwg_x / p.nsplit;
    const int q8 = p.n_tiles >> 3, r8 = p.n_tiles & 7;
    const int t_begin = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int t_end = t_begin + (xcd < r8 ? q8 + 1 : q8);
    const int stride = ngrp * WAVES;
    const int r16 = lane & 15, kq = lane >> 4;
    const int S = p.S, sl = lane < S ? lane : S - 1;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int nch = p.nks / P3S_KC;
    int wsrc[WF], wdst[WF];                                        // 16-byte piece offsets of this wave's fragments of a chunk
#pragma unroll
    for (int j = 0; j < WF; ++j) {
        const int f = WF * wave + j, pl = f % 3, n = (f / 3) % NT, kk = f / (3 * NT);
        wsrc[j] = ((kk * p.nt_tot + slice * NT + n) * 3 + pl) * 64 + lane;      // + chunk * KC * nt_tot * 192
        wdst[j] = f * 64 + lane;
    }
    const int wchunk = P3S_KC * p.nt_tot * 192;
    const unsigned rowmul = (unsigned)(p.x_vb >> 4);
    const int skip_key = (int)((unsigned)p.skip_row * rowmul);
    const int rounds = (t_end - t_begin - lj * WAVES + stride - 1) / stride;      // of wave 0 of this workgroup: the most any wave has
    int t = t_begin + lj * WAVES + wave;
    for (int rd = 0; rd < rounds; ++rd, t += stride) {
        const bool live = t < t_end;
        const int tt = live ? t : t_end - 1;
        const int bs = tt / p.n_vg, vg = tt - bs * p.n_vg;
        const int v0 = vg * RT;
        int tv[RT];
#pragma unroll
        for (int m = 0; m < RT; ++m) tv[m] = (int)((unsigned)p.table[(long)(v0 + m < p.R ? v0 + m : p.R - 1) * S + sl] * rowmul);      // pre-multiplied, as in conv_p3_kernel
        const char* xl = p.xp + (long)bs * p.x_bgb + lane * 16;
        int ls = 0, lc = 0;
        u32x4 ring[D][RT][3];
        auto issue = [&](u32x4 (&a)[RT][3]) {
            const int s = ls < S ? ls : S - 1;
#pragma unroll
            for (int m = 0; m < RT; ++m) {
                const int row = __builtin_amdgcn_readlane(tv[m], s);
                const bool f32row = false;                                        // (this form's layers image their pre-summed rows)
                const char* src = f32row ? reinterpret_cast<const char*>(p.xf + (long)row * p.xf_sv + (long)(bs * 16 + r16) * p.xf_sb + lc * 32 + kq * 8)
                                         : xl + (long)((unsigned long)(unsigned)row << 4) + (long)lc * 3072;
                const int o1 = f32row ? 16 : 1024, o2 = f32row ? 0 : 2048;
                a[m][0] = *reinterpret_cast<const u32x4*>(src);
                a[m][1] = *reinterpret_cast<const u32x4*>(src + o1);
                a[m][2] = *reinterpret_cast<const u32x4*>(src + o2);
            }
            if (++lc >= p.ncg) { lc = 0; ++ls; }
        };
        f32x4 acc[RT][NT];
#pragma unroll
        for (int m = 0; m < RT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = zero4;
        int cs = 0, cc = 0;
        auto compute = [&](const u32x4* wk, u32x4 (&a)[RT][3]) {
            bool live[RT];                                 // see conv_p3_kernel: products of "no source" entries are skipped
            bool any = false;
#pragma unroll
            for (int m = 0; m < RT; ++m) {
                const int row = BWD ? __builtin_amdgcn_readlane(tv[m], cs) : 0;
                live[m] = !BWD || p.skip_row < 0 || row != skip_key;
                any = any || live[m];
            }
            if (BWD && ++cc >= p.ncg) { cc = 0; if (cs + 1 < S) ++cs; }
            if (!any) return;
            // one scalar branch per (channel tile, row tile) instead of one around every MFMA (two branches per MFMA in the
            // instruction stream: 366 around 192 MFMAs), the products of an accumulator in ascending magnitude (weight plane,
            // x plane) as ONE chain: v_mfma_f32_16x16x32_bf16 issues every 16 cycles on one accumulator, every 22 on rotating ones
            // (tools/exp/mfma_rate.hip)
            constexpr int PW[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0}, PX[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const u32x4 r0 = wk[n * 192], r1 = wk[n * 192 + 64], r2 = wk[n * 192 + 128];
                const bf16x8 w[3] = {*reinterpret_cast<const bf16x8*>(&r0), *reinterpret_cast<const bf16x8*>(&r1),
                                     *reinterpret_cast<const bf16x8*>(&r2)};
#pragma unroll
                for (int m = 0; m < RT; ++m) {
                    if (RT > 1 && !live[m]) continue;
                    f32x4 c = acc[m][n];
#pragma unroll
                    for (int q = 9 - NP; q < 9; ++q)
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[PW[q]], *reinterpret_cast<const bf16x8*>(&a[m][PX[q]]), c, 0, 0, 0);
                    acc[m][n] = c;
                }
            }
        };
        // chunk 0 of the weight -> buffer 0 (the previous round's last reads of it ended before that round's last barrier)
        u32x4 wreg[WF];
#pragma unroll
        for (int j = 0; j < WF; ++j) wreg[j] = p.wfrag[wsrc[j]];
#pragma unroll
        for (int d = 0; d < D - 1; ++d) issue(ring[d]);
#pragma unroll
        for (int j = 0; j < WF; ++j) Wl[wdst[j]] = wreg[j];
        __syncthreads();
        for (int c = 0; c < nch; ++c) {
            const bool more = c + 1 < nch;
            if (more) {
#pragma unroll
                for (int j = 0; j < WF; ++j) wreg[j] = p.wfrag[wsrc[j] + (c + 1) * wchunk];
            }
            const u32x4* wb = Wl + (c & 1) * CHUNK_PIECES + lane;
            auto step = [&](auto J, int) {
                constexpr int j = decltype(J)::value;
                issue(ring[(j + D - 1) % D]);             // (past the end of K: a clamped, unused load)
                __builtin_amdgcn_sched_barrier(0);
                compute(wb + j * NT * 192, ring[j]);
            };
            p3_ring_steps<0, D>(0, D, step);
            if (more) {
                u32x4* wn = Wl + ((c + 1) & 1) * CHUNK_PIECES;
#pragma unroll
                for (int j = 0; j < WF; ++j) wn[wdst[j]] = wreg[j];
            }
            __syncthreads();
        }
        // ---- epilogue (as conv_p3_kernel)
        const int b = bs * 16 + r16;
#pragma unroll
        for (int m = 0; m < RT; ++m) {
            const int v = v0 + m;
            if (!live || v >= p.R || b >= p.B) continue;
            const bool zero = v == p.zero_row;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int c0 = (slice * NT + n) * 16 + kq * 4;
                if (c0 >= p.Nout) continue;
                f32x4 a = acc[m][n];
                if (!BWD) {
                    if (p.bias) a += *reinterpret_cast<const f32x4*>(p.bias + c0);
                    a = sh_act_fwd4(a, p.act);
                } else if (p.yprev_img) {
                    const f32x4 yv = p3_quad_from_image(p.yprev_img, p.yvi_vb, p.yvi_bgb, v, bs, r16, c0, p.Nout);
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[j] *= sh_act_grad_from_out(yv[j], p.act);
                } else if (p.yprev) {
                    const f32x4 yv = *reinterpret_cast<const f32x4*>(p.yprev + (long)v * p.yv_sv + (long)b * p.yv_sb + c0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[j] *= sh_act_grad_from_out(yv[j], p.act);
                }
                if (zero) a = zero4;
                if (p.y) *reinterpret_cast<f32x4*>(p.y + (long)v * p.y_sv + (long)b * p.y_sb + c0) = a;
                if (p.yp) {
                    u32x2 h, mm, l;
                    sh_split3_quad(a, h, mm, l);
                    const bool o16 = p.Nout == 16;
                    char* dst = p.yp + (long)v * p.yp_vb + (long)bs * p.yp_bgb +
                                (o16 ? ((c0 >> 3) * 16 + r16) * 16 : (c0 >> 5) * 3072 + (((c0 & 31) >> 3) * 16 + r16) * 16) + (kq & 1) * 8;
                    const int opb = o16 ? 512 : 1024;
                    *reinterpret_cast<u32x2*>(dst) = h;
                    *reinterpret_cast<u32x2*>(dst + opb) = mm;
                    *reinterpret_cast<u32x2*>(dst + 2 * opb) = l;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// fp32 tensor [rows][B][C] (element strides sv, sb; channels contiguous) -> its plane image.  One 16-byte piece of each
// plane per thread.
struct ToP3Args { const float* x; long sv, sb; char* out; int nbg, C; long total; };
__global__ __launch_bounds__(256) void to_p3_kernel(const ToP3Args a) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= a.total) return;
    int l, kb, cbase; long blk; char* d; int pb;
    if (a.C == 16) {
        l = (int)(q & 31); blk = q >> 5; kb = l >> 4; cbase = kb * 8;
        d = a.out + blk * 1536 + l * 16; pb = 512;
    } else {
        l = (int)(q & 63); const long f = q >> 6; const int ncg = a.C >> 5;
        blk = f / ncg; const int cg = (int)(f - blk * ncg); kb = l >> 4; cbase = cg * 32 + kb * 8;
        d = a.out + f * 3072 + l * 16; pb = 1024;
    }
    const long v = blk / a.nbg; const int bg = (int)(blk - v * a.nbg), r = l & 15;
    const float* s = a.x + v * a.sv + (long)(bg * 16 + r) * a.sb + cbase;
    const f32x4 lo = *reinterpret_cast<const f32x4*>(s), hi = *reinterpret_cast<const f32x4*>(s + 4);
    u32x4 h, m, lw;
    sh_split3(lo, hi, h, m, lw);
    *reinterpret_cast<u32x4*>(d) = h;
    *reinterpret_cast<u32x4*>(d + pb) = m;
    *reinterpret_cast<u32x4*>(d + 2 * pb) = lw;
}

// fp32 master weight -> three planes of fragment-ordered bf16 (one launch for all layers; geometry of sh_frag_geom):
//   frag3[ks][nt][plane][lane][j] = plane of W'[16 nt + (lane & 15)][32 ks + 8 (lane >> 4) + j]
constexpr int WF3_MAX = 24;
struct WFrag3Args {
    const float* w[WF3_MAX]; u32x4* out[WF3_MAX];
    int S[WF3_MAX], Cin[WF3_MAX], Cout[WF3_MAX], tr[WF3_MAX], nks[WF3_MAX], nt_tot[WF3_MAX], block0[WF3_MAX + 1];
    int nd;
};
__global__ __launch_bounds__(256) void wfrag3_prep_kernel(const WFrag3Args a) {
    int d = 0;
    while (d + 1 < a.nd && a.block0[d + 1] <= (int)blockIdx.x) ++d;
    const long i = (long)((int)blockIdx.x - a.block0[d]) * 256 + threadIdx.x;
    const long total = (long)a.nks[d] * a.nt_tot[d] * 64;
    if (i >= total) return;
    const int lane = (int)(i & 63), f = (int)(i >> 6), nt = f % a.nt_tot[d], ks = f / a.nt_tot[d];
    const int S = a.S[d], Cin = a.Cin[d], Cout = a.Cout[d], tr = a.tr[d];
    const int Cg = tr ? Cout : Cin, Nout = tr ? Cin : Cout;
    const int row = nt * 16 + (lane & 15), k0 = 32 * ks + 8 * (lane >> 4);
    const float* w = a.w[d];
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        const int s = k / Cg, c = k - s * Cg;
        v[j] = 0.f;
        if (row < Nout && s < S) {
            const int co = tr ? c : row, ci = tr ? row : c;
            v[j] = w[(long)co * S * Cin + (long)s * Cin + ci];
        }
    }
    u32x4 h, m, l;
    sh_split3((f32x4){v[0], v[1], v[2], v[3]}, (f32x4){v[4], v[5], v[6], v[7]}, h, m, l);
    u32x4* o = a.out[d] + (long)f * 192 + lane;
    o[0] = h; o[64] = m; o[128] = l;
}

int p3_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// channel tiles per workgroup and output-channel slices for an LDS-resident three-plane weight (<= 150 KiB)
struct P3Geom { int nks, nt_tot, nt, nsplit; };
inline P3Geom p3_geom(int S, int Cg, int Nout) {
    const ShFragGeom g = sh_frag_geom(S, Cg, Nout);
    P3Geom r{g.nks, g.nt_tot, g.nt_tot > 8 ? 8 : g.nt_tot, 1};
    r.nsplit = g.nt_tot / r.nt;
    while (r.nt > 1 && (long)g.nks * r.nt * 3 > 150) { r.nt >>= 1; r.nsplit <<= 1; }
    return r;
}
// the streaming form: gathered channels in 32s; whole slices of 64 output channels, or 32 output channels (two tiles)
inline bool p3s_shape_ok(int S, int Cg, int Nout) {
    static const int on = sh_env_int("SH_P3_STREAM", 1, 0, 1);
    static const int pad_on = sh_env_int("SH_P3S_PAD", 1, 0, 1), nt2_on = sh_env_int("SH_P3S_NT2", 1, 0, 1);
    if (!on || Cg % 32 || !(Nout % 64 == 0 || (nt2_on && Nout == 32))) return false;
    const ShFragGeom g = sh_frag_geom(S, Cg, Nout);
    return (pad_on || g.nks % P3S_KC == 0) && (Nout == 32 || g.nt_tot % P3S_NT == 0);
}
inline bool p3_resident_ok(int S, int Cg, int Nout) {
    const P3Geom g = p3_geom(S, Cg, Nout);
    static const int max_split = sh_env_int("SH_P3_MAX_SPLIT", 1, 1, 8);      // output-channel slices re-gather the input
    return (long)g.nks * g.nt * 3 <= 150 && g.nsplit <= max_split;
}
// k-steps of a layer's three-plane fragment buffer: sh_frag_geom's, rounded up to whole chunks where the weight is streamed
inline int p3_nks(int S, int Cg, int Nout) {
    const ShFragGeom g = sh_frag_geom(S, Cg, Nout);
    if (p3_resident_ok(S, Cg, Nout) || !p3s_shape_ok(S, Cg, Nout)) return g.nks;
    return (g.nks + P3S_KC - 1) / P3S_KC * P3S_KC;
}
inline bool p3_shape_ok(int B, int S, int Cg, int Nout) {
    if (B <= 0 || B % 16 || S <= 0 || S > 64 || Nout % 4) return false;
    if (!(Cg == 16 || (Cg > 0 && Cg % 32 == 0))) return false;
    return p3_resident_ok(S, Cg, Nout) || p3s_shape_ok(S, Cg, Nout);
}

template <int RT, bool BWD, int NP, int NT = P3S_NT>
int launch_p3s(P3Params& p, hipStream_t st) {
    auto kern = conv_p3s_kernel<RT, BWD, NP, NT>;
    constexpr int WAVES = 16 / RT;
    const size_t smem = (size_t)2 * P3S_KC * NT * 3072;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            sh_set_error("conv_p3s: cannot raise the dynamic LDS limit to %zu bytes", smem);
            return SH_ERR_LAUNCH;
        }
        attr_set = true;
    }
    p.nsplit = p.nt_tot / NT;
    p.n_vg = sh_cdiv(p.R, RT);
    const long tiles = (long)p.n_vg * (p.B / 16);
    SH_REQUIRE(tiles < (1L << 30), SH_ERR_UNSUPPORTED, "conv_p3s: %ld work items", tiles);
    p.n_tiles = (int)tiles;
    long groups = (tiles + WAVES - 1) / WAVES;                         // workgroups per channel slice that still get an item
    const long cap = (long)p3_num_cus() / p.nsplit;                    // one workgroup per CU (96 KiB of LDS)
    if (groups > cap) groups = cap;
    if (groups < 8) groups = 8;
    groups = (groups + 7) / 8 * 8;
    const int grid = (int)groups * p.nsplit;
    ShProfScope ps(st, "conv_p3s_kernel<%d, %s, %d, %d>|R=%d B=%d K=%d N=%d grid=%dx%d f32=%d", RT, BWD ? "true" : "false", NP, NT, p.R, p.B,
                   p.S * p.Cg, p.Nout, grid, WAVES * 64, p.y ? 1 : 0);
    SH_LAUNCH_PS(ps, kern, dim3(grid), dim3(WAVES * 64), smem, st, p);
    SH_CHECK_LAUNCH("conv_p3s");
    g_p3_launches.fetch_add(1, std::memory_order_relaxed);
    return SH_OK;
}
template <bool BWD, int NP>
int dispatch_p3s(P3Params& p, hipStream_t st) {
    static const int rt = sh_env_int("SH_P3S_RT", 2, 1, 2);
    if (p.nt_tot == 2) return launch_p3s<2, BWD, NP, 2>(p, st);          // 32 output channels: two tiles, two vertices per wave
    return rt == 2 ? launch_p3s<2, BWD, NP>(p, st) : launch_p3s<1, BWD, NP>(p, st);
}

template <int NT, int RT, bool C16, bool BWD, int NP, bool F32R = false>
int launch_p3(P3Params& p, hipStream_t st) {
    auto kern = conv_p3_kernel<NT, RT, C16, BWD, NP, F32R>;
    const size_t smem = (size_t)p.nks * NT * 3072;
    static size_t attr_set = 0;
    if (smem > 65536 && smem > attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            sh_set_error("conv_p3: cannot raise the dynamic LDS limit to %zu bytes", smem);
            return SH_ERR_LAUNCH;
        }
        attr_set = 160 * 1024;
    }
    p.n_vg = sh_cdiv(p.R, RT);
    const long tiles = (long)p.n_vg * (p.B / 16);
    SH_REQUIRE(tiles < (1L << 30), SH_ERR_UNSUPPORTED, "conv_p3: %ld work items", tiles);
    p.n_tiles = (int)tiles;
    const int per_cu = smem <= 36 * 1024 ? 4 : smem <= 76 * 1024 ? 2 : 1;
    static const int waves_cu = sh_env_int("SH_P3_WAVES_CU", 4 * P3_WAVES_PER_EU, 16, 32);      // resident waves per CU the registers allow
    int nw = waves_cu / per_cu;
    if (nw > 16) nw = 16;
    while (nw > 4 && (nw & 1) == 0 && (long)p3_num_cus() * per_cu * (nw >> 1) >= tiles * p.nsplit) nw >>= 1;
    long groups = (tiles + nw - 1) / nw;
    const long cap = (long)p3_num_cus() * per_cu / p.nsplit;
    if (groups > cap) groups = cap;
    if (groups < 8) groups = 8;
    groups = (groups + 7) / 8 * 8;
    const int grid = (int)groups * p.nsplit;
    ShProfScope ps(st, "conv_p3_kernel<%d, %d, %s, %s, %d, %s>|R=%d B=%d K=%d N=%d grid=%dx%d f32=%d", NT, RT, C16 ? "true" : "false",
                   BWD ? "true" : "false", NP, F32R ? "true" : "false", p.R, p.B, p.S * p.Cg, p.Nout, grid, nw * 64, p.y ? 1 : 0);
    SH_LAUNCH_PS(ps, kern, dim3(grid), dim3(nw * 64), smem, st, p);
    SH_CHECK_LAUNCH("conv_p3");
    g_p3_launches.fetch_add(1, std::memory_order_relaxed);
    return SH_OK;
}

template <int NT, int NP>
int launch_p3r(P3Params& p, hipStream_t st) {
    auto kern = conv_p3r_kernel<NT, NP>;
    const size_t smem = (size_t)p.nks * NT * 3072;
    static size_t attr_set = 0;
    if (smem > 65536 && smem > attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            sh_set_error("conv_p3r: cannot raise the dynamic LDS limit to %zu bytes", smem);
            return SH_ERR_LAUNCH;
        }
        attr_set = 160 * 1024;
    }
    p.n_vg = p.R;
    const long tiles = (long)p.n_vg * (p.B / 16);
    SH_REQUIRE(tiles < (1L << 30), SH_ERR_UNSUPPORTED, "conv_p3r: %ld work items", tiles);
    p.n_tiles = (int)tiles;
    const int per_cu = smem <= 36 * 1024 ? 4 : smem <= 76 * 1024 ? 2 : 1;
    static const int waves_cu = sh_env_int("SH_P3_WAVES_CU", 4 * P3_WAVES_PER_EU, 16, 32);
    int nw = waves_cu / per_cu;
    if (nw > 16) nw = 16;
    while (nw > 4 && (nw & 1) == 0 && (long)p3_num_cus() * per_cu * (nw >> 1) >= tiles * p.nsplit) nw >>= 1;
    long groups = (tiles + nw - 1) / nw;
    const long cap = (long)p3_num_cus() * per_cu / p.nsplit;
    if (groups > cap) groups = cap;
    if (groups < 8) groups = 8;
    groups = (groups + 7) / 8 * 8;
    const int grid = (int)groups * p.nsplit;
    ShProfScope ps(st, "conv_p3r_kernel<%d, %d>|R=%d B=%d K=%d N=%d grid=%dx%d L=%d f32=%d", NT, NP, p.R, p.B, p.S * p.Cg, p.Nout, grid, nw * 64, p.rag_L, p.y ? 1 : 0);
    SH_LAUNCH_PS(ps, kern, dim3(grid), dim3(nw * 64), smem, st, p);
    SH_CHECK_LAUNCH("conv_p3r");
    g_p3_launches.fetch_add(1, std::memory_order_relaxed);
    return SH_OK;
}

template <int NT, int G, bool BWD, int NP, bool C16 = false>
int launch_p3g(P3Params& p, hipStream_t st) {
    auto kern = conv_p3g_kernel<NT, G, BWD, NP, C16>;
    const size_t smem = (size_t)p.nks * NT * 3072;
    static size_t attr_set = 0;
    if (smem > 65536 && smem > attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            sh_set_error("conv_p3g: cannot raise the dynamic LDS limit to %zu bytes", smem);
            return SH_ERR_LAUNCH;
        }
        attr_set = 160 * 1024;
    }
    p.n_vg = p.n_grp;
    const long tiles = (long)p.n_grp * (p.B / 16);
    SH_REQUIRE(tiles < (1L << 30), SH_ERR_UNSUPPORTED, "conv_p3g: %ld work items", tiles);
    p.n_tiles = (int)tiles;
    const int per_cu = smem <= 36 * 1024 ? 4 : smem <= 76 * 1024 ? 2 : 1;
    static const int waves_cu = sh_env_int("SH_P3_WAVES_CU", 4 * P3_WAVES_PER_EU, 16, 32);
    int nw = waves_cu / per_cu;
    if (nw > 16) nw = 16;
    while (nw > 4 && (nw & 1) == 0 && (long)p3_num_cus() * per_cu * (nw >> 1) >= tiles * p.nsplit) nw >>= 1;
    long groups = (tiles + nw - 1) / nw;
    const long cap = (long)p3_num_cus() * per_cu / p.nsplit;
    if (groups > cap) groups = cap;
    if (groups < 8) groups = 8;
    groups = (groups + 7) / 8 * 8;
    const int grid = (int)groups * p.nsplit;
    ShProfScope ps(st, "conv_p3g_kernel<%d, %d, %s, %d, %s>|R=%d B=%d K=%d N=%d grid=%dx%d groups=%d L=%d f32=%d", NT, G, BWD ? "true" : "false", NP,
                   C16 ? "true" : "false", p.R, p.B, p.S * p.Cg, p.Nout, grid, nw * 64, p.n_grp, p.g_L, p.y ? 1 : 0);
    SH_LAUNCH_PS(ps, kern, dim3(grid), dim3(nw * 64), smem, st, p);
    SH_CHECK_LAUNCH("conv_p3g");
    g_p3_launches.fetch_add(1, std::memory_order_relaxed);
    return SH_OK;
}

template <bool C16, bool BWD, int NP, bool F32R = false>
int dispatch_p3_nt(P3Params& p, int nt, hipStream_t st) {
    const long tiles16 = (long)p.R * (p.B / 16);
    const long fill = 2L * p3_num_cus() * 16;
    static const int rt_force = sh_env_int("SH_P3_RT", 0, 0, 2);
    const bool two = rt_force ? rt_force == 2 : tiles16 / 2 >= fill;
    if (nt == 1) return two ? launch_p3<1, 2, C16, BWD, NP, F32R>(p, st) : launch_p3<1, 1, C16, BWD, NP, F32R>(p, st);
    if (nt == 2) return two ? launch_p3<2, 2, C16, BWD, NP, F32R>(p, st) : launch_p3<2, 1, C16, BWD, NP, F32R>(p, st);
    if constexpr (!C16) {
        if (nt == 4) return two ? launch_p3<4, 2, C16, BWD, NP, F32R>(p, st) : launch_p3<4, 1, C16, BWD, NP, F32R>(p, st);
        if (nt == 8) return launch_p3<8, 1, C16, BWD, NP, F32R>(p, st);
    }
    sh_set_error("conv_p3: %d channel tiles per workgroup with %d gathered channels is not built", nt, p.Cg);
    return SH_ERR_UNSUPPORTED;
}

template <bool BWD>
int dispatch_p3(P3Params& p, hipStream_t st) {
    SH_REQUIRE(p3_shape_ok(p.B, p.S, p.Cg, p.Nout), SH_ERR_UNSUPPORTED,
               "conv_p3: B=%d S=%d gathered channels=%d output channels=%d is outside the three-plane kernels (sh_spiral_conv_p3_ok)",
               p.B, p.S, p.Cg, p.Nout);
    const P3Geom g = p3_geom(p.S, p.Cg, p.Nout);
    p.nks = g.nks; p.nt_tot = g.nt_tot; p.nsplit = g.nsplit; p.ncg = p.Cg / 32;
    const int nbg = p.B / 16;
    p.x_bgb = p.Cg == 16 ? 1536 : (long)p.ncg * 3072;
    p.x_vb = p.x_bgb * nbg;
    if (p.yp) {
        SH_REQUIRE(p.Nout == 16 || p.Nout % 32 == 0, SH_ERR_UNSUPPORTED, "conv_p3: a plane image has 16 or a multiple of 32 channels (%d)", p.Nout);
        p.yp_bgb = p.Nout == 16 ? 1536 : (long)(p.Nout / 32) * 3072;
        p.yp_vb = p.yp_bgb * nbg;
    }
    SH_REQUIRE(p.y || p.yp, SH_ERR_INVALID_ARG, "conv_p3: no output");
    SH_REQUIRE(((reinterpret_cast<uintptr_t>(p.xp) | reinterpret_cast<uintptr_t>(p.yp) | reinterpret_cast<uintptr_t>(p.y) |
                 reinterpret_cast<uintptr_t>(p.yprev) | reinterpret_cast<uintptr_t>(p.bias)) & 15) == 0 &&
               ((p.y_sv | p.y_sb | p.yv_sv | p.yv_sb) & 3) == 0, SH_ERR_INVALID_ARG, "conv_p3: tensors must be 16-byte aligned with strides %% 4 == 0");
    static const int np = sh_env_int("SH_P3_NP", 6, 6, 9);
    if (!p3_resident_ok(p.S, p.Cg, p.Nout)) {
        p.nks = p3_nks(p.S, p.Cg, p.Nout);                     // whole chunks (the fragments past K are zeros)
        SH_REQUIRE(!p.xf, SH_ERR_UNSUPPORTED, "conv_p3: the weight-streaming form takes imaged rows only (sh_spiral_conv_p3_kind() == 2)");
        return np == 9 ? dispatch_p3s<BWD, 9>(p, st) : dispatch_p3s<BWD, 6>(p, st);
    }
    if constexpr (BWD) {
        if (p.xf) {                                            // rows without an image: the six-product kernels that split them
            SH_REQUIRE(np == 6, SH_ERR_UNSUPPORTED, "conv_p3: fp32 rows are built for the six-product form (SH_P3_NP=6)");
            return p.Cg == 16 ? dispatch_p3_nt<true, true, 6, true>(p, g.nt, st) : dispatch_p3_nt<false, true, 6, true>(p, g.nt, st);
        }
    }
    if (p.Cg == 16) return np == 9 ? dispatch_p3_nt<true, BWD, 9>(p, g.nt, st) : dispatch_p3_nt<true, BWD, 6>(p, g.nt, st);
    return np == 9 ? dispatch_p3_nt<false, BWD, 9>(p, g.nt, st) : dispatch_p3_nt<false, BWD, 6>(p, g.nt, st);
}

}  // namespace

extern "C" {

size_t sh_p3_bytes(int rows, int B, int C) {
    if (rows <= 0 || B <= 0 || B % 16 || !(C == 16 || (C > 0 && C % 32 == 0))) return 0;
    return (size_t)rows * (B / 16) * (C == 16 ? 1536 : (size_t)(C / 32) * 3072);
}

int sh_to_p3(const float* x, int64_t x_sv, int64_t x_sb, void* planes, int B, int rows, int C, sh_stream_t stream) {
    SH_REQUIRE(x && planes && rows > 0, SH_ERR_INVALID_ARG, "sh_to_p3: null pointer or no rows");
    SH_REQUIRE(sh_p3_bytes(rows, B, C) > 0, SH_ERR_UNSUPPORTED, "sh_to_p3: B=%d C=%d has no plane image (B %% 16 == 0; C == 16 or C %% 32 == 0)", B, C);
    SH_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(planes)) & 15) == 0 && ((x_sv | x_sb) & 3) == 0,
               SH_ERR_INVALID_ARG, "sh_to_p3: tensors must be 16-byte aligned with strides %% 4 == 0");
    ToP3Args a{x, (long)x_sv, (long)x_sb, static_cast<char*>(planes), B / 16, C, (long)rows * (B / 16) * (C == 16 ? 32 : (C / 32) * 64)};
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long blocks = (a.total + 255) / 256;
    SH_REQUIRE(blocks < (1L << 31), SH_ERR_UNSUPPORTED, "sh_to_p3: tensor too large");
    ShProfScope ps(st, "to_p3_kernel|rows=%d B=%d C=%d", rows, B, C);
    SH_LAUNCH_PS(ps, to_p3_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
    SH_CHECK_LAUNCH("to_p3");
    return SH_OK;
}

size_t sh_conv_wfrag3_bytes(int S, int Cg, int Nout) {
    if (S <= 0 || Cg <= 0 || Nout <= 0) return 0;
    const ShFragGeom g = sh_frag_geom(S, Cg, Nout);
    return (size_t)p3_nks(S, Cg, Nout) * g.nt_tot * 3072;
}

int sh_conv_wfrag3_prep_multi(int n_layers, const float* const* weight, void* const* wfrag3, const int* S, const int* Cin,
                              const int* Cout, const int* transpose, sh_stream_t stream) {
    SH_REQUIRE(n_layers > 0 && weight && wfrag3 && S && Cin && Cout && transpose, SH_ERR_INVALID_ARG, "sh_conv_wfrag3_prep_multi: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int l0 = 0; l0 < n_layers; l0 += WF3_MAX) {
        WFrag3Args a{};
        a.nd = n_layers - l0 < WF3_MAX ? n_layers - l0 : WF3_MAX;
        int blocks = 0;
        for (int i = 0; i < a.nd; ++i) {
            const int k = l0 + i;
            SH_REQUIRE(weight[k] && wfrag3[k] && S[k] > 0 && Cin[k] > 0 && Cout[k] > 0, SH_ERR_INVALID_ARG,
                       "sh_conv_wfrag3_prep_multi: bad layer %d", k);
            SH_REQUIRE((reinterpret_cast<uintptr_t>(wfrag3[k]) & 15) == 0, SH_ERR_INVALID_ARG, "sh_conv_wfrag3_prep_multi: wfrag3 %d misaligned", k);
            const int Cg = transpose[k] ? Cout[k] : Cin[k];
            SH_REQUIRE(Cg % 8 == 0, SH_ERR_UNSUPPORTED, "sh_conv_wfrag3_prep_multi: %d gathered channels", Cg);
            const ShFragGeom g = transpose[k] ? sh_frag_geom(S[k], Cout[k], Cin[k]) : sh_frag_geom(S[k], Cin[k], Cout[k]);
            a.w[i] = weight[k]; a.out[i] = static_cast<u32x4*>(wfrag3[k]);
            const int nks = transpose[k] ? p3_nks(S[k], Cout[k], Cin[k]) : p3_nks(S[k], Cin[k], Cout[k]);
            a.S[i] = S[k]; a.Cin[i] = Cin[k]; a.Cout[i] = Cout[k]; a.tr[i] = transpose[k] ? 1 : 0; a.nks[i] = nks; a.nt_tot[i] = g.nt_tot;
            a.block0[i] = blocks;
            blocks += (nks * g.nt_tot * 64 + 255) / 256;
        }
        a.block0[a.nd] = blocks;
        ShProfScope ps(st, "wfrag3_prep_kernel|layers=%d", a.nd);
        SH_LAUNCH_PS(ps, wfrag3_prep_kernel, dim3(blocks), dim3(256), 0, st, a);
        SH_CHECK_LAUNCH("wfrag3_prep");
    }
    return SH_OK;
}

int64_t sh_p3_launch_count(void) { return (int64_t)g_p3_launches.load(std::memory_order_relaxed); }

int sh_spiral_conv_p3_ok(int B, int S, int Cg, int Nout) { return p3_shape_ok(B, S, Cg, Nout) ? 1 : 0; }
int sh_spiral_conv_p3_kind(int B, int S, int Cg, int Nout) {
    if (!p3_shape_ok(B, S, Cg, Nout)) return 0;
    return p3_resident_ok(S, Cg, Nout) ? 1 : 2;
}

int sh_spiral_conv_fwd_p3(const void* xp, const int32_t* table, const void* wfrag3, const float* bias, float* y, int64_t y_sv,
                          int64_t y_sb, void* yp, int B, int R, int S, int Cin, int Cout, int act, int zero_row, sh_stream_t stream) {
    SH_REQUIRE(xp && table && wfrag3, SH_ERR_INVALID_ARG, "sh_spiral_conv_fwd_p3: null pointer");
    SH_REQUIRE(B > 0 && R > 0 && S > 0 && Cin > 0 && Cout > 0, SH_ERR_INVALID_ARG, "sh_spiral_conv_fwd_p3: non-positive size");
    SH_REQUIRE(act >= SH_ACT_IDENTITY && act <= SH_ACT_TANH, SH_ERR_INVALID_ARG, "sh_spiral_conv_fwd_p3: unknown activation %d", act);
    P3Params p{};
    p.xp = static_cast<const char*>(xp); p.table = table; p.wfrag = static_cast<const u32x4*>(wfrag3); p.bias = bias;
    p.y = y; p.y_sv = y_sv; p.y_sb = y_sb; p.yp = static_cast<char*>(yp);
    p.B = B; p.R = R; p.S = S; p.Cg = Cin; p.Nout = Cout; p.act = act; p.zero_row = zero_row; p.skip_row = -1; p.n_img = 0x7fffffff;
    return dispatch_p3<false>(p, static_cast<hipStream_t>(stream));
}

int sh_spiral_conv_bwd_data_p3(const void* dprep, int dpre_zero_row, const float* dpre_f32, int64_t dp_sv, int64_t dp_sb, int n_image_rows,
                               const int32_t* table_t, const void* wfrag3_t, float* dx, int64_t dx_sv, int64_t dx_sb, void* dxp,
                               const float* yprev, int64_t yp_sv, int64_t yp_sb, const void* yprev_planes, int act_prev, int zero_row, int B,
                               int n_in, int S, int Cin, int Cout, sh_stream_t stream) {
    SH_REQUIRE(dprep && table_t && wfrag3_t, SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_data_p3: null pointer");
    SH_REQUIRE(B > 0 && n_in > 0 && S > 0 && Cin > 0 && Cout > 0, SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_data_p3: non-positive size");
    SH_REQUIRE(act_prev >= SH_ACT_IDENTITY && act_prev <= SH_ACT_TANH, SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_data_p3: unknown activation");
    P3Params p{};
    p.xp = static_cast<const char*>(dprep); p.table = table_t; p.wfrag = static_cast<const u32x4*>(wfrag3_t); p.bias = nullptr;
    p.y = dx; p.y_sv = dx_sv; p.y_sb = dx_sb; p.yp = static_cast<char*>(dxp);
    p.yprev = yprev; p.yv_sv = yp_sv; p.yv_sb = yp_sb;
    if (yprev_planes) {
        SH_REQUIRE(sh_p3_bytes(1, B, Cin) && (reinterpret_cast<uintptr_t>(yprev_planes) & 15) == 0, SH_ERR_UNSUPPORTED,
                   "sh_spiral_conv_bwd_data_p3: B=%d Cin=%d has no plane image (yprev_planes)", B, Cin);
        p.yprev_img = static_cast<const char*>(yprev_planes);
        p.yvi_bgb = Cin == 16 ? 1536 : (long)(Cin / 32) * 3072; p.yvi_vb = p.yvi_bgb * (B / 16);
    }
    p.B = B; p.R = n_in; p.S = S; p.Cg = Cout; p.Nout = Cin; p.act = act_prev; p.zero_row = zero_row;
    static const int skip_on = sh_env_int("SH_P3_SKIP", 1, 0, 1);
    p.skip_row = skip_on ? dpre_zero_row : -1;
    p.n_img = 0x7fffffff;
    if (dpre_f32) {
        SH_REQUIRE(n_image_rows >= 0 && (reinterpret_cast<uintptr_t>(dpre_f32) & 15) == 0 && ((dp_sv | dp_sb) & 3) == 0, SH_ERR_INVALID_ARG,
                   "sh_spiral_conv_bwd_data_p3: the fp32 gradient must be 16-byte aligned with strides %% 4 == 0");
        p.xf = dpre_f32; p.xf_sv = dp_sv; p.xf_sb = dp_sb; p.n_img = n_image_rows;
    }
    return dispatch_p3<true>(p, static_cast<hipStream_t>(stream));
}

int sh_spiral_conv_p3_rag_ok(int B, int S, int Cg, int Nout, int rag_L) {
    return p3_shape_ok(B, S, Cg, Nout) && p3_resident_ok(S, Cg, Nout) && p3_geom(S, Cg, Nout).nt <= 4 && Cg % 32 == 0 && rag_L > 0 && rag_L <= 64;
}

int sh_spiral_conv_bwd_data_p3_rag(const void* dprep, const int32_t* rag_rows, const int32_t* rag_pos, int rag_L, const void* wfrag3_t, float* dx,
                                   int64_t dx_sv, int64_t dx_sb, void* dxp, const float* yprev, int64_t yp_sv, int64_t yp_sb,
                                   const void* yprev_planes, int act_prev, int zero_row, int B, int n_in, int S, int Cin, int Cout,
                                   sh_stream_t stream) {
    SH_REQUIRE(dprep && rag_rows && rag_pos && wfrag3_t, SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_data_p3_rag: null pointer");
    SH_REQUIRE(B > 0 && n_in > 0 && S > 0 && Cin > 0 && Cout > 0, SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_data_p3_rag: non-positive size");
    SH_REQUIRE(act_prev >= SH_ACT_IDENTITY && act_prev <= SH_ACT_TANH, SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_data_p3_rag: unknown activation");
    SH_REQUIRE(sh_spiral_conv_p3_rag_ok(B, S, Cout, Cin, rag_L), SH_ERR_UNSUPPORTED,
               "sh_spiral_conv_bwd_data_p3_rag: B=%d S=%d gathered channels=%d output channels=%d lists of %d is not taken (resident three-plane "
               "weight, gathered channels %% 32 == 0, lists of at most 64 sources)", B, S, Cout, Cin, rag_L);
    P3Params p{};
    p.xp = static_cast<const char*>(dprep); p.wfrag = static_cast<const u32x4*>(wfrag3_t);
    p.y = dx; p.y_sv = dx_sv; p.y_sb = dx_sb; p.yp = static_cast<char*>(dxp);
    p.yprev = yprev; p.yv_sv = yp_sv; p.yv_sb = yp_sb;
    p.B = B; p.R = n_in; p.S = S; p.Cg = Cout; p.Nout = Cin; p.act = act_prev; p.zero_row = zero_row;
    p.rag_rows = rag_rows; p.rag_pos = rag_pos; p.rag_L = rag_L;
    if (yprev_planes) {
        SH_REQUIRE(sh_p3_bytes(1, B, Cin) && (reinterpret_cast<uintptr_t>(yprev_planes) & 15) == 0, SH_ERR_UNSUPPORTED,
                   "sh_spiral_conv_bwd_data_p3_rag: B=%d Cin=%d has no plane image (yprev_planes)", B, Cin);
        p.yprev_img = static_cast<const char*>(yprev_planes);
        p.yvi_bgb = Cin == 16 ? 1536 : (long)(Cin / 32) * 3072; p.yvi_vb = p.yvi_bgb * (B / 16);
    }
    const P3Geom g = p3_geom(p.S, p.Cg, p.Nout);
    p.nks = g.nks; p.nt_tot = g.nt_tot; p.nsplit = g.nsplit; p.ncg = p.Cg / 32;
    const int nbg = B / 16;
    p.x_bgb = (long)p.ncg * 3072; p.x_vb = p.x_bgb * nbg;
    if (p.yp) {
        SH_REQUIRE(p.Nout == 16 || p.Nout % 32 == 0, SH_ERR_UNSUPPORTED, "conv_p3r: a plane image has 16 or a multiple of 32 channels (%d)", p.Nout);
        p.yp_bgb = p.Nout == 16 ? 1536 : (long)(p.Nout / 32) * 3072; p.yp_vb = p.yp_bgb * nbg;
    }
    SH_REQUIRE(p.y || p.yp, SH_ERR_INVALID_ARG, "conv_p3r: no output");
    SH_REQUIRE(((reinterpret_cast<uintptr_t>(p.xp) | reinterpret_cast<uintptr_t>(p.yp) | reinterpret_cast<uintptr_t>(p.y) | reinterpret_cast<uintptr_t>(p.yprev)) & 15) == 0 &&
               ((p.y_sv | p.y_sb | p.yv_sv | p.yv_sb) & 3) == 0, SH_ERR_INVALID_ARG, "conv_p3r: tensors must be 16-byte aligned with strides %% 4 == 0");
    static const int np = sh_env_int("SH_P3_NP", 6, 6, 9);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nt = g.nt;
#define SH_P3R_CASE(N) (np == 9 ? launch_p3r<N, 9>(p, st) : launch_p3r<N, 6>(p, st))
    if (nt == 1) return SH_P3R_CASE(1);
    if (nt == 2) return SH_P3R_CASE(2);
    if (nt == 4) return SH_P3R_CASE(4);
#undef SH_P3R_CASE
    sh_set_error("conv_p3r: %d channel tiles per workgroup is not built", nt);
    return SH_ERR_UNSUPPORTED;
}

int sh_spiral_conv_p3_grp_ok(int B, int S, int Cg, int Nout, int g_L) {
    static const int c16_on = sh_env_int("SH_P3_GRP_C16", 1, 0, 1);
    if (!(Cg % 32 == 0 || (Cg == 16 && c16_on && p3_geom(S, Cg, Nout).nt <= 2))) return 0;
    return p3_shape_ok(B, S, Cg, Nout) && p3_resident_ok(S, Cg, Nout) && p3_geom(S, Cg, Nout).nt <= 4 && S < 255 && g_L > 0 && g_L <= 64;
}

// members per group the kernel of this shape is built for (the host groups accordingly): four while their accumulators leave room
// for the load ring (<= 2 channel tiles per workgroup), two with four tiles; 0: the shape is not taken
int sh_spiral_conv_p3_grp_members(int B, int S, int Cg, int Nout) {
    if (!sh_spiral_conv_p3_grp_ok(B, S, Cg, Nout, 1)) return 0;
    return p3_geom(S, Cg, Nout).nt <= 2 ? 4 : 2;
}

// does grouping pay for this launch?  Measured (profiles/r06_kernel_experiments.txt item 12, 6890 vertices x 64): a group is one wave
// item of up to four rows - with fewer than ~one item per resident wave the launch loses more to its coarser, uneven items than the
// halved gather brings (449-493 groups x 4 batch groups: +2.3 ... +4.0 us; 886-1760 x 4: -0.8 ... -8.2 us).
int sh_spiral_conv_p3_grp_pays(int B, int n_groups) {
    static const int min_per_cu = sh_env_int("SH_P3_GRP_MIN_ITEMS_PER_CU", 12, 0, 1 << 20);
    return B >= 16 && (long)n_groups * (B / 16) >= (long)min_per_cu * p3_num_cus();
}

int sh_spiral_conv_p3_grp(const void* xp, const int32_t* g_rows, const uint32_t* g_pos, const int32_t* g_out, int n_groups, int g_L,
                          const void* wfrag3, const float* bias, float* y, int64_t y_sv, int64_t y_sb, void* yp, const float* yprev,
                          int64_t yp_sv, int64_t yp_sb, const void* yprev_planes, int act, int zero_row, int backward, int B, int R, int S,
                          int Cg, int Nout, sh_stream_t stream) {
    SH_REQUIRE(xp && g_rows && g_pos && g_out && wfrag3, SH_ERR_INVALID_ARG, "sh_spiral_conv_p3_grp: null pointer");
    SH_REQUIRE(B > 0 && R > 0 && S > 0 && Cg > 0 && Nout > 0 && n_groups > 0, SH_ERR_INVALID_ARG, "sh_spiral_conv_p3_grp: non-positive size");
    SH_REQUIRE(act >= SH_ACT_IDENTITY && act <= SH_ACT_TANH, SH_ERR_INVALID_ARG, "sh_spiral_conv_p3_grp: unknown activation");
    SH_REQUIRE(sh_spiral_conv_p3_grp_ok(B, S, Cg, Nout, g_L), SH_ERR_UNSUPPORTED,
               "sh_spiral_conv_p3_grp: B=%d S=%d gathered channels=%d output channels=%d lists of %d is not taken (resident three-plane weight, "
               "gathered channels 16 or %% 32 == 0, lists of at most 64 rows)", B, S, Cg, Nout, g_L);
    SH_REQUIRE(backward || !(yprev || yprev_planes), SH_ERR_INVALID_ARG, "sh_spiral_conv_p3_grp: yprev belongs to the backward form");
    P3Params p{};
    p.xp = static_cast<const char*>(xp); p.wfrag = static_cast<const u32x4*>(wfrag3); p.bias = backward ? nullptr : bias;
    p.y = y; p.y_sv = y_sv; p.y_sb = y_sb; p.yp = static_cast<char*>(yp);
    p.yprev = yprev; p.yv_sv = yp_sv; p.yv_sb = yp_sb;
    p.B = B; p.R = R; p.S = S; p.Cg = Cg; p.Nout = Nout; p.act = act; p.zero_row = zero_row;
    p.g_rows = g_rows; p.g_pos = g_pos; p.g_out = g_out; p.g_L = g_L; p.n_grp = n_groups;
    if (yprev_planes) {
        SH_REQUIRE(sh_p3_bytes(1, B, Nout) && (reinterpret_cast<uintptr_t>(yprev_planes) & 15) == 0, SH_ERR_UNSUPPORTED,
                   "sh_spiral_conv_p3_grp: B=%d channels=%d has no plane image (yprev_planes)", B, Nout);
        p.yprev_img = static_cast<const char*>(yprev_planes);
        p.yvi_bgb = Nout == 16 ? 1536 : (long)(Nout / 32) * 3072; p.yvi_vb = p.yvi_bgb * (B / 16);
    }
    const P3Geom g = p3_geom(p.S, p.Cg, p.Nout);
    p.nks = g.nks; p.nt_tot = g.nt_tot; p.nsplit = g.nsplit; p.ncg = p.Cg / 32;
    const int nbg = B / 16;
    const bool c16 = Cg == 16;
    p.x_bgb = c16 ? 1536 : (long)p.ncg * 3072; p.x_vb = p.x_bgb * nbg;
    if (p.yp) {
        SH_REQUIRE(p.Nout == 16 || p.Nout % 32 == 0, SH_ERR_UNSUPPORTED, "conv_p3g: a plane image has 16 or a multiple of 32 channels (%d)", p.Nout);
        p.yp_bgb = p.Nout == 16 ? 1536 : (long)(p.Nout / 32) * 3072; p.yp_vb = p.yp_bgb * nbg;
    }
    SH_REQUIRE(p.y || p.yp, SH_ERR_INVALID_ARG, "conv_p3g: no output");
    SH_REQUIRE(((reinterpret_cast<uintptr_t>(p.xp) | reinterpret_cast<uintptr_t>(p.yp) | reinterpret_cast<uintptr_t>(p.y) | reinterpret_cast<uintptr_t>(p.yprev) |
                 reinterpret_cast<uintptr_t>(p.bias)) & 15) == 0 && ((p.y_sv | p.y_sb | p.yv_sv | p.yv_sb) & 3) == 0, SH_ERR_INVALID_ARG,
               "conv_p3g: tensors must be 16-byte aligned with strides %% 4 == 0");
    static const int np = sh_env_int("SH_P3_NP", 6, 6, 9);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nt = g.nt;
    // four members per group while their accumulators leave room for the load ring (<= 2 channel tiles), two beyond
#define SH_P3G_CASE(N, GG) (backward ? (np == 9 ? launch_p3g<N, GG, true, 9>(p, st) : launch_p3g<N, GG, true, 6>(p, st)) \
                                     : (np == 9 ? launch_p3g<N, GG, false, 9>(p, st) : launch_p3g<N, GG, false, 6>(p, st)))
#define SH_P3G_CASE16(N) (backward ? (np == 9 ? launch_p3g<N, 4, true, 9, true>(p, st) : launch_p3g<N, 4, true, 6, true>(p, st)) \
                                   : (np == 9 ? launch_p3g<N, 4, false, 9, true>(p, st) : launch_p3g<N, 4, false, 6, true>(p, st)))
    if (c16 && nt == 1) return SH_P3G_CASE16(1);
    if (c16 && nt == 2) return SH_P3G_CASE16(2);
#undef SH_P3G_CASE16
    if (nt == 1) return SH_P3G_CASE(1, 4);
    if (nt == 2) return SH_P3G_CASE(2, 4);
    if (nt == 4) return SH_P3G_CASE(4, 2);
#undef SH_P3G_CASE
    sh_set_error("conv_p3g: %d channel tiles per workgroup is not built", nt);
    return SH_ERR_UNSUPPORTED;
}

}  // extern "C"
