// Spiral-convolution weight gradient of the fp32 path in the THREE-PLANE form (SH_MMA_PLANES3; round 6):
//
//   dW[co][s*Cin + ci] = sum_{v,b} dpre[v,b,co] * x[table[v,s], b, ci]        dbias[co] = sum_{v,b} dpre[v,b,co]
//                                                                            (autograd of reference models.py:45)
// Both operands are read through the plane images that already exist when the backward pass gets here - the image of the
// layer's input that its forward plane conv gathered, and the image of the pre-activation gradient that its backward-data
// plane conv gathers (csrc/p3_conv.hip: every fp32 value an EXACT sum of three bf16 numbers h + m + l, fragment-major 1-KiB
// blocks of 16 batch entries x 32 channels per plane) - and a product is the six leading terms of (Xh + Xm + Xl)^T (Dh + Dm + Dl)
// on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: the arithmetic of the plane convs (fp32-level error, gated against
// float64 next to the exact fp32 MFMA kernel in tests/test_p3.py).  The exact kernel (wgrad_stream_kernel, spiral_conv.hip)
// is paced by the issue rate of v_mfma_f32_16x16x4_f32 (0.43-0.55 of that pipe's peak); here the matrix time is 2.7x less
// and the kernel is a gather stream.
//
// The reduction index is (vertex, batch) - in the images the index ACROSS lanes, where the matrix instruction wants it inside
// a lane's eight values - so every operand block takes the hardware transpose on its way from LDS (ds_read_b64_tr_b16), as
// in the bf16 path's LDS-DMA weight gradient (bf16_wgrad.hip).  What the image layout buys: one LDS-DMA instruction
// (global_load_lds_dwordx4, lane l -> LDS base + 16 l) copies one whole 1-KiB fragment - 16 batch entries x 32 channels of one
// plane, CONTIGUOUS in memory, four consecutive lanes on 64 consecutive bytes (eight full lines per instruction; the vector L1
// serves a quad of lanes that touches two lines as two accesses: the first form of this kernel, which had lane 2 r + half
// fetch piece `half` of row r, ran at 2.2 accesses per 64 bytes) - into LDS as it is; no staging registers, no ds_write.  The
// transposed read of a 16-channel block then takes, per lane, 8 bytes of the piece (8-channel group kb, batch entry b): the
// 32 reduction rows of a matrix instruction are the 16 batch entries of two fragments (batch groups 2 bp and 2 bp + 1 of one
// vertex).  Bank conflicts: the two 8-channel groups of a 16-channel block are 256 bytes = all 64 banks apart, so the copy
// swaps the upper and lower eight batch entries of every ODD group (lane l fetches piece l ^ 8 when l & 16: still 64
// contiguous bytes per quad of lanes) and the reads undo it - conflict-free.
//
// A WAVE owns an output tile - QF gathered 32-column groups (a 32-channel fragment of one spiral position; two positions of a
// 16-channel image) x PT 16-channel tiles of dpre - over a contiguous range of stages (one stage = one vertex x 32 batch
// entries) and never synchronises inside its loop.  Its LDS ring holds G groups of six TR blocks (6 KiB: two 16-column
// halves x three planes); per stage PT/2 groups of dpre (kept in registers as B operands for the whole stage) and QF groups
// of gathered rows (A operands: 2 x PT x 6 MFMAs each) pass through it, G - 1 groups in flight behind a counted
// s_waitcnt vmcnt.  The four waves of a workgroup take four consecutive row chunks of one tile and add their tiles in LDS
// (fixed order): one fp32 partial slab per workgroup, reduced by the slab reduction every weight-gradient kernel shares.
#include "sh_bf16.h"

#include <type_traits>

namespace {

struct WP3Params {
    const char* xp; long x_vb, x_bgb;          // image of the gathered tensor: bytes per row / per 16-batch group
    const char* dp; long d_vb, d_bgb;          // image of dpre
    const int* table;                          // [R][S]
    float* slab; long slab_stride, bias_off;   // [nslab][Cout * K], then [nslab][Cout]
    int B, R, S, Cin, Cout, K;
    int nxg;                                   // gathered 32-column groups: S * Cin / 32, or ceil(S / 2) for 16-channel images
    int n_qg, n_pt, sgx, nslab, n_items;       // sgx: workgroups (slabs) per XCD and tile
    int nbg;                                   // 16-batch groups per row (B / 16)
    long n_units;                              // R * nbg (vertex, batch group) units; a stage = units 2 t and 2 t + 1
    long zero_unit16;                          // dpre-image offset (16-byte units) of batch group 0 of an all-zero row: the partner of
                                               // the last unit when n_units is odd (-1: none - the plan refuses an odd count)
    // tail job (sh_spiral_conv_bwd_wgt_p3_presum): workgroups n_items .. n_items + tail_blocks - 1 of the launch fill the pre-summed
    // rows the layer's backward-data pass reads through its transposed table - y[r] = sum_e val[e] dpre[col[e]] over the FP32
    // gradient rows, sh_spmm's arithmetic entry for entry (the rider of wgrad_stream_kernel, spiral_conv.hip)
    const float* df; long df_sv, df_sb;        // fp32 dpre, element strides of (row, batch entry)
    int tail_blocks, tail_rows;
    const int* tail_rowptr; const int* tail_col; const float* tail_val;
    float* tail_y;                             // same strides as df
    char* tail_img; long tail_img_vb, tail_img_bgb;   // three-plane image of the tail rows, or NULL
    long n_stages;
};

constexpr int WP_BLK = 1024;                   // one TR block: [32 r][16 idx] bf16
constexpr int WP_GRP = 6 * WP_BLK;             // a group: two 16-column halves x three planes
#ifndef WP_G
#define WP_G 6                                 // ring slots per wave: G - 1 groups (30 KiB) requested ahead
#endif
constexpr int WP_TBL_INTS = 1024;              // source offsets a wave may hold (stages x (1 + its column groups))
constexpr int WP_WAVE_LDS = WP_G * WP_GRP + WP_TBL_INTS * 4;      // 40 KiB: four waves = the CU's 160 KiB

// transposed operand fragment: lane (i = lane & 15, g = lane >> 4) receives channel i of batch entries {4g..4g+3} of the
// fragment at LDS byte address `lo` (batch group 2 bp) and of the one at `hi` (batch group 2 bp + 1) - its eight reduction rows.
// Fragment image in LDS: [8-channel group kb][batch entry b ^ (8 (kb & 1))][8 bf16]; the lane-dependent part of the address
// (wp_lane_part: 8-channel group p >> 1 of the block, batch entry 4g + q, half p & 1) is formed once per group by the caller,
// everything else is an immediate offset of the read.
__device__ __forceinline__ unsigned wp_lane_part(int lane) {
    const int g = lane >> 4, l16 = lane & 15, q = l16 >> 2, p = l16 & 3;
    return (unsigned)((p >> 1) * 256 + (((4 * g + q) ^ ((p >> 1) << 3)) << 4) + (p & 1) * 8);
}
__device__ __forceinline__ bf16x8 wp_frag(unsigned lo, unsigned hi) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)lo);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)hi);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return *reinterpret_cast<const bf16x8*>(&v);
}

// The tail job.  A tail workgroup only starts when a main workgroup has left its CU (every workgroup of the launch is allotted the
// 160 KiB of LDS), so it runs ALONE there, four waves per CU: what it needs is loads in flight per thread, not occupancy - each
// thread works on four output pieces (four (row, 256-element part) items) at once, up to four entries of each requested before
// the first sum.  Per element: acc = fma(val[e], dpre[col[e]], acc) in entry order - sh_spmm's / ws_presum_tail's sum, bit for bit.
__device__ __forceinline__ void wp_presum_tail(const WP3Params& p) {
    const int CW = p.Cout >> 2, per_row = p.B * CW;
    const int parts = (per_row + 255) >> 8;
    const long items = (long)p.tail_rows * parts;
    const long stride = p.tail_blocks;
    for (long it0 = (long)blockIdx.x - p.n_items; it0 < items; it0 += 4 * stride) {
        int rr[4], e0[4], cnt[4];
        long xo[4];
        bool ok[4];
        f32x4 xv[4][4];
        float wv[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long it = it0 + u * stride;
            const bool live = it < items;                                       // uniform
            const long itc = live ? it : 0;
            const int r = (int)(itc / parts), part = (int)(itc - (long)r * parts);
            rr[u] = r;
            e0[u] = p.tail_rowptr[r];
            cnt[u] = live ? p.tail_rowptr[r + 1] - e0[u] : 0;
            const int j = part * 256 + (int)threadIdx.x;
            ok[u] = live && j < per_row;
            const int jc = j < per_row ? j : 0;
            const int b = jc / CW, co = 4 * (jc - b * CW);
            xo[u] = (long)b * p.df_sb + co;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < cnt[u]) {                                               // uniform
                    xv[u][k] = *reinterpret_cast<const f32x4*>(p.df + (long)p.tail_col[e0[u] + k] * p.df_sv + xo[u]);
                    wv[u][k] = p.tail_val[e0[u] + k];
                }
            }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (cnt[u] <= 0 && !(it0 + u * stride < items)) continue;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < cnt[u]) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] = fmaf(wv[u][k], xv[u][k][q], acc[q]);
                }
            }
            for (int e = e0[u] + 4; e < e0[u] + cnt[u]; ++e) {                  // long lists: the rest, one entry at a time
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(p.df + (long)p.tail_col[e] * p.df_sv + xo[u]);
                const float w1 = p.tail_val[e];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = fmaf(w1, x1[q], acc[q]);
            }
            if (!ok[u]) continue;
            *reinterpret_cast<f32x4*>(p.tail_y + (long)rr[u] * p.df_sv + xo[u]) = acc;
            if (p.tail_img) {                                                   // the image of the row just written, as spmm_kernel<true, true>
                u32x2 h, m, l;
                sh_split3_quad(acc, h, m, l);
                const int b = (int)(xo[u] / p.df_sb), co = (int)(xo[u] - (long)b * p.df_sb);
                const bool c16 = p.Cout == 16;
                char* d = p.tail_img + (long)rr[u] * p.tail_img_vb + (long)(b >> 4) * p.tail_img_bgb +
                          (c16 ? ((co >> 3) * 16 + (b & 15)) * 16 : (co >> 5) * 3072 + (((co & 31) >> 3) * 16 + (b & 15)) * 16) + ((co >> 2) & 1) * 8;
                const int pb = c16 ? 512 : 1024;
                *reinterpret_cast<u32x2*>(d) = h;
                *reinterpret_cast<u32x2*>(d + pb) = m;
                *reinterpret_cast<u32x2*>(d + 2 * pb) = l;
            }
        }
    }
}

// Work assignment.  The reduction rows are cut into UNITS of one vertex x one 16-batch group (one fragment per plane and channel
// group); a STAGE is two consecutive units (2 t, 2 t + 1) - the 32 reduction rows of a matrix instruction; for batches that are
// multiples of 32 both belong to one vertex, otherwise (16, 48, ...) a stage may pair two vertices: every unit brings its own
// gather offsets, nothing else changes.  An odd unit count is completed by a unit of an all-zero dpre row (its products are 0).
// XCD x (= blockIdx & 7: workgroups are dealt round-robin) owns a contiguous eighth of the stages and its C = 4 sgx waves per tile
// walk it TOGETHER: wave c takes stages lo + c, lo + c + C, ... - at any moment the XCD gathers the neighbourhoods of a short run
// of consecutive vertices.  (Measured, profiles/r06_wgrad_p3.txt: what this kernel waits for is the L2 -> CU gather itself, 11-15
// TB/s of plane bytes; a perfectly local table, one batch pair per XCD, or non-temporal dpre loads change its time by < 2 %.)
template <int QF, int PT, bool XC16>
__global__ __launch_bounds__(256) void wgrad_p3_kernel(const WP3Params p) {
    static_assert(PT % 2 == 0, "dpre channels come in 32-channel fragments");
    constexpr int NGD = PT / 2, NG = NGD + QF, G = WP_G;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x >= p.n_items) { wp_presum_tail(p); return; }        // tail job (whole workgroups; before any barrier)
    const int lane = threadIdx.x & 63;
    const int wave = sh_wave_id();
    char* ring = smem + wave * WP_WAVE_LDS;
    int* Tl = reinterpret_cast<int*>(ring + G * WP_GRP);
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int tiles = p.n_qg * p.n_pt;
    const int tile = local % tiles, sgl = local / tiles;
    const int qg = tile % p.n_qg, pt = tile / p.n_qg;
    const int sgroup = xcd * p.sgx + sgl;
    const int C = 4 * p.sgx, c = sgl * 4 + wave;
    const int S = p.S, nbg = p.nbg;
    const int S0 = (int)((long)xcd * p.n_stages / 8), S1 = (int)((long)(xcd + 1) * p.n_stages / 8);
    const int nst = c < S1 - S0 ? (S1 - S0 - c + C - 1) / C : 0;
    // this wave's source offsets -> LDS, all in 16-byte units (32 bits reach 64 GiB): per stage n and unit h (0 / 1)
    //   Tl[n * NE + h]                          dpre:  v * d_vb + bg * d_bgb   (the completing unit: the zero row's)
    //   Tl[n * NE + 2 + 2 (k * H + hb) + h]     gathered group k (16-channel images: its half hb): table[v][position] * x_vb + bg * x_bgb +
    //                                           the offset of its 32-channel group   (groups past the end: a duplicate, never stored)
    // so a fragment's source address is ONE LDS read away and the loop holds no division
    constexpr int H = XC16 ? 2 : 1, NE = 2 + 2 * QF * H;
    const int f0 = qg * QF;
    for (int i = lane; i < nst * NE; i += 64) {
        const int n = i / NE, e = i - n * NE, hu = e & 1;
        const long u = 2L * (S0 + c + n * C) + hu;
        const bool pad = u >= p.n_units;                                    // only the very last unit of an odd count
        const long uc = pad ? u - 1 : u;
        const int v = (int)(uc / nbg), bg = (int)(uc - (long)v * nbg);
        unsigned val;
        if (e < 2) {
            val = pad ? (unsigned)p.zero_unit16 : (unsigned)v * (unsigned)(p.d_vb >> 4) + (unsigned)bg * (unsigned)(p.d_bgb >> 4);
        } else {
            const int kk = (e - 2) >> 1, k = kk / H, hb = kk - k * H;
            const int f = f0 + k < p.nxg ? f0 + k : p.nxg - 1;
            int sp, off16;
            if (XC16) { sp = 2 * f + hb < S ? 2 * f + hb : S - 1; off16 = 0; }
            else { const int ncg = p.Cin >> 5; sp = f / ncg; off16 = (f - sp * ncg) * 192; }
            val = (unsigned)p.table[(long)v * S + sp] * (unsigned)(p.x_vb >> 4) + (unsigned)bg * (unsigned)(p.x_bgb >> 4) + (unsigned)off16;
        }
        Tl[i] = (int)val;
    }
    // which 16-byte piece of a 1-KiB instruction this lane copies (see the header): l, with the low / high eight batch entries of
    // every odd 8-channel group swapped (16-channel images: a 512-byte plane has two groups: the same rule)
    const long lane_off = (long)((lane ^ ((lane >> 1) & 8)) << 4);
    typedef __attribute__((address_space(3))) char* lptr_t;
    auto dma16 = [](const char* gsrc, unsigned lds_dst) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory", "m0");
    };
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lptr_t)ring);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");        // the offsets are in LDS

    // issue side: the stage whose groups are being requested; advanced when a stage's last group went out
    int iss_n = 0, issued = 0;
    const char* xbase = p.xp + lane_off;
    const char* dbase = p.dp + (long)(pt * NGD) * 3072 + lane_off;
    auto issue = [&](auto JJ) {
        constexpr int jj = decltype(JJ)::value;
        const unsigned slot = ring_lds + (unsigned)((issued % G) * WP_GRP);
        if constexpr (jj < NGD) {                                           // fragments (unit h, plane pl) of 32 dpre channels
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned o = (unsigned)__builtin_amdgcn_readfirstlane(Tl[iss_n * NE + h]);
                const char* base = dbase + ((unsigned long)o << 4) + jj * 3072;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) dma16(base + pl * 1024, slot + (unsigned)((h * 3 + pl) * WP_BLK));
            }
        } else {
            constexpr int k = jj - NGD;
            if constexpr (XC16) {
                // per half hb: the two units' three 512-byte planes = six planes [lo h, lo m, lo l, hi h, hi m, hi l], two per
                // instruction (lanes 0-31 the first, 32-63 the second; lane_off already carries the second plane's + 512):
                // (lo h | lo m), (lo l | hi h), (hi m | hi l) - only the middle one takes its halves from different units
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    const unsigned olo = (unsigned)__builtin_amdgcn_readfirstlane(Tl[iss_n * NE + 2 + 2 * (k * 2 + hb)]);
                    const unsigned ohi = (unsigned)__builtin_amdgcn_readfirstlane(Tl[iss_n * NE + 2 + 2 * (k * 2 + hb) + 1]);
                    const char* blo = xbase + ((unsigned long)olo << 4);
                    const char* bhi = xbase + ((unsigned long)ohi << 4);
                    dma16(blo, slot + (unsigned)((hb * 3 + 0) * WP_BLK));
                    dma16(lane < 32 ? blo + 1024 : bhi - 512, slot + (unsigned)((hb * 3 + 1) * WP_BLK));
                    dma16(bhi + 512, slot + (unsigned)((hb * 3 + 2) * WP_BLK));
                }
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const unsigned o = (unsigned)__builtin_amdgcn_readfirstlane(Tl[iss_n * NE + 2 + 2 * k + h]);
                    const char* base = xbase + ((unsigned long)o << 4);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) dma16(base + pl * 1024, slot + (unsigned)((h * 3 + pl) * WP_BLK));
                }
            }
        }
        ++issued;
        if constexpr (jj == NG - 1) {                                       // the stage is out: next one (the last one again past the end)
            if (iss_n + 1 < nst) ++iss_n;
        }
    };

    f32x4 acc[QF][2][PT], accb[PT];
#pragma unroll
    for (int k = 0; k < QF; ++k)
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int j = 0; j < PT; ++j) acc[k][hb][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < PT; ++j) accb[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16x8 ones = {(__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f};
    const bool want_bias = qg == 0;
    bf16x8 fb[PT][3];
    // Operand fragments are read from LDS ONE GROUP AHEAD (one wave per SIMD: nothing else would hide the LDS round trip in front
    // of every group's products): fr[b] holds the six fragments - two 16-channel blocks x three planes - of a group, whatever its
    // kind; the group being multiplied and the one being read alternate between the two sets (a stage with an odd number of
    // groups ends with one register copy of the set, so that the parity is a compile-time constant).
    bf16x8 fr[2][2][3];
    const unsigned lane_part = wp_lane_part(lane);
    auto read_group = [&](unsigned slot_lds, bf16x8 (&f)[2][3]) {
        unsigned vb = slot_lds + lane_part;
        asm volatile("" : "+v"(vb));                                        // ONE address register per group; the rest are immediates
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                if constexpr (XC16) {
                    // (a gathered 16-channel group: half hb = [bg 0: h m l][bg 1: h m l] x 512 bytes, a plane = one block)
                    f[hb][pl] = wp_frag(vb + hb * 3 * WP_BLK + pl * 512, vb + hb * 3 * WP_BLK + pl * 512 + 1536);
                } else {
                    f[hb][pl] = wp_frag(vb + pl * WP_BLK + hb * 512, vb + (3 + pl) * WP_BLK + hb * 512);
                }
            }
    };
    // dpre fragments always have the 32-channel layout
    auto read_dgroup = [&](unsigned slot_lds, bf16x8 (&f)[2][3]) {
        unsigned vb = slot_lds + lane_part;
        asm volatile("" : "+v"(vb));
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) f[cb][pl] = wp_frag(vb + pl * WP_BLK + cb * 512, vb + (3 + pl) * WP_BLK + cb * 512);
    };

    auto multiply = [&](auto J, const bf16x8 (&f)[2][3]) {
        constexpr int j0 = decltype(J)::value;
        if constexpr (j0 < NGD) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) fb[2 * j0 + cb][pl] = f[cb][pl];
            if (want_bias) {                                                // column sums of the dpre tile: ones^T . (Dl + Dm + Dh)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    f32x4 cc = accb[2 * j0 + cb];
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, f[cb][2], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, f[cb][1], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, f[cb][0], cc, 0, 0, 0);
                    accb[2 * j0 + cb] = cc;
                }
            }
        } else {
            constexpr int k = j0 - NGD;
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int j = 0; j < PT; ++j) {
                    f32x4 cc = acc[k][hb][j];                               // smallest terms first, one dependent chain (the fast form: DESIGN 4e)
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[hb][2], fb[j][0], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[hb][0], fb[j][2], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[hb][1], fb[j][1], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[hb][1], fb[j][0], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[hb][0], fb[j][1], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[hb][0], fb[j][0], cc, 0, 0, 0);
                    acc[k][hb][j] = cc;
                }
        }
    };

    // flat group sequence t = stage * NG + j; group t lives in slot t % G.  Iteration t: group t + 1 has landed -> its fragments
    // into the idle register set; group t + G - 1 requested into the slot group t - 1 was read from (an iteration ago); the
    // products of group t.  Groups t + 2 .. t + G - 1 are in flight meanwhile.
    auto prologue = [&](auto self, auto D) -> void {
        constexpr int d = decltype(D)::value;
        if constexpr (d < G - 1) {
            issue(std::integral_constant<int, d % NG>{});
            self(self, std::integral_constant<int, d + 1>{});
        }
    };
    int t = 0;
    if (nst > 0) {
        prologue(prologue, std::integral_constant<int, 0>{});
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((G - 2) * 6) : "memory");
        read_dgroup(ring_lds, fr[0]);                                        // group 0 is a dpre group
    }
    auto stage_groups = [&](auto self, auto J, auto PAR) -> void {
        constexpr int j = decltype(J)::value, par = decltype(PAR)::value;
        if constexpr (j < NG) {
            constexpr int cur = (j + par) & 1, jn = (j + 1) % NG;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((G - 3) * 6) : "memory");
            const unsigned nslot = ring_lds + (unsigned)(((t + 1) % G) * WP_GRP);
            if constexpr (jn < NGD) read_dgroup(nslot, fr[cur ^ 1]);
            else read_group(nslot, fr[cur ^ 1]);
            issue(std::integral_constant<int, (j + G - 1) % NG>{});
            multiply(J, fr[cur]);
            ++t;
            self(self, std::integral_constant<int, j + 1>{}, PAR);
        }
    };
    for (int st = 0; st < nst; ++st) {
        stage_groups(stage_groups, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        if constexpr (NG % 2 != 0) {                                         // the next stage's first group was read into set 1: every stage starts on set 0
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) fr[0][hb][pl] = fr[1][hb][pl];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // the clamped tail loads still write this wave's LDS
    __syncthreads();                                                         // every wave is done with its ring: reuse it for the sum

    // four waves -> one slab: passes of up to 32 tiles per wave (128 KiB for the four).  (A form whose summing threads store whole
    // runs of a channel's columns - 64 contiguous bytes per four threads instead of one 16-byte piece per channel row and lane -
    // measured the same on one box, 18.4 / 19.5 / 27.3 / 45.6 / 29.1 / 37.9 against 18.3 / 19.2 / 27.1 / 45.1 / 29.0 / 38.2 us: the
    // slab stores are not what the launch waits for.  Not kept.)
    constexpr int NT = QF * 2 * PT, TPP = 32, NPASS = (NT + TPP - 1) / TPP;
    float* red = reinterpret_cast<float*>(smem);
    float* slab = p.slab + (long)sgroup * p.slab_stride;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        constexpr int WAVE_F = TPP * 256;
        float* mine = red + wave * WAVE_F;
#pragma unroll
        for (int k = 0; k < QF; ++k)
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int j = 0; j < PT; ++j) {
                    const int tl = (k * 2 + hb) * PT + j;
                    if (tl / TPP == ps) *reinterpret_cast<f32x4*>(mine + (tl % TPP) * 256 + lane * 4) = acc[k][hb][j];
                }
        __syncthreads();
        const int ntile = NT - ps * TPP < TPP ? NT - ps * TPP : TPP;
        for (int e = threadIdx.x; e < ntile * 64; e += 256) {
            const int tl = ps * TPP + (e >> 6), l = e & 63;
            const int k = tl / (2 * PT), hb = (tl / PT) & 1, j = tl % PT;
            const f32x4 v = ((*reinterpret_cast<const f32x4*>(red + e * 4) + *reinterpret_cast<const f32x4*>(red + WAVE_F + e * 4)) +
                             *reinterpret_cast<const f32x4*>(red + 2 * WAVE_F + e * 4)) + *reinterpret_cast<const f32x4*>(red + 3 * WAVE_F + e * 4);
            const int f = f0 + k;
            const int q = f * 32 + hb * 16 + 4 * (l >> 4), co = (pt * PT + j) * 16 + (l & 15);
            const bool col_ok = f < p.nxg && (XC16 ? 2 * f + hb < S : true) && q < p.K;
            if (col_ok && co < p.Cout) *reinterpret_cast<f32x4*>(slab + (long)co * p.K + q) = v;
        }
        __syncthreads();
    }
    if (want_bias) {                                                         // (uniform per workgroup)
        if (lane < 16) {
#pragma unroll
            for (int j = 0; j < PT; ++j) red[wave * (PT * 16) + j * 16 + lane] = accb[j][0];      // row 0 of ones^T . D
        }
        __syncthreads();
        if (threadIdx.x < PT * 16) {
            const int e = threadIdx.x, co = pt * PT * 16 + e;
            const float v = ((red[e] + red[PT * 16 + e]) + red[2 * PT * 16 + e]) + red[3 * PT * 16 + e];
            if (co < p.Cout) p.slab[p.bias_off + (long)sgroup * p.Cout + co] = v;
        }
    }
}

// column groups per wave the kernel is built for (PT = 2: <= 16, PT = 4: <= 8: 256 accumulator registers)
constexpr int WP_QF2[] = {4, 6, 8, 9, 11, 12, 16}, WP_QF4[] = {4, 6, 8};

struct WP3Plan { int ok, qf, pt, xc16, nxg, n_qg, n_pt, sgx, nslab; long n_stages, n_units; };
WP3Plan plan_wp3(int B, int R, int S, int Cin, int Cout) {
    WP3Plan w{};
    if (B <= 0 || B % 16 != 0 || R <= 0 || S <= 0) return w;
    if (!(Cin == 16 || Cin % 32 == 0) || Cout % 32 != 0) return w;
    w.xc16 = Cin == 16;
    w.nxg = w.xc16 ? (S + 1) / 2 : S * (Cin / 32);
    w.pt = Cout % 64 == 0 ? 4 : 2;
    w.n_pt = Cout / (16 * w.pt);
    w.n_units = (long)R * (B / 16);
    w.n_stages = (w.n_units + 1) / 2;
    // Column groups per wave (QF), out of the widths the kernel is built for: one workgroup (160 KiB of LDS) per CU, so
    // 8 XCDs x tiles x sgx <= 256 workgroups; a wave's time is its stage count x the groups of a stage (dpre + QF gathered),
    // and with only a handful of stages per wave the rounding of that count decides (1724 rows x 16 groups: 3.4 stages of 17
    // groups per wave = 4 x 17, or split in two column groups 6.7 stages of 9 = 7 x 9).  Minimise it; ties: the wider group
    // (dpre is re-read once per column group).
    static const int wg_target = sh_env_int("SH_WP3_BLOCKS", 256, 8, 1 << 16);
    static const int slab_mb = sh_env_int("SH_WP3_SLAB_MB", 32, 1, 4096);
    static const int q_force = sh_env_int("SH_WP3_QF", 0, 0, 16);
    const long cap = (((long)slab_mb << 20) / ((long)Cout * S * Cin * 4)) / 8;
    const long per_xcd = (w.n_stages + 7) / 8;
    const int ngd = w.pt / 2;
    long best = -1, sgx = 1;
    auto consider = [&](int q) {
        if (q_force && q != q_force) return;
        const int n_qg = sh_cdiv(w.nxg, q);
        if (q > w.nxg && q != (w.pt == 2 ? WP_QF2[0] : WP_QF4[0])) return;       // wider than the layer: only the narrowest form
        const long tiles = (long)n_qg * w.n_pt;
        long sg = wg_target / (8 * tiles);
        if (sg > cap) sg = cap;
        if (sg < 1) sg = 1;
        for (;; ++sg) {                                                         // a wave's source offsets must fit its LDS area
            const long nst = (per_xcd + 4 * sg - 1) / (4 * sg);
            if (nst * (2 + 2 * q * (w.xc16 ? 2 : 1)) <= WP_TBL_INTS) break;
            if (sg > per_xcd) return;
        }
        const long rounds = (8 * tiles * sg + wg_target - 1) / wg_target;        // more workgroups than CUs: they run in turns
        // + the workgroup's epilogue (its 2 q pt accumulator tiles through LDS, summed over the four waves, stored as a slab): half a
        // group's time per tile pair - calibrated on a sweep of forced widths (tools/exp/r06_wp3_qf_sweep.sh: 863 x 256 -> 64 and
        // 432 x 512 -> 128 are 2.4 / 2.9 us faster with four column groups per wave than with eight; in units of half a group)
        const long cost = 2 * rounds * ((per_xcd + 4 * sg - 1) / (4 * sg)) * (ngd + q) + (long)q * w.pt;
        if (best < 0 || cost < best || (cost == best && q > w.qf)) { best = cost; w.qf = q; w.n_qg = n_qg; sgx = sg; }
    };
    if (w.pt == 2) { for (int q : WP_QF2) consider(q); }
    else { for (int q : WP_QF4) consider(q); }
    if (best < 0) return w;
    w.sgx = (int)sgx;
    w.nslab = 8 * w.sgx;
    w.ok = 1;
    return w;
}

template <int QF, int PT, bool XC16>
int launch_wp3(const WP3Params& p, hipStream_t st) {
    auto kern = wgrad_p3_kernel<QF, PT, XC16>;
    const size_t smem = (size_t)4 * WP_WAVE_LDS;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            sh_set_error("wgrad_p3: cannot raise the dynamic LDS limit to %zu bytes", smem);
            return SH_ERR_LAUNCH;
        }
        attr_set = true;
    }
    ShProfScope ps(st, "wgrad_p3_kernel<%d, %d, %s>|R=%d B=%d K=%d N=%d grid=%d presum=%d", QF, PT, XC16 ? "true" : "false", p.R, p.B, p.K, p.Cout,
                   p.n_items, p.tail_blocks ? p.tail_rows : 0);
    SH_LAUNCH_PS(ps, kern, dim3(p.n_items + p.tail_blocks), dim3(256), smem, st, p);
    SH_CHECK_LAUNCH("wgrad_p3");
    return SH_OK;
}

template <int PT, bool XC16>
int dispatch_wp3(int qf, const WP3Params& p, hipStream_t st) {
    if constexpr (PT == 2) {
        switch (qf) {
            case 4: return launch_wp3<4, 2, XC16>(p, st);
            case 6: return launch_wp3<6, 2, XC16>(p, st);
            case 8: return launch_wp3<8, 2, XC16>(p, st);
            case 9: return launch_wp3<9, 2, XC16>(p, st);
            case 11: return launch_wp3<11, 2, XC16>(p, st);
            case 12: return launch_wp3<12, 2, XC16>(p, st);
            case 16: return launch_wp3<16, 2, XC16>(p, st);
        }
    } else {
        switch (qf) {
            case 4: return launch_wp3<4, 4, XC16>(p, st);
            case 6: return launch_wp3<6, 4, XC16>(p, st);
            case 8: return launch_wp3<8, 4, XC16>(p, st);
        }
    }
    sh_set_error("wgrad_p3: no kernel for %d column groups x %d channel tiles", qf, PT);
    return SH_ERR_UNSUPPORTED;
}

}  // namespace

// slab count of the plane-form plan (0: the shape is not taken); shared with the slab reduction in spiral_conv.hip
int sh_wgrad_p3_nslab(int B, int R, int S, int Cin, int Cout) {
    const WP3Plan w = plan_wp3(B, R, S, Cin, Cout);
    return w.ok ? w.nslab : 0;
}

extern "C" {

int sh_spiral_conv_bwd_wgt_p3_ok(int B, int R, int S, int Cin, int Cout) { return plan_wp3(B, R, S, Cin, Cout).ok; }

size_t sh_spiral_conv_bwd_wgt_p3_workspace(int B, int R, int S, int Cin, int Cout) {
    const WP3Plan w = plan_wp3(B, R, S, Cin, Cout);
    if (!w.ok) return 0;
    return (size_t)w.nslab * ((size_t)Cout * S * Cin + Cout) * sizeof(float);
}

int sh_spiral_conv_bwd_wgt_p3(const void* dpre_planes, int dpre_zero_row, const void* x_planes, const int32_t* table, void* workspace,
                              size_t workspace_bytes, int B, int R, int S, int Cin, int Cout, sh_stream_t stream) {
    return sh_spiral_conv_bwd_wgt_p3_presum(dpre_planes, dpre_zero_row, x_planes, table, workspace, workspace_bytes, nullptr, 0, 0, nullptr, nullptr,
                                            nullptr, nullptr, nullptr, 0, B, R, S, Cin, Cout, stream);
}

int sh_spiral_conv_bwd_wgt_p3_presum(const void* dpre_planes, int dpre_zero_row, const void* x_planes, const int32_t* table, void* workspace,
                                     size_t workspace_bytes, const float* dpre, int64_t dp_sv, int64_t dp_sb, const int32_t* sum_rowptr,
                                     const int32_t* sum_col, const float* sum_val, float* sum_out, void* sum_out_planes, int sum_rows, int B,
                                     int R, int S, int Cin, int Cout, sh_stream_t stream) {
    SH_REQUIRE(dpre_planes && x_planes && table && workspace, SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_wgt_p3: null pointer");
    SH_REQUIRE(sum_rows == 0 || (sum_rows > 0 && dpre && sum_rowptr && sum_col && sum_val && sum_out), SH_ERR_INVALID_ARG,
               "sh_spiral_conv_bwd_wgt_p3_presum: incomplete pre-sum job");
    const WP3Plan w = plan_wp3(B, R, S, Cin, Cout);
    SH_REQUIRE(w.ok, SH_ERR_UNSUPPORTED,
               "sh_spiral_conv_bwd_wgt_p3: B=%d S=%d Cin=%d Cout=%d is not taken (batch %% 16 == 0; Cin 16 or %% 32 == 0; Cout %% 32 == 0)", B, S,
               Cin, Cout);
    SH_REQUIRE(w.n_units % 2 == 0 || (dpre_zero_row >= 0 && dpre_zero_row < R), SH_ERR_UNSUPPORTED,
               "sh_spiral_conv_bwd_wgt_p3: %d rows x %d batch groups is an odd number of units - name an all-zero row of dpre (dpre_zero_row)", R,
               B / 16);
    SH_REQUIRE(workspace_bytes >= sh_spiral_conv_bwd_wgt_p3_workspace(B, R, S, Cin, Cout), SH_ERR_WORKSPACE,
               "sh_spiral_conv_bwd_wgt_p3: workspace too small");
    SH_REQUIRE(((reinterpret_cast<uintptr_t>(dpre_planes) | reinterpret_cast<uintptr_t>(x_planes) | reinterpret_cast<uintptr_t>(workspace)) & 15) == 0,
               SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_wgt_p3: images and workspace must be 16-byte aligned");
    WP3Params p{};
    const int nbg = B / 16;
    p.xp = static_cast<const char*>(x_planes);
    p.x_bgb = Cin == 16 ? 1536 : (long)(Cin / 32) * 3072; p.x_vb = p.x_bgb * nbg;
    p.dp = static_cast<const char*>(dpre_planes);
    p.d_bgb = (long)(Cout / 32) * 3072; p.d_vb = p.d_bgb * nbg;
    p.table = table; p.slab = static_cast<float*>(workspace);
    p.B = B; p.R = R; p.S = S; p.Cin = Cin; p.Cout = Cout; p.K = S * Cin;
    p.slab_stride = (long)Cout * p.K; p.bias_off = (long)w.nslab * p.slab_stride;
    p.nxg = w.nxg; p.n_qg = w.n_qg; p.n_pt = w.n_pt; p.sgx = w.sgx; p.nslab = w.nslab; p.n_stages = w.n_stages;
    p.n_items = w.n_qg * w.n_pt * w.nslab;
    p.nbg = nbg; p.n_units = w.n_units;
    p.zero_unit16 = dpre_zero_row >= 0 ? (long)dpre_zero_row * (p.d_vb >> 4) : -1;
    if (sum_rows > 0) {
        // the pre-sum job: a launch of its own in front of this one (default), or tail workgroups of this launch (SH_WP3_TAIL=1).
        // Measured (6890 vertices, batch 64, profiles/r06_wgrad_p3.txt): as tail workgroups the job costs MORE than its own launch -
        // every workgroup of this launch is allotted the CU's whole LDS, so a tail workgroup only starts when a main one has left and
        // then runs alone on its CU, four waves deep, on a chain of dependent loads (row pointer -> column -> rows): the five hosting
        // launches 37 / 31 / 47 / 30 / 22 us -> 85 / 48 / 83 / 44 / 32, step 1.416 -> 1.483 ms.  The fp32 kernels host the job beside
        // their one wave per SIMD (no LDS to speak of); this kernel cannot.
        static const int tail_on = sh_env_int("SH_WP3_TAIL", 0, 0, 1), tail_cap = sh_env_int("SH_WP3_TAIL_BLOCKS", 256, 1, 1 << 16);
        const bool sum_vec = (Cout % 4 == 0) && (dp_sv % 4 == 0) && (dp_sb % 4 == 0) &&
                             ((reinterpret_cast<uintptr_t>(dpre) | reinterpret_cast<uintptr_t>(sum_out)) % 16 == 0);
        if (tail_on && sum_vec) {
            const long items = (long)sum_rows * (((long)B * (Cout / 4) + 255) / 256);
            const long want = (items + 3) / 4;
            p.tail_blocks = (int)(want < tail_cap ? want : tail_cap);
            p.df = dpre; p.df_sv = dp_sv; p.df_sb = dp_sb;
            p.tail_rows = sum_rows; p.tail_rowptr = sum_rowptr; p.tail_col = sum_col; p.tail_val = sum_val; p.tail_y = sum_out;
            if (sum_out_planes) {
                SH_REQUIRE(dp_sb == Cout && dp_sv == (int64_t)B * Cout && sh_p3_bytes(1, B, Cout) && (reinterpret_cast<uintptr_t>(sum_out_planes) & 15) == 0,
                           SH_ERR_UNSUPPORTED, "sh_spiral_conv_bwd_wgt_p3_presum: B=%d Cout=%d has no plane image", B, Cout);
                p.tail_img = static_cast<char*>(sum_out_planes);
                p.tail_img_bgb = (long)(Cout / 32) * 3072;
                p.tail_img_vb = p.tail_img_bgb * (B / 16);
            }
        } else {
            const int rc = sh_spmm_p3(sum_rowptr, sum_col, sum_val, dpre, dp_sv, dp_sb, sum_out, dp_sv, dp_sb, sum_out_planes, nullptr, 0, 0, 0, -1, B,
                                      sum_rows, Cout, stream);
            if (rc != SH_OK) return rc;
        }
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (w.xc16) return w.pt == 4 ? dispatch_wp3<4, true>(w.qf, p, st) : dispatch_wp3<2, true>(w.qf, p, st);
    return w.pt == 4 ? dispatch_wp3<4, false>(w.qf, p, st) : dispatch_wp3<2, false>(w.qf, p, st);
}

}  // extern "C"
