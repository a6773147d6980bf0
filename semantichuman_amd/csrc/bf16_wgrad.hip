// Spiral-convolution weight gradient in bf16 (BASELINE config 3):
//
//   dW[co][s*Cin + ci] = sum_{v,b} dpre[v,b,co] * x[table[v,s], b, ci]        dbias[co] = sum_{v,b} dpre[v,b,co]
//                                                                            (autograd of reference models.py:45)
// The reduction runs over the (vertex, batch) rows, which is the SLOW index of both operands in memory (channels are
// contiguous), so both go through the transposed LDS image of sh_bf16_tiles.h and come back through
// ds_read_b64_tr_b16 as v_mfma_f32_16x16x32_bf16 fragments; the gathered [rows, S*Cin] matrix is re-gathered on the
// fly, 16 bytes (8 channels of one neighbour) per lane.  A workgroup owns 64 weight columns x 64 output channels and a
// contiguous range of 64-row stages; it writes an fp32 partial slab, and the slabs are summed in a fixed order by the
// same reduction kernel the fp32 path uses (deterministic, no atomics).  fp32 master gradients out.
// 3-channel fp32 operands (the xyz input of the first layer, the xyz gradient of the last) are converted on the way in.
#include "sh_bf16_tiles.h"

namespace {

struct BWParams {
    const char* dpre; long dp_rb, dp_bb;      // byte strides (row, batch entry)
    const char* x; long x_rb, x_bb;
    const int* table;                          // [R][S]
    float* slab; long slab_stride, bias_off;   // [nsplit][Cout * K], then [nsplit][Cout]
    int B, R, S, Cin, Cout, K, Kq;             // Kq = gathered columns as staged (K, or 4 S for 3-channel inputs)
    int n_qt, n_pt, nsplit, stages_per_split;
    long rows;                                 // R * B
};

struct __attribute__((packed, aligned(4))) bw_f3 { float a, b, c; };
__device__ __forceinline__ u32x4 bw_pack(const bw_f3& u, const bw_f3& v) {
    const bf16x8 o = {(__bf16)u.a, (__bf16)u.b, (__bf16)u.c, (__bf16)0.f, (__bf16)v.a, (__bf16)v.b, (__bf16)v.c, (__bf16)0.f};
    return *reinterpret_cast<const u32x4*>(&o);
}

// XC3: x is fp32 with 3 channels (columns staged as zero-padded quads q' = 4 s + c); PC3: dpre is fp32 with 3 channels
template <bool XC3, bool PC3>
__global__ __launch_bounds__(256) void wgrad_bf16_kernel(const BWParams p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TG_IMG_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = sh_wave_id();
    int bid = blockIdx.x;
    const int qt = bid % p.n_qt; bid /= p.n_qt;
    const int pt = bid % p.n_pt; const int split = bid / p.n_pt;
    const int q0 = qt * 64, p0 = pt * 64;
    const long st0 = (long)split * p.stages_per_split;
    const long nst_all = (p.rows + 63) >> 6;
    const int nst = (int)min((long)p.stages_per_split, nst_all - st0);
    const int qh = wave & 1, ph = wave >> 1;

    // this thread's two pieces per operand and stage: rows (tid >> 3) and (tid >> 3) + 32, 8-column group c8 = tid & 7
    const int c8 = tid & 7, prow = tid >> 3;
    // gathered operand: column group -> (spiral position, channel) is the same for every stage
    const int kq = q0 + 8 * c8;
    const bool a_ok = kq < p.Kq;
    int sa, sb = 0, ca = 0;                    // XC3: two neighbours sa, sb per piece; else one neighbour sa, channel ca
    if (XC3) { sa = a_ok ? kq >> 2 : 0; sb = sa + 1 < p.S ? sa + 1 : sa; }
    else { sa = a_ok ? kq / p.Cin : 0; ca = a_ok ? kq - sa * p.Cin : 0; }
    const bool sb_ok = XC3 && a_ok && (kq >> 2) + 1 < p.S;
    const int cb = p0 + 8 * c8;                // dpre channel group
    const bool b_ok = cb < p.Cout;

    struct Idx { int v[2], b[2], ta[2], tb[2]; bool ok[2]; };
    auto rows_of = [&](long st, Idx& ix) {     // (vertex, batch) of the stage's two rows + their gather-table entries
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const long r = (st << 6) + prow + 32 * k;
            ix.ok[k] = r < p.rows;
            const long rc = ix.ok[k] ? r : 0;
            ix.v[k] = (int)(rc / p.B); ix.b[k] = (int)(rc - (long)ix.v[k] * p.B);
            ix.ta[k] = p.table[(long)ix.v[k] * p.S + sa];
            ix.tb[k] = XC3 ? p.table[(long)ix.v[k] * p.S + sb] : 0;
        }
    };
    auto load = [&](const Idx& ix, u32x4 (&ra)[2], u32x4 (&rb)[2]) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const u32x4 z = {0u, 0u, 0u, 0u};
            if (XC3) {
                const char* xb = p.x + (long)ix.b[k] * p.x_bb;
                const bw_f3 u = *reinterpret_cast<const bw_f3*>(xb + (long)ix.ta[k] * p.x_rb);
                bw_f3 w = *reinterpret_cast<const bw_f3*>(xb + (long)ix.tb[k] * p.x_rb);
                if (!sb_ok) w = bw_f3{0.f, 0.f, 0.f};
                ra[k] = (ix.ok[k] && a_ok) ? bw_pack(u, w) : z;
            } else {
                const u32x4 g = *reinterpret_cast<const u32x4*>(p.x + (long)ix.ta[k] * p.x_rb + (long)ix.b[k] * p.x_bb + 2 * ca);
                ra[k] = (ix.ok[k] && a_ok) ? g : z;
            }
            const char* dp = p.dpre + (long)ix.v[k] * p.dp_rb + (long)ix.b[k] * p.dp_bb;
            if (PC3) {
                const bw_f3 u = *reinterpret_cast<const bw_f3*>(dp);
                const bf16x8 o = {(__bf16)u.a, (__bf16)u.b, (__bf16)u.c, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
                rb[k] = (ix.ok[k] && c8 == 0 && p0 == 0) ? *reinterpret_cast<const u32x4*>(&o) : z;
            } else {
                const u32x4 g = *reinterpret_cast<const u32x4*>(dp + (b_ok ? 2 * cb : 0));
                rb[k] = (ix.ok[k] && b_ok) ? g : z;
            }
        }
    };
    auto store = [&](char* img, const u32x4 (&ra)[2], const u32x4 (&rb)[2]) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            *reinterpret_cast<u32x4*>(img + tg_tr_piece(prow + 32 * k, c8)) = ra[k];
            *reinterpret_cast<u32x4*>(img + TG_IMG_BYTES + tg_tr_piece(prow + 32 * k, c8)) = rb[k];
        }
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float csum = 0.f;

    // pipeline: gather-table entries two stages ahead, operand loads one stage ahead, LDS double-buffered
    Idx ixa, ixb;
    u32x4 ra[2], rb[2];
    if (nst > 0) {
        rows_of(st0, ixa);
        load(ixa, ra, rb);
        rows_of(st0 + (nst > 1 ? 1 : 0), ixb);
        store(smem, ra, rb);
    }
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
        const int cur = st & 1;
        if (st + 1 < nst) {
            load(ixb, ra, rb);                                          // stage st + 1 (its table entries arrived a stage ago)
            rows_of(st0 + (st + 2 < nst ? st + 2 : st + 1), ixb);       // stage st + 2
        }
        const char* ia = smem + cur * 2 * TG_IMG_BYTES;
        const char* ib = ia + TG_IMG_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = tg_tr_frag(ia, 2 * qh + i, kk, lane);
                fb[i] = tg_tr_frag(ib, 2 * ph + i, kk, lane);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (qt == 0 && tid < 64) {                                      // bias gradient: column sums of the dpre tile
            const char* col = ib + (tid >> 4) * TG_TR_BLK + (tid & 15) * 2;
#pragma unroll 8
            for (int r = 0; r < 64; ++r) csum += (float)*reinterpret_cast<const __bf16*>(col + (r >> 5) * TG_TR_KS + (r & 31) * 32);
        }
        if (st + 1 < nst) store(smem + (cur ^ 1) * 2 * TG_IMG_BYTES, ra, rb);
        __syncthreads();
    }

    // lane holds columns q .. q+3 of output channel co
    float* slab = p.slab + (long)split * p.slab_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = q0 + 32 * qh + 16 * i + 4 * (lane >> 4), co = p0 + 32 * ph + 16 * j + (lane & 15);
            if (q >= p.Kq || co >= p.Cout) continue;
            if (XC3) {                                                  // quad q / 4 = spiral position; element 3 is the padding
                float* dst = slab + (long)co * p.K + 3 * (q >> 2);
                dst[0] = acc[i][j][0]; dst[1] = acc[i][j][1]; dst[2] = acc[i][j][2];
            } else {
                *reinterpret_cast<f32x4*>(slab + (long)co * p.K + q) = acc[i][j];
            }
        }
    if (qt == 0 && tid < 64 && p0 + tid < p.Cout) p.slab[p.bias_off + (long)split * p.Cout + p0 + tid] = csum;
}


// ------------------------------------------------------------------------------------------------------------------
// LDS-DMA form (bf16 x and dpre, channel counts % 16 == 0, (R * B) % 32 == 0).  The staged kernel above is bound by the
// round trip of one stage's loads (registers hold one stage, a barrier per stage).  Here a WAVE owns a whole output tile
// (QT x 16 weight columns by PT x 16 output channels) over a range of 32-row stages and never synchronises with another
// wave: operand tiles go from memory straight into the wave's private LDS ring by `global_load_lds_dwordx4` (no staging
// registers, no ds_write; per-lane source addresses make the A tile a gather, and lane = 2 row + half lands every 16-byte
// piece exactly where the transposed image [16-column block][32 rows][16] wants it), R - 1 stages in flight behind a
// counted s_waitcnt; fragments come back through ds_read_b64_tr_b16.  The gather-table lines of the wave's vertices are
// copied into LDS first, so the loop contains no ordinary vector load (one would make hipcc drain the DMA queue).
struct BDParams {
    const char* dpre; long dp_rb, dp_bb;
    const char* x; long x_rb, x_bb;
    const int* table;
    float* slab; long slab_stride, bias_off;
    int B, R, S, Cin, Cout, K;
    int n_qg, n_pt, nsplit, stages_per_split, n_items;
    long n_stages;
};
constexpr int BD_TBL_INTS = 256;                      // table lines a wave may hold (planner: vertices per wave * S <= this)

template <int QT, int PT>
constexpr int bd_ring() { return (QT + PT) <= 9 ? 4 : 3; }       // 4 waves x ring <= 160 KiB of LDS
template <int QT, int PT>
constexpr int bd_wave_lds() { return bd_ring<QT, PT>() * (QT + PT) * TG_TR_BLK + BD_TBL_INTS * 4; }

template <int QT, int PT>
__global__ __launch_bounds__(256) void wgrad_bf16_dma_kernel(const BDParams p) {
    constexpr int R = bd_ring<QT, PT>(), NL = QT + PT, SLOT = NL * TG_TR_BLK;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // XCD-contiguous item order: an XCD works through a contiguous range of row chunks (all column groups of a chunk side by
    // side), so the rows it gathers - a 1/8 slice of x plus its spiral neighbourhood - stay in its own 4 MiB L2
    // The four waves of a workgroup take four consecutive row chunks of ONE output tile and add their results in LDS at the
    // end (fixed order), so a workgroup writes one partial slab instead of four.
    char* ring = smem + wave * bd_wave_lds<QT, PT>();
    int* Tl = reinterpret_cast<int*>(ring + R * SLOT);
    int it = sh_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int qg = it % p.n_qg; it /= p.n_qg;
    const int pt = it % p.n_pt; const int sgroup = it / p.n_pt;
    const int split = sgroup * 4 + wave;
    const long st0 = (long)split * p.stages_per_split;
    const int nst = (int)max(0L, min((long)p.stages_per_split, p.n_stages - st0));
    const int S = p.S, B = p.B;
    const int v_first = (int)((st0 * 32) / B);
    if (nst > 0) {   // this wave's table lines -> LDS (ordinary loads, finished before the DMA loop starts)
        const int v_last = (int)(((st0 + nst) * 32 - 1) / B);
        const int n = (v_last - v_first + 1) * S;
        // pre-multiplied by the row stride of x in 16-byte units (the stride is a multiple of 16 bytes; 32 bits reach 64 GB): the
        // gather addresses are one shift and add away from the LDS read instead of behind a 64-bit multiply each
        for (int i = lane; i < n; i += 64) Tl[i] = (int)((unsigned)p.table[(long)v_first * S + i] * (unsigned)(p.x_rb >> 4));
    }
    // per 16-column block: (spiral position, first channel); blocks past K / Cout duplicate block 0 (their products are never stored)
    int s_blk[QT], c_blk[QT], co_blk[PT];
    const int q0 = qg * (16 * QT), p0 = pt * (16 * PT);
#pragma unroll
    for (int k = 0; k < QT; ++k) {
        const int col = q0 + 16 * k < p.K ? q0 + 16 * k : q0;
        s_blk[k] = col / p.Cin; c_blk[k] = col - s_blk[k] * p.Cin;
    }
#pragma unroll
    for (int k = 0; k < PT; ++k) co_blk[k] = p0 + 16 * k < p.Cout ? p0 + 16 * k : p0;
    const int row = lane >> 1, half8 = (lane & 1) * 8;
    typedef __attribute__((address_space(3))) char* lptr_t;
    // LDS-DMA through inline asm: issued through the builtin, hipcc protects every later LDS read with s_waitcnt vmcnt(0)
    // (it cannot see that the read slot is not the slot being filled) and the ring never holds more than one stage in
    // flight.  M0 = wave-uniform LDS byte address of the 1-KiB block, lane l lands at + 16 l (cdna_hip_programming.md 5.7).
    auto dma16 = [](const char* gsrc, unsigned lds_dst) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory", "m0");
    };
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lptr_t)ring);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");        // the table lines are in LDS

    // (vertex, batch entry) of this lane's row in the NEXT stage to issue: advanced by 32 rows per stage, no division in the loop
    int iv, ib;
    {
        const long r_lin = st0 * 32 + row;
        iv = (int)(r_lin / B); ib = (int)(r_lin - (long)iv * B);
    }
    int issued = 0;
    auto issue = [&]() {                                             // next stage (the last one again past the end) -> slot issued % R
        const unsigned slot = ring_lds + (unsigned)((issued % R) * SLOT);
        const int* tl = Tl + (iv - v_first) * S;
        const char* xb = p.x + (long)ib * p.x_bb + 2 * half8;
        const char* db = p.dpre + (long)iv * p.dp_rb + (long)ib * p.dp_bb + 2 * half8;
#pragma unroll
        for (int k = 0; k < QT; ++k) dma16(xb + ((unsigned long)(unsigned)tl[s_blk[k]] << 4) + 2 * c_blk[k], slot + (unsigned)(k * TG_TR_BLK));
#pragma unroll
        for (int k = 0; k < PT; ++k) dma16(db + 2 * co_blk[k], slot + (unsigned)((QT + k) * TG_TR_BLK));
        ++issued;
        if (issued < nst) {                                          // advance by 32 rows (uniform branch; B >= 1)
            ib += 32;
            while (ib >= B) { ib -= B; ++iv; }
        }
    };

    f32x4 acc[QT][PT], accb[PT];
#pragma unroll
    for (int i = 0; i < QT; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < PT; ++j) accb[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16x8 ones = {(__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f};
    const bool want_bias = qg == 0;

    if (nst > 0) {
#pragma unroll
    for (int d = 0; d < R - 1; ++d) issue();
    }
    for (int st = 0; st < nst; ++st) {
        // stages st .. st+R-2 are in flight (clamped repeats past the end keep the count constant): wait for the oldest
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * NL) : "memory");
        issue();                                                     // stage st+R-1 into the slot stage st-1 was read from
        const char* slot = ring + (st % R) * SLOT;
        bf16x8 fb[PT];
#pragma unroll
        for (int j = 0; j < PT; ++j) fb[j] = tg_tr_frag(slot + QT * TG_TR_BLK, j, 0, lane);
#pragma unroll
        for (int i = 0; i < QT; ++i) {
            const bf16x8 fa = tg_tr_frag(slot, i, 0, lane);
#pragma unroll
            for (int j = 0; j < PT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[j], acc[i][j], 0, 0, 0);
        }
        if (want_bias) {                                             // column sums of the dpre tile: ones^T . P
#pragma unroll
            for (int j = 0; j < PT; ++j) accb[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fb[j], accb[j], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // this slot's reads are done before a later DMA may overwrite it
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the clamped tail loads still write this wave's LDS
    __syncthreads();                                                 // every wave is done with its ring: reuse it for the sum
    constexpr int TILE_F = QT * PT * 256, WAVE_F = TILE_F + PT * 16;
    float* red = reinterpret_cast<float*>(smem);                     // [4][tiles as 64 lanes x f32x4 | bias PT x 16]
    {
        float* mine = red + wave * WAVE_F;
#pragma unroll
        for (int i = 0; i < QT; ++i)
#pragma unroll
            for (int j = 0; j < PT; ++j) *reinterpret_cast<f32x4*>(mine + (i * PT + j) * 256 + lane * 4) = acc[i][j];
        if (lane < 16) {
#pragma unroll
            for (int j = 0; j < PT; ++j) mine[TILE_F + j * 16 + lane] = accb[j][0];            // row 0 of ones^T . P
        }
    }
    __syncthreads();
    float* slab = p.slab + (long)sgroup * p.slab_stride;
    for (int t = threadIdx.x; t < QT * PT * 64; t += 256) {
        const int tile = t >> 6, l = t & 63, i = tile / PT, j = tile - i * PT;
        const f32x4 v = ((*reinterpret_cast<const f32x4*>(red + t * 4) + *reinterpret_cast<const f32x4*>(red + WAVE_F + t * 4)) +
                         *reinterpret_cast<const f32x4*>(red + 2 * WAVE_F + t * 4)) + *reinterpret_cast<const f32x4*>(red + 3 * WAVE_F + t * 4);
        const int q = q0 + 16 * i + 4 * (l >> 4), co = p0 + 16 * j + (l & 15);
        if (q < p.K && co < p.Cout) *reinterpret_cast<f32x4*>(slab + (long)co * p.K + q) = v;
    }
    if (want_bias && threadIdx.x < PT * 16) {
        const int t = threadIdx.x, co = p0 + t;
        const float v = ((red[TILE_F + t] + red[WAVE_F + TILE_F + t]) + red[2 * WAVE_F + TILE_F + t]) + red[3 * WAVE_F + TILE_F + t];
        if (co < p.Cout) p.slab[p.bias_off + (long)sgroup * p.Cout + co] = v;
    }
}

struct BWPlan { int dma, qt, pt, n_qg, n_pt, nsplit, nslab, sps; long n_stages; };   // nsplit row chunks (one per wave), nslab = ceil(nsplit / 4) slabs
BWPlan plan_bw(int B, int R, int S, int Cin, int Cout) {
    BWPlan w{};
    static const int dma_on = sh_env_int("SH_BW_DMA", 1, 0, 1);
    const int K = S * Cin;
    const long rows = (long)R * B;
    w.dma = dma_on && Cin % 16 == 0 && Cout % 16 == 0 && rows % 32 == 0;
    if (!w.dma) return w;
    const int pb = sh_cdiv(Cout, 16);
    w.pt = pb >= 4 ? 4 : pb >= 2 ? 2 : 1;
    w.qt = w.pt == 4 ? 4 : w.pt == 2 ? 6 : 8;                            // QT + PT <= 9: a ring of 4 stages per wave fits
    if (K <= 16 * 4) w.qt = 4;
    w.n_qg = sh_cdiv(K, 16 * w.qt);
    w.n_pt = sh_cdiv(Cout, 16 * w.pt);
    w.n_stages = rows / 32;
    // 1024 waves = 256 workgroups = ONE round of the chip (a workgroup's 160 KiB of LDS fills a CU): 2048 left a second round
    // of 240 workgroups behind the first 256 (6890 vertices, batch 64: the five launches 37.5 / 26.0 / 21.2 / 26.1 / 19.6 ->
    // 33.1 / 21.5 / 17.0 / 22.0 / 15.4 us, step 0.965 -> 0.945 ms; 768 and 1280 are both slower)
    static const int wave_target = sh_env_int("SH_BW_WAVES", 1024, 64, 1 << 16);
    static const int slab_mb = sh_env_int("SH_BW_SLAB_MB", 32, 1, 4096);
    const long tiles = (long)w.n_qg * w.n_pt;
    long ns = wave_target / tiles;
    const long cap = 4 * (((long)slab_mb << 20) / ((long)Cout * K * 4));  // four row chunks share a slab
    if (ns > cap) ns = cap;
    if (ns > w.n_stages / 6) ns = w.n_stages / 6;                        // >= 6 stages per wave (the ring holds 3-4)
    if (ns < 1) ns = 1;
    long sps = (w.n_stages + ns - 1) / ns;
    // the wave's table lines must fit its LDS area: vertices per wave <= BD_TBL_INTS / S
    const long max_v = BD_TBL_INTS / S - 2;
    const long max_sps = max_v * B / 32;
    if (max_sps < 1) { w.dma = 0; return w; }
    if (sps > max_sps) sps = max_sps;
    w.sps = (int)sps;
    w.nsplit = (int)((w.n_stages + sps - 1) / sps);
    if ((long)w.nsplit > 4 * cap + 64) { w.dma = 0; return w; }          // a table-bound split would need too many slabs: staged form
    w.nslab = sh_cdiv(w.nsplit, 4);
    return w;
}

template <int QT, int PT>
int launch_bd(BDParams& p, hipStream_t st) {
    auto kern = wgrad_bf16_dma_kernel<QT, PT>;
    const size_t smem = (size_t)4 * bd_wave_lds<QT, PT>();
    static bool attr_set = false;
    if (smem > 65536 && !attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            sh_set_error("wgrad_bf16_dma: cannot raise the dynamic LDS limit to %zu bytes", smem);
            return SH_ERR_LAUNCH;
        }
        attr_set = true;
    }
    const int grid = p.n_items;
    ShProfScope ps(st, "wgrad_bf16_dma_kernel<%d, %d>|R=%d B=%d S=%d Cin=%d N=%d grid=%d split=%d", QT, PT, p.R, p.B, p.S, p.Cin, p.Cout, grid,
                   p.nsplit);
    SH_LAUNCH_PS(ps, kern, dim3(grid), dim3(256), smem, st, p);
    SH_CHECK_LAUNCH("wgrad_bf16_dma");
    return SH_OK;
}

}  // namespace

// shared with the slab reduction in spiral_conv.hip
int sh_wgrad_bf16_nsplit(int B, int R, int S, int Cin, int Cout) {
    {
        const BWPlan w = plan_bw(B, R, S, Cin, Cout);
        if (w.dma) return w.nslab;
    }
    const int Kq = Cin == 3 ? 4 * S : S * Cin, K = S * Cin;
    const long tiles = (long)sh_cdiv(Kq, 64) * sh_cdiv(Cout, 64);
    const long nst = ((long)R * B + 63) >> 6;
    static const int wg_target = sh_env_int("SH_BW_BLOCKS", 2048, 64, 1 << 16);
    static const int slab_mb = sh_env_int("SH_BW_SLAB_MB", 32, 1, 4096);
    long ns = wg_target / tiles;
    const long cap = ((long)slab_mb << 20) / ((long)Cout * K * 4);       // partial slabs are written and re-read once
    if (ns > cap) ns = cap;
    if (ns > nst / 4) ns = nst / 4;                                      // >= 4 stages per split
    if (ns < 1) ns = 1;
    const long sps = (nst + ns - 1) / ns;
    return (int)((nst + sps - 1) / sps);
}

extern "C" {

size_t sh_spiral_conv_bwd_wgt_workspace_bf16(int B, int R, int S, int Cin, int Cout) {
    if (B <= 0 || R <= 0 || S <= 0 || Cin <= 0 || Cout <= 0) return 0;
    return (size_t)sh_wgrad_bf16_nsplit(B, R, S, Cin, Cout) * ((size_t)Cout * S * Cin + Cout) * sizeof(float);
}

int sh_spiral_conv_bwd_wgt_bf16(const void* dpre, int dp_dtype, int64_t dp_sv, int64_t dp_sb, const void* x, int x_dtype, int64_t x_sv,
                                int64_t x_sb, const int32_t* table, void* workspace, size_t workspace_bytes, int B, int R, int S, int Cin,
                                int Cout, sh_stream_t stream) {
    SH_REQUIRE(dpre && x && table && workspace, SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_wgt_bf16: null pointer");
    SH_REQUIRE(B > 0 && R > 0 && S > 0 && Cin > 0 && Cout > 0, SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_wgt_bf16: non-positive size");
    SH_REQUIRE(workspace_bytes >= sh_spiral_conv_bwd_wgt_workspace_bf16(B, R, S, Cin, Cout), SH_ERR_WORKSPACE,
               "sh_spiral_conv_bwd_wgt_bf16: workspace too small");
    const bool xc3 = x_dtype == SH_DTYPE_F32, pc3 = dp_dtype == SH_DTYPE_F32;
    SH_REQUIRE(xc3 ? Cin == 3 : (x_dtype == SH_DTYPE_BF16 && Cin % 8 == 0), SH_ERR_UNSUPPORTED,
               "sh_spiral_conv_bwd_wgt_bf16: input must be bf16 with channels %% 8 == 0 or fp32 with 3 channels (got %d)", Cin);
    SH_REQUIRE(pc3 ? Cout == 3 : (dp_dtype == SH_DTYPE_BF16 && Cout % 8 == 0), SH_ERR_UNSUPPORTED,
               "sh_spiral_conv_bwd_wgt_bf16: gradient must be bf16 with channels %% 8 == 0 or fp32 with 3 channels (got %d)", Cout);
    SH_REQUIRE(!(xc3 && pc3), SH_ERR_UNSUPPORTED, "sh_spiral_conv_bwd_wgt_bf16: 3 -> 3 channel layers are not built in bf16");
    const long xe = xc3 ? 4 : 2, pe = pc3 ? 4 : 2;
    SH_REQUIRE(xc3 || ((reinterpret_cast<uintptr_t>(x) | (uintptr_t)(x_sv * xe) | (uintptr_t)(x_sb * xe)) & 15) == 0, SH_ERR_INVALID_ARG,
               "sh_spiral_conv_bwd_wgt_bf16: x must be 16-byte aligned with 16-byte-multiple strides");
    SH_REQUIRE(pc3 || ((reinterpret_cast<uintptr_t>(dpre) | (uintptr_t)(dp_sv * pe) | (uintptr_t)(dp_sb * pe)) & 15) == 0, SH_ERR_INVALID_ARG,
               "sh_spiral_conv_bwd_wgt_bf16: dpre must be 16-byte aligned with 16-byte-multiple strides");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!xc3 && !pc3) {
        const BWPlan w = plan_bw(B, R, S, Cin, Cout);
        if (w.dma) {
            BDParams d{};
            d.dpre = static_cast<const char*>(dpre); d.dp_rb = dp_sv * 2; d.dp_bb = dp_sb * 2;
            d.x = static_cast<const char*>(x); d.x_rb = x_sv * 2; d.x_bb = x_sb * 2;
            d.table = table; d.slab = static_cast<float*>(workspace);
            d.B = B; d.R = R; d.S = S; d.Cin = Cin; d.Cout = Cout; d.K = S * Cin;
            d.slab_stride = (long)Cout * d.K; d.bias_off = (long)w.nslab * d.slab_stride;
            d.n_qg = w.n_qg; d.n_pt = w.n_pt; d.nsplit = w.nsplit; d.stages_per_split = w.sps; d.n_stages = w.n_stages;
            d.n_items = w.n_qg * w.n_pt * w.nslab;                  // workgroups
            if (w.qt == 4 && w.pt == 4) return launch_bd<4, 4>(d, st);
            if (w.qt == 6 && w.pt == 2) return launch_bd<6, 2>(d, st);
            if (w.qt == 8 && w.pt == 1) return launch_bd<8, 1>(d, st);
            if (w.qt == 4 && w.pt == 2) return launch_bd<4, 2>(d, st);
            return launch_bd<4, 1>(d, st);
        }
    }
    BWParams p{};
    p.dpre = static_cast<const char*>(dpre); p.dp_rb = dp_sv * pe; p.dp_bb = dp_sb * pe;
    p.x = static_cast<const char*>(x); p.x_rb = x_sv * xe; p.x_bb = x_sb * xe;
    p.table = table;
    p.B = B; p.R = R; p.S = S; p.Cin = Cin; p.Cout = Cout; p.K = S * Cin; p.Kq = xc3 ? 4 * S : p.K;
    p.rows = (long)R * B;
    p.n_qt = sh_cdiv(p.Kq, 64); p.n_pt = sh_cdiv(Cout, 64);
    p.nsplit = sh_wgrad_bf16_nsplit(B, R, S, Cin, Cout);
    const long nst = (p.rows + 63) >> 6;
    p.stages_per_split = (int)((nst + p.nsplit - 1) / p.nsplit);
    p.slab = static_cast<float*>(workspace);
    p.slab_stride = (long)Cout * p.K;
    p.bias_off = (long)p.nsplit * p.slab_stride;
    const int grid = p.n_qt * p.n_pt * p.nsplit;
    ShProfScope ps(st, "wgrad_bf16_kernel<%d, %d>|R=%d B=%d S=%d Cin=%d N=%d grid=%d split=%d", (int)xc3, (int)pc3, R, B, S, Cin, Cout, grid, p.nsplit);
    if (xc3) SH_LAUNCH_PS(ps, (wgrad_bf16_kernel<true, false>), dim3(grid), dim3(256), 0, st, p);
    else if (pc3) SH_LAUNCH_PS(ps, (wgrad_bf16_kernel<false, true>), dim3(grid), dim3(256), 0, st, p);
    else SH_LAUNCH_PS(ps, (wgrad_bf16_kernel<false, false>), dim3(grid), dim3(256), 0, st, p);
    SH_CHECK_LAUNCH("wgrad_bf16");
    return SH_OK;
}

}  // extern "C"
