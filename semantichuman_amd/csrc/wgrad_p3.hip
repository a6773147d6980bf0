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
// (global_load_lds_dwordx4, lane l -> LDS base + 16 l) moves one "TR block" = 32 reduction rows (two 16-batch groups of one
// vertex) x 16 channels of one plane = two contiguous 512-byte pieces of HBM/L2 into exactly the [32 r][16 idx] image the
// transposed read wants (lane l = 2 r + half fetches piece (2 cb + half) * 16 + (r & 15) of batch group r >> 4): whole
// 128-byte lines, no staging registers, no ds_write, conflict-free reads.
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
    int n_qg, n_pt, nsplit, nslab, n_items;
    long n_stages;
};

constexpr int WP_BLK = 1024;                   // one TR block: [32 r][16 idx] bf16
constexpr int WP_GRP = 6 * WP_BLK;             // a group: two 16-column halves x three planes
constexpr int WP_G = 5;                        // ring slots per wave
constexpr int WP_TBL_INTS = 2048;              // table lines a wave may hold
constexpr int WP_WAVE_LDS = WP_G * WP_GRP + WP_TBL_INTS * 4;

// transposed fragment of one TR block: lane (i = lane & 15, g = lane >> 4) receives r = {4g..4g+3} u {16+4g..16+4g+3} of idx i
__device__ __forceinline__ bf16x8 wp_frag(const char* blk, int lane) {
    const int g = lane >> 4, l16 = lane & 15, q = l16 >> 2, p = l16 & 3;
    const char* a = blk + (4 * g + q) * 32 + p * 8;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 16 * 32));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return *reinterpret_cast<const bf16x8*>(&v);
}

template <int QF, int PT, bool XC16>
__global__ __launch_bounds__(256) void wgrad_p3_kernel(const WP3Params p) {
    static_assert(PT % 2 == 0, "dpre channels come in 32-channel fragments");
    constexpr int NGD = PT / 2, NG = NGD + QF, G = WP_G;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = sh_wave_id();
    char* ring = smem + wave * WP_WAVE_LDS;
    int* Tl = reinterpret_cast<int*>(ring + G * WP_GRP);
    // XCD-contiguous item order: an XCD works through a contiguous range of row chunks, all column groups of a chunk side by side
    int it = sh_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int qg = it % p.n_qg; it /= p.n_qg;
    const int pt = it % p.n_pt; const int sgroup = it / p.n_pt;
    const int split = sgroup * 4 + wave;
    const long st0 = (long)split * p.n_stages / p.nsplit, st1 = (long)(split + 1) * p.n_stages / p.nsplit;
    const int nst = split < p.nsplit ? (int)(st1 - st0) : 0;
    const int S = p.S, nbp = p.B >> 5;
    const int v_first = (int)(st0 / nbp);
    if (nst > 0) {   // this wave's table lines -> LDS, pre-multiplied by the image's row stride in 16-byte units
        const int v_last = (int)((st0 + nst - 1) / nbp);
        const int n = (v_last - v_first + 1) * S;
        for (int i = lane; i < n; i += 64) Tl[i] = (int)((unsigned)p.table[(long)v_first * S + i] * (unsigned)(p.x_vb >> 4));
    }
    // per gathered group: spiral position(s) and byte offset of its channel group inside a (row, batch group)
    const int f0 = qg * QF;
    int s_of[QF][2], cgo[QF];
#pragma unroll
    for (int k = 0; k < QF; ++k) {
        const int f = f0 + k < p.nxg ? f0 + k : p.nxg - 1;                  // groups past the end: a duplicate, never stored
        if (XC16) {
            s_of[k][0] = 2 * f < S ? 2 * f : S - 1; s_of[k][1] = 2 * f + 1 < S ? 2 * f + 1 : S - 1; cgo[k] = 0;
        } else {
            const int ncg = p.Cin >> 5;
            s_of[k][0] = s_of[k][1] = f / ncg; cgo[k] = (f % ncg) * 3072;
        }
    }
    // lane l = 2 r + half: reduction row r (batch group r >> 4, entry r & 15), 8-channel piece `half` of a 16-channel block
    const int r = lane >> 1, half = lane & 1;
    const long lo_x = (long)(r >> 4) * p.x_bgb + (long)((half * 16 + (r & 15)) * 16);
    const long lo_d = (long)(r >> 4) * p.d_bgb + (long)((half * 16 + (r & 15)) * 16);
    typedef __attribute__((address_space(3))) char* lptr_t;
    auto dma16 = [](const char* gsrc, unsigned lds_dst) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory", "m0");
    };
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lptr_t)ring);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");        // the table lines are in LDS

    // issue side: (local vertex, batch pair) of the stage whose groups are being requested; advanced when a stage's last group went out
    int iss_v = 0, iss_bp = (int)(st0 - (long)v_first * nbp), iss_st = 0, issued = 0;
    auto issue = [&](auto JJ) {
        constexpr int jj = decltype(JJ)::value;
        const unsigned slot = ring_lds + (unsigned)((issued % G) * WP_GRP);
        if constexpr (jj < NGD) {
            const char* base = p.dp + (long)(v_first + iss_v) * p.d_vb + (long)(2 * iss_bp) * p.d_bgb + (long)(pt * NGD + jj) * 3072 + lo_d;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) dma16(base + cb * 512 + pl * 1024, slot + (unsigned)((cb * 3 + pl) * WP_BLK));
        } else {
            constexpr int k = jj - NGD;
            if constexpr (XC16) {
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    const unsigned row = (unsigned)__builtin_amdgcn_readfirstlane(Tl[iss_v * S + s_of[k][hb]]);
                    const char* base = p.xp + ((unsigned long)row << 4) + (long)(2 * iss_bp) * p.x_bgb + lo_x;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) dma16(base + pl * 512, slot + (unsigned)((hb * 3 + pl) * WP_BLK));
                }
            } else {
                const unsigned row = (unsigned)__builtin_amdgcn_readfirstlane(Tl[iss_v * S + s_of[k][0]]);
                const char* base = p.xp + ((unsigned long)row << 4) + (long)(2 * iss_bp) * p.x_bgb + cgo[k] + lo_x;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) dma16(base + cb * 512 + pl * 1024, slot + (unsigned)((cb * 3 + pl) * WP_BLK));
            }
        }
        ++issued;
        if constexpr (jj == NG - 1) {                                       // the stage is out: next one (the last one again past the end)
            if (iss_st + 1 < nst) {
                ++iss_st;
                if (++iss_bp == nbp) { iss_bp = 0; ++iss_v; }
            }
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

    auto consume = [&](auto J, const char* slot) {
        constexpr int j0 = decltype(J)::value;
        if constexpr (j0 < NGD) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) fb[2 * j0 + cb][pl] = wp_frag(slot + (cb * 3 + pl) * WP_BLK, lane);
            if (want_bias) {                                                // column sums of the dpre tile: ones^T . (Dl + Dm + Dh)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    f32x4 c = accb[2 * j0 + cb];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fb[2 * j0 + cb][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fb[2 * j0 + cb][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fb[2 * j0 + cb][0], c, 0, 0, 0);
                    accb[2 * j0 + cb] = c;
                }
            }
        } else {
            constexpr int k = j0 - NGD;
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                const bf16x8 xh = wp_frag(slot + (hb * 3 + 0) * WP_BLK, lane), xm = wp_frag(slot + (hb * 3 + 1) * WP_BLK, lane),
                             xl = wp_frag(slot + (hb * 3 + 2) * WP_BLK, lane);
#pragma unroll
                for (int j = 0; j < PT; ++j) {
                    f32x4 c = acc[k][hb][j];                                // smallest terms first, one dependent chain (the fast form: DESIGN 4e)
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, fb[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, fb[j][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xm, fb[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xm, fb[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, fb[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, fb[j][0], c, 0, 0, 0);
                    acc[k][hb][j] = c;
                }
            }
        }
    };

    // flat group sequence t = stage * NG + j; group t lives in slot t % G; groups t+1 .. t+G-2 are in flight while t is read
    auto prologue = [&](auto self, auto D) -> void {
        constexpr int d = decltype(D)::value;
        if constexpr (d < G - 1) {
            issue(std::integral_constant<int, d % NG>{});
            self(self, std::integral_constant<int, d + 1>{});
        }
    };
    if (nst > 0) prologue(prologue, std::integral_constant<int, 0>{});
    int t = 0;
    auto stage_groups = [&](auto self, auto J) -> void {
        constexpr int j = decltype(J)::value;
        if constexpr (j < NG) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((G - 2) * 6) : "memory");
            issue(std::integral_constant<int, (j + G - 1) % NG>{});          // group t + G - 1 into the slot group t - 1 was read from
            consume(J, ring + (t % G) * WP_GRP);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // this slot's reads are done before a later DMA may overwrite it
            ++t;
            self(self, std::integral_constant<int, j + 1>{});
        }
    };
    for (int st = 0; st < nst; ++st) stage_groups(stage_groups, std::integral_constant<int, 0>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // the clamped tail loads still write this wave's LDS
    __syncthreads();                                                         // every wave is done with its ring: reuse it for the sum

    // four waves -> one slab: passes of up to 16 tiles per wave (64 KiB for the four)
    constexpr int NT = QF * 2 * PT, TPP = 16, NPASS = (NT + TPP - 1) / TPP;
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
                    const int tile = (k * 2 + hb) * PT + j;
                    if (tile / TPP == ps) *reinterpret_cast<f32x4*>(mine + (tile % TPP) * 256 + lane * 4) = acc[k][hb][j];
                }
        __syncthreads();
        const int ntile = NT - ps * TPP < TPP ? NT - ps * TPP : TPP;
        for (int e = threadIdx.x; e < ntile * 64; e += 256) {
            const int tile = ps * TPP + (e >> 6), l = e & 63;
            const int k = tile / (2 * PT), hb = (tile / PT) & 1, j = tile % PT;
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

struct WP3Plan { int ok, qf, pt, xc16, nxg, n_qg, n_pt, nslab, nsplit; long n_stages; };
WP3Plan plan_wp3(int B, int R, int S, int Cin, int Cout) {
    WP3Plan w{};
    if (B <= 0 || B % 32 != 0 || R <= 0 || S <= 0) return w;
    if (!(Cin == 16 || Cin % 32 == 0) || Cout % 32 != 0) return w;
    w.xc16 = Cin == 16;
    w.nxg = w.xc16 ? (S + 1) / 2 : S * (Cin / 32);
    w.pt = Cout % 64 == 0 ? 4 : 2;
    static const int qf_env = sh_env_int("SH_WP3_QF", 4, 2, 4);
    w.qf = qf_env == 3 ? 4 : qf_env;
    if (w.nxg <= 2) w.qf = 2;
    w.n_qg = sh_cdiv(w.nxg, w.qf);
    w.n_pt = Cout / (16 * w.pt);
    w.n_stages = (long)R * (B / 32);
    // one workgroup (~152 KiB of LDS) per CU: tiles x slabs <= 256
    static const int wg_target = sh_env_int("SH_WP3_BLOCKS", 256, 8, 1 << 16);
    static const int slab_mb = sh_env_int("SH_WP3_SLAB_MB", 32, 1, 4096);
    const long tiles = (long)w.n_qg * w.n_pt;
    long ns = wg_target / tiles;
    const long cap = ((long)slab_mb << 20) / ((long)Cout * S * Cin * 4);
    if (ns > cap) ns = cap;
    if (ns > w.n_stages / (4 * 4)) ns = w.n_stages / (4 * 4);               // >= 4 stages per wave
    if (ns < 1) ns = 1;
    // a wave's table lines must fit its LDS area
    const int nbp = B / 32;
    for (;; ++ns) {
        const long sps = (w.n_stages + 4 * ns - 1) / (4 * ns);
        if (((sps + nbp - 1) / nbp + 2) * S <= WP_TBL_INTS) break;
        if (ns > w.n_stages) return w;
    }
    w.nslab = (int)ns;
    w.nsplit = (int)(4 * ns < w.n_stages ? 4 * ns : w.n_stages);
    w.nslab = sh_cdiv(w.nsplit, 4);
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
    ShProfScope ps(st, "wgrad_p3_kernel<%d, %d, %s>|R=%d B=%d K=%d N=%d grid=%d split=%d", QF, PT, XC16 ? "true" : "false", p.R, p.B, p.K, p.Cout,
                   p.n_items, p.nsplit);
    SH_LAUNCH_PS(ps, kern, dim3(p.n_items), dim3(256), smem, st, p);
    SH_CHECK_LAUNCH("wgrad_p3");
    return SH_OK;
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

int sh_spiral_conv_bwd_wgt_p3(const void* dpre_planes, const void* x_planes, const int32_t* table, void* workspace, size_t workspace_bytes,
                              int B, int R, int S, int Cin, int Cout, sh_stream_t stream) {
    SH_REQUIRE(dpre_planes && x_planes && table && workspace, SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_wgt_p3: null pointer");
    const WP3Plan w = plan_wp3(B, R, S, Cin, Cout);
    SH_REQUIRE(w.ok, SH_ERR_UNSUPPORTED,
               "sh_spiral_conv_bwd_wgt_p3: B=%d S=%d Cin=%d Cout=%d is not taken (batch %% 32 == 0; Cin 16 or %% 32 == 0; Cout %% 32 == 0)", B, S,
               Cin, Cout);
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
    p.nxg = w.nxg; p.n_qg = w.n_qg; p.n_pt = w.n_pt; p.nsplit = w.nsplit; p.nslab = w.nslab; p.n_stages = w.n_stages;
    p.n_items = w.n_qg * w.n_pt * w.nslab;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (w.xc16) return w.pt == 4 ? (w.qf == 4 ? launch_wp3<4, 4, true>(p, st) : launch_wp3<2, 4, true>(p, st))
                                 : (w.qf == 4 ? launch_wp3<4, 2, true>(p, st) : launch_wp3<2, 2, true>(p, st));
    return w.pt == 4 ? (w.qf == 4 ? launch_wp3<4, 4, false>(p, st) : launch_wp3<2, 4, false>(p, st))
                     : (w.qf == 4 ? launch_wp3<4, 2, false>(p, st) : launch_wp3<2, 2, false>(p, st));
}

}  // extern "C"
