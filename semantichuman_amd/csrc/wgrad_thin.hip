// Weight gradient of a spiral convolution with THREE output channels and 16 input channels (the decoder's last layer,
// reference models.py:146-153: conv[-1] maps filters_dec[-1] = 16 -> 3) in ROLE-SWAPPED form, fp32 and bf16 paths.
//
//   dW[co][s][ci] = sum_{v,b} dpre[v,b,co] * x[table[v,s],b,ci]                    (what the general kernels compute: x, the
//                                                                                  16-channel operand, is gathered S times)
//                 = sum_{u,b} x[u,b,ci] * dpre_ext[table_t[u,s],b,co]             (this kernel: x is read ONCE, the 3-channel
//                                                                                  gradient is gathered through the transposed table)
//
// table_t / dpre_ext are exactly what backward-data uses (mesh_ops.transpose_table_dense: one source row per (u, s), the
// rows of irregular vertices pre-summed into extra rows behind the R real ones, "no source" -> the zero row), so the
// stack sequencer runs the pre-sum launches first and hands the same buffer to both.  Gathered bytes per pass fall from
// S * 64 (fp32 x) or S * 32 (bf16 x) per (vertex, batch entry) to S * 12.
//
// A WAVE streams 32-row items (one vertex u, 32 batch entries): x[u] (1 KiB bf16 / 2 KiB fp32, contiguous) and the S + 1
// gradient chunks dpre_ext[table_t[u,s]] / dpre[u] (384 contiguous bytes each, packed back to back: 2.67 per instruction) go from memory
// straight into the wave's LDS ring by global_load_lds_dwordx4, a counted s_waitcnt keeps RING - 1 items in flight, no
// barrier in the loop.  Per item: out[ci][(s,co)] += x^T . G on the matrix cores (bf16: one 16x16x32 k-step per 16-column
// tile with x through ds_read_b64_tr_b16; fp32: eight 16x16x4 steps), bias sums in fp32 on the VALU.  The waves of a
// workgroup add their tiles in LDS in a fixed order and write ONE slab in the layout / count the general weight-gradient
// kernels use, so the slab reduction launch is unchanged.
#include "sh_bf16.h"
#include "sh_bf16_tiles.h"

int sh_wgrad_bf16_nsplit(int B, int R, int S, int Cin, int Cout);      // bf16_wgrad.hip
int sh_wgrad_f32_nsplit(int B, int R, int S, int Cin, int Cout);       // spiral_conv.hip

namespace {

struct WTParams {
    const char* g; long g_rb;            // dpre_ext fp32 [rows][B][3], a row's B * 12 bytes contiguous
    const char* x; long x_rb;            // x [n_in][B][16], a row contiguous
    const int* tt;                       // table_t [n_in][S]
    float* slab; long slab_stride, bias_off;
    int B, n_in, S, nblk, n_items, ipw;  // nblk = ceil(B / 32) items per vertex; ipw = items per wave
    int half;                            // B % 32 == 16: a vertex's last item holds 16 batch entries (its upper half is masked)
    // optional backward-data of the same layer, from the same staged gradient chunks: dx[u,b,ci] = act'(x[u,b,ci]) * sum_{s,co}
    // dpre_ext[table_t[u,s],b,co] * W[co][s][ci]  (dx null: weight gradient only)
    const float* w;                      // fp32 master weight [3][S * 16]
    char* dx; long dx_rb;                // [n_in (+ extra rows)][B][16] of the path's dtype, rows contiguous
    char* dx_img;                        // fp32 path: three-plane image of dx's rows (csrc/p3_conv.hip, 16-channel layout), or NULL
    // a launch covers spiral positions s0 .. s0 + S - 1 of the S_tot the table / weight / slab have (spirals longer than 10 run as
    // several launches, fp32 path): `first` writes the bias sums and STORES the partial input gradient, later ones add to it,
    // `last` applies the activation derivative / zero row / image
    int S_tot, s0, first, last;
    int act_prev, zero_prev;             // activation whose output x is (identity: no factor), row of dx forced to zero (-1: none)
};

template <bool XB16> constexpr int wt_ring() { return XB16 ? 4 : 3; }     // items in flight per wave + 1
constexpr int WT_NG = 5;                               // DMA instructions for the S + 1 <= 11 chunks of 384 bytes, packed
constexpr int WT_G_BYTES = 4352;                      // 11 chunks = 4224 bytes (the last instruction carries 128 of them), 256-byte multiple
constexpr int WT_TBL_INTS = 256;
template <bool XB16> constexpr int wt_x_bytes() { return XB16 ? 1024 : 2048; }
template <bool XB16> constexpr int wt_stage() { return wt_x_bytes<XB16>() + WT_G_BYTES; }
template <bool XB16> constexpr int wt_wave_lds() { return wt_ring<XB16>() * wt_stage<XB16>() + WT_TBL_INTS * 4; }

template <bool XB16>
__global__ __launch_bounds__(256) void wgrad_thin_kernel(const WTParams p) {
    constexpr int R = wt_ring<XB16>(), XB = wt_x_bytes<XB16>(), STAGE = wt_stage<XB16>(), NL = (XB16 ? 1 : 2) + WT_NG;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    char* ring = smem + wave * wt_wave_lds<XB16>();
    int* Tl = reinterpret_cast<int*>(ring + R * STAGE);
    const int S = p.S, nblk = p.nblk;
    // XCD-contiguous item ranges: an XCD works through a contiguous range of vertices
    const int wid = sh_xcd_remap((int)blockIdx.x, (int)gridDim.x) * nw + wave;
    const int it0 = min(wid * p.ipw, p.n_items), it1 = min(it0 + p.ipw, p.n_items);
    const int nst = it1 - it0;
    const int u_first = it0 / nblk;
    if (nst > 0) {   // this wave's table lines -> LDS (ordinary loads, finished before the DMA loop starts)
        const int u_last = (it1 - 1) / nblk;
        const int n = (u_last - u_first + 1) * S;
        for (int i = lane; i < n; i += 64) {
            const int v = i / S, j = i - v * S;
            Tl[i] = p.tt[(long)(u_first + v) * p.S_tot + p.s0 + j];
        }
    }
    typedef __attribute__((address_space(3))) char* lptr_t;
    auto dma16 = [](const char* gsrc, unsigned lds_dst) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory", "m0");
    };
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lptr_t)ring);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    // lane's 16 bytes of DMA instruction q sit at byte q * 1024 + 16 lane of the packed chunk area: chunk (that / 384), offset
    // (that % 384).  Chunk S is the vertex's own gradient row (bias); lanes past it re-read it into the pad.
    int cq[WT_NG], oq[WT_NG];
#pragma unroll
    for (int q = 0; q < WT_NG; ++q) {
        const int off = q * 1024 + lane * 16;
        const int c = off / 384;
        cq[q] = c < S ? c : S;
        oq[q] = off - c * 384;
    }
    int iu = u_first, ib = it0 - u_first * nblk;                       // (vertex, 32-entry batch block) of the next item to issue
    int issued = 0;
    auto issue = [&]() {
        const unsigned slot = ring_lds + (unsigned)((issued % R) * STAGE);
        // a half item's upper 16 entries lie behind the row: those lanes re-read the lower half (never read past the buffer; the
        // consumer masks them)
        const bool hb = p.half && ib == nblk - 1;
        const char* xs = p.x + (long)iu * p.x_rb + (long)ib * XB + lane * 16 - ((XB16 && hb && lane >= 32) ? 512 : 0);
        dma16(xs, slot);
        if (!XB16) dma16(hb ? xs : xs + 1024, slot + 1024u);
        const int* tl = Tl + (iu - u_first) * S;
        const long goff = (long)ib * 384;
#pragma unroll
        for (int q = 0; q < WT_NG; ++q) {
            const int row = cq[q] < S ? tl[cq[q]] : iu;
            const char* src = p.g + (unsigned long)(unsigned)row * (unsigned long)p.g_rb + goff + oq[q] - ((hb && oq[q] >= 192) ? 192 : 0);
            // lanes whose 16 bytes lie past the S + 1 chunks stay idle; lane 0 of every instruction is kept so that the
            // instruction (and the vmcnt it is counted with) exists for every S
            if (q * 1024 + lane * 16 < (S + 1) * 384 || lane == 0) dma16(src, slot + (unsigned)(XB + q * 1024));
        }
        ++issued;
        if (issued < nst) { if (++ib == nblk) { ib = 0; ++iu; } }
    };

    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    // this lane's output column n' = 16 nt + (lane & 15) -> (s, co) -> byte offset of G[s][.][co] inside the stage's G area
    const int g4 = lane >> 4;
    int goffs[2]; bool gvalid[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = 16 * nt + (lane & 15), s = n / 3, co = n - 3 * s;
        gvalid[nt] = n < 3 * S;
        const int sc = gvalid[nt] ? s : 0;
        goffs[nt] = XB + sc * 384 + co * 4;
    }
    const int self_off = XB + S * 384 + lane * 16;    // lanes < 24: 4 floats of dpre[u][32 b][3]

    // fused backward-data: D[ci][b] = sum_n W[n][ci] * G[b][n], n = 3 s + co; operands swapped so that a lane ends up with
    // four consecutive channels of one batch entry (one 8 / 16-byte store).  bf16: one 16x16x32 step, lane (ci, kq) holds
    // W[8 kq + j][ci]; fp32: eight 16x16x4 steps, lane (ci, k) holds W[4 t + k][ci].
    const bool want_dx = p.dx != nullptr;
    bf16x8 wf16 = {};
    float wf32[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int doff[8];                                                       // byte offset of G[n][.] for this lane's eight n
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int n = XB16 ? 8 * g4 + j : 4 * j + g4;
        const bool ok = n < 3 * S;
        const int s = ok ? n / 3 : 0, co = ok ? n - 3 * s : 0;
        doff[j] = XB + s * 384 + co * 4;                                // n past 3 S: reads column 0 against a zero weight
        const float wv = (want_dx && ok) ? p.w[(long)co * p.S_tot * 16 + (p.s0 + s) * 16 + (lane & 15)] : 0.f;
        if (XB16) wf16[j] = (__bf16)wv; else wf32[j] = wv;
    }
    int cu = u_first, cb = it0 - u_first * nblk;                       // (vertex, batch block) of the item being consumed

    if (nst > 0) {
#pragma unroll
        for (int d = 0; d < R - 1; ++d) issue();
        for (int st = 0; st < nst; ++st) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * NL) : "memory");
            issue();
            const char* slot = ring + (st % R) * STAGE;
            const bool hc = p.half && cb == nblk - 1;                     // half item: entries 16..31 contribute nothing
            if constexpr (XB16) {
                bf16x8 fa = tg_tr_frag(slot, 0, 0, lane);                 // k order: rows {4g..4g+3} u {16+4g..16+4g+3}
                if (hc) { fa[4] = (__bf16)0.f; fa[5] = (__bf16)0.f; fa[6] = (__bf16)0.f; fa[7] = (__bf16)0.f; }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const char* gp = slot + goffs[nt];
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float*>(gp + ((j < 4 ? 4 * g4 + j : 12 + 4 * g4 + j)) * 12);
                    bf16x8 fb;
#pragma unroll
                    for (int j = 0; j < 8; ++j) fb[j] = (__bf16)(gvalid[nt] ? v[j] : 0.f);
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[nt], 0, 0, 0);
                }
            } else {
                const float* xs = reinterpret_cast<const float*>(slot) + (lane & 15);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int b = 4 * j + g4;
                    const float a = (hc && j >= 4) ? 0.f : xs[b * 16];
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const float gv = *reinterpret_cast<const float*>(slot + goffs[nt] + b * 12);
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, gvalid[nt] ? gv : 0.f, acc[nt], 0, 0, 0);
                    }
                }
            }
            if (lane < (hc ? 12 : 24)) bsum += *reinterpret_cast<const f32x4*>(slot + self_off);
            if (want_dx) {
                const bool zrow = cu == p.zero_prev;
#pragma unroll
                for (int h = 0; h < (hc ? 1 : 2); ++h) {
                    const int b = 16 * h + (lane & 15);
                    f32x4 d = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (XB16) {
                        bf16x8 fg;
#pragma unroll
                        for (int j = 0; j < 8; ++j) fg[j] = (__bf16)*reinterpret_cast<const float*>(slot + doff[j] + b * 12);
                        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf16, fg, d, 0, 0, 0);
                        const bf16x4 yv = *reinterpret_cast<const bf16x4*>(slot + b * 32 + g4 * 8);
                        f32x4 o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = zrow ? 0.f : d[r] * sh_act_grad_from_out((float)yv[r], p.act_prev);
                        *reinterpret_cast<bf16x4*>(p.dx + (long)cu * p.dx_rb + (long)(cb * 32 + b) * 32 + g4 * 8) = sh_to_bf16x4(o);
                    } else {
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            const float gv = *reinterpret_cast<const float*>(slot + doff[t] + b * 12);
                            d = __builtin_amdgcn_mfma_f32_16x16x4f32(wf32[t], gv, d, 0, 0, 0);
                        }
                        const f32x4 yv = *reinterpret_cast<const f32x4*>(slot + b * 64 + g4 * 16);
                        f32x4* dxp = reinterpret_cast<f32x4*>(p.dx + (long)cu * p.dx_rb + (long)(cb * 32 + b) * 64 + g4 * 16);
                        if (!p.first) d += *dxp;               // the partial sum of the positions earlier launches covered
                        f32x4 o = d;
                        if (p.last) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) o[r] = zrow ? 0.f : d[r] * sh_act_grad_from_out(yv[r], p.act_prev);
                        }
                        *dxp = o;
                        if (p.dx_img && p.last) {             // the exact split of the values just stored
                            u32x2 ph, pm, pl;
                            sh_split3_quad(o, ph, pm, pl);
                            char* di = p.dx_img + ((long)cu * (p.B >> 4) + cb * 2 + h) * 1536 + ((g4 >> 1) * 16 + (lane & 15)) * 16 + (g4 & 1) * 8;
                            *reinterpret_cast<u32x2*>(di) = ph;
                            *reinterpret_cast<u32x2*>(di + 512) = pm;
                            *reinterpret_cast<u32x2*>(di + 1024) = pl;
                        }
                    }
                }
            }
            if (++cb == nblk) { cb = 0; ++cu; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();                                                     // every wave is done with its ring: reuse it
    float* red = reinterpret_cast<float*>(smem);                         // [nw][512 + 96]
    {
        float* mine = red + wave * 608;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) *reinterpret_cast<f32x4*>(mine + (16 * nt + (lane & 15)) * 16 + 4 * g4) = acc[nt];
        if (lane < 24) *reinterpret_cast<f32x4*>(mine + 512 + 4 * lane) = bsum;
    }
    __syncthreads();
    float* slab = p.slab + (long)blockIdx.x * p.slab_stride;
    for (int t = threadIdx.x; t < 512; t += blockDim.x) {
        const int n = t >> 4, ci = t & 15;
        if (n >= 3 * S) continue;
        float v = 0.f;
        for (int w = 0; w < nw; ++w) v += red[w * 608 + t];
        const int s = n / 3, co = n - 3 * s;
        slab[(long)co * p.S_tot * 16 + (p.s0 + s) * 16 + ci] = v;
    }
    if (threadIdx.x < 3 && p.first) {
        float v = 0.f;
        for (int w = 0; w < nw; ++w)
            for (int e = threadIdx.x; e < 96; e += 3) v += red[w * 608 + 512 + e];
        p.slab[p.bias_off + (long)blockIdx.x * 3 + threadIdx.x] = v;
    }
}

}  // namespace

// waves per workgroup: as many as keep every workgroup of the grid resident (the grid is the slab count of the path's plan)
static int wt_waves(int nslab, bool b16) {
    const int L = b16 ? wt_wave_lds<true>() : wt_wave_lds<false>();
    for (int nw = 4; nw > 1; --nw)
        if ((long)((160 * 1024) / (nw * L)) * 256 >= nslab) return nw;
    return 1;
}

extern "C" {

int sh_spiral_conv_bwd_wgt_thin_ok(int B, int n_in, int S, int Cin, int Cout, int path_dtype) {
    // spirals of 11..30 positions (data-dependent lengths, utils_spiral.py:72-82; BASELINE config 4 forces 18): fp32 path, as two or
    // three launches over equal shares of the positions
    const int smax = path_dtype == SH_DTYPE_F32 ? 30 : 10;
    if (!(Cout == 3 && Cin == 16 && S >= 1 && S <= smax && B > 0 && B % 16 == 0 && n_in > 0)) return 0;
    static const int on = sh_env_int("SH_WGRAD_THIN", 1, 0, 1);
    if (!on) return 0;
    const int nslab = path_dtype == SH_DTYPE_BF16 ? sh_wgrad_bf16_nsplit(B, n_in, S, Cin, Cout) : sh_wgrad_f32_nsplit(B, n_in, S, Cin, Cout);
    const int nw = wt_waves(nslab, path_dtype == SH_DTYPE_BF16);
    const int nblk = (B + 31) / 32, npass = (S + 9) / 10;
    const long items = (long)n_in * nblk;
    const long ipw = (items + (long)nslab * nw - 1) / ((long)nslab * nw);
    const long max_v = WT_TBL_INTS / ((S + npass - 1) / npass) - 2;        // table lines a wave can hold (per launch)
    return ipw / nblk + 2 <= max_v;
}

int sh_spiral_conv_bwd_wgt_thin(const float* dpre_ext, int64_t dp_sv, int64_t dp_sb, const void* x, int x_dtype, int64_t x_sv, int64_t x_sb,
                                const int32_t* table_t, void* workspace, size_t workspace_bytes, const float* weight, void* dx, int64_t dx_sv,
                                int64_t dx_sb, void* dx_planes, int act_prev, int zero_prev, int B, int R, int n_in, int S, int Cin, int Cout,
                                int path_dtype, sh_stream_t stream) {
    SH_REQUIRE(R == n_in, SH_ERR_UNSUPPORTED, "sh_spiral_conv_bwd_wgt_thin: the layer must keep the vertex count (R %d, n_in %d)", R, n_in);
    SH_REQUIRE(dpre_ext && x && table_t && workspace, SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_wgt_thin: null pointer");
    SH_REQUIRE(path_dtype == SH_DTYPE_F32 || path_dtype == SH_DTYPE_BF16, SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_wgt_thin: unknown path dtype");
    SH_REQUIRE(x_dtype == path_dtype, SH_ERR_UNSUPPORTED, "sh_spiral_conv_bwd_wgt_thin: x must have the path's dtype");
    SH_REQUIRE(sh_spiral_conv_bwd_wgt_thin_ok(B, n_in, S, Cin, Cout, path_dtype), SH_ERR_UNSUPPORTED,
               "sh_spiral_conv_bwd_wgt_thin: shape not covered (needs Cout 3, Cin 16, S <= 10 - fp32: 30 -, B %% 16 == 0)");
    const long xe = x_dtype == SH_DTYPE_BF16 ? 2 : 4;
    SH_REQUIRE(dp_sb == 3 && dp_sv == (int64_t)B * 3 && x_sb == 16 && x_sv == (int64_t)B * 16, SH_ERR_UNSUPPORTED,
               "sh_spiral_conv_bwd_wgt_thin: vertex-major contiguous operands required");
    SH_REQUIRE(((reinterpret_cast<uintptr_t>(dpre_ext) | reinterpret_cast<uintptr_t>(x)) & 15) == 0, SH_ERR_INVALID_ARG,
               "sh_spiral_conv_bwd_wgt_thin: operands must be 16-byte aligned");
    const int nslab = path_dtype == SH_DTYPE_BF16 ? sh_wgrad_bf16_nsplit(B, n_in, S, Cin, Cout) : sh_wgrad_f32_nsplit(B, n_in, S, Cin, Cout);
    const size_t need = (size_t)nslab * ((size_t)Cout * S * Cin + Cout) * sizeof(float);
    SH_REQUIRE(workspace_bytes >= need, SH_ERR_WORKSPACE, "sh_spiral_conv_bwd_wgt_thin: workspace too small");
    SH_REQUIRE(!dx || (weight && dx_sb == 16 && dx_sv == (int64_t)B * 16 && (reinterpret_cast<uintptr_t>(dx) & 15) == 0 &&
                       act_prev >= SH_ACT_IDENTITY && act_prev <= SH_ACT_TANH), SH_ERR_INVALID_ARG,
               "sh_spiral_conv_bwd_wgt_thin: the fused input gradient needs the weight and a contiguous vertex-major 16-channel buffer");
    WTParams p{};
    SH_REQUIRE(!dx_planes || (dx && path_dtype == SH_DTYPE_F32 && B % 16 == 0 && (reinterpret_cast<uintptr_t>(dx_planes) & 15) == 0), SH_ERR_UNSUPPORTED,
               "sh_spiral_conv_bwd_wgt_thin: the plane image of dx needs the fp32 path, the fused dx and B %% 16 == 0");
    p.w = weight; p.dx = static_cast<char*>(dx); p.dx_rb = dx_sv * xe; p.act_prev = act_prev; p.zero_prev = zero_prev;
    p.dx_img = static_cast<char*>(dx_planes);
    p.g = reinterpret_cast<const char*>(dpre_ext); p.g_rb = dp_sv * 4;
    p.x = static_cast<const char*>(x); p.x_rb = x_sv * xe;
    p.tt = table_t; p.slab = static_cast<float*>(workspace);
    p.slab_stride = (long)Cout * S * Cin; p.bias_off = (long)nslab * p.slab_stride;
    p.B = B; p.n_in = n_in; p.S = S; p.nblk = (B + 31) / 32; p.half = B % 32 == 16; p.n_items = n_in * p.nblk;
    const int nw = wt_waves(nslab, path_dtype == SH_DTYPE_BF16);
    p.ipw = (int)(((long)p.n_items + (long)nslab * nw - 1) / ((long)nslab * nw));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool b16 = path_dtype == SH_DTYPE_BF16;
    const size_t smem = (size_t)nw * (b16 ? wt_wave_lds<true>() : wt_wave_lds<false>());
    static bool attr_set[2] = {false, false};
    if (smem > 65536 && !attr_set[b16]) {
        const void* k = b16 ? reinterpret_cast<const void*>(wgrad_thin_kernel<true>) : reinterpret_cast<const void*>(wgrad_thin_kernel<false>);
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            sh_set_error("wgrad_thin: cannot raise the dynamic LDS limit to %zu bytes", smem);
            return SH_ERR_LAUNCH;
        }
        attr_set[b16] = true;
    }
    // positions in launches of at most 10 (the kernel's two 16-column output tiles hold 3 x 10 columns)
    const int npass = (S + 9) / 10, s_per = (S + npass - 1) / npass;
    p.S_tot = S;
    for (int k = 0; k < npass; ++k) {
        p.s0 = k * s_per;
        p.S = k == npass - 1 ? S - p.s0 : s_per;
        p.first = k == 0; p.last = k == npass - 1;
        ShProfScope ps(st, "wgrad_thin_kernel<%s>|R=%d B=%d S=%d Cin=%d N=%d grid=%d waves=%d dx=%d pass=%d/%d", b16 ? "true" : "false", n_in, B, S,
                       Cin, Cout, nslab, nw, dx ? 1 : 0, k + 1, npass);
        if (b16) SH_LAUNCH_PS(ps, wgrad_thin_kernel<true>, dim3(nslab), dim3(64 * nw), smem, st, p);
        else SH_LAUNCH_PS(ps, wgrad_thin_kernel<false>, dim3(nslab), dim3(64 * nw), smem, st, p);
        SH_CHECK_LAUNCH("wgrad_thin");
    }
    return SH_OK;
}

}  // extern "C"
