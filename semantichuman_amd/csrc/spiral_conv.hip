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
#include "sh_bf16.h"

#include <type_traits>

int sh_wgrad_bf16_nsplit(int B, int R, int S, int Cin, int Cout);      // csrc/bf16_wgrad.hip
int sh_wgrad_p3_nslab(int B, int R, int S, int Cin, int Cout);         // csrc/wgrad_p3.hip

namespace {

constexpr int TM = 128;
constexpr int KC = 32;
constexpr int NTHREADS = 256;

struct GGParams {
    const float* x; long x_sv, x_sb;
    const int* table;   // [R][S] rows of x to gather
    const float* w;     // [Nout][K] row-major
    const float* bias;  // [Nout] or null
    float* y; long y_sv, y_sb;
    const float* yprev; long yp_sv, yp_sb;
    int B, R, S, Cg, Nout, K;
    int Kw;             // row stride of w (== K except in the 3-channel mode, where K counts padded quads)
    int act;            // forward: activation of this layer; backward: activation that produced x
    int zero_row;
    int log2TB, n_btiles, n_vtiles, nchunks;
    int nsplit;         // workgroups per row tile (each owns NT*16 output channels)
    int vec_out;        // Nout % 4 == 0 and output strides 16-B aligned
    // backward-data: row of x (= dpre) that is known to be all zero and that "no source" entries of the transposed table
    // point at (mesh_ops.transpose_table_dense: 51-59 % of the entries of the down-sampling levels, 22-30 % of the
    // others).  skip_on: a 16-row MFMA tile is one vertex (batch slice of 16) and a half-chunk lies inside one spiral
    // position (gathered channels % 16 == 0), so "this tile's entry is a no-source entry" is wave-uniform and the tile's
    // MFMAs of that half-chunk are skipped - they would add exact zeros.
    unsigned skip_off;  // zero row * x_sv (element offset, like the entries of the LDS table tile)
    int skip_on;
    // conv_out3_linewise_kernel over a RANGE of spiral positions (spirals longer than its register budget run as two passes):
    // positions s0 .. s0 + <template S> - 1 of the p.S the table and the weight have; pass 0 = the only one, 1 = first of two
    // (stores the raw partial sums), 2 = second (adds them, then bias / activation / mask)
    int s0, pass;
    // three-plane image of the output rows (csrc/p3_conv.hip), written by the STAGED kernel's epilogue beside y; or NULL
    char* y_img; long img_vb, img_bgb;
};

// 3-channel rows (the xyz input of the first encoder layer, the xyz gradient entering the last decoder layer):
// a 12-byte row is loaded as one 4-byte-aligned dwordx3 and treated as a 16-byte quad whose 4th channel is zero,
// so these layers run the vector path over K' = 4 S columns instead of the scalar one over 3 S.
struct __attribute__((packed, aligned(4))) sh_f3 { float a, b, c; };
__device__ __forceinline__ f32x4 sh_ld3(const float* p) {
    const sh_f3 v = *reinterpret_cast<const sh_f3*>(p);
    return (f32x4){v.a, v.b, v.c, 0.f};
}

template <int NT, bool VEC4, bool BWD_EPI, bool TB16, bool C3 = false>
__global__ __launch_bounds__(NTHREADS) void gather_gemm_kernel(const GGParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);            // [2][TM][KC]
    float* Ws = As + 2 * TM * KC;                          // [2][NT*16][KC]
    int* Ts = reinterpret_cast<int*>(Ws + 2 * NT * 16 * KC);   // table / list-pointer tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int TB = 1 << p.log2TB, TV = TM >> p.log2TB;
    // work item = (row tile, output-channel split).  Layers with few row tiles (coarse mesh levels)
    // split their output channels over `nsplit` workgroups so that the launch still fills the chip;
    // the splits of one tile are adjacent in the XCD-contiguous order (they gather the same rows).
    const int item = sh_xcd_remap(blockIdx.x, gridDim.x);
    const int tile = item / p.nsplit;
    const int n_base = (item - tile * p.nsplit) * (NT * 16);
    const int bt = tile / p.n_vtiles, vt = tile - bt * p.n_vtiles;     // batch-slice-major order
    const int v0 = vt * TV, b0 = bt * TB;
    const int S = p.S;

    {   // table tile: rows v0 .. v0+TV-1 are contiguous in the table
        const int nT = TV * S;
        const long lim = (long)p.R * S;
        for (int i = tid; i < nT; i += NTHREADS) {
            const long g = (long)v0 * S + i;
            Ts[i] = g < lim ? (int)(unsigned)((long)p.table[g] * p.x_sv) : 0;     // element offset of the gathered row (< 2^32)
        }
    }
    __syncthreads();

    // ---- staging assignment.
    // All hot-path loads are BRANCH-FREE: out-of-range rows / columns read a valid dummy address,
    // so the compiler can keep two chunks of loads in flight behind counted s_waitcnt vmcnt(N) (a
    // load under a branch would force vmcnt(0)).  Garbage in rows/channels outside the valid range
    // only feeds outputs the epilogue never stores; only K columns past the end of K need zeroing,
    // and that is done on the weight operand when the chunk is written to LDS.
    //
    // FAST path (TB16: batch slice of 16, vector loads): thread -> (vertex vl = tid>>5, batch rows
    // bl0 + 4j, quad q).  One gather-table lookup and one base address per thread and chunk; the four
    // loads differ only by a per-thread constant batch offset.  The generic path (any slice width,
    // scalar channels) maps thread -> quad q of rows rbase + 32*i with a lookup per row.
    constexpr bool FAST = TB16 && VEC4;
    const int q = tid & 7, rbase = tid >> 3;
    int a_ts[4];        // offset of the row's table line
    long a_boff[4];     // batch offset (elements)
    bool a_ok[4];
    int a_lds[4];       // LDS float offset of this thread's quad in row i (swizzled)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = FAST ? ((tid >> 5) << 4) + ((tid & 31) >> 3) + 4 * i : rbase + 32 * i;
        const int vl = row >> p.log2TB, bl = row & (TB - 1);
        a_ok[i] = (v0 + vl) < p.R && (b0 + bl) < p.B;
        a_ts[i] = vl * S;
        a_boff[i] = (b0 + bl) < p.B ? (long)(b0 + bl) * p.x_sb : 0;
        a_lds[i] = row * KC + ((q ^ (row & 7)) << 2);
    }
    constexpr int WQ = NT >= 2 ? NT / 2 : 1;
    const bool w_thread = (NT >= 2) || tid < 128;
    long w_off[WQ];
    bool w_ok[WQ];
#pragma unroll
    for (int i = 0; i < WQ; ++i) {
        const int n = n_base + rbase + 32 * i;
        w_ok[i] = n < p.Nout;
        w_off[i] = w_ok[i] ? (long)n * p.Kw : 0;
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // running (k, s, channel) of this thread's quad for the NEXT chunk to be loaded: chunks are
    // loaded strictly in order, so the split of k into (spiral position, channel) advances by
    // additions instead of a division per chunk
    int k_next = 4 * q, s_next = k_next / p.Cg, ch_next = k_next - s_next * p.Cg;
    const int adv_s = KC / p.Cg, adv_c = KC - adv_s * p.Cg;

    auto load_chunk = [&](f32x4 (&ra)[4], f32x4 (&rw)[WQ], unsigned& mask) {
        const int k = k_next;
        const bool kok = k < p.K;                           // also false for prefetches past the last chunk
        mask = kok ? 1u : 0u;
        if (VEC4) {
            const int s = kok ? s_next : 0, ch = kok ? ch_next : 0, kc = kok ? k : 0;
            if (FAST) {
                const float* src = p.x + (unsigned)Ts[a_ts[0] + s] + ch;      // Ts holds row * x_sv (element offsets)
#pragma unroll
                for (int i = 0; i < 4; ++i) ra[i] = C3 ? sh_ld3(src + a_boff[i]) : *reinterpret_cast<const f32x4*>(src + a_boff[i]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float* src = p.x + (unsigned)Ts[a_ts[i] + s] + a_boff[i] + ch;
                    ra[i] = C3 ? sh_ld3(src) : *reinterpret_cast<const f32x4*>(src);
                }
            }
#pragma unroll
            for (int i = 0; i < WQ; ++i)
                rw[i] = C3 ? sh_ld3(p.w + w_off[i] + 3 * s) : *reinterpret_cast<const f32x4*>(p.w + w_off[i] + kc);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = zero4;
#pragma unroll
            for (int i = 0; i < WQ; ++i) rw[i] = zero4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kk = k + j;
                if (kk < p.K) {
                    const int s = kk / p.Cg, ch = kk - s * p.Cg;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (a_ok[i]) ra[i][j] = p.x[(long)(unsigned)Ts[a_ts[i] + s] + a_boff[i] + ch];
#pragma unroll
                    for (int i = 0; i < WQ; ++i)
                        if (w_ok[i]) rw[i][j] = p.w[w_off[i] + kk];
                }
            }
            mask = 1u;
        }
        // branch-free advance by KC columns (a loop here would make hipcc drain vmcnt)
        k_next += KC;
        ch_next += adv_c;
        s_next += adv_s;
        const bool wrap = ch_next >= p.Cg;
        ch_next = wrap ? ch_next - p.Cg : ch_next;
        s_next = wrap ? s_next + 1 : s_next;
    };
    auto store_chunk = [&](int buf, const f32x4 (&ra)[4], const f32x4 (&rw)[WQ], unsigned mask) {
        float* Ab = As + buf * TM * KC;
        float* Wb = Ws + buf * NT * 16 * KC;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(Ab + a_lds[i]) = ra[i];
        if (w_thread) {
            const int pq = (q ^ (rbase & 7)) << 2;     // (rbase + 32 i) & 7 == rbase & 7
#pragma unroll
            for (int i = 0; i < WQ; ++i)
                *reinterpret_cast<f32x4*>(Wb + (rbase + 32 * i) * KC + pq) = mask ? rw[i] : zero4;
        }
    };

    f32x4 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = zero4;

    const int lrow = lane & 15, lq = lane >> 4;
    // backward-data: skip the MFMAs of a (tile, half-chunk) whose table entry is a no-source entry (GGParams::skip_on; with a
    // batch slice of 16 the wave's two tiles are the vertices 2 wave and 2 wave + 1 of the workgroup's eight)
    const bool skip_on = BWD_EPI && TB16 && VEC4 && !C3 && p.skip_on;
    int sc_k[2] = {0, 16}, sc_s[2], sc_c[2];                 // running (k, position, channel) of the half-chunks being multiplied
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) { sc_s[ks] = sc_k[ks] / p.Cg; sc_c[ks] = sc_k[ks] - sc_s[ks] * p.Cg; }
    auto compute = [&](int buf) {
        const float* Ab = As + buf * TM * KC + (32 * wave + lrow) * KC;
        const float* Wb = Ws + buf * NT * 16 * KC + lrow * KC;
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
            const int pq = ((lq + 4 * ks) ^ (lane & 7)) << 2;
            bool l0 = true, l1 = true;
            if (BWD_EPI && skip_on) {
                const int sp = sc_k[ks] < p.K ? sc_s[ks] : 0;
                l0 = (unsigned)__builtin_amdgcn_readfirstlane(Ts[(2 * wave) * S + sp]) != p.skip_off;
                l1 = (unsigned)__builtin_amdgcn_readfirstlane(Ts[(2 * wave + 1) * S + sp]) != p.skip_off;
                sc_k[ks] += KC; sc_c[ks] += adv_c; sc_s[ks] += adv_s;
                const bool wrap = sc_c[ks] >= p.Cg;
                sc_c[ks] = wrap ? sc_c[ks] - p.Cg : sc_c[ks];
                sc_s[ks] = wrap ? sc_s[ks] + 1 : sc_s[ks];
                if (!(l0 || l1)) continue;
            }
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(Ab + pq);
            const f32x4 g1 = *reinterpret_cast<const f32x4*>(Ab + 16 * KC + pq);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const f32x4 wq = *reinterpret_cast<const f32x4*>(Wb + n * 16 * KC + pq);
                if (!(BWD_EPI && skip_on) || (l0 && l1)) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[t], g0[t], acc[0][n], 0, 0, 0);
                        acc[1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[t], g1[t], acc[1][n], 0, 0, 0);
                    }
                } else if (l0) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[t], g0[t], acc[0][n], 0, 0, 0);
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[t], g1[t], acc[1][n], 0, 0, 0);
                }
            }
        }
    };

    // ---- main loop.  LDS holds chunk c (buffer c&1); register set A/B hold the loads of chunks
    // c+1 and c+2, so every load has two MFMA phases + a barrier to land (the kernel is otherwise
    // bound by loaded-memory latency: 3 workgroups/CU x 20 KB in flight / ~2 us).  Unrolled by two so
    // that both register sets are statically named.
    f32x4 raA[4], raB[4], rwA[WQ], rwB[WQ];
    unsigned mA, mB;
    load_chunk(raA, rwA, mA);                // chunk 0
    store_chunk(0, raA, rwA, mA);
    __syncthreads();
    load_chunk(raA, rwA, mA);                // chunk 1
    for (int c = 0; c < p.nchunks; c += 2) {
        load_chunk(raB, rwB, mB);            // chunk c+2
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch loads AHEAD of the MFMA phase
        compute(0);
        store_chunk(1, raA, rwA, mA);        // chunk c+1 (a clamped duplicate past the end is never read)
        __syncthreads();
        if (c + 1 >= p.nchunks) break;
        load_chunk(raA, rwA, mA);            // chunk c+3
        __builtin_amdgcn_sched_barrier(0);
        compute(1);
        store_chunk(0, raB, rwB, mB);        // chunk c+2
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
            const int n0 = n_base + n * 16 + lq * 4;
            if (n0 >= p.Nout) continue;
            f32x4 a = acc[m][n];
            if (p.vec_out) {
                if (!BWD_EPI) {
                    if (p.bias) a += *reinterpret_cast<const f32x4*>(p.bias + n0);
                    a = sh_act_fwd4(a, p.act);
                } else if (yp) {
                    const f32x4 yv = *reinterpret_cast<const f32x4*>(yp + n0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[j] *= sh_act_grad_from_out(yv[j], p.act);
                }
                if (zero) a = (f32x4){0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(yrow + n0) = a;
                if (p.y_img) {                                     // the exact split of the values just stored
                    u32x2 ph, pm, pl;
                    sh_split3_quad(a, ph, pm, pl);
                    const bool c16 = p.Nout == 16;
                    char* d = p.y_img + (long)v * p.img_vb + (long)(b >> 4) * p.img_bgb +
                              (c16 ? ((n0 >> 3) * 16 + (b & 15)) * 16 : (n0 >> 5) * 3072 + (((n0 & 31) >> 3) * 16 + (b & 15)) * 16) + ((n0 >> 2) & 1) * 8;
                    const int pb = c16 ? 512 : 1024;
                    *reinterpret_cast<u32x2*>(d) = ph;
                    *reinterpret_cast<u32x2*>(d + pb) = pm;
                    *reinterpret_cast<u32x2*>(d + 2 * pb) = pl;
                }
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

// ------------------------------------------------------------------------------------------
// Direct form of the same kernel (Cg % 4 == 0): the gathered operand never touches LDS.  A wave's
// 32 tile rows are private to it, so staging them through LDS only re-laid them out for the MFMA;
// instead lane (r = lane & 15, kq = lane >> 4) loads the 16-byte quad x[row r][k0 + 16 ks + 4 kq ..+3]
// it will feed to the matrix pipe itself (element t of the quad is the B operand of MFMA t, the
// weight quad W[n][same k] from LDS the A operand).  Only the weight chunk, shared by the four
// waves, goes through LDS (triple-buffered, one barrier per chunk); gathered chunks c+1 and c+2
// are in flight in registers while chunk c multiplies.  Same tiling, K order and epilogue as
// gather_gemm_kernel, hence bit-identical results.
// SKIP (backward-data, RT == 1): skip the matrix products of no-source table entries (GGParams::skip_on)
template <int NT, bool BWD_EPI, int RT, bool SKIP = false>      // RT = 16-row tiles per wave: the workgroup covers 64*RT rows
__global__ __launch_bounds__(NTHREADS) void gather_gemm_direct_kernel(const GGParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Ws = reinterpret_cast<float*>(smem);                 // [3][NT*16][KC]
    int* Ts = reinterpret_cast<int*>(Ws + 3 * NT * 16 * KC);    // table tile (element offsets)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int TB = 1 << p.log2TB, TV = (64 * RT) >> p.log2TB;
    const int item = sh_xcd_remap(blockIdx.x, gridDim.x);
    const int tile = item / p.nsplit;
    const int n_base = (item - tile * p.nsplit) * (NT * 16);
    const int bt = tile / p.n_vtiles, vt = tile - bt * p.n_vtiles;
    const int v0 = vt * TV, b0 = bt * TB;
    const int S = p.S;

    const int lrow = lane & 15, lq = lane >> 4;
    int a_ts[RT];
    long a_boff[RT];
#pragma unroll
    for (int m = 0; m < RT; ++m) {
        const int row = 16 * RT * wave + 16 * m + lrow;
        const int vl = row >> p.log2TB, bl = row & (TB - 1);
        a_ts[m] = vl * S;
        a_boff[m] = (b0 + bl) < p.B ? (long)(b0 + bl) * p.x_sb : 0;      // rows past B read row 0 of the slice; never stored
    }
    // running (k, s, channel) of this lane's quads (ks = 0, 1) for the next chunk to load
    int k_n[2], s_n[2], c_n[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        k_n[ks] = 16 * ks + 4 * lq;
        s_n[ks] = k_n[ks] / p.Cg;
        c_n[ks] = k_n[ks] - s_n[ks] * p.Cg;
    }
    const int adv_s = KC / p.Cg, adv_c = KC - adv_s * p.Cg;
    auto load_a = [&](f32x4 (&ra)[RT][2], unsigned& live) {
        unsigned toff[RT][2];
        int ch[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {                        // all table lookups first: one LDS round trip per chunk
            const bool kok = k_n[ks] < p.K;                     // K tail / prefetch past the end: any valid address
            const int s = kok ? s_n[ks] : 0;
            ch[ks] = kok ? c_n[ks] : 0;
#pragma unroll
            for (int m = 0; m < RT; ++m) toff[m][ks] = (unsigned)Ts[a_ts[m] + s];
        }
        live = ~0u;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int m = 0; m < RT; ++m) {
                const float* src = p.x + toff[m][ks] + a_boff[m] + ch[ks];
                if (SKIP) {
                    // a no-source entry (wave-uniform, see GGParams::skip_on): every lane reads the SAME 16 bytes of the zero
                    // row - one L1 access instead of 64 - and compute() skips the tile's MFMAs of this half-chunk
                    const bool dead = (unsigned)__builtin_amdgcn_readfirstlane((int)toff[m][ks]) == p.skip_off;
                    if (dead) live &= ~(1u << (2 * m + ks));
                    src = dead ? p.x + p.skip_off : src;
                }
                ra[m][ks] = *reinterpret_cast<const f32x4*>(src);
            }
            k_n[ks] += KC;
            c_n[ks] += adv_c;
            s_n[ks] += adv_s;
            const bool wrap = c_n[ks] >= p.Cg;
            c_n[ks] = wrap ? c_n[ks] - p.Cg : c_n[ks];
            s_n[ks] = wrap ? s_n[ks] + 1 : s_n[ks];
        }
    };

    // weight chunk staging: thread -> quad q of rows rbase + 32 i (as in gather_gemm_kernel)
    const int q = tid & 7, rbase = tid >> 3;
    constexpr int WQ = NT >= 2 ? NT / 2 : 1;
    const bool w_thread = (NT >= 2) || tid < 128;
    long w_off[WQ];
#pragma unroll
    for (int i = 0; i < WQ; ++i) {
        const int n = n_base + rbase + 32 * i;
        w_off[i] = n < p.Nout ? (long)n * p.Kw : 0;
    }
    int kw_n = 4 * q;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto load_w = [&](f32x4 (&rw)[WQ], unsigned& mask) {
        const bool kok = kw_n < p.K;
        mask = kok ? 1u : 0u;
        const int kc = kok ? kw_n : 0;
#pragma unroll
        for (int i = 0; i < WQ; ++i) rw[i] = *reinterpret_cast<const f32x4*>(p.w + w_off[i] + kc);
        kw_n += KC;
    };
    auto store_w = [&](int buf, const f32x4 (&rw)[WQ], unsigned mask) {
        if (w_thread) {
            float* Wb = Ws + buf * NT * 16 * KC;
            const int pq = (q ^ (rbase & 7)) << 2;
#pragma unroll
            for (int i = 0; i < WQ; ++i) *reinterpret_cast<f32x4*>(Wb + (rbase + 32 * i) * KC + pq) = mask ? rw[i] : zero4;
        }
    };

    f32x4 acc[RT][NT];
#pragma unroll
    for (int m = 0; m < RT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = zero4;
    auto compute = [&](int buf, const f32x4 (&ra)[RT][2], unsigned live) {
        const float* Wb = Ws + buf * NT * 16 * KC + lrow * KC;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int pq = ((lq + 4 * ks) ^ (lane & 7)) << 2;
            unsigned lv = ~0u;                                   // bit m: tile m has a source in this half-chunk
            if (SKIP) {
                lv = 0;
#pragma unroll
                for (int m = 0; m < RT; ++m) lv |= ((live >> (2 * m + ks)) & 1u) << m;
                if (lv == 0) continue;                           // scalar branch: nothing to add
            }
            // all weight quads of the step first, then t outermost: consecutive MFMAs go to DIFFERENT accumulators (a dependent
            // v_mfma_f32_16x16x4_f32 issues 40 cycles after its predecessor, an independent one after 32)
            f32x4 wq[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) wq[n] = *reinterpret_cast<const f32x4*>(Wb + n * 16 * KC + pq);
            if (!SKIP || lv == (1u << RT) - 1) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int m = 0; m < RT; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[n][t], ra[m][ks][t], acc[m][n], 0, 0, 0);
            } else {
#pragma unroll
                for (int m = 0; m < RT; ++m) {
                    if (!((lv >> m) & 1u)) continue;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[n][t], ra[m][ks][t], acc[m][n], 0, 0, 0);
                }
            }
        }
    };

    f32x4 ra[3][RT][2], rw[3][WQ];
    unsigned mw[3], lva[3];
    load_w(rw[0], mw[0]);                                // chunk 0 of the weight does not need the table: in flight under its load
    {
        const int nT = TV * S;
        const long lim = (long)p.R * S;
        for (int i = tid; i < nT; i += NTHREADS) {
            const long g = (long)v0 * S + i;
            Ts[i] = g < lim ? (int)(unsigned)((long)p.table[g] * p.x_sv) : 0;
        }
    }
    __syncthreads();
    load_a(ra[0], lva[0]);                               // chunk 0
    store_w(0, rw[0], mw[0]);
    __syncthreads();
    load_a(ra[1], lva[1]); load_w(rw[1], mw[1]);         // chunk 1
    // one chunk: J = c mod 3 names the register sets / LDS buffers statically
    auto step = [&](auto J) {
        constexpr int j = decltype(J)::value;
        load_a(ra[(j + 2) % 3], lva[(j + 2) % 3]); load_w(rw[(j + 2) % 3], mw[(j + 2) % 3]);               // chunk c+2
        __builtin_amdgcn_sched_barrier(0);               // keep the prefetch loads AHEAD of the MFMA phase
        compute(j, ra[j], lva[j]);
        store_w((j + 1) % 3, rw[(j + 1) % 3], mw[(j + 1) % 3]);                           // chunk c+1
        __syncthreads();
    };
    // explicit early exits (not three independent `if`s): the compiler must see that a skipped step ends
    // the loop, or it drains every outstanding load (vmcnt(0)) where the paths merge
    for (int c = 0; c < p.nchunks; c += 3) {
        step(std::integral_constant<int, 0>{});
        if (c + 1 >= p.nchunks) break;
        step(std::integral_constant<int, 1>{});
        if (c + 2 >= p.nchunks) break;
        step(std::integral_constant<int, 2>{});
    }

    // ---- epilogue (identical to gather_gemm_kernel): lane holds channels n0..n0+3 of tile row 16*RT*wave + 16*m + lrow
#pragma unroll
    for (int m = 0; m < RT; ++m) {
        const int row = 16 * RT * wave + 16 * m + lrow;
        const int vl = row >> p.log2TB, bl = row & (TB - 1);
        const int v = v0 + vl, b = b0 + bl;
        if (v >= p.R || b >= p.B) continue;
        float* yrow = p.y + (long)v * p.y_sv + (long)b * p.y_sb;
        const float* yp = (BWD_EPI && p.yprev) ? p.yprev + (long)v * p.yp_sv + (long)b * p.yp_sb : nullptr;
        const bool zero = v == p.zero_row;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int n0 = n_base + n * 16 + lq * 4;
            if (n0 >= p.Nout) continue;
            f32x4 a = acc[m][n];
            if (p.vec_out) {
                if (!BWD_EPI) {
                    if (p.bias) a += *reinterpret_cast<const f32x4*>(p.bias + n0);
                    a = sh_act_fwd4(a, p.act);
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

// ------------------------------------------------------------------------------------------
// bf16x3 form of the direct kernel (mma_mode SH_MMA_SPLIT3; Cg % 8 == 0).  Every fp32 operand is written as an
// EXACT sum of three bf16 numbers, x = h + m + l (h = rne_bf16(x), m = rne_bf16(x - h), l = x - h - m: 8 + 8 + 8 significand
// bits), and a product row is evaluated as the six leading terms of (Wh + Wm + Wl)(Xh + Xm + Xl) -
//     Wh Xh + (Wh Xm + Wm Xh) + (Wh Xl + Wl Xh + Wm Xm)
// - on v_mfma_f32_16x16x32_bf16 with fp32 accumulation.  bf16 x bf16 products are exact in fp32; the three dropped terms are
// below 2^-24 of |w||x|, i.e. under the rounding of the fp32 FMA chain the exact form executes (the tolerances in tests/ are
// the same for both forms; identity weights still copy bit for bit: 1.0 = (1, 0, 0) and h + m + l sums back exactly).  Six
// 16-cycle MFMAs cover a 32-deep k-step that takes eight 32-cycle v_mfma_f32_16x16x4_f32: 2.7x less matrix-pipe time, paid
// for with ~44 VALU operations per 8 gathered values (the split runs in the lane that loaded them).
// Same tiling as the direct kernel: the gathered operand stays in the registers of the lane that loaded it (lane = row r,
// k-block kq: the 32 contiguous bytes x[row r][k0 + 8 kq .. +7]); the weight chunk is split by the staging threads and kept
// in LDS as three planes of MFMA fragments (1 KiB per 16-channel tile and plane: slot = lane, XOR-permuted inside aligned
// groups of 8 so that the staging writes - two channels x four k-blocks per 8 threads - are conflict-free as well).
// ALL9: all nine partial products (the product of the split operands is then EXACT; only the fp32 accumulation rounds)
template <int NT, bool BWD_EPI, int RT, bool ALL9 = false>
__global__ __launch_bounds__(NTHREADS) void gather_gemm_split3_kernel(const GGParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* Wl = reinterpret_cast<u32x4*>(smem);                  // [3 buffers][3 planes][NT][64 slots]
    int* Ts = reinterpret_cast<int*>(Wl + 3 * 3 * NT * 64);      // table tile (element offsets)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int TB = 1 << p.log2TB, TV = (64 * RT) >> p.log2TB;
    const int item = sh_xcd_remap(blockIdx.x, gridDim.x);
    const int tile = item / p.nsplit;
    const int n_base = (item - tile * p.nsplit) * (NT * 16);
    const int bt = tile / p.n_vtiles, vt = tile - bt * p.n_vtiles;
    const int v0 = vt * TV, b0 = bt * TB;
    const int S = p.S;

    const int lrow = lane & 15, lq = lane >> 4;
    int a_ts[RT];
    long a_boff[RT];
#pragma unroll
    for (int m = 0; m < RT; ++m) {
        const int row = 16 * RT * wave + 16 * m + lrow;
        const int vl = row >> p.log2TB, bl = row & (TB - 1);
        a_ts[m] = vl * S;
        a_boff[m] = (b0 + bl) < p.B ? (long)(b0 + bl) * p.x_sb : 0;      // rows past B read row 0 of the slice; never stored
    }
    // running (k, s, channel) of this lane's 8 gathered columns for the next chunk to load
    int k_n = 8 * lq, s_n = k_n / p.Cg, c_n = k_n - s_n * p.Cg;
    const int adv_s = KC / p.Cg, adv_c = KC - adv_s * p.Cg;
    auto load_a = [&](f32x4 (&ra)[RT][2]) {
        const bool kok = k_n < p.K;                              // K tail / prefetch past the end: any valid address
        const int s = kok ? s_n : 0, ch = kok ? c_n : 0;
        unsigned toff[RT];
#pragma unroll
        for (int m = 0; m < RT; ++m) toff[m] = (unsigned)Ts[a_ts[m] + s];
#pragma unroll
        for (int m = 0; m < RT; ++m) {
            const float* src = p.x + toff[m] + a_boff[m] + ch;
            ra[m][0] = *reinterpret_cast<const f32x4*>(src);
            ra[m][1] = *reinterpret_cast<const f32x4*>(src + 4);
        }
        k_n += KC; c_n += adv_c; s_n += adv_s;
        const bool wrap = c_n >= p.Cg;
        c_n = wrap ? c_n - p.Cg : c_n;
        s_n = wrap ? s_n + 1 : s_n;
    };

    // weight chunk staging: piece pid = tid + 256 i -> weight row pid >> 2, k-block pid & 3 (8 columns, 32 bytes)
    constexpr int PIECES = NT * 64;
    constexpr int WP = PIECES >= NTHREADS ? PIECES / NTHREADS : 1;
    const bool w_thread = PIECES >= NTHREADS || tid < PIECES;
    long w_off[WP];
    int w_slot[WP];
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        const int pid = tid + NTHREADS * i, n = pid >> 2, kp = pid & 3;
        w_off[i] = (n_base + n) < p.Nout ? (long)(n_base + n) * p.Kw + 8 * kp : 8 * kp;
        w_slot[i] = (n >> 4) * 64 + (((n & 15) + 16 * kp) ^ (kp << 1));
    }
    int kw_n = 8 * (tid & 3);
    auto load_w = [&](f32x4 (&rw)[WP][2], unsigned& mask) {
        const bool kok = kw_n < p.K;
        mask = kok ? 1u : 0u;
        const int kc = kok ? kw_n - 8 * (tid & 3) : 0;
#pragma unroll
        for (int i = 0; i < WP; ++i) {
            rw[i][0] = *reinterpret_cast<const f32x4*>(p.w + w_off[i] + kc);
            rw[i][1] = *reinterpret_cast<const f32x4*>(p.w + w_off[i] + kc + 4);
        }
        kw_n += KC;
    };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto store_w = [&](int buf, const f32x4 (&rw)[WP][2], unsigned mask) {
        if (w_thread) {
            u32x4* Wb = Wl + buf * 3 * NT * 64;
#pragma unroll
            for (int i = 0; i < WP; ++i) {
                u32x4 h, m, l;
                sh_split3(mask ? rw[i][0] : zero4, mask ? rw[i][1] : zero4, h, m, l);
                Wb[w_slot[i]] = h;
                Wb[NT * 64 + w_slot[i]] = m;
                Wb[2 * NT * 64 + w_slot[i]] = l;
            }
        }
    };

    f32x4 acc[RT][NT];
#pragma unroll
    for (int m = 0; m < RT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = zero4;
    const int rslot = lane ^ ((lane >> 4) << 1);
    auto compute = [&](int buf, const f32x4 (&ra)[RT][2]) {
        bf16x8 xh[RT], xm[RT], xl[RT];
#pragma unroll
        for (int m = 0; m < RT; ++m) {
            u32x4 h, mm, l;
            sh_split3(ra[m][0], ra[m][1], h, mm, l);
            xh[m] = __builtin_bit_cast(bf16x8, h); xm[m] = __builtin_bit_cast(bf16x8, mm); xl[m] = __builtin_bit_cast(bf16x8, l);
        }
        const u32x4* Wb = Wl + buf * 3 * NT * 64 + rslot;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const bf16x8 wh = __builtin_bit_cast(bf16x8, Wb[n * 64]);
            const bf16x8 wm = __builtin_bit_cast(bf16x8, Wb[(NT + n) * 64]);
            const bf16x8 wl = __builtin_bit_cast(bf16x8, Wb[(2 * NT + n) * 64]);
#pragma unroll
            for (int m = 0; m < RT; ++m) {                       // smallest terms first
                f32x4 c = acc[m][n];
                if (ALL9) {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xl[m], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xm[m], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xl[m], c, 0, 0, 0);
                }
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh[m], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl[m], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xm[m], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xh[m], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xm[m], c, 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh[m], c, 0, 0, 0);
            }
        }
    };

    f32x4 ra[3][RT][2], rw[3][WP][2];
    unsigned mw[3];
    load_w(rw[0], mw[0]);                                // chunk 0 of the weight does not need the table: in flight under its load
    {
        const int nT = TV * S;
        const long lim = (long)p.R * S;
        for (int i = tid; i < nT; i += NTHREADS) {
            const long g = (long)v0 * S + i;
            Ts[i] = g < lim ? (int)(unsigned)((long)p.table[g] * p.x_sv) : 0;
        }
    }
    __syncthreads();
    load_a(ra[0]);                                       // chunk 0
    store_w(0, rw[0], mw[0]);
    __syncthreads();
    load_a(ra[1]); load_w(rw[1], mw[1]);                 // chunk 1
    auto step = [&](auto J) {
        constexpr int j = decltype(J)::value;
        load_a(ra[(j + 2) % 3]); load_w(rw[(j + 2) % 3], mw[(j + 2) % 3]);               // chunk c+2
        __builtin_amdgcn_sched_barrier(0);               // keep the prefetch loads AHEAD of the MFMA phase
        compute(j, ra[j]);
        store_w((j + 1) % 3, rw[(j + 1) % 3], mw[(j + 1) % 3]);                           // chunk c+1
        __syncthreads();
    };
    for (int c = 0; c < p.nchunks; c += 3) {
        step(std::integral_constant<int, 0>{});
        if (c + 1 >= p.nchunks) break;
        step(std::integral_constant<int, 1>{});
        if (c + 2 >= p.nchunks) break;
        step(std::integral_constant<int, 2>{});
    }

    // ---- epilogue (identical to gather_gemm_kernel): lane holds channels n0..n0+3 of tile row 16*RT*wave + 16*m + lrow
#pragma unroll
    for (int m = 0; m < RT; ++m) {
        const int row = 16 * RT * wave + 16 * m + lrow;
        const int vl = row >> p.log2TB, bl = row & (TB - 1);
        const int v = v0 + vl, b = b0 + bl;
        if (v >= p.R || b >= p.B) continue;
        float* yrow = p.y + (long)v * p.y_sv + (long)b * p.y_sb;
        const float* yp = (BWD_EPI && p.yprev) ? p.yprev + (long)v * p.yp_sv + (long)b * p.yp_sb : nullptr;
        const bool zero = v == p.zero_row;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int n0 = n_base + n * 16 + lq * 4;
            if (n0 >= p.Nout) continue;
            f32x4 a = acc[m][n];
            if (!BWD_EPI) {
                if (p.bias) a += *reinterpret_cast<const f32x4*>(p.bias + n0);
                a = sh_act_fwd4(a, p.act);
            } else if (yp) {
                const f32x4 yv = *reinterpret_cast<const f32x4*>(yp + n0);
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] *= sh_act_grad_from_out(yv[j], p.act);
            }
            if (zero) a = (f32x4){0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(yrow + n0) = a;
        }
    }
}

template <int NT, bool BWD_EPI>
int launch_s3(const GGParams& p_in, int rt, hipStream_t st) {
    GGParams p = p_in;
    const int tb = 1 << p.log2TB;
    if (tb > 64 * rt) rt = 2;
    const int TV = (64 * rt) >> p.log2TB;
    p.n_vtiles = sh_cdiv(p.R, TV);
    const int nblocks = p.n_vtiles * p.n_btiles * p.nsplit;
    const size_t smem = (size_t)(3 * 3 * NT * 64) * 16 + (size_t)(TV * p.S) * sizeof(int);
    static const int all9 = sh_env_int("SH_S3_ALL9", 0, 0, 1);
    ShProfScope ps(st, "gather_gemm_split3_kernel<%d, %s, %d, %s>|R=%d B=%d K=%d N=%d grid=%d", NT, BWD_EPI ? "true" : "false", rt,
                   all9 ? "true" : "false", p.R, p.B, p.K, p.Nout, nblocks);
    if (all9) {
        if (rt == 1) SH_LAUNCH_PS(ps, (gather_gemm_split3_kernel<NT, BWD_EPI, 1, true>), dim3(nblocks), dim3(NTHREADS), smem, st, p);
        else SH_LAUNCH_PS(ps, (gather_gemm_split3_kernel<NT, BWD_EPI, 2, true>), dim3(nblocks), dim3(NTHREADS), smem, st, p);
    } else if (rt == 1) SH_LAUNCH_PS(ps, (gather_gemm_split3_kernel<NT, BWD_EPI, 1>), dim3(nblocks), dim3(NTHREADS), smem, st, p);
    else SH_LAUNCH_PS(ps, (gather_gemm_split3_kernel<NT, BWD_EPI, 2>), dim3(nblocks), dim3(NTHREADS), smem, st, p);
    SH_CHECK_LAUNCH("gather_gemm_split3");
    return SH_OK;
}

template <int NT, bool BWD_EPI>
int launch_ggd(const GGParams& p_in, int nblocks128 /* workgroups if the tiles had 128 rows */, hipStream_t st) {
    // 16-row tiles per wave.  Measured on MI355X (B = 64): one tile per wave (64-row workgroups, twice as many of them)
    // is 3-8 % faster while the launch has <= 1024 workgroups of 128 rows, two tiles are level above that, four are
    // 10-30 % slower (registers: fewer waves per SIMD to hide the LDS and global latencies).
    static const int rt_pref = sh_env_int("SH_GG_RT", 0, 0, 2);
    GGParams p = p_in;
    const int tb = 1 << p.log2TB;
    int rt = rt_pref == 0 ? (nblocks128 <= 1024 ? 1 : 2) : rt_pref;
    if (tb > 64 * rt) rt = 2;
    const int TV = (64 * rt) >> p.log2TB;
    p.n_vtiles = sh_cdiv(p.R, TV);
    const int nblocks = p.n_vtiles * p.n_btiles * p.nsplit;
    const size_t smem = (size_t)(3 * NT * 16 * KC) * sizeof(float) + (size_t)(TV * p.S) * sizeof(int);
    const bool skip = rt == 1 && BWD_EPI && p.skip_on;
    ShProfScope ps(st, "gather_gemm_direct_kernel<%d, %s, %d, %s>|R=%d B=%d K=%d N=%d grid=%d", NT, BWD_EPI ? "true" : "false", rt,
                   skip ? "true" : "false", p.R, p.B, p.K, p.Nout, nblocks);
    // one tile per wave + known zero row: the form that skips no-source entries (the two-tile launches are the fine levels, where
    // 5-22 % of the entries have no source and the extra branches cost more than they save: 58.6 -> 63.8 us measured)
    if (skip) SH_LAUNCH_PS(ps, (gather_gemm_direct_kernel<NT, BWD_EPI, 1, BWD_EPI>), dim3(nblocks), dim3(NTHREADS), smem, st, p);
    else if (rt == 1) SH_LAUNCH_PS(ps, (gather_gemm_direct_kernel<NT, BWD_EPI, 1>), dim3(nblocks), dim3(NTHREADS), smem, st, p);
    else SH_LAUNCH_PS(ps, (gather_gemm_direct_kernel<NT, BWD_EPI, 2>), dim3(nblocks), dim3(NTHREADS), smem, st, p);
    SH_CHECK_LAUNCH("gather_gemm_direct");
    return SH_OK;
}

// ------------------------------------------------------------------------------------------
// Forward of a layer with <= 3 output channels over 16-channel rows (the autoencoder's last decoder layer, 16 -> xyz at the
// finest level).  One 16-wide channel tile would push 13 zero columns through the matrix pipe, but that is not what the MFMA
// kernels lose on this layer: they are bound by the vector L1, which serves a gathered operand in MFMA layout as 64 separate
// 16-byte accesses per load instruction (6.8-8.7 TB/s measured, tools/exp/ta_probe.hip).  Here the gather is LINE-WISE: four
// lanes read one 64-byte row, a wave's load instruction covers sixteen consecutive batch entries of one gathered vertex = 1 KiB
// contiguous (16.6-21.9 TB/s from L2), and the arithmetic - 12 FMAs per 16-byte quad - runs on the VALU with the layer's whole
// weight (S x 16 x 3 floats, 12 per position per lane) in registers.  No LDS, no barrier; a wave walks rows of its XCD's
// contiguous slice (the gathered neighbourhoods stay in that XCD's L2), one 16-batch unit in flight under the one it
// multiplies; the table entries of a row are wave-uniform scalars.  Fixed summation order: positions in sequence, the
// lane's four channels in sequence, then the four channel groups by two butterfly steps.
// (The mirror case - first encoder layer, 3-channel rows -> 16 channels, lane = (batch entry, channel quad) - measured no
// gain in this form: 20.3 vs 18.6 us, its 12-byte rows are scattered over the batch-major input; not kept.)
template <int S>
__global__ __launch_bounds__(256) void conv_out3_linewise_kernel(const GGParams p) {
    const int lane = threadIdx.x & 63;
    const int wave = sh_wave_id();
    const int cq = lane & 3, bl = lane >> 2;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int nw = (int)(gridDim.x >> 3) * 4;                          // waves per XCD (the grid is a multiple of 8)
    const int lo = (int)((long)p.R * xcd / 8), hi = (int)((long)p.R * (xcd + 1) / 8);
    const int wl = local * 4 + wave;
    const int nch = (p.B + 15) >> 4;
    // the weight: w[n][s][4 cq .. 4 cq + 3] for n < Nout (rows of w beyond Nout do not exist: read row 0, never used)
    f32x4 w[S][3];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int n = 0; n < 3; ++n) w[s][n] = *reinterpret_cast<const f32x4*>(p.w + (long)(n < p.Nout ? n : 0) * p.Kw + (p.s0 + s) * 16 + 4 * cq);
    const long lane_off = 4 * cq;
    const int nrows = lo + wl < hi ? (hi - lo - wl + nw - 1) / nw : 0;
    // unit = (batch slice of 64, row, 16-batch chunk of the slice), slice-major: a large batch is walked in slices so that the
    // rows an XCD gathers from stay inside its 4 MiB L2 (861 rows x 4 KiB per slice at 6890 vertices)
    const int cps = nch < 4 ? nch : 4, per_slice = nrows * cps;
    const int units = ((nch + cps - 1) / cps) * per_slice;
    if (units == 0) return;
    auto unit_row = [&](int u) { return lo + wl + ((u % per_slice) / cps) * nw; };
    auto unit_chunk = [&](int u) { return (u / per_slice) * cps + (u % per_slice) % cps; };      // may be >= nch in the last slice: not stored
    const float bias = (p.bias && cq < p.Nout) ? p.bias[cq] : 0.f;
    // Straight-line pipeline (no branch around a vector load, so the waits stay counted): unit u multiplies from one buffer
    // while unit u + 1 is in flight in the other and the table line of unit u + 2 is in flight in scalar registers; units past
    // the end repeat the last one and are not stored.
    auto load_table = [&](int u, long (&tt)[S]) {
        const int r = unit_row(min(u, units - 1));
#pragma unroll
        for (int s = 0; s < S; ++s) tt[s] = (long)p.table[(long)r * p.S + p.s0 + s] * p.x_sv;
    };
    auto issue = [&](const long (&tt)[S], int u, f32x4 (&buf)[S]) {
        const int b = min(16 * unit_chunk(min(u, units - 1)) + bl, p.B - 1);
        const float* base = p.x + (long)b * p.x_sb + lane_off;
#pragma unroll
        for (int s = 0; s < S; ++s) buf[s] = *reinterpret_cast<const f32x4*>(base + tt[s]);
    };
    auto finish = [&](int u, const f32x4 (&buf)[S]) {
        const int r = unit_row(u), j = unit_chunk(u);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a0 = fmaf(buf[s][i], w[s][0][i], a0);
                a1 = fmaf(buf[s][i], w[s][1][i], a1);
                a2 = fmaf(buf[s][i], w[s][2][i], a2);
            }
        a0 += __shfl_xor(a0, 1, 64); a1 += __shfl_xor(a1, 1, 64); a2 += __shfl_xor(a2, 1, 64);
        a0 += __shfl_xor(a0, 2, 64); a1 += __shfl_xor(a1, 2, 64); a2 += __shfl_xor(a2, 2, 64);
        const int b = 16 * j + bl;
        if (u < units && cq < p.Nout && b < p.B) {                     // lane cq stores channel cq
            float* dst = p.y + (long)r * p.y_sv + (long)b * p.y_sb + cq;
            float acc = cq == 0 ? a0 : cq == 1 ? a1 : a2;
            if (p.pass == 1) { *dst = acc; return; }                   // first of two passes: the raw partial sum
            if (p.pass == 2) acc += *dst;
            const float v = sh_act_fwd(acc + bias, p.act);
            *dst = r == p.zero_row ? 0.f : v;
        }
    };
    f32x4 bufA[S], bufB[S];
    long t0[S], t1[S];
    load_table(0, t0);
    issue(t0, 0, bufA);
    load_table(1, t1);
    for (int u = 0; u < units; u += 2) {
        issue(t1, u + 1, bufB);
        load_table(u + 2, t0);
        finish(u, bufA);
        issue(t0, u + 2, bufA);
        load_table(u + 3, t1);
        finish(u + 1, bufB);
    }
}

template <int S>
int launch_out3(const GGParams& p, hipStream_t st) {
    static const int grid = 8 * sh_env_int("SH_OUT3_WG_PER_XCD", 64, 1, 1024);       // 2 workgroups per CU
    ShProfScope ps(st, "conv_out3_linewise_kernel<%d>|R=%d B=%d K=%d N=%d grid=%d pass=%d", S, p.R, p.B, p.pass == 1 ? S * 16 : p.K, p.Nout, grid, p.pass);
    SH_LAUNCH_PS(ps, conv_out3_linewise_kernel<S>, dim3(grid), dim3(256), 0, st, p);
    SH_CHECK_LAUNCH("conv_out3_linewise");
    return SH_OK;
}

thread_local bool tl_img_written = false;                  // did the kernel a dispatch picked write GGParams::y_img?
template <int NT, bool VEC4, bool BWD_EPI, bool C3 = false>
int launch_gg(const GGParams& p, int nblocks, hipStream_t st) {
    if (p.y_img && p.vec_out) tl_img_written = true;
    const int TV = TM >> p.log2TB;
    const size_t smem = (size_t)(2 * TM * KC + 2 * NT * 16 * KC) * sizeof(float) + (size_t)(TV * p.S) * sizeof(int);
    {
        ShProfScope ps(st, "gather_gemm_kernel<%d, %s, %s, %s, %s>|R=%d B=%d K=%d N=%d grid=%d", NT, VEC4 ? "true" : "false",
                       BWD_EPI ? "true" : "false", p.log2TB == 4 ? "true" : "false", C3 ? "true" : "false", p.R, p.B, p.K, p.Nout,
                       nblocks);
        if (p.log2TB == 4)
            SH_LAUNCH_PS(ps, (gather_gemm_kernel<NT, VEC4, BWD_EPI, true, C3>), dim3(nblocks), dim3(NTHREADS), smem, st, p);
        else
            SH_LAUNCH_PS(ps, (gather_gemm_kernel<NT, VEC4, BWD_EPI, false, C3>), dim3(nblocks), dim3(NTHREADS), smem, st, p);
    }
    SH_CHECK_LAUNCH("gather_gemm");
    return SH_OK;
}

template <bool BWD_EPI>
int dispatch_gg(GGParams& p, hipStream_t st) {
    // batch slice per tile: a power of two <= SH_GG_TB (default 16).  Narrow slices keep the
    // working set of an XCD's sweep (neighbour rows x slice width) inside its 4 MiB L2.
    static const int tb_pref = sh_env_int("SH_GG_TB", 16, 1, TM);
    int tb = 1;
    while (tb < p.B && tb < tb_pref) tb <<= 1;
    p.log2TB = sh_ilog2_floor(tb);
    static const int skip_env = sh_env_int("SH_GG_SKIP", 1, 0, 1);
    p.skip_on = (BWD_EPI && skip_env && p.skip_on && tb == 16 && p.Cg % 16 == 0) ? 1 : 0;
    const int TV = TM >> p.log2TB;
    p.n_btiles = sh_cdiv(p.B, tb);
    p.n_vtiles = sh_cdiv(p.R, TV);
    p.Kw = p.K;
    // channel tiles of 16: up to 8 per workgroup, more output channels are split over workgroups from the start (each
    // workgroup gathers the rows again; the reference's configurations stay <= 128)
    const int nt_all = sh_cdiv(p.Nout, 16);
    int nt = nt_all <= 1 ? 1 : nt_all <= 2 ? 2 : nt_all <= 4 ? 4 : 8;
    const int nsplit0 = sh_cdiv(nt_all, nt);
    // 3-channel gathered rows (built for one channel tile): dwordx3 loads, K counted in zero-padded quads
    static const int c3_on = sh_env_int("SH_GG_C3", 1, 0, 1);
    const bool c3 = c3_on && p.Cg == 3 && nt == 1 && reinterpret_cast<uintptr_t>(p.w) % 4 == 0;
    if (c3) { p.Cg = 4; p.K = 4 * p.S; }
    p.nchunks = sh_cdiv(p.K, KC);
    const long nblocks = (long)p.n_vtiles * p.n_btiles;
    SH_REQUIRE(nblocks > 0 && nblocks < (1L << 31), SH_ERR_UNSUPPORTED, "gather_gemm: grid %ld out of range", nblocks);
    SH_REQUIRE(p.S <= 64, SH_ERR_UNSUPPORTED, "gather_gemm: spiral length %d > 64", p.S);
    const bool vec4 = (p.Cg % 4 == 0) && (p.x_sv % 4 == 0) && (p.x_sb % 4 == 0) &&
                      ((reinterpret_cast<uintptr_t>(p.x) | reinterpret_cast<uintptr_t>(p.w)) % 16 == 0);
    p.vec_out = (p.Nout % 4 == 0) && (p.y_sv % 4 == 0) && (p.y_sb % 4 == 0) &&
                (reinterpret_cast<uintptr_t>(p.y) % 16 == 0) &&
                (!p.bias || reinterpret_cast<uintptr_t>(p.bias) % 16 == 0) &&
                (!p.yprev || ((p.yp_sv % 4 == 0) && (p.yp_sb % 4 == 0) && reinterpret_cast<uintptr_t>(p.yprev) % 16 == 0));
    // <= 3 output channels over 16-channel rows: the line-wise VALU kernel (both arithmetic forms: nothing for a split to win)
    static const int out3_on = sh_env_int("SH_GG_OUT3", 1, 0, 1);
    if (!BWD_EPI && out3_on && p.Nout <= 3 && p.Cg == 16 && vec4 && p.S >= 6 && p.S <= 24) {
        p.Kw = p.K;
        auto run = [&](int n) -> int {
            switch (n) {
                case 6: return launch_out3<6>(p, st);
                case 7: return launch_out3<7>(p, st);
                case 8: return launch_out3<8>(p, st);
                case 9: return launch_out3<9>(p, st);
                case 10: return launch_out3<10>(p, st);
                case 11: return launch_out3<11>(p, st);
                default: return launch_out3<12>(p, st);
            }
        };
        if (p.S <= 12) { p.s0 = 0; p.pass = 0; return run(p.S); }
        // 13..24 positions (data-dependent spiral lengths, utils_spiral.py:72-82; BASELINE config 4 forces 18): two passes of
        // 7..12 over the same rows - the second adds the first's partial sums (12 bytes per row) before bias / activation
        const int s1 = (p.S + 1) / 2;
        p.s0 = 0; p.pass = 1;
        int rc = run(s1);
        if (rc != SH_OK) return rc;
        p.s0 = s1; p.pass = 2;
        return run(p.S - s1);
    }
    // bf16x3 form (mma_mode SH_MMA_SPLIT3): up to four channel tiles per workgroup, the rest split over workgroups
    static const int s3_min_nt = sh_env_int("SH_S3_MIN_NT", 4, 1, 8);      // layers with fewer channel tiles keep the exact form (no gain there)
    if (sh_f32_mma_mode() != SH_MMA_EXACT && vec4 && p.vec_out && !c3 && p.Cg % 8 == 0 && nt >= s3_min_nt) {
        static const int s3_nt = sh_env_int("SH_S3_NT", 4, 1, 8), s3_rt = sh_env_int("SH_S3_RT", 0, 0, 2);
        static const int s3_rt2_at = sh_env_int("SH_S3_RT2_AT", 2048, 1, 1 << 30);
        int ntw = nt;
        p.nsplit = nsplit0;
        while (ntw > s3_nt) { ntw >>= 1; p.nsplit <<= 1; }
        const long wg64 = (long)sh_cdiv(p.R, 64 >> (p.log2TB < 6 ? p.log2TB : 6)) * p.n_btiles * p.nsplit;      // workgroups of 64 rows
        // two row tiles per wave halve the weight-split work per MFMA: pays from ~1000 reduction columns on (67.7 -> 64.7, 64.2 -> 60.5 us)
        const int rt = s3_rt ? s3_rt : ((wg64 >= s3_rt2_at || (p.K >= 1024 && wg64 >= 512)) ? 2 : 1);
        switch (ntw) {
            case 1: return launch_s3<1, BWD_EPI>(p, rt, st);
            case 2: return launch_s3<2, BWD_EPI>(p, rt, st);
            case 4: return launch_s3<4, BWD_EPI>(p, rt, st);
            default: return launch_s3<8, BWD_EPI>(p, rt, st);
        }
    }
    // too few row tiles to fill 256 CUs x ~3 workgroups: split the output channels over workgroups
    static const int fill_target = sh_env_int("SH_GG_FILL", 768, 1, 1 << 20);
    static const int direct_on = sh_env_int("SH_GG_DIRECT", 1, 0, 1);
    p.nsplit = nsplit0;
    while (nt > 2 && nblocks * p.nsplit < fill_target) { nt >>= 1; p.nsplit <<= 1; }
    // two channel tiles run in the direct form with 64-row workgroups (twice the count): only split them further if even
    // that does not fill the chip - one tile per workgroup is the slower staged form
    if (nt == 2 && nblocks * p.nsplit * ((vec4 && direct_on) ? 2 : 1) < fill_target) { nt = 1; p.nsplit <<= 1; }
    // eight channel tiles only exist staged; two workgroups of four tiles in the direct form are faster although each
    // gathers the rows again (27 554-vertex template, 128 channels: 88 -> ~105 TFLOP/s)
    if (nt == 8 && vec4 && direct_on) { nt = 4; p.nsplit <<= 1; }
    const long nitems = nblocks * p.nsplit;
#define SH_GG_CASE(NTV)                                                                  \
    return vec4 ? launch_gg<NTV, true, BWD_EPI>(p, (int)nitems, st)              \
                : launch_gg<NTV, false, BWD_EPI>(p, (int)nitems, st)
    if (c3) return launch_gg<1, true, BWD_EPI, true>(p, (int)nitems, st);
    // the direct form serves 2 and 4 channel tiles; one tile (16 channels) measured 25-30 % slower in it than staged
    if (nt == 4 && vec4 && direct_on) return launch_ggd<4, BWD_EPI>(p, (int)nitems, st);
    if (nt <= 1) { SH_GG_CASE(1); }
    // the direct form wins for two channel tiles (2-8 us per launch on MI355X) and loses for 1 and 4: a chunk of a
    // one-tile layer has too few MFMAs between the weight-chunk barriers, four tiles are bound elsewhere
    if (nt <= 2 && vec4 && direct_on) return launch_ggd<2, BWD_EPI>(p, (int)nitems, st);
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
#ifndef SH_WG_ABLATE
#define SH_WG_ABLATE 0      // diagnostic builds only (tools/ablate_wgrad.sh): 1 no global loads, 2 no MFMA, 4 no LDS stores, 8 no LDS reads
#endif
constexpr int WG_TABLE_CAP = 8192;      // ints of gather table a workgroup keeps in LDS (32 KiB)

struct WGParams {
    const float* dpre; long dp_sv, dp_sb;
    const float* x; long x_sv, x_sb;
    const int* table;
    float* slab;        // [nrc][Cout*K] then [nrc][Cout] bias partials
    long slab_stride;   // Cout*K
    long bias_off;      // nrc*Cout*K
    int B, R, S, Cin, Cout, K;
    int log2TB, n_btiles, n_vtiles, nvc, steps_per_block, ncg;
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
    int* Tl = reinterpret_cast<int*>(Ps + 2 * TMW * LDP);   // gather-table lines of this block's vertices

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // linear block id -> XCD-contiguous order, column group fastest: the column groups of one row
    // chunk (same dpre rows, same gathered input rows) run on the same XCD, and an XCD sweeps a
    // contiguous vertex range of one batch slice.
    const int lin = sh_xcd_remap(blockIdx.x, gridDim.x);
    const int rc = lin / p.ncg, cg = lin - rc * p.ncg;
    const int bt = rc / p.nvc, vc = rc - bt * p.nvc;
    const int TB = 1 << p.log2TB, TV = TMW >> p.log2TB;
    const int step0 = vc * p.steps_per_block;
    const int nsteps = min(p.steps_per_block, p.n_vtiles - step0);      // may be <= 0 for the last chunk
    const int vbase = step0 * TV, b0 = bt * TB;
    const int S = p.S;

    {   // table lines of vertices vbase .. vbase + nsteps*TV - 1 (clipped to R; padding -> row 0)
        const int nT = max(nsteps, 0) * TV * S;
        const long lim = (long)p.R * S;
        for (int i = tid; i < nT; i += NTHREADS) {
            const long g = (long)vbase * S + i;
            Tl[i] = g < lim ? p.table[g] : 0;
        }
    }
    __syncthreads();

    // fixed (s, channel) of this thread's G quad: the column group never changes.  Columns past
    // K read (s=0, c=0): they only feed weight columns that are never stored.
    const int gq = tid % GQ, grow0 = tid / GQ;
    const int k = cg * KCW + 4 * gq;
    const bool k_in = k < p.K;
    int s4[4], c4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int kk = k + j;
        s4[j] = kk < p.K ? kk / p.Cin : 0;
        c4[j] = kk < p.K ? kk - s4[j] * p.Cin : 0;
    }
    // per-pass row decomposition (does not depend on the step)
    int g_vl[GP], p_vl[PP], p_q[PP], p_row[PP];
    long g_boff[GP], p_boff[PP];
    unsigned g_bok = 0, p_bok = 0;                       // bit i: batch index of pass i is < B
#pragma unroll
    for (int i = 0; i < GP; ++i) {
        const int row = grow0 + GROWS * i;
        const int bl = row & (TB - 1);
        g_vl[i] = row >> p.log2TB;
        const bool ok = b0 + bl < p.B;
        g_bok |= (ok ? 1u : 0u) << i;
        g_boff[i] = ok ? (long)(b0 + bl) * p.x_sb : 0;
    }
    const bool p_vec = (p.Cout & 3) == 0 && ((p.dp_sv | p.dp_sb) & 3) == 0;
#pragma unroll
    for (int i = 0; i < PP; ++i) {
        const int idx = tid + NTHREADS * i;
        const int row = idx < PTOT ? idx / PQ : 0;
        p_row[i] = row;
        p_q[i] = idx < PTOT ? idx - row * PQ : 0;
        p_vl[i] = row >> p.log2TB;
        const int bl = row & (TB - 1);
        const bool ok = idx < PTOT && b0 + bl < p.B && 4 * p_q[i] < p.Cout;
        p_bok |= (ok ? 1u : 0u) << i;
        p_boff[i] = ok ? (long)(b0 + bl) * p.dp_sb + 4 * p_q[i] : 0;
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // Branch-free loads (addresses clamped to valid memory); the row-validity masks are applied
    // when the tile is written to LDS two steps later.  Invalid rows MUST be zeroed here because
    // rows are the reduction dimension.
    auto load_step = [&](int st, f32x4 (&rg)[GP], f32x4 (&rp)[PP], unsigned& mg, unsigned& mp) {
        if (SH_WG_ABLATE & 1) {
            mg = mp = ~0u;
            for (int i = 0; i < GP; ++i) rg[i] = (f32x4){1.f, 2.f, 3.f, (float)st};
            for (int i = 0; i < PP; ++i) rp[i] = (f32x4){1.f, 2.f, 3.f, (float)st};
            return;
        }
        st = st < nsteps ? st : nsteps - 1;
        const int vloc = st * TV;                       // local vertex index of the step's first vertex
        mg = 0; mp = 0;
#pragma unroll
        for (int i = 0; i < GP; ++i) {
            const int vl = vloc + g_vl[i];
            const bool ok = vbase + vl < p.R && ((g_bok >> i) & 1u);
            mg |= (ok ? 1u : 0u) << i;
            if (VEC4) {
                // columns past K (padding of the last column group) all read ONE shared line
                const long off = k_in ? (long)Tl[vl * S + s4[0]] * p.x_sv + g_boff[i] + c4[0] : 0;
                rg[i] = *reinterpret_cast<const f32x4*>(p.x + off);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int u = Tl[vl * S + s4[j]];
                    rg[i][j] = p.x[(long)u * p.x_sv + g_boff[i] + c4[j]];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < PP; ++i) {
            const int vl = vloc + p_vl[i];
            const bool ok = vbase + vl < p.R && ((p_bok >> i) & 1u);
            mp |= (ok ? 1u : 0u) << i;
            const int v = ok ? vbase + vl : 0;
            const float* src = p.dpre + (long)v * p.dp_sv + p_boff[i];
            if (p_vec) {
                rp[i] = *reinterpret_cast<const f32x4*>(src);
            } else {
                rp[i] = zero4;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (ok && 4 * p_q[i] + j < p.Cout) rp[i][j] = src[j];
            }
        }
    };
    auto store_step = [&](int buf, const f32x4 (&rg)[GP], const f32x4 (&rp)[PP], unsigned mg, unsigned mp) {
        if ((SH_WG_ABLATE & 4) && rg[0][0] != 12345.f) return;
        float* Gb = Gs + buf * TMW * LDG;
        float* Pb = Ps + buf * TMW * LDP;
#pragma unroll
        for (int i = 0; i < GP; ++i)
            *reinterpret_cast<f32x4*>(Gb + (grow0 + GROWS * i) * LDG + 4 * gq) = ((mg >> i) & 1u) ? rg[i] : zero4;
#pragma unroll
        for (int i = 0; i < PP; ++i)
            if (tid + NTHREADS * i < PTOT)
                *reinterpret_cast<f32x4*>(Pb + p_row[i] * LDP + 4 * p_q[i]) = ((mp >> i) & 1u) ? rp[i] : zero4;
    };

    f32x4 acc[CTW][COT];
#pragma unroll
    for (int a = 0; a < CTW; ++a)
#pragma unroll
        for (int b = 0; b < COT; ++b) acc[a][b] = zero4;
    float bsum = 0.f;

    const int lcol = lane & 15, lk = lane >> 4;
    auto compute = [&](int buf) {
        const float* Gb = Gs + buf * TMW * LDG + (wave * CTW) * 16 + lcol;
        const float* Pb = Ps + buf * TMW * LDP + lcol;
        // operands of 4 k-steps (16 rows) are read from LDS in one batch, then their MFMAs issue
        // back to back: the ds_read latency is paid twice per tile instead of eight times
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float gf[4][CTW], pf[4][COT];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int row = 16 * half + 4 * kk + lk;
#pragma unroll
                for (int a = 0; a < CTW; ++a) gf[kk][a] = (SH_WG_ABLATE & 8) ? (float)(row + a) : Gb[row * LDG + a * 16];
#pragma unroll
                for (int b = 0; b < COT; ++b) pf[kk][b] = (SH_WG_ABLATE & 8) ? (float)(row - b) : Pb[row * LDP + b * 16];
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int a = 0; a < CTW; ++a)
#pragma unroll
                    for (int b = 0; b < COT; ++b)
                        if (SH_WG_ABLATE & 2) acc[a][b][0] += gf[kk][a] * pf[kk][b];
                        else acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(gf[kk][a], pf[kk][b], acc[a][b], 0, 0, 0);
        }
        if (cg == 0 && tid < COT * 16) {
            const float* Pc = Ps + buf * TMW * LDP + tid;
#pragma unroll 8
            for (int r = 0; r < TMW; ++r) bsum += Pc[r * LDP];
        }
    };

    // two steps of loads in flight (register sets A/B), LDS double-buffered, one barrier per step
    if (nsteps > 0) {
        f32x4 rgA[GP], rgB[GP], rpA[PP], rpB[PP];
        unsigned mgA, mgB, mpA, mpB;
        load_step(0, rgA, rpA, mgA, mpA);
        store_step(0, rgA, rpA, mgA, mpA);
        __syncthreads();
        load_step(1, rgA, rpA, mgA, mpA);
        for (int st = 0; st < nsteps; st += 2) {
            load_step(st + 2, rgB, rpB, mgB, mpB);
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch loads AHEAD of the MFMA phase
            compute(0);
            store_step(1, rgA, rpA, mgA, mpA);
            __syncthreads();
            if (st + 1 >= nsteps) break;
            load_step(st + 3, rgA, rpA, mgA, mpA);
            __builtin_amdgcn_sched_barrier(0);
            compute(1);
            store_step(0, rgB, rpB, mgB, mpB);
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

// out[i] = sum_r slab[r][i], r in increasing order within each of 16 interleaved lanes, lanes then
// combined in a fixed order: deterministic.  Block = 64 outputs x 16 slab lanes.
__global__ __launch_bounds__(1024) void slab_reduce_kernel(const float* __restrict__ slab, long stride, int nslab, long n,
                                                           float* __restrict__ out) {
    __shared__ float red[16][64];
    const int ox = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + ox;
    float s = 0.f;
    if (i < n) {
        int r = ry;
        for (; r + 48 < nslab; r += 64) {
            const float a = slab[(long)r * stride + i], b = slab[(long)(r + 16) * stride + i];
            const float c = slab[(long)(r + 32) * stride + i], d = slab[(long)(r + 48) * stride + i];
            s += (a + b) + (c + d);
        }
        for (; r < nslab; r += 16) s += slab[(long)r * stride + i];
    }
    red[ry][ox] = s;
    __syncthreads();
    if (ry == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][ox];
        out[i] = t;
    }
}

// Several layers' slab reductions / weight transposes in ONE launch (descriptors passed by value as
// kernel arguments, so the launch is graph-capturable): the step has ~9 of each and each is far too
// small to fill the chip, so what they cost separately is launch latency.
constexpr int MR_MAX = 32;
struct MultiReduce {
    const float* slab[MR_MAX]; long stride[MR_MAX]; long n[MR_MAX]; float* out[MR_MAX];
    int nslab[MR_MAX]; int block0[MR_MAX]; int nd;
    unsigned vec_mask;                  // bit d: entry d is reduced four outputs per thread (16-byte loads: 1 KiB of a slab per workgroup)
};
// One workgroup = 64 (or, four per thread, 256) outputs x 16 slab phases.  Per output the sum is the same in both forms and in
// every launch: phase ry takes slabs ry, ry + 16, ... in groups of four ((a + b) + (c + e)), then the 16 phases are added in order.
__global__ __launch_bounds__(1024) void slab_reduce_multi_kernel(const MultiReduce m) {
    __shared__ f32x4 red[16][64];
    int d = 0;
    while (d + 1 < m.nd && m.block0[d + 1] <= (int)blockIdx.x) ++d;
    const float* __restrict__ slab = m.slab[d];
    const long stride = m.stride[d], n = m.n[d];
    const int nslab = m.nslab[d];
    const int ox = threadIdx.x & 63, ry = threadIdx.x >> 6;
    if ((m.vec_mask >> d) & 1u) {                         // round 5: n, stride multiples of 4, 16-byte aligned slabs and output
        const long i = ((long)(blockIdx.x - m.block0[d]) * 64 + ox) * 4;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (i < n) {
            int r = ry;
            for (; r + 48 < nslab; r += 64) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(slab + (long)r * stride + i), b = *reinterpret_cast<const f32x4*>(slab + (long)(r + 16) * stride + i);
                const f32x4 c = *reinterpret_cast<const f32x4*>(slab + (long)(r + 32) * stride + i), e = *reinterpret_cast<const f32x4*>(slab + (long)(r + 48) * stride + i);
                s += (a + b) + (c + e);
            }
            for (; r < nslab; r += 16) s += *reinterpret_cast<const f32x4*>(slab + (long)r * stride + i);
        }
        red[ry][ox] = s;
        __syncthreads();
        if (ry == 0 && i < n) {
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 16; ++k) t += red[k][ox];
            *reinterpret_cast<f32x4*>(m.out[d] + i) = t;
        }
        return;
    }
    const long i = (long)(blockIdx.x - m.block0[d]) * 64 + ox;
    float s = 0.f;
    if (i < n) {
        int r = ry;
        for (; r + 48 < nslab; r += 64) {
            const float a = slab[(long)r * stride + i], b = slab[(long)(r + 16) * stride + i];
            const float c = slab[(long)(r + 32) * stride + i], e = slab[(long)(r + 48) * stride + i];
            s += (a + b) + (c + e);
        }
        for (; r < nslab; r += 16) s += slab[(long)r * stride + i];
    }
    red[ry][ox][0] = s;
    __syncthreads();
    if (ry == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][ox][0];
        m.out[d][i] = t;
    }
}

struct MultiTranspose {
    const float* w[MR_MAX]; float* wt[MR_MAX]; int S[MR_MAX]; int Cin[MR_MAX]; int Cout[MR_MAX]; int block0[MR_MAX]; int nd;
};
__device__ __forceinline__ void weight_transpose_block(const MultiTranspose& m, int blk) {
    int d = 0;
    while (d + 1 < m.nd && m.block0[d + 1] <= blk) ++d;
    const int S = m.S[d], Cin = m.Cin[d], Cout = m.Cout[d];
    const long n = (long)S * Cin * Cout;
    const long i = (long)(blk - m.block0[d]) * 256 + threadIdx.x;
    if (i >= n) return;
    const int co = (int)(i % Cout);
    const long t = i / Cout;
    const int s = (int)(t % S);
    const int ci = (int)(t / S);
    m.wt[d][i] = m.w[d][(long)co * S * Cin + (long)s * Cin + ci];      // wt[ci][s*Cout + co] = w[co][s*Cin + ci]
}
__global__ __launch_bounds__(256) void weight_transpose_multi_kernel(const MultiTranspose m) { weight_transpose_block(m, (int)blockIdx.x); }

// ------------------------------------------------------------------------------------------
// weight gradient, streaming form (Cin % 4 == 0).  The staged kernel above moves the gathered
// tile through LDS only to transpose it; measured on MI355X its global loads, LDS traffic and
// MFMAs barely overlap (profiles/r01_ablation_wgrad.txt).  Here every WAVE owns 64 weight columns
// and a row chunk, and feeds the matrix pipe straight from its 16-byte global loads - no LDS for
// the operands, no barrier in the loop:
//   lane (a = lane & 15, kq = lane >> 4) loads x[nbr(v, s_a)][b0 + 4g + kq][c_a .. c_a+3], the four
//   gathered columns k_a .. k_a+3 (k_a = 64 cg + 4a) of row (v, b0 + 4g + kq).  MFMA t takes element t
//   as its A operand: A_t[i = a][k = kq] = G[row kq][col 4a + t], so one load feeds 4 MFMAs whose
//   outputs are the column sets {4i + t}.  The B operand is dpre[row kq][cout = a + 16 b], one
//   4-byte load per output-channel tile.  acc[t][b][j] = dW[cout 16b + a][col 64cg + 16kq + 4j + t].
// The loads of the next DEPTH-1 vertices are in flight while a vertex's MFMAs issue.
struct WSParams {
    const float* dpre; long dp_sv, dp_sb;
    const float* x; long x_sv, x_sb;
    const int* table;
    float* slab; long slab_stride, bias_off;
    int B, R, S, Cin, Cout, K;
    int log2TB, n_btiles, nvc, vpc, ncg, n_items;     // vpc = vertices per chunk
    int co0;                                          // first output channel of this launch (groups of <= 128 channels)
    // vertex assignment of a row chunk: 0 = vc * vpc .. + vpc - 1 (a contiguous block per wave), 1 = vc, vc + nvc, vc + 2 nvc, ...
    // (the waves of an XCD walk the mesh TOGETHER: at any time they gather from ~nvc neighbouring vertices, which stay in the
    // XCD's L2, instead of from one block of vpc vertices each - measured: the level-0 launch fetched 270 MB for 85 MB of input)
    int vstride;
    // vstride == 2 (round 5): the strided walk INSIDE one vertex range per XCD.  With 1 / 2 / 4 / 8 batch slices, G = 8 / slices
    // XCDs share a slice; the plain strided walk had each of them stream EVERY vertex of the slice through its own L2 (PMC: the
    // level-0 launch fetched 318 MB for 85 MB of input = the G = 4 XCDs of a slice each reading all of it).  Here XCD x (= blockIdx & 7:
    // workgroups are dealt round-robin) owns batch slice x / G and the vertex range [(x % G) Vg, (x % G + 1) Vg) of it; its cpg row
    // chunks walk that range together (chunk j: lo + j, lo + j + cpg, ...), its items are (chunk, column group) pairs, slab index
    // x cpg + j.  Every gathered row is then fetched by ONE XCD (plus the range borders' neighbours).
    int xg_G, xg_cpg, xg_Vg;
    // tail job (sh_spiral_conv_bwd_wgt_presum): workgroups grid_main .. grid_main + tail_blocks - 1 of the launch fill the
    // pre-summed rows the layer's backward-data pass reads through its transposed table - y[r] = sum_e val[e] dpre[col[e]],
    // sh_spmm's arithmetic entry for entry - beside the weight-gradient workgroups instead of in a launch of their own
    int grid_main, tail_blocks, tail_rows;
    const int* tail_rowptr; const int* tail_col; const float* tail_val;
    float* tail_y;                                    // same (row, batch) strides as dpre
    char* tail_img; long tail_img_vb, tail_img_bgb;   // three-plane image of the tail rows (csrc/p3_conv.hip), or NULL
};

// the tail job: one 256-element part of an output row per step, as spmm_kernel<true> (bitwise the same sums)
__device__ __forceinline__ void ws_presum_tail(const WSParams& p) {
    const int CW = p.Cout >> 2, per_row = p.B * CW;
    const int parts = (per_row + 255) >> 8;
    const long items = (long)p.tail_rows * parts;
    for (long it = (long)blockIdx.x - p.grid_main; it < items; it += p.tail_blocks) {
        const int r = (int)(it / parts), part = (int)(it - (long)r * parts);
        const int e0 = p.tail_rowptr[r], e1 = p.tail_rowptr[r + 1];
        const int j = part * 256 + threadIdx.x;
        if (j >= per_row) continue;
        const int b = j / CW, co = 4 * (j - b * CW);
        const long xo = (long)b * p.dp_sb + co;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        int e = e0;
        for (; e + 3 < e1; e += 4) {                          // 4 independent 16-B loads in flight
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(p.dpre + (long)p.tail_col[e] * p.dp_sv + xo);
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(p.dpre + (long)p.tail_col[e + 1] * p.dp_sv + xo);
            const f32x4 x2 = *reinterpret_cast<const f32x4*>(p.dpre + (long)p.tail_col[e + 2] * p.dp_sv + xo);
            const f32x4 x3 = *reinterpret_cast<const f32x4*>(p.dpre + (long)p.tail_col[e + 3] * p.dp_sv + xo);
            const float w0 = p.tail_val[e], w1 = p.tail_val[e + 1], w2 = p.tail_val[e + 2], w3 = p.tail_val[e + 3];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = fmaf(w3, x3[k], fmaf(w2, x2[k], fmaf(w1, x1[k], fmaf(w0, x0[k], acc[k]))));
        }
        for (; e < e1; ++e) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(p.dpre + (long)p.tail_col[e] * p.dp_sv + xo);
            const float w = p.tail_val[e];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = fmaf(w, xv[k], acc[k]);
        }
        *reinterpret_cast<f32x4*>(p.tail_y + (long)r * p.dp_sv + xo) = acc;
        if (p.tail_img) {                                     // the image of the row just written, as spmm_kernel<true, true>
            u32x2 h, m, l;
            sh_split3_quad(acc, h, m, l);
            const bool c16 = p.Cout == 16;
            char* d = p.tail_img + (long)r * p.tail_img_vb + (long)(b >> 4) * p.tail_img_bgb +
                      (c16 ? ((co >> 3) * 16 + (b & 15)) * 16 : (co >> 5) * 3072 + (((co & 31) >> 3) * 16 + (b & 15)) * 16) + ((co >> 2) & 1) * 8;
            const int pb = c16 ? 512 : 1024;
            *reinterpret_cast<u32x2*>(d) = h;
            *reinterpret_cast<u32x2*>(d + pb) = m;
            *reinterpret_cast<u32x2*>(d + 2 * pb) = l;
        }
    }
}

// the part of slab rc an item with column group cg owns (all output channels of this launch x its 64 columns; the bias sums
// with cg == 0), zeroed - for an active item without vertices
__device__ __forceinline__ void ws_zero_slab(const WSParams& p, int rc, int cg, int lane) {
    float* slab = p.slab + (long)rc * p.slab_stride;
    const int co_end = min(p.Cout, p.co0 + 128);
    const int c_lo = p.Cin == 3 ? 48 * cg : 64 * cg, c_hi = min(p.K, c_lo + (p.Cin == 3 ? 48 : 64));      // (3-channel inputs: 16 quads = 48 columns)
    const int w = c_hi - c_lo;
    if (w > 0)
        for (int i = lane; i < (co_end - p.co0) * w; i += 64) slab[(long)(p.co0 + i / w) * p.K + c_lo + i % w] = 0.f;
    if (cg == 0)
        for (int co = p.co0 + lane; co < co_end; co += 64) p.slab[p.bias_off + (long)rc * p.Cout + co] = 0.f;
}

// work item of a wave: (slab index rc, column group cg, first batch entry, vertices v_begin + vl * v_step for vl < nv)
struct WSItem { bool active; int rc, cg, b0, v_begin, v_step, nv; };
__device__ __forceinline__ WSItem ws_item(const WSParams& p, int wave) {
    WSItem it;
    if (p.vstride == 2) {
        const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
        const int local = li * 4 + wave;
        it.active = local < p.xg_cpg * p.ncg;
        const int l2 = it.active ? local : 0;
        const int j = l2 / p.ncg;
        it.cg = l2 - j * p.ncg;
        const int bt = xcd / p.xg_G, g = xcd - bt * p.xg_G;
        const int lo = g * p.xg_Vg, hi = min(p.R, lo + p.xg_Vg);
        it.rc = xcd * p.xg_cpg + j;
        it.b0 = bt << p.log2TB;
        it.v_begin = lo + j; it.v_step = p.xg_cpg;
        it.nv = (it.active && it.v_begin < hi) ? (hi - it.v_begin + p.xg_cpg - 1) / p.xg_cpg : 0;
        return it;
    }
    const int item_raw = sh_xcd_remap(blockIdx.x, p.grid_main) * 4 + wave;
    it.active = item_raw < p.n_items;
    const int item = it.active ? item_raw : 0;
    it.rc = item / p.ncg; it.cg = item - it.rc * p.ncg;
    const int bt = it.rc / p.nvc, vc = it.rc - bt * p.nvc;
    it.b0 = bt << p.log2TB;
    it.v_begin = p.vstride ? vc : vc * p.vpc; it.v_step = p.vstride ? p.nvc : 1;      // vertex of local index vl: v_begin + vl * v_step
    it.nv = !it.active ? 0 : p.vstride ? (vc < p.R ? (p.R - vc + p.nvc - 1) / p.nvc : 0) : min(p.vpc, p.R - it.v_begin);
    return it;
}

// C3: Cin == 3, columns counted in zero-padded quads (k' = 4 s + c), dwordx3 gathers, scalar slab stores.
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ILV (round 5): the COT channel tiles of a wave are INTERLEAVED - lane la's B operands are the channels co0 + COT la + b instead
// of co0 + 16 b + la - so its COT gradient values of a row are contiguous and arrive by ONE load of 4 COT bytes (COT = 8: two
// 16-byte loads) instead of COT 4-byte loads: NG instead of NG x COT gradient loads (and their address arithmetic) per vertex.
// Only which output channel an accumulator belongs to changes (the slab stores below); needs all 16 COT channels of the launch.
template <int COT, int NG, int DEPTH, bool FULL, bool C3 = false, bool ILV = false>
__global__ __launch_bounds__(NTHREADS) void wgrad_stream_kernel(const WSParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x >= p.grid_main) { ws_presum_tail(p); return; }    // tail job (whole workgroups; before any barrier)
    const int lane = threadIdx.x & 63;
    // the wave index as a SCALAR (round 5): the item, its column group and loop bounds are then uniform values - real branches
    // instead of predicated code (the bias sums of the cg == 0 items were executed, masked, by every wave): the four-tile launches
    // 82.5 -> 75 and 30.2 -> 28 us
    const int wave = sh_wave_id();
    int* Tl = reinterpret_cast<int*>(smem) + wave * p.vpc * p.S;          // this wave's table lines
    const WSItem wi = ws_item(p, wave);
    const int rc = wi.rc, cg = wi.cg, b0 = wi.b0, v_begin = wi.v_begin, v_step = wi.v_step, nv = wi.nv;
    const int S = p.S;
    // table lines, PRE-MULTIPLIED by the row stride of x (in 16-byte units - the stride is a multiple of 4 floats - so that 32 bits
    // reach 64 GB; the 3-channel form in elements, its launcher checks the range): the gather
    // addresses of a vertex are then one add away from the LDS read instead of behind two 64-bit multiplies - these kernels live
    // on how early their gathers are issued
    for (int i = lane; i < nv * S; i += 64) {
        const int vl = i / S, j = i - vl * S;
        Tl[i] = (int)((unsigned)p.table[(long)(v_begin + vl * v_step) * S + j] * (unsigned)(C3 ? p.x_sv : p.x_sv >> 2));
    }
    __syncthreads();
    if (nv <= 0) {                    // nothing to sum; an ACTIVE item whose vertex range is empty still owns a slab: zeros
        if (wi.active) ws_zero_slab(p, rc, cg, lane);
        return;                       // (a separate exit, so that the main path below stays the straight-line code it was: with the
    }                                 //  prefetch inside an `if` the four-tile instance ran 25 % slower)

    const int la = lane & 15, kq = lane >> 4;
    const int k0 = cg * 64 + 4 * la;
    const bool k_in = C3 ? k0 < 4 * p.S : k0 < p.K;              // columns past K: read column 0, never stored
    const int s_l = k_in ? (C3 ? k0 >> 2 : k0 / p.Cin) : 0, c_l = (k_in && !C3) ? k0 - s_l * p.Cin : 0;
    // batch entry of this lane in group g: b0 + 4g + kq
    int bcl[NG];
    unsigned bok = 0;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int b = b0 + 4 * g + kq;
        bok |= (b < p.B ? 1u : 0u) << g;
        bcl[g] = b < p.B ? b : p.B - 1;
    }
    const float* pb = p.dpre + (long)v_begin * p.dp_sv;
    int pco[COT];
#pragma unroll
    for (int b = 0; b < COT; ++b) pco[b] = ILV ? p.co0 + COT * la + b : min(p.co0 + 16 * b + la, p.Cout - 1);      // channels past Cout: duplicate, never stored
    // loop-invariant parts of the addresses, formed once: x column + batch entry, dpre batch entry
    const float* xg[NG];
    long pbo[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) { xg[g] = p.x + c_l + (long)bcl[g] * p.x_sb; pbo[g] = (long)bcl[g] * p.dp_sb; }
    const long pv_step = (long)v_step * p.dp_sv;

    f32x4 gr[DEPTH][NG];
    float pr[DEPTH][NG][COT];
    auto load_v = [&](int vl, f32x4 (&g4)[NG], float (&pp)[NG][COT]) {
        vl = vl < nv ? vl : nv - 1;
        const long goff = C3 ? (long)Tl[vl * S + s_l] : (long)(unsigned)Tl[vl * S + s_l] << 2;
        const float* psrc = pb + (long)vl * pv_step;
#pragma unroll
        for (int g = 0; g < NG; ++g) g4[g] = C3 ? sh_ld3(xg[g] + goff) : *reinterpret_cast<const f32x4*>(xg[g] + goff);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if constexpr (ILV && COT == 2) {
                const f32x2 v = *reinterpret_cast<const f32x2*>(psrc + pbo[g] + pco[0]);
                pp[g][0] = v[0]; pp[g][1] = v[1];
            } else if constexpr (ILV && COT >= 4) {
#pragma unroll
                for (int q = 0; q < COT / 4; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(psrc + pbo[g] + pco[0] + 4 * q);
                    pp[g][4 * q] = v[0]; pp[g][4 * q + 1] = v[1]; pp[g][4 * q + 2] = v[2]; pp[g][4 * q + 3] = v[3];
                }
            } else {
#pragma unroll
                for (int b = 0; b < COT; ++b) pp[g][b] = psrc[pbo[g] + pco[b]];
            }
        }
    };

    f32x4 acc[4][COT];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int b = 0; b < COT; ++b) acc[t][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bs[COT];
#pragma unroll
    for (int b = 0; b < COT; ++b) bs[b] = 0.f;

    auto mma_v = [&](const f32x4 (&g4)[NG], const float (&pp)[NG][COT]) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float pv[COT];
#pragma unroll
            for (int b = 0; b < COT; ++b) pv[b] = (FULL || ((bok >> g) & 1u)) ? pp[g][b] : 0.f;   // rows are the reduction index
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int b = 0; b < COT; ++b) acc[t][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(g4[g][t], pv[b], acc[t][b], 0, 0, 0);
            if (cg == 0) {
#pragma unroll
                for (int b = 0; b < COT; ++b) bs[b] += pv[b];
            }
        }
    };

#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) load_v(d, gr[d], pr[d]);
    // (Round 5, measured and put back: the steps as straight-line whole rounds + a separate partial round - no `if` per step, so
    // that hipcc's s_waitcnt insertion stops draining every load at the top of each step (it merges the "previous step ran" and
    // "was skipped" paths there: `s_waitcnt vmcnt(3) .. vmcnt(0)` in front of each step's address arithmetic).  The counted waits
    // came out as intended (vmcnt(16) with 24 loads in flight) and every launch got SLOWER - 74 -> 80, 43 -> 48, 84 -> 88 us, the
    // eight-tile instance spilled - : one vertex of prefetch distance already covers the L2 latency here, and the compiler's
    // freer schedule moved loads in between the MFMAs.  profiles/r05_kernel_experiments.txt.)
    for (int vl = 0; vl < nv; vl += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (vl + d < nv) {                                   // wave-uniform
                load_v(vl + d + DEPTH - 1, gr[(d + DEPTH - 1) % DEPTH], pr[(d + DEPTH - 1) % DEPTH]);
                __builtin_amdgcn_sched_barrier(0);               // prefetch loads stay ahead of the MFMAs
                mma_v(gr[d], pr[d]);
            }
        }
    }

    float* slab = p.slab + (long)rc * p.slab_stride;
#pragma unroll
    for (int b = 0; b < COT; ++b) {
        const int co = ILV ? pco[b] : p.co0 + 16 * b + la;
        if (co >= p.Cout) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = cg * 64 + 16 * kq + 4 * j;
            if (C3) {                                            // quad col/4 = spiral position; element 3 is the zero padding
                const int sp = col >> 2;
                if (sp < p.S) {
                    float* dst = slab + (long)co * p.K + 3 * sp;
                    dst[0] = acc[0][b][j]; dst[1] = acc[1][b][j]; dst[2] = acc[2][b][j];
                }
            } else if (col < p.K) {
                *reinterpret_cast<f32x4*>(slab + (long)co * p.K + col) = (f32x4){acc[0][b][j], acc[1][b][j], acc[2][b][j], acc[3][b][j]};
            }
        }
    }
    if (cg == 0) {
#pragma unroll
        for (int b = 0; b < COT; ++b) {
            float v = bs[b];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int co = ILV ? pco[b] : p.co0 + 16 * b + la;
            if (kq == 0 && co < p.Cout) p.slab[p.bias_off + (long)rc * p.Cout + co] = v;
        }
    }
}

struct WGPlan {
    int log2TB, n_btiles, n_vtiles, nvc, steps_per_block, nrc, ncg, ctw, cot;
    int stream, vpc;    // streaming form: batch slice 1 << log2TB, nvc chunks of vpc vertices, ncg = ceil(K / 64)
    int xg_G, xg_cpg, xg_Vg;      // > 0: one vertex range per XCD (WSParams::vstride == 2)
};

WGPlan plan_wgrad(int B, int R, int S, int Cin, int Cout) {
    WGPlan w;
    w.xg_G = w.xg_cpg = w.xg_Vg = 0;
    const int K = S * Cin;
    static const int tb_pref = sh_env_int("SH_WG_TB", 8, 1, TMW);
    static const int blocks_target = sh_env_int("SH_WG_BLOCKS", 1024, 64, 65536);
    int tb = 1;
    while (tb < B && tb < tb_pref) tb <<= 1;
    w.log2TB = sh_ilog2_floor(tb);
    const int tv = TMW >> w.log2TB;
    w.n_btiles = sh_cdiv(B, tb);
    w.n_vtiles = sh_cdiv(R, tv);
    const int cot = sh_cdiv(Cout < 128 ? Cout : 128, 16);      // more than 128 output channels: launches of <= 128 (streaming form)
    w.cot = cot <= 1 ? 1 : cot <= 2 ? 2 : cot <= 4 ? 4 : 8;
    // column-group width (64 or 128 weight columns): minimise the staged work per row, i.e. column
    // groups x (gathered columns incl. the padding of the last group + the dpre channels re-read by
    // every group); 128 channels x 128 columns would need 360 registers (1 wave/SIMD) -> 64 there.
    {
        const int cp = w.cot * 16;
        const long cost1 = (long)sh_cdiv(K, 64) * (64 + cp), cost2 = (long)sh_cdiv(K, 128) * (128 + cp);
        static const int force = sh_env_int("SH_WG_CTW", 0, 0, 2);
        w.ctw = force ? force : ((w.cot == 8 || cost1 <= cost2) ? 1 : 2);
    }
    w.ncg = sh_cdiv(K, 64 * w.ctw);
    // vertex chunks per batch slice: enough blocks to fill the chip, few enough that the partial
    // slabs stay small, and a table slice that fits its LDS area
    int nvc = blocks_target / (w.ncg * w.n_btiles);
    if (nvc < 1) nvc = 1;
    if (nvc > w.n_vtiles) nvc = w.n_vtiles;
    const int max_steps = WG_TABLE_CAP / (tv * S) > 0 ? WG_TABLE_CAP / (tv * S) : 1;
    w.steps_per_block = sh_cdiv(w.n_vtiles, nvc);
    if (w.steps_per_block > max_steps) w.steps_per_block = max_steps;
    w.nvc = sh_cdiv(w.n_vtiles, w.steps_per_block);
    w.nrc = w.nvc * w.n_btiles;
    // streaming form (wgrad_stream_kernel): work item = (64-column group, batch slice, vertex chunk), one per wave
    static const int stream_on = sh_env_int("SH_WG_STREAM", 1, 0, 1);
    w.stream = (stream_on && (Cin % 4 == 0 || Cin == 3)) ? 1 : 0;
    w.vpc = 0;
    if (w.stream) {
        static const int items_env = sh_env_int("SH_WS_ITEMS", 1024, 64, 1 << 20);
        // the 16 -> 3 layer's slabs are written by the role-swapped kernel (wgrad_thin.hip: one workgroup of four waves per slab, each
        // wave paced by the latency of its own load ring): 512 slabs = two waves per SIMD instead of 340 = 1.33 (45 -> 36 us at 6890
        // vertices, batch 64; the slabs are 1.9 KB each)
        const int items_target = (Cin == 16 && Cout <= 3 && items_env < 1536) ? 1536 : items_env;
        static const int slab_mb = sh_env_int("SH_WS_SLAB_MB", 32, 1, 4096);
        // batch slice: 1 or 4 groups of 4 rows; 8 groups for ONE channel tile (a vertex is then 32 MFMAs instead of 16 behind the
        // same table read and address arithmetic: 69.5 -> 62 us on the level-0 32 -> 16 layer; with two tiles the registers of the
        // deeper prefetch cost more than that: 72 -> 78 us.  SH_WS_TBS=16 keeps the narrow slice everywhere)
        static const int tbs_env = sh_env_int("SH_WS_TBS", 32, 16, 32);
        const int tbs = B <= 4 ? 4 : (tbs_env == 32 && B % 32 == 0 && Cout <= 16 && Cin != 3) ? 32 : 16;
        w.log2TB = sh_ilog2_floor(tbs);
        w.n_btiles = sh_cdiv(B, tbs);
        w.ncg = sh_cdiv(Cin == 3 ? 4 * S : K, 64);            // 3-channel inputs: columns counted in padded quads
        long nrc_t = items_target / w.ncg;
        const long cap = ((long)slab_mb << 20) / ((long)Cout * K * 4);     // partial slabs are written and re-read once
        if (nrc_t > cap) nrc_t = cap;
        if (nrc_t < 1) nrc_t = 1;
        long nvc_t = nrc_t / w.n_btiles;
        if (nvc_t < 1) nvc_t = 1;
        if (nvc_t > R) nvc_t = R;
        int vpc = sh_cdiv(R, (int)nvc_t);
        const int vcap = 2048 / S > 0 ? 2048 / S : 1;          // table lines of a wave: <= 8 KiB of LDS
        if (vpc > vcap) vpc = vcap;
        // The table cap can force more items than the target (long spirals on fine levels).  All of them are resident at
        // once, so what matters is that every CU gets the same number: round the item count up to a whole multiple of
        // the target (measured at 27 554 vertices, spiral 18: 2.1 "rounds" ran at 67 TFLOP/s, 3.0 at ~95).
        {
            const long per_chunk = (long)w.ncg * w.n_btiles;
            const long items = per_chunk * sh_cdiv(R, vpc);
            if (items > items_target) {
                const long rounds = (items + items_target - 1) / items_target;
                long nvc_goal = rounds * items_target / per_chunk;
                const long cap2 = 2 * (((long)slab_mb << 20) / ((long)Cout * K * 4)) / w.n_btiles;
                if (nvc_goal > cap2) nvc_goal = cap2;
                if (nvc_goal > sh_cdiv(R, vpc)) vpc = sh_cdiv(R, (int)nvc_goal);
            }
        }
        w.vpc = vpc;
        w.nvc = sh_cdiv(R, vpc);
        w.nrc = w.nvc * w.n_btiles;
        // one vertex range per XCD (WSParams::xg_*): 1 / 2 / 4 / 8 batch slices, not the layer the role-swapped kernel serves
        static const int xg_on = sh_env_int("SH_WS_XCD_RANGES", 1, 0, 1);
        w.xg_G = w.xg_cpg = w.xg_Vg = 0;
        if (xg_on && (w.n_btiles == 1 || w.n_btiles == 2 || w.n_btiles == 4 || w.n_btiles == 8) && Cout > 3) {
            const int G = 8 / w.n_btiles;
            const int Vg = sh_cdiv(R, G);
            // chunks per XCD: whole "rounds" of the item target (one wave per SIMD = items_target / 8 items per XCD), as many rounds
            // as the general plan above settled on (table cap, slab budget), more if a chunk's table lines would not fit
            const long per_xcd = items_target / 8 > 0 ? items_target / 8 : 1;
            long rounds = ((long)w.ncg * w.n_btiles * w.nvc + items_target / 2) / items_target;
            if (rounds < 1) rounds = 1;
            const int vcap2 = 2048 / S > 0 ? 2048 / S : 1;          // table lines of a wave: <= 8 KiB of LDS
            int cpg = 1;
            for (;; ++rounds) {
                cpg = (int)(rounds * per_xcd / w.ncg);
                if (cpg < 1) cpg = 1;
                if (cpg >= Vg) { cpg = Vg; break; }
                if (sh_cdiv(Vg, cpg) <= vcap2) break;
            }
            w.xg_G = G; w.xg_cpg = cpg; w.xg_Vg = Vg;
            w.vpc = sh_cdiv(Vg, cpg);
            w.nvc = cpg * G;
            w.nrc = 8 * cpg;
        }
    }
    return w;
}

template <int COT, int NG, bool FULL>
int launch_ws(const WSParams& p, hipStream_t st) {
    constexpr int DEPTH = COT <= 2 ? 3 : 2;       // vertices of loads in flight (deeper measured no faster)
    const size_t smem = (size_t)4 * p.vpc * p.S * sizeof(int);
    const int grid = p.grid_main + p.tail_blocks;
    const bool c3 = p.Cin == 3;
    ShProfScope ps(st, "wgrad_stream_kernel<%d, %d, %d, %s, %s>|R=%d B=%d K=%d N=%d grid=%d presum=%d", COT, NG, DEPTH, FULL ? "true" : "false",
                   c3 ? "true" : "false", p.R, p.B, p.K, p.Cout, p.grid_main, p.tail_blocks ? p.tail_rows : 0);
    // interleaved channel tiles (one gradient load per batch group instead of COT): all 16 COT channels of the launch exist and
    // a lane's COT values are COT-float aligned
    static const int ilv_on = sh_env_int("SH_WS_ILV", 1, 0, 1);
    const bool ilv = ilv_on && COT >= 2 && !c3 && p.Cout - p.co0 >= 16 * COT && p.dp_sb % COT == 0 && p.dp_sv % COT == 0 &&
                     (reinterpret_cast<uintptr_t>(p.dpre) & 15) == 0;
    if constexpr (COT >= 2) {
        if (ilv) {
            snprintf(ps.name, sizeof ps.name, "wgrad_stream_kernel<%d, %d, %d, %s, ilv>|R=%d B=%d K=%d N=%d grid=%d presum=%d", COT, NG, DEPTH,
                     FULL ? "true" : "false", p.R, p.B, p.K, p.Cout, p.grid_main, p.tail_blocks ? p.tail_rows : 0);
            SH_LAUNCH_PS(ps, (wgrad_stream_kernel<COT, NG, DEPTH, FULL, false, true>), dim3(grid), dim3(NTHREADS), smem, st, p);
            SH_CHECK_LAUNCH("wgrad_stream");
            return SH_OK;
        }
    }
    if (c3) SH_LAUNCH_PS(ps, (wgrad_stream_kernel<COT, NG, DEPTH, FULL, true>), dim3(grid), dim3(NTHREADS), smem, st, p);
    else SH_LAUNCH_PS(ps, (wgrad_stream_kernel<COT, NG, DEPTH, FULL, false>), dim3(grid), dim3(NTHREADS), smem, st, p);
    SH_CHECK_LAUNCH("wgrad_stream");
    return SH_OK;
}

// ------------------------------------------------------------------------------------------
// Weight gradient in the bf16x3 form (mma_mode SH_MMA_SPLIT3; Cin % 4 == 0, batch % 16 == 0, >= 2 output-channel
// tiles).  Same work items, slabs and output mapping as wgrad_stream_kernel; the reduction runs in steps of 32 rows = two
// vertices x 16 batch entries on v_mfma_f32_16x16x32_bf16: lane (a = lane & 15, kb = lane >> 4) loads the quads
// x[nbr(v + (kb >> 1), s_a)][b0 + 8 (kb & 1) + j][c_a .. c_a+3], j = 0..7 - its eight reduction rows of the four gathered
// columns 4a .. 4a+3 - and the matching eight rows of dpre[.][16 b + a]; column t of the eight quads, split exactly into three
// bf16 terms (sh_split3), is the A operand of the products for the column set {4i + t}, the dpre values the B operand.  Six
// partial products per fp32 product, fp32 accumulation: 24 COT MFMAs of 16 cycles per 32 rows against 32 COT of 32 cycles in
// the exact form; the splits (44 VALU operations per eight values) ride in the MFMAs' issue shadow for >= 4 channel tiles.
template <int COT>
__global__ __launch_bounds__(NTHREADS) void wgrad_split3_kernel(const WSParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x >= p.grid_main) { ws_presum_tail(p); return; }    // tail job (whole workgroups; before any barrier)
    const int lane = threadIdx.x & 63, wave = sh_wave_id();
    int* Tl = reinterpret_cast<int*>(smem) + wave * p.vpc * p.S;
    const WSItem wi = ws_item(p, wave);                          // (log2TB == 4: a slice of 16 batch entries)
    const int rc = wi.rc, cg = wi.cg, b0 = wi.b0, v_begin = wi.v_begin, v_step = wi.v_step, nv = wi.nv;
    const int S = p.S;
    for (int i = lane; i < nv * S; i += 64) {
        const int vl = i / S, j = i - vl * S;
        Tl[i] = p.table[(long)(v_begin + vl * v_step) * S + j];
    }
    __syncthreads();
    if (nv <= 0) {                    // as wgrad_stream_kernel
        if (wi.active) ws_zero_slab(p, rc, cg, lane);
        return;
    }

    const int la = lane & 15, kb = lane >> 4;
    const int k0 = cg * 64 + 4 * la;
    const bool k_in = k0 < p.K;
    const int s_l = k_in ? k0 / p.Cin : 0, c_l = k_in ? k0 - s_l * p.Cin : 0;
    const int dv = kb >> 1;                                      // which of the step's two vertices this lane's rows belong to
    const long xo = (long)(b0 + 8 * (kb & 1)) * p.x_sb + c_l;    // first of this lane's eight batch rows
    const long po = (long)(b0 + 8 * (kb & 1)) * p.dp_sb;
    const float* pb = p.dpre + (long)v_begin * p.dp_sv + po;
    int pco[COT];
#pragma unroll
    for (int b = 0; b < COT; ++b) pco[b] = min(p.co0 + 16 * b + la, p.Cout - 1);

    f32x4 g4[2][8];
    float pr[2][8][COT];
    auto load_step = [&](int vl, f32x4 (&g)[8], float (&pp)[8][COT]) {
        int vv = vl + dv;
        vv = vv < nv ? vv : nv - 1;                              // past the chunk: a valid row; its dpre values are zeroed below
        const float* gsrc = p.x + (long)Tl[vv * S + s_l] * p.x_sv + xo;
        const float* psrc = pb + (long)vv * v_step * p.dp_sv;
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = *reinterpret_cast<const f32x4*>(gsrc + (long)j * p.x_sb);
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int b = 0; b < COT; ++b) pp[j][b] = psrc[(long)j * p.dp_sb + pco[b]];
    };

    f32x4 acc[4][COT];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int b = 0; b < COT; ++b) acc[t][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bs[COT];
#pragma unroll
    for (int b = 0; b < COT; ++b) bs[b] = 0.f;

    auto mma_step = [&](int vl, const f32x4 (&g)[8], const float (&pp)[8][COT]) {
        const bool live = vl + dv < nv;                          // the second vertex of the last, odd step does not exist
        bf16x8 ah[4], am[4], al[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            u32x4 h, m, l;
            sh_split3((f32x4){g[0][t], g[1][t], g[2][t], g[3][t]}, (f32x4){g[4][t], g[5][t], g[6][t], g[7][t]}, h, m, l);
            ah[t] = __builtin_bit_cast(bf16x8, h); am[t] = __builtin_bit_cast(bf16x8, m); al[t] = __builtin_bit_cast(bf16x8, l);
        }
#pragma unroll
        for (int b = 0; b < COT; ++b) {
            float pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = live ? pp[j][b] : 0.f;
            if (cg == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) bs[b] += pv[j];
            }
            u32x4 h, m, l;
            sh_split3((f32x4){pv[0], pv[1], pv[2], pv[3]}, (f32x4){pv[4], pv[5], pv[6], pv[7]}, h, m, l);
            const bf16x8 bh = __builtin_bit_cast(bf16x8, h), bm = __builtin_bit_cast(bf16x8, m), bl = __builtin_bit_cast(bf16x8, l);
#pragma unroll
            for (int t = 0; t < 4; ++t) {                        // smallest terms first
                f32x4 c = acc[t][b];
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[t], bh, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[t], bl, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[t], bm, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[t], bh, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[t], bm, c, 0, 0, 0);
                acc[t][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[t], bh, c, 0, 0, 0);
            }
        }
    };

    load_step(0, g4[0], pr[0]);
    for (int vl = 0; vl < nv; vl += 4) {
        load_step(vl + 2, g4[1], pr[1]);
        __builtin_amdgcn_sched_barrier(0);                       // prefetch loads stay ahead of the MFMAs
        mma_step(vl, g4[0], pr[0]);
        if (vl + 2 >= nv) break;                                 // wave-uniform
        load_step(vl + 4, g4[0], pr[0]);
        __builtin_amdgcn_sched_barrier(0);
        mma_step(vl + 2, g4[1], pr[1]);
    }

    float* slab = p.slab + (long)rc * p.slab_stride;
#pragma unroll
    for (int b = 0; b < COT; ++b) {
        const int co = p.co0 + 16 * b + la;
        if (co >= p.Cout) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = cg * 64 + 16 * kb + 4 * j;
            if (col < p.K)
                *reinterpret_cast<f32x4*>(slab + (long)co * p.K + col) = (f32x4){acc[0][b][j], acc[1][b][j], acc[2][b][j], acc[3][b][j]};
        }
    }
    if (cg == 0) {
#pragma unroll
        for (int b = 0; b < COT; ++b) {
            float v = bs[b];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int co = p.co0 + 16 * b + la;
            if (kb == 0 && co < p.Cout) p.slab[p.bias_off + (long)rc * p.Cout + co] = v;
        }
    }
}

template <int COT>
int launch_ws3(const WSParams& p, hipStream_t st) {
    const size_t smem = (size_t)4 * p.vpc * p.S * sizeof(int);
    const int grid = p.grid_main + p.tail_blocks;
    ShProfScope ps(st, "wgrad_split3_kernel<%d>|R=%d B=%d K=%d N=%d grid=%d presum=%d", COT, p.R, p.B, p.K, p.Cout, p.grid_main,
                   p.tail_blocks ? p.tail_rows : 0);
    SH_LAUNCH_PS(ps, (wgrad_split3_kernel<COT>), dim3(grid), dim3(NTHREADS), smem, st, p);
    SH_CHECK_LAUNCH("wgrad_split3");
    return SH_OK;
}

// does a streaming weight-gradient launch with `cot` channel tiles run in the bf16x3 form?
// 2 and 4 channel tiles (8 would need more than the 512 registers of a one-wave-per-SIMD kernel; one tile: the splits outweigh
// the matrix time saved)
bool ws_uses_split3(int cot, const WSParams& p) {
    static const int s3_min_cot = sh_env_int("SH_S3_WG_MIN_COT", 2, 1, 16);
    const bool full = (p.B & ((1 << p.log2TB) - 1)) == 0;
    // (the three-plane form keeps the exact weight-gradient kernels: they host the pre-sum riders up to four channel tiles, the
    // split ones only with two - measured 30 us per step in their favour once the riders also write plane images)
    static const int p3_wg_split = sh_env_int("SH_P3_WG_SPLIT3", 0, 0, 1);
    const int mode = sh_f32_mma_mode();
    const bool split = mode == SH_MMA_SPLIT3 || (mode == SH_MMA_PLANES3 && p3_wg_split);
    return (cot == 2 || cot == 4) && split && cot >= s3_min_cot && full && p.log2TB == 4 && p.Cin % 4 == 0;
}

template <int COT>
int dispatch_ws(const WSParams& p, hipStream_t st) {
    const bool full = (p.B & ((1 << p.log2TB) - 1)) == 0;
    if ((COT == 2 || COT == 4) && ws_uses_split3(COT, p)) return launch_ws3<(COT == 4 ? 4 : 2)>(p, st);
    if (p.log2TB == 2) return full ? launch_ws<COT, 1, true>(p, st) : launch_ws<COT, 1, false>(p, st);
    if constexpr (COT == 1) {
        if (p.log2TB == 5) return launch_ws<COT, 8, true>(p, st);
    }
    return full ? launch_ws<COT, 4, true>(p, st) : launch_ws<COT, 4, false>(p, st);
}

template <int COT, int CTW>
int launch_wg(const WGParams& p, const WGPlan& w, bool vec4, hipStream_t st) {
    constexpr int KCW = 64 * CTW;
    constexpr int LDG = KCW + 16;
    constexpr int LDP = COT == 1 ? 16 : COT * 16 + 16;
    const int tv = TMW >> w.log2TB;
    const size_t smem = (size_t)2 * TMW * (LDG + LDP) * sizeof(float) + (size_t)w.steps_per_block * tv * p.S * sizeof(int);
    dim3 grid(w.ncg * w.nrc);
    ShProfScope ps(st, "wgrad_kernel<%d, %d, %s>|R=%d B=%d K=%d N=%d grid=%d", COT, CTW, vec4 ? "true" : "false", p.R, p.B, p.K,
                   p.Cout, w.ncg * w.nrc);
    if (vec4) SH_LAUNCH_PS(ps, (wgrad_kernel<COT, CTW, true>), grid, dim3(NTHREADS), smem, st, p);
    else SH_LAUNCH_PS(ps, (wgrad_kernel<COT, CTW, false>), grid, dim3(NTHREADS), smem, st, p);
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

// The weight transposes of a stack's backward pass ride in the launch that opens it (sh_act_backward_tr): workgroups
// grid_main .. of the activation-backward launch are weight_transpose_multi_kernel's workgroups - two launches of a few
// microseconds each become one (a launch boundary costs ~3 us whatever the kernel does).  tr.nd == 0: no rider.
// one workgroup per row (vertex), 32-bit index arithmetic, 16-byte accesses when the channel count allows
template <bool VEC4>
__global__ void act_backward_kernel(const float* __restrict__ dy, long dy_sv, long dy_sb,
                                    const float* __restrict__ y, long y_sv, long y_sb,
                                    float* __restrict__ dp, long dp_sv, long dp_sb,
                                    int B, int R, int C, int act, int zero_row, int grid_main, const MultiTranspose tr,
                                    char* __restrict__ img, long img_vb, long img_bgb) {
    if ((int)blockIdx.x >= grid_main) { weight_transpose_block(tr, (int)blockIdx.x - grid_main); return; }
    const int cq = VEC4 ? C >> 2 : C;                 // elements (or quads) per (row, batch) entry
    const int per_row = B * cq;
    const int parts = (per_row + 255) >> 8;           // work item = 256-element part of a row (coarse levels: few long rows)
    const long items = (long)R * parts;
    for (long it = blockIdx.x; it < items; it += grid_main) {
        const int r = (int)(it / parts), part = (int)(it - (long)r * parts);
        const bool zero = r == zero_row;
        const float* dyr = dy + (long)r * dy_sv;
        const float* yr = y + (long)r * y_sv;
        float* dpr = dp + (long)r * dp_sv;
        for (int i = part * 256 + threadIdx.x; i < per_row && i < (part + 1) * 256; i += 256) {
            const int b = i / cq, c = i - b * cq;
            if (VEC4) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(dyr + (long)b * dy_sb + 4 * c);
                const f32x4 yv = *reinterpret_cast<const f32x4*>(yr + (long)b * y_sb + 4 * c);
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = zero ? 0.f : g[j] * sh_act_grad_from_out(yv[j], act);
                *reinterpret_cast<f32x4*>(dpr + (long)b * dp_sb + 4 * c) = o;
                if (img) {                                         // three-plane image of the row (vertex-major dpre)
                    u32x2 ph, pm, pl;
                    sh_split3_quad(o, ph, pm, pl);
                    const int co = 4 * c;
                    const bool c16 = C == 16;
                    char* d = img + (long)r * img_vb + (long)(b >> 4) * img_bgb +
                              (c16 ? ((co >> 3) * 16 + (b & 15)) * 16 : (co >> 5) * 3072 + (((co & 31) >> 3) * 16 + (b & 15)) * 16) + ((co >> 2) & 1) * 8;
                    const int pb = c16 ? 512 : 1024;
                    *reinterpret_cast<u32x2*>(d) = ph;
                    *reinterpret_cast<u32x2*>(d + pb) = pm;
                    *reinterpret_cast<u32x2*>(d + 2 * pb) = pl;
                }
            } else {
                const float g = dyr[(long)b * dy_sb + c] * sh_act_grad_from_out(yr[(long)b * y_sb + c], act);
                dpr[(long)b * dp_sb + c] = zero ? 0.f : g;
            }
        }
    }
}

// The same for a batch-major gradient / output pair and a vertex-major result (the stack's last layer: x_hat and its
// gradient are [B][rows][C] like the reference's tensors, dpre is [rows][B][C]) with few channels: the element-per-thread
// form reads one 12-byte entry per cache line.  Here a workgroup takes TV consecutive rows: it reads each batch entry's
// TV * C contiguous floats, turns the tile in LDS and writes TV contiguous B * C rows.
constexpr int AB_TV = 16;
__global__ __launch_bounds__(256) void act_backward_turn_kernel(const float* __restrict__ dy, const float* __restrict__ y, long src_sb,
                                                                float* __restrict__ dp, int B, int R, int C, int act, int zero_row,
                                                                int grid_main, const MultiTranspose tr) {
    if ((int)blockIdx.x >= grid_main) { weight_transpose_block(tr, (int)blockIdx.x - grid_main); return; }
    extern __shared__ float tile[];                    // [TV][B * C + 1]
    const int r0 = blockIdx.x * AB_TV, tv = min(AB_TV, R - r0);
    const int seg = tv * C, pitch = B * C + 1;
    for (int i = threadIdx.x; i < B * seg; i += 256) {
        const int b = i / seg, j = i - b * seg;        // j = (row, channel) inside the segment
        const long o = (long)b * src_sb + (long)r0 * C + j;
        const int v = j / C, c = j - v * C;
        const float g = dy[o] * sh_act_grad_from_out(y[o], act);
        tile[v * pitch + b * C + c] = (r0 + v == zero_row) ? 0.f : g;
    }
    __syncthreads();
    const int row_len = B * C;
    for (int i = threadIdx.x; i < tv * row_len; i += 256) {
        const int v = i / row_len, k = i - v * row_len;
        dp[(long)(r0 + v) * row_len + k] = tile[v * pitch + k];
    }
}

}  // namespace

extern "C" {

int sh_spiral_conv_fwd(const float* x, int64_t x_sv, int64_t x_sb, const int32_t* table, const float* weight,
                       const float* bias, float* y, int64_t y_sv, int64_t y_sb, int B, int R, int S, int Cin,
                       int Cout, int act, int zero_row, int mma_mode, sh_stream_t stream) {
    return sh_spiral_conv_fwd_img(x, x_sv, x_sb, table, weight, bias, y, y_sv, y_sb, nullptr, B, R, S, Cin, Cout, act, zero_row, mma_mode, stream);
}

int sh_spiral_conv_fwd_img(const float* x, int64_t x_sv, int64_t x_sb, const int32_t* table, const float* weight,
                           const float* bias, float* y, int64_t y_sv, int64_t y_sb, void* y_planes, int B, int R, int S, int Cin,
                           int Cout, int act, int zero_row, int mma_mode, sh_stream_t stream) {
    SH_REQUIRE(x && table && weight && y, SH_ERR_INVALID_ARG, "sh_spiral_conv_fwd: null pointer");
    SH_REQUIRE(sh_mma_mode_valid(mma_mode), SH_ERR_INVALID_ARG, "sh_spiral_conv_fwd: unknown mma_mode %d", mma_mode);
    ShMmaScope mma_scope(mma_mode);
    SH_REQUIRE(B > 0 && R > 0 && S > 0 && Cin > 0 && Cout > 0, SH_ERR_INVALID_ARG,
               "sh_spiral_conv_fwd: non-positive size B=%d R=%d S=%d Cin=%d Cout=%d", B, R, S, Cin, Cout);
    SH_REQUIRE(act >= SH_ACT_IDENTITY && act <= SH_ACT_TANH, SH_ERR_INVALID_ARG, "sh_spiral_conv_fwd: unknown activation %d", act);
    GGParams p{};
    p.x = x; p.x_sv = x_sv; p.x_sb = x_sb;
    p.table = table; p.w = weight; p.bias = bias;
    p.y = y; p.y_sv = y_sv; p.y_sb = y_sb;
    p.yprev = nullptr;
    p.B = B; p.R = R; p.S = S; p.Cg = Cin; p.Nout = Cout; p.K = S * Cin;
    p.act = act; p.zero_row = zero_row;
    if (y_planes) {
        SH_REQUIRE(y_sb == Cout && y_sv == (int64_t)B * Cout && sh_p3_bytes(1, B, Cout) && (reinterpret_cast<uintptr_t>(y_planes) & 15) == 0,
                   SH_ERR_UNSUPPORTED, "sh_spiral_conv_fwd_img: B=%d Cout=%d has no plane image (vertex-major y; B %% 16 == 0; Cout 16 or %% 32 == 0)", B, Cout);
        p.y_img = static_cast<char*>(y_planes);
        p.img_bgb = Cout == 16 ? 1536 : (long)(Cout / 32) * 3072;
        p.img_vb = p.img_bgb * (B / 16);
    }
    tl_img_written = false;
    const int rc = dispatch_gg<false>(p, static_cast<hipStream_t>(stream));
    if (rc != SH_OK || !y_planes || tl_img_written) return rc;
    return sh_to_p3(y, y_sv, y_sb, y_planes, B, R, Cout, stream);          // the dispatch picked a kernel without the image epilogue
}

int sh_spiral_conv_bwd_data(const float* dpre, int64_t dp_sv, int64_t dp_sb, const int32_t* table_t, const float* weight_t,
                            float* dx, int64_t dx_sv, int64_t dx_sb, const float* yprev, int64_t yp_sv, int64_t yp_sb,
                            int act_prev, int zero_row, int B, int n_in, int S, int Cin, int Cout, int mma_mode, sh_stream_t stream) {
    return sh_spiral_conv_bwd_data_z(dpre, dp_sv, dp_sb, -1, table_t, weight_t, dx, dx_sv, dx_sb, yprev, yp_sv, yp_sb, act_prev, zero_row,
                                     B, n_in, S, Cin, Cout, mma_mode, stream);
}

int sh_spiral_conv_bwd_data_z(const float* dpre, int64_t dp_sv, int64_t dp_sb, int dpre_zero_row, const int32_t* table_t,
                              const float* weight_t, float* dx, int64_t dx_sv, int64_t dx_sb, const float* yprev, int64_t yp_sv,
                              int64_t yp_sb, int act_prev, int zero_row, int B, int n_in, int S, int Cin, int Cout, int mma_mode,
                              sh_stream_t stream) {
    SH_REQUIRE(dpre && table_t && weight_t && dx, SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_data: null pointer");
    SH_REQUIRE(sh_mma_mode_valid(mma_mode), SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_data: unknown mma_mode %d", mma_mode);
    ShMmaScope mma_scope(mma_mode);
    SH_REQUIRE(B > 0 && n_in > 0 && S > 0 && Cin > 0 && Cout > 0, SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_data: non-positive size");
    SH_REQUIRE(act_prev >= SH_ACT_IDENTITY && act_prev <= SH_ACT_TANH, SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_data: unknown activation %d", act_prev);
    GGParams p{};
    p.x = dpre; p.x_sv = dp_sv; p.x_sb = dp_sb;
    p.table = table_t; p.w = weight_t; p.bias = nullptr;
    p.y = dx; p.y_sv = dx_sv; p.y_sb = dx_sb;
    p.yprev = yprev; p.yp_sv = yp_sv; p.yp_sb = yp_sb;
    p.B = B; p.R = n_in; p.S = S; p.Cg = Cout; p.Nout = Cin; p.K = S * Cout;
    p.act = act_prev; p.zero_row = zero_row;
    p.skip_on = dpre_zero_row >= 0 && (long)dpre_zero_row * dp_sv < (1L << 32);
    p.skip_off = p.skip_on ? (unsigned)((long)dpre_zero_row * dp_sv) : 0u;
    return dispatch_gg<true>(p, static_cast<hipStream_t>(stream));
}

int sh_weight_transpose(const float* weight, float* weight_t, int S, int Cin, int Cout, sh_stream_t stream) {
    SH_REQUIRE(weight && weight_t && S > 0 && Cin > 0 && Cout > 0, SH_ERR_INVALID_ARG, "sh_weight_transpose: bad argument");
    const long n = (long)S * Cin * Cout;
    hipLaunchKernelGGL(weight_transpose_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), weight, weight_t, S, Cin, Cout);
    SH_CHECK_LAUNCH("weight_transpose");
    return SH_OK;
}

}  // extern "C"
// slab count of the fp32 weight-gradient plan; shared with the thin-layer kernel in wgrad_thin.hip
int sh_wgrad_f32_nsplit(int B, int R, int S, int Cin, int Cout) { return plan_wgrad(B, R, S, Cin, Cout).nrc; }
extern "C" {

size_t sh_spiral_conv_bwd_wgt_workspace(int B, int R, int S, int Cin, int Cout) {
    if (B <= 0 || R <= 0 || S <= 0 || Cin <= 0 || Cout <= 0) return 0;
    const WGPlan w = plan_wgrad(B, R, S, Cin, Cout);
    return (size_t)w.nrc * ((size_t)Cout * S * Cin + Cout) * sizeof(float);
}

int sh_spiral_conv_bwd_wgt(const float* dpre, int64_t dp_sv, int64_t dp_sb, const float* x, int64_t x_sv, int64_t x_sb,
                           const int32_t* table, float* dW, float* dbias, void* workspace, size_t workspace_bytes, int B,
                           int R, int S, int Cin, int Cout, int mma_mode, sh_stream_t stream) {
    return sh_spiral_conv_bwd_wgt_presum(dpre, dp_sv, dp_sb, x, x_sv, x_sb, table, dW, dbias, workspace, workspace_bytes, nullptr, nullptr,
                                         nullptr, nullptr, nullptr, 0, B, R, S, Cin, Cout, mma_mode, stream);
}

int sh_spiral_conv_bwd_wgt_presum(const float* dpre, int64_t dp_sv, int64_t dp_sb, const float* x, int64_t x_sv, int64_t x_sb,
                                  const int32_t* table, float* dW, float* dbias, void* workspace, size_t workspace_bytes,
                                  const int32_t* sum_rowptr, const int32_t* sum_col, const float* sum_val, float* sum_out,
                                  void* sum_out_planes, int sum_rows, int B, int R, int S, int Cin, int Cout, int mma_mode,
                                  sh_stream_t stream) {
    SH_REQUIRE(dpre && x && table && workspace, SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_wgt: null pointer");
    SH_REQUIRE(sh_mma_mode_valid(mma_mode), SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_wgt: unknown mma_mode %d", mma_mode);
    ShMmaScope mma_scope(mma_mode);
    SH_REQUIRE(sum_rows == 0 || (sum_rows > 0 && sum_rowptr && sum_col && sum_val && sum_out), SH_ERR_INVALID_ARG,
               "sh_spiral_conv_bwd_wgt_presum: incomplete pre-sum job");
    SH_REQUIRE(B > 0 && R > 0 && S > 0 && Cin > 0 && Cout > 0, SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_wgt: non-positive size");
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
    p.log2TB = w.log2TB; p.n_btiles = w.n_btiles; p.n_vtiles = w.n_vtiles; p.nvc = w.nvc; p.steps_per_block = w.steps_per_block; p.ncg = w.ncg;
    const bool vec4 = (Cin % 4 == 0) && (x_sv % 4 == 0) && (x_sb % 4 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0) &&
                      (reinterpret_cast<uintptr_t>(dpre) % 16 == 0);
    int rc;
    if (w.stream) {
        SH_REQUIRE(Cin == 3 || ((x_sv % 4 == 0) && (x_sb % 4 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0)), SH_ERR_UNSUPPORTED,
                   "sh_spiral_conv_bwd_wgt: x must be 16-byte aligned with strides that are multiples of 4 floats");
        SH_REQUIRE(Cin != 3 || (long)2 * (R + 1) * x_sv < 0x7fffffffL, SH_ERR_UNSUPPORTED,
                   "sh_spiral_conv_bwd_wgt: 3-channel input too large for 32-bit row offsets (R %d, row stride %ld)", R, (long)x_sv);
        WSParams s{};
        s.dpre = dpre; s.dp_sv = dp_sv; s.dp_sb = dp_sb; s.x = x; s.x_sv = x_sv; s.x_sb = x_sb; s.table = table;
        s.slab = p.slab; s.slab_stride = p.slab_stride; s.bias_off = p.bias_off;
        s.B = B; s.R = R; s.S = S; s.Cin = Cin; s.Cout = Cout; s.K = p.K;
        s.log2TB = w.log2TB; s.n_btiles = w.n_btiles; s.nvc = w.nvc; s.vpc = w.vpc; s.ncg = w.ncg;
        s.n_items = w.nrc * w.ncg;
        s.grid_main = sh_cdiv(s.n_items, 4);
        static const int vstride = sh_env_int("SH_WS_VSTRIDE", 1, 0, 1);
        s.vstride = vstride;
        if (w.xg_G > 0) {
            s.vstride = 2; s.xg_G = w.xg_G; s.xg_cpg = w.xg_cpg; s.xg_Vg = w.xg_Vg;
            s.grid_main = 8 * sh_cdiv(w.xg_cpg * w.ncg, 4);
        }
        // The pre-sum job rides as tail workgroups of this launch when a second wave of the kernel fits beside the first on
        // a SIMD (exact form: up to four channel tiles; bf16x3 form: two) and the rows take 16-byte accesses; otherwise it is
        // the launch of its own it used to be.  Measured: the seven foldable launches of a step were 58 us + their gaps.
        static const int tail_on = sh_env_int("SH_WS_TAIL", 1, 0, 1), tail_cap = sh_env_int("SH_WS_TAIL_BLOCKS", 768, 1, 1 << 16);
        const bool sum_vec = (Cout % 4 == 0) && (dp_sv % 4 == 0) && (dp_sb % 4 == 0) &&
                             ((reinterpret_cast<uintptr_t>(dpre) | reinterpret_cast<uintptr_t>(sum_out)) % 16 == 0);
        bool fold = tail_on && sum_rows > 0 && sum_vec && Cout <= 128 && !(Cin == 3) &&
                    (ws_uses_split3(w.cot, s) ? w.cot == 2 : w.cot <= 4);
        if (sum_rows > 0 && !fold) {
            rc = sh_spmm_p3(sum_rowptr, sum_col, sum_val, dpre, dp_sv, dp_sb, sum_out, dp_sv, dp_sb, sum_out_planes, nullptr, 0, 0, 0, -1, B,
                            sum_rows, Cout, stream);
            if (rc != SH_OK) return rc;
        }
        if (fold) {
            const long items = (long)sum_rows * (((long)B * (Cout / 4) + 255) / 256);
            s.tail_blocks = (int)(items < tail_cap ? items : tail_cap);
            s.tail_rows = sum_rows; s.tail_rowptr = sum_rowptr; s.tail_col = sum_col; s.tail_val = sum_val; s.tail_y = sum_out;
            if (sum_out_planes) {
                SH_REQUIRE(dp_sb == Cout && dp_sv == (int64_t)B * Cout && sh_p3_bytes(1, B, Cout) && (reinterpret_cast<uintptr_t>(sum_out_planes) & 15) == 0,
                           SH_ERR_UNSUPPORTED, "sh_spiral_conv_bwd_wgt_presum: B=%d Cout=%d has no plane image", B, Cout);
                s.tail_img = static_cast<char*>(sum_out_planes);
                s.tail_img_bgb = Cout == 16 ? 1536 : (long)(Cout / 32) * 3072;
                s.tail_img_vb = s.tail_img_bgb * (B / 16);
            }
        }
        rc = SH_OK;
        for (int co0 = 0; co0 < Cout && rc == SH_OK; co0 += 128) {          // one launch per group of <= 128 output channels
            s.co0 = co0;
            const int tiles = sh_cdiv((Cout - co0 < 128 ? Cout - co0 : 128), 16);
            const int cg_t = tiles <= 1 ? 1 : tiles <= 2 ? 2 : tiles <= 4 ? 4 : 8;
            rc = cg_t == 1 ? dispatch_ws<1>(s, st) : cg_t == 2 ? dispatch_ws<2>(s, st) : cg_t == 4 ? dispatch_ws<4>(s, st) : dispatch_ws<8>(s, st);
        }
    } else {
        if (sum_rows > 0) {
            rc = sh_spmm_p3(sum_rowptr, sum_col, sum_val, dpre, dp_sv, dp_sb, sum_out, dp_sv, dp_sb, sum_out_planes, nullptr, 0, 0, 0, -1, B,
                            sum_rows, Cout, stream);
            if (rc != SH_OK) return rc;
        }
        SH_REQUIRE(Cout <= 128, SH_ERR_UNSUPPORTED,
                   "sh_spiral_conv_bwd_wgt: more than 128 output channels (%d) need input channels that are a multiple of 4 (or 3)", Cout);
#define SH_WG_CASE(C, T) rc = launch_wg<C, T>(p, w, vec4, st)
    if (w.ctw == 1) {
        if (w.cot == 1) SH_WG_CASE(1, 1); else if (w.cot == 2) SH_WG_CASE(2, 1); else if (w.cot == 4) SH_WG_CASE(4, 1); else SH_WG_CASE(8, 1);
    } else {
        if (w.cot == 1) SH_WG_CASE(1, 2); else if (w.cot == 2) SH_WG_CASE(2, 2); else if (w.cot == 4) SH_WG_CASE(4, 2); else SH_WG_CASE(8, 2);
    }
    }
#undef SH_WG_CASE
    if (rc != SH_OK) return rc;
    if (!dW) return SH_OK;            // deferred: the caller reduces several layers at once (.._reduce_multi)
    const long n = p.slab_stride;
    ShProfScope ps(st, "slab_reduce_kernel");
    SH_LAUNCH_PS(ps, slab_reduce_kernel, dim3((unsigned)((n + 63) / 64)), dim3(1024), 0, st, p.slab, p.slab_stride, w.nrc, n, dW);
    if (dbias)
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((Cout + 63) / 64)), dim3(1024), 0, st, p.slab + p.bias_off,
                           (long)Cout, w.nrc, (long)Cout, dbias);
    SH_CHECK_LAUNCH("slab_reduce");
    return SH_OK;
}

static int reduce_multi_impl(int n_layers, const void* const* workspaces, float* const* dW, float* const* dbias, const int* B,
                             const int* R, const int* S, const int* Cin, const int* Cout, bool bf16_plan_all, sh_stream_t stream,
                             const int* kinds = nullptr) {
    SH_REQUIRE(n_layers > 0 && 2 * n_layers <= MR_MAX && workspaces && dW && dbias && B && R && S && Cin && Cout,
               SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_wgt_reduce_multi: bad argument (at most %d layers)", MR_MAX / 2);
    MultiReduce m{};
    int nd = 0, blocks = 0;
    for (int i = 0; i < n_layers; ++i) {
        SH_REQUIRE(workspaces[i] && dW[i], SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_wgt_reduce_multi: null pointer in layer %d", i);
        const int kind = kinds ? kinds[i] : (bf16_plan_all ? 1 : 0);
        SH_REQUIRE(kind >= 0 && kind <= 2, SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_wgt_reduce_multi: unknown plan kind %d of layer %d", kind, i);
        const bool bf16_plan = kind == 1;                      // (slab-walk form below; the three-plane plan has the fp32 plan's slab counts)
        const int nrc = kind == 1 ? sh_wgrad_bf16_nsplit(B[i], R[i], S[i], Cin[i], Cout[i])
                        : kind == 2 ? sh_wgrad_p3_nslab(B[i], R[i], S[i], Cin[i], Cout[i]) : plan_wgrad(B[i], R[i], S[i], Cin[i], Cout[i]).nrc;
        SH_REQUIRE(nrc > 0, SH_ERR_UNSUPPORTED, "sh_spiral_conv_bwd_wgt_reduce_multi: layer %d has no plan of kind %d", i, kind);
        const long stride = (long)Cout[i] * S[i] * Cin[i];
        const float* slab = static_cast<const float*>(workspaces[i]);
        m.slab[nd] = slab; m.stride[nd] = stride; m.nslab[nd] = nrc; m.n[nd] = stride; m.out[nd] = dW[i]; m.block0[nd] = blocks;
        // measured (tools/exp/r05_quick.sh): four outputs per thread win where a workgroup has many slabs to walk (27 554 vertices, 720-760
        // slabs: 33.0 / 22.2 -> 24.5 / 16.5 us) and in the bf16 plan (10.1 / 10.3 -> 9.2 / 9.8), and lose on the headline's 256 slabs (13.4 /
        // 16.8 -> 14.6 / 18.6: a quarter of the workgroups, each as long): 1 = that rule, 0 = never, 2 = always
        static const int vec_on = sh_env_int("SH_SLAB_REDUCE_VEC", 1, 0, 2);
        if ((vec_on == 2 || (vec_on == 1 && (nrc >= 512 || bf16_plan))) && stride % 4 == 0 && ((reinterpret_cast<uintptr_t>(slab) | reinterpret_cast<uintptr_t>(dW[i])) & 15) == 0) {
            m.vec_mask |= 1u << nd;
            blocks += (int)((stride / 4 + 63) / 64);
        } else
            blocks += (int)((stride + 63) / 64);
        ++nd;
        if (dbias[i]) {
            m.slab[nd] = slab + (long)nrc * stride; m.stride[nd] = Cout[i]; m.nslab[nd] = nrc; m.n[nd] = Cout[i];
            m.out[nd] = dbias[i]; m.block0[nd] = blocks;
            blocks += (Cout[i] + 63) / 64; ++nd;
        }
    }
    m.nd = nd;
    hipStream_t st = static_cast<hipStream_t>(stream);
    ShProfScope ps(st, "slab_reduce_multi_kernel");
    SH_LAUNCH_PS(ps, slab_reduce_multi_kernel, dim3((unsigned)blocks), dim3(1024), 0, st, m);
    SH_CHECK_LAUNCH("slab_reduce_multi");
    return SH_OK;
}

int sh_spiral_conv_bwd_wgt_reduce_multi(int n_layers, const void* const* workspaces, float* const* dW, float* const* dbias,
                                        const int* B, const int* R, const int* S, const int* Cin, const int* Cout,
                                        sh_stream_t stream) {
    return reduce_multi_impl(n_layers, workspaces, dW, dbias, B, R, S, Cin, Cout, false, stream);
}

int sh_spiral_conv_bwd_wgt_reduce_multi_kinds(int n_layers, const void* const* workspaces, float* const* dW, float* const* dbias,
                                              const int* B, const int* R, const int* S, const int* Cin, const int* Cout,
                                              const int* kinds, sh_stream_t stream) {
    SH_REQUIRE(kinds, SH_ERR_INVALID_ARG, "sh_spiral_conv_bwd_wgt_reduce_multi_kinds: no kinds");
    return reduce_multi_impl(n_layers, workspaces, dW, dbias, B, R, S, Cin, Cout, false, stream, kinds);
}

int sh_spiral_conv_bwd_wgt_reduce_multi_bf16(int n_layers, const void* const* workspaces, float* const* dW, float* const* dbias,
                                             const int* B, const int* R, const int* S, const int* Cin, const int* Cout,
                                             sh_stream_t stream) {
    return reduce_multi_impl(n_layers, workspaces, dW, dbias, B, R, S, Cin, Cout, true, stream);
}

int sh_weight_transpose_multi(int n_layers, const float* const* weight, float* const* weight_t, const int* S, const int* Cin,
                              const int* Cout, sh_stream_t stream) {
    SH_REQUIRE(n_layers > 0 && n_layers <= MR_MAX && weight && weight_t && S && Cin && Cout, SH_ERR_INVALID_ARG,
               "sh_weight_transpose_multi: bad argument (at most %d layers)", MR_MAX);
    MultiTranspose m{};
    int blocks = 0;
    for (int i = 0; i < n_layers; ++i) {
        SH_REQUIRE(weight[i] && weight_t[i] && S[i] > 0 && Cin[i] > 0 && Cout[i] > 0, SH_ERR_INVALID_ARG,
                   "sh_weight_transpose_multi: bad layer %d", i);
        m.w[i] = weight[i]; m.wt[i] = weight_t[i]; m.S[i] = S[i]; m.Cin[i] = Cin[i]; m.Cout[i] = Cout[i]; m.block0[i] = blocks;
        blocks += (int)(((long)S[i] * Cin[i] * Cout[i] + 255) / 256);
    }
    m.nd = n_layers;
    hipLaunchKernelGGL(weight_transpose_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), m);
    SH_CHECK_LAUNCH("weight_transpose_multi");
    return SH_OK;
}

int sh_act_backward(const float* dy, int64_t dy_sv, int64_t dy_sb, const float* y, int64_t y_sv, int64_t y_sb, float* dpre,
                    int64_t dp_sv, int64_t dp_sb, int B, int R, int C, int act, int zero_row, sh_stream_t stream) {
    return sh_act_backward_tr(dy, dy_sv, dy_sb, y, y_sv, y_sb, dpre, dp_sv, dp_sb, B, R, C, act, zero_row, 0, nullptr, nullptr, nullptr, nullptr,
                              nullptr, stream);
}

int sh_act_backward_tr(const float* dy, int64_t dy_sv, int64_t dy_sb, const float* y, int64_t y_sv, int64_t y_sb, float* dpre,
                       int64_t dp_sv, int64_t dp_sb, int B, int R, int C, int act, int zero_row, int n_layers, const float* const* weight,
                       float* const* weight_t, const int* S, const int* Cin, const int* Cout, sh_stream_t stream) {
    return sh_act_backward_tr_img(dy, dy_sv, dy_sb, y, y_sv, y_sb, dpre, dp_sv, dp_sb, nullptr, B, R, C, act, zero_row, n_layers, weight, weight_t,
                                  S, Cin, Cout, stream);
}

int sh_act_backward_tr_img(const float* dy, int64_t dy_sv, int64_t dy_sb, const float* y, int64_t y_sv, int64_t y_sb, float* dpre,
                           int64_t dp_sv, int64_t dp_sb, void* dpre_planes, int B, int R, int C, int act, int zero_row, int n_layers,
                           const float* const* weight, float* const* weight_t, const int* S, const int* Cin, const int* Cout,
                           sh_stream_t stream) {
    SH_REQUIRE(dy && y && dpre && B > 0 && R > 0 && C > 0, SH_ERR_INVALID_ARG, "sh_act_backward: bad argument");
    SH_REQUIRE(n_layers >= 0 && n_layers <= MR_MAX && (n_layers == 0 || (weight && weight_t && S && Cin && Cout)), SH_ERR_INVALID_ARG,
               "sh_act_backward_tr: bad transpose list (at most %d layers)", MR_MAX);
    MultiTranspose tr{};
    int tr_blocks = 0;
    for (int i = 0; i < n_layers; ++i) {
        SH_REQUIRE(weight[i] && weight_t[i] && S[i] > 0 && Cin[i] > 0 && Cout[i] > 0, SH_ERR_INVALID_ARG, "sh_act_backward_tr: bad layer %d", i);
        tr.w[i] = weight[i]; tr.wt[i] = weight_t[i]; tr.S[i] = S[i]; tr.Cin[i] = Cin[i]; tr.Cout[i] = Cout[i]; tr.block0[i] = tr_blocks;
        tr_blocks += (int)(((long)S[i] * Cin[i] * Cout[i] + 255) / 256);
    }
    tr.nd = n_layers;
    SH_REQUIRE(act >= SH_ACT_IDENTITY && act <= SH_ACT_TANH, SH_ERR_INVALID_ARG, "sh_act_backward: unknown activation %d", act);
    const bool vec = (C % 4 == 0) && ((dy_sv | dy_sb | y_sv | y_sb | dp_sv | dp_sb) % 4 == 0) &&
                     ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dpre)) % 16 == 0);
    const long items = (long)R * (((long)B * (vec ? C / 4 : C) + 255) / 256);
    const int blocks = (int)(items < 8192 ? items : 8192);
    hipStream_t st = static_cast<hipStream_t>(stream);
    // batch-major in, vertex-major out, few channels: turn tiles through LDS
    const bool turn = C <= 8 && dy_sv == C && y_sv == C && dy_sb == y_sb && dy_sb >= (int64_t)R * C && dp_sb == C && dp_sv == (int64_t)B * C &&
                      (size_t)AB_TV * ((size_t)B * C + 1) * sizeof(float) <= 64 * 1024;
    char* img = static_cast<char*>(dpre_planes);
    long img_bgb = 0;
    if (img) {
        SH_REQUIRE(vec && !turn && dp_sb == C && dp_sv == (int64_t)B * C && sh_p3_bytes(1, B, C) && (reinterpret_cast<uintptr_t>(img) & 15) == 0,
                   SH_ERR_UNSUPPORTED, "sh_act_backward_tr_img: B=%d C=%d has no plane image (vertex-major dpre; B %% 16 == 0; C 16 or %% 32 == 0)", B, C);
        img_bgb = C == 16 ? 1536 : (long)(C / 32) * 3072;
    }
    if (turn) {
        const int gm = (R + AB_TV - 1) / AB_TV;
        hipLaunchKernelGGL(act_backward_turn_kernel, dim3(gm + tr_blocks), dim3(256), (size_t)AB_TV * ((size_t)B * C + 1) * sizeof(float), st,
                           dy, y, (long)dy_sb, dpre, B, R, C, act, zero_row, gm, tr);
        SH_CHECK_LAUNCH("act_backward");
        return SH_OK;
    }
    if (vec)
        hipLaunchKernelGGL(act_backward_kernel<true>, dim3(blocks + tr_blocks), dim3(256), 0, st, dy, dy_sv, dy_sb, y, y_sv, y_sb, dpre, dp_sv, dp_sb,
                           B, R, C, act, zero_row, blocks, tr, img, img_bgb * (B / 16), img_bgb);
    else
        hipLaunchKernelGGL(act_backward_kernel<false>, dim3(blocks + tr_blocks), dim3(256), 0, st, dy, dy_sv, dy_sb, y, y_sv, y_sb, dpre, dp_sv, dp_sb,
                           B, R, C, act, zero_row, blocks, tr, (char*)nullptr, 0L, 0L);
    SH_CHECK_LAUNCH("act_backward");
    return SH_OK;
}

}  // extern "C"
