// HBM-bound companions of the spiral convolution: sparse mesh re-sampling (D / U and their
// transposes) and the loss / metric reductions.  All are streaming kernels: 16-byte accesses
// where the layout allows, grid capped at ~8 workgroups per CU with grid-stride loops, two-stage
// fixed-order reductions (no atomics -> bitwise reproducible).
#include "sh_bf16.h"

namespace {

constexpr int RED_BLOCKS = 2048;   // partial sums of the first reduction stage

// y[r,b,:] = sum_e val[e] * x[col[e],b,:]   (+ optional act'(yprev) epilogue, zero_row)
// One workgroup per output row (grid-stride over rows): the row's CSR entries are wave-uniform
// scalars, threads sweep the row's B*C elements (contiguous in the vertex-major layout) with
// 32-bit index math only.
// P3: the result's three-plane image (csrc/p3_conv.hip: fragment-major bf16 planes of the rows this launch writes) is
// written beside the fp32 rows - the exact split of the value just stored, by the thread that holds it.
template <bool VEC, bool P3 = false>
__global__ __launch_bounds__(256) void spmm_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                   const float* __restrict__ val, const float* __restrict__ x, long x_sv, long x_sb,
                                                   float* __restrict__ y, long y_sv, long y_sb, const float* __restrict__ yprev,
                                                   long yp_sv, long yp_sb, int act, int zero_row, int B, int rows, int C,
                                                   char* __restrict__ img = nullptr, long img_vb = 0, long img_bgb = 0) {
    const int CW = VEC ? C >> 2 : C;
    const int per_row = B * CW;
    // work item = (row, 256-element part of the row): coarse levels have few, long rows (432 x 32 KB) and would not
    // fill the chip with one workgroup per row
    const int parts = (per_row + 255) >> 8;
    const long items = (long)rows * parts;
    for (long it = blockIdx.x; it < items; it += gridDim.x) {
        const int r = (int)(it / parts), part = (int)(it - (long)r * parts);
        const int e0 = rowptr[r], e1 = rowptr[r + 1];
        const bool zero = r == zero_row;
        for (int j = part * 256 + threadIdx.x; j < per_row && j < (part + 1) * 256; j += 256) {
            const int b = j / CW, cw = j - b * CW;
            const int co = VEC ? 4 * cw : cw;
            const long xo = (long)b * x_sb + co;
            const long yo = (long)r * y_sv + (long)b * y_sb + co;
            if (VEC) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                int e = e0;
                for (; e + 3 < e1; e += 4) {         // 4 independent 16-B loads in flight
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(x + (long)col[e] * x_sv + xo);
                    const f32x4 x1 = *reinterpret_cast<const f32x4*>(x + (long)col[e + 1] * x_sv + xo);
                    const f32x4 x2 = *reinterpret_cast<const f32x4*>(x + (long)col[e + 2] * x_sv + xo);
                    const f32x4 x3 = *reinterpret_cast<const f32x4*>(x + (long)col[e + 3] * x_sv + xo);
                    const float w0 = val[e], w1 = val[e + 1], w2 = val[e + 2], w3 = val[e + 3];
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[k] = fmaf(w3, x3[k], fmaf(w2, x2[k], fmaf(w1, x1[k], fmaf(w0, x0[k], acc[k]))));
                }
                for (; e < e1; ++e) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (long)col[e] * x_sv + xo);
                    const float w = val[e];
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[k] = fmaf(w, xv[k], acc[k]);
                }
                if (yprev) {
                    const f32x4 yv = *reinterpret_cast<const f32x4*>(yprev + (long)r * yp_sv + (long)b * yp_sb + co);
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[k] *= sh_act_grad_from_out(yv[k], act);
                }
                if (zero) acc = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (!P3 || y) *reinterpret_cast<f32x4*>(y + yo) = acc;        // (P3: the image alone when nobody reads the fp32 rows)
                if constexpr (P3) {
                    u32x2 h, m, l;
                    sh_split3_quad(acc, h, m, l);
                    const bool c16 = C == 16;
                    char* d = img + (long)r * img_vb + (long)(b >> 4) * img_bgb +
                              (c16 ? ((co >> 3) * 16 + (b & 15)) * 16 : (co >> 5) * 3072 + (((co & 31) >> 3) * 16 + (b & 15)) * 16) + ((co >> 2) & 1) * 8;
                    const int pb = c16 ? 512 : 1024;
                    *reinterpret_cast<u32x2*>(d) = h;
                    *reinterpret_cast<u32x2*>(d + pb) = m;
                    *reinterpret_cast<u32x2*>(d + 2 * pb) = l;
                }
            } else {
                float acc = 0.f;
                for (int e = e0; e < e1; ++e) acc = fmaf(val[e], x[(long)col[e] * x_sv + xo], acc);
                if (yprev) acc *= sh_act_grad_from_out(yprev[(long)r * yp_sv + (long)b * yp_sb + co], act);
                y[yo] = zero ? 0.f : acc;
            }
        }
    }
}

// The image-writing form with EIGHT channels per thread (round 5): a thread's result is one whole 16-byte piece of each plane
// (the quad form above writes 8-byte halves: two store instructions per piece, each touching half of every line), its fp32
// row segment two 16-byte stores; the sums are the quad form's, entry for entry (bitwise the same rows and image).
__global__ __launch_bounds__(256) void spmm_p3x8_kernel(const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ val,
                                                        const float* __restrict__ x, long x_sv, long x_sb, float* __restrict__ y, long y_sv,
                                                        long y_sb, const float* __restrict__ yprev, long yp_sv, long yp_sb, int act, int zero_row,
                                                        int B, int rows, int C, char* __restrict__ img, long img_vb, long img_bgb) {
    const int CW = C >> 3;
    const int per_row = B * CW;
    const int parts = (per_row + 255) >> 8;
    const long items = (long)rows * parts;
    const bool c16 = C == 16;
    const int pb = c16 ? 512 : 1024;
    for (long it = blockIdx.x; it < items; it += gridDim.x) {
        const int r = (int)(it / parts), part = (int)(it - (long)r * parts);
        const int e0 = rowptr[r], e1 = rowptr[r + 1];
        const int j = part * 256 + threadIdx.x;
        if (j >= per_row) continue;
        const int b = j / CW, co = 8 * (j - b * CW);
        const long xo = (long)b * x_sb + co;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
        int e = e0;
        for (; e + 1 < e1; e += 2) {                         // two entries = four independent 16-byte loads in flight
            const float* s0 = x + (long)col[e] * x_sv + xo;
            const float* s1 = x + (long)col[e + 1] * x_sv + xo;
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(s0), p1 = *reinterpret_cast<const f32x4*>(s0 + 4);
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(s1), q1 = *reinterpret_cast<const f32x4*>(s1 + 4);
            const float w0 = val[e], w1 = val[e + 1];
#pragma unroll
            for (int k = 0; k < 4; ++k) { a0[k] = fmaf(w1, q0[k], fmaf(w0, p0[k], a0[k])); a1[k] = fmaf(w1, q1[k], fmaf(w0, p1[k], a1[k])); }
        }
        if (e < e1) {
            const float* s0 = x + (long)col[e] * x_sv + xo;
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(s0), p1 = *reinterpret_cast<const f32x4*>(s0 + 4);
            const float w0 = val[e];
#pragma unroll
            for (int k = 0; k < 4; ++k) { a0[k] = fmaf(w0, p0[k], a0[k]); a1[k] = fmaf(w0, p1[k], a1[k]); }
        }
        if (yprev) {
            const float* yp = yprev + (long)r * yp_sv + (long)b * yp_sb + co;
            const f32x4 y0 = *reinterpret_cast<const f32x4*>(yp), y1 = *reinterpret_cast<const f32x4*>(yp + 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) { a0[k] *= sh_act_grad_from_out(y0[k], act); a1[k] *= sh_act_grad_from_out(y1[k], act); }
        }
        if (r == zero_row) { a0 = (f32x4){0.f, 0.f, 0.f, 0.f}; a1 = a0; }
        if (y) {
            float* d = y + (long)r * y_sv + (long)b * y_sb + co;
            *reinterpret_cast<f32x4*>(d) = a0;
            *reinterpret_cast<f32x4*>(d + 4) = a1;
        }
        u32x4 h, m, l;
        sh_split3(a0, a1, h, m, l);
        char* d = img + (long)r * img_vb + (long)(b >> 4) * img_bgb +
                  (c16 ? ((co >> 3) * 16 + (b & 15)) * 16 : (co >> 5) * 3072 + (((co & 31) >> 3) * 16 + (b & 15)) * 16);
        *reinterpret_cast<u32x4*>(d) = h;
        *reinterpret_cast<u32x4*>(d + pb) = m;
        *reinterpret_cast<u32x4*>(d + 2 * pb) = l;
    }
}

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = sh_wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0)
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    return t;   // valid in thread 0
}

__global__ void l1_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, long n, float* __restrict__ part) {
    __shared__ float red[16];
    float s = 0.f;
    const long n4 = n >> 2;
    const f32x4* a4 = reinterpret_cast<const f32x4*>(a);
    const f32x4* b4 = reinterpret_cast<const f32x4*>(b);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 d = a4[i] - b4[i];
        s += (fabsf(d[0]) + fabsf(d[1])) + (fabsf(d[2]) + fabsf(d[3]));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) s += fabsf(a[(n4 << 2) + threadIdx.x] - b[(n4 << 2) + threadIdx.x]);
    const float t = block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// out[0] = (sum of part[0..np)) * scale, summed in a fixed order in double
__global__ void final_sum_kernel(const float* __restrict__ part, int np, double scale, float* __restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < np; i += 256) s += (double)part[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(red[0] * scale);
}

__global__ void l1_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, long n, const float* __restrict__ gscale,
                              float* __restrict__ g) {
    const float sc = gscale[0] / (float)n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float d = b[i] - a[i];
        g[i] = d > 0.f ? sc : (d < 0.f ? -sc : 0.f);
    }
}

__global__ void vertex_l2_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, int B, int N1, int N, float scale,
                                         float* __restrict__ part) {
    __shared__ float red[16];
    float s = 0.f;
    const long n = (long)B * N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long bb = i / N, v = i - bb * N;
        const long o = (bb * N1 + v) * 3;
        const float dx = (a[o] - b[o]) * scale, dy = (a[o + 1] - b[o + 1]) * scale, dz = (a[o + 2] - b[o + 2]) * scale;
        s += sqrtf(dx * dx + dy * dy + dz * dz);
    }
    const float t = block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// one 12-byte point / index triple as a single dwordx3 access (4-byte aligned)
struct __attribute__((packed, aligned(4))) P3 { float x, y, z; };
struct __attribute__((packed, aligned(4))) I3 { int a, b, c; };
__device__ __forceinline__ P3 ld3(const float* p, int i) { return *reinterpret_cast<const P3*>(p + 3 * i); }
__device__ __forceinline__ float edge_len3(const P3& a, const P3& b) {
    const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return sqrtf(dx * dx + dy * dy + dz * dz);
}
// same arithmetic as edge_grad below on points already in registers
__device__ __forceinline__ void edge_grad3(const P3& hv, const P3& gv, const P3& ho, const P3& go, float* g) {
    const float dx = hv.x - ho.x, dy = hv.y - ho.y, dz = hv.z - ho.z;
    const float len = sqrtf(dx * dx + dy * dy + dz * dz);
    const float t = edge_len3(gv, go) + 0.00001f;
    const float r = len / t - 1.f;
    const float sg = r > 0.f ? 1.f : (r < 0.f ? -1.f : 0.f);
    if (len > 0.f) {
        const float k = sg / (len * t);
        g[0] += k * dx; g[1] += k * dy; g[2] += k * dz;
    }
}

__device__ __forceinline__ float edge_len(const float* p, int i, int j) {
    const float dx = p[3 * i] - p[3 * j], dy = p[3 * i + 1] - p[3 * j + 1], dz = p[3 * i + 2] - p[3 * j + 2];
    return sqrtf(dx * dx + dy * dy + dz * dz);
}

__global__ void edge_loss_partial_kernel(const float* __restrict__ xh, const float* __restrict__ x, const int* __restrict__ faces,
                                         int B, int N1, int F, float* __restrict__ part) {
    __shared__ float red[16];
    float s = 0.f;
    const long n = (long)B * F;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long bb = i / F;
        const int f = (int)(i - bb * F);
        const int ia = faces[3 * f], ib = faces[3 * f + 1], ic = faces[3 * f + 2];
        const float* ph = xh + bb * N1 * 3;
        const float* pg = x + bb * N1 * 3;
        s += fabsf(edge_len(ph, ia, ib) / (edge_len(pg, ia, ib) + 0.00001f) - 1.f);
        s += fabsf(edge_len(ph, ib, ic) / (edge_len(pg, ib, ic) + 0.00001f) - 1.f);
        s += fabsf(edge_len(ph, ia, ic) / (edge_len(pg, ia, ic) + 0.00001f) - 1.f);
    }
    const float t = block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// d/dp_i | |p_i - p_j| / t - 1 |  accumulated into g[3]
__device__ __forceinline__ void edge_grad(const float* ph, const float* pg, int i, int j, float* g) {
    const float dx = ph[3 * i] - ph[3 * j], dy = ph[3 * i + 1] - ph[3 * j + 1], dz = ph[3 * i + 2] - ph[3 * j + 2];
    const float len = sqrtf(dx * dx + dy * dy + dz * dz);
    const float t = edge_len(pg, i, j) + 0.00001f;
    const float r = len / t - 1.f;
    const float sg = r > 0.f ? 1.f : (r < 0.f ? -1.f : 0.f);
    if (len > 0.f) {
        const float k = sg / (len * t);
        g[0] += k * dx; g[1] += k * dy; g[2] += k * dz;
    }
}

__global__ void edge_loss_bwd_kernel(const float* __restrict__ xh, const float* __restrict__ x, const int* __restrict__ faces,
                                     const int* __restrict__ vptr, const int* __restrict__ vcorner, int B, int N1, int F,
                                     const float* __restrict__ gscale, float* __restrict__ grad) {
    const float sc = gscale[0] / ((float)B * (float)F);
    const long n = (long)B * N1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long bb = i / N1;
        const int v = (int)(i - bb * N1);
        const float* ph = xh + bb * N1 * 3;
        const float* pg = x + bb * N1 * 3;
        float g[3] = {0.f, 0.f, 0.f};
        for (int e = vptr[v]; e < vptr[v + 1]; ++e) {
            const int c = vcorner[e], f = c / 3, kx = c - 3 * f;
            // the two face edges incident to corner kx
            const int o1 = faces[3 * f + (kx + 1) % 3], o2 = faces[3 * f + (kx + 2) % 3];
            edge_grad(ph, pg, v, o1, g);
            edge_grad(ph, pg, v, o2, g);
        }
        grad[i * 3] = sc * g[0]; grad[i * 3 + 1] = sc * g[1]; grad[i * 3 + 2] = sc * g[2];
    }
}

// ---- fused training loss: total = mean|x - x_hat| + w * edge_ratio(x_hat, x) in three launches instead of eleven
// (each small launch costs a ~5 us slot of a replayed graph, whatever it computes).
// One grid computes the partial sums of BOTH terms; part[0..nb) = L1 partials, part[RED_BLOCKS..+nb) = edge partials.
__global__ void recon_partial_kernel(const float* __restrict__ xh, const float* __restrict__ x, const int* __restrict__ faces,
                                     int B, int N1, int F, float* __restrict__ part) {
    __shared__ float red[16];
    float s = 0.f;
    const long n = (long)B * N1 * 3;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += fabsf(x[i] - xh[i]);
    const float t1 = block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = t1;
    __syncthreads();
    s = 0.f;
    const long nf = (long)B * F;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nf; i += (long)gridDim.x * blockDim.x) {
        const long bb = i / F;
        const int f = (int)(i - bb * F);
        const I3 fc = *reinterpret_cast<const I3*>(faces + 3 * f);
        const float* ph = xh + bb * N1 * 3;
        const float* pg = x + bb * N1 * 3;
        const P3 ha = ld3(ph, fc.a), hb = ld3(ph, fc.b), hc = ld3(ph, fc.c), ga = ld3(pg, fc.a), gb = ld3(pg, fc.b), gc = ld3(pg, fc.c);
        s += fabsf(edge_len3(ha, hb) / (edge_len3(ga, gb) + 0.00001f) - 1.f);
        s += fabsf(edge_len3(hb, hc) / (edge_len3(gb, gc) + 0.00001f) - 1.f);
        s += fabsf(edge_len3(ha, hc) / (edge_len3(ga, gc) + 0.00001f) - 1.f);
    }
    const float t2 = block_sum(s, red);
    if (threadIdx.x == 0) part[RED_BLOCKS + blockIdx.x] = t2;
}

// total[0] = rec + w * edge, parts[0] = rec, parts[1] = edge (fixed-order double sums)
__global__ void recon_final_kernel(const float* __restrict__ part, int np, double scale1, double scale2, float w, float* __restrict__ total,
                                   float* __restrict__ parts) {
    __shared__ double red[2][256];
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < np; i += 256) { s1 += (double)part[i]; s2 += (double)part[RED_BLOCKS + i]; }
    red[0][threadIdx.x] = s1; red[1][threadIdx.x] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float rec = (float)(red[0][0] * scale1), edge = (float)(red[1][0] * scale2);
        total[0] = rec + w * edge; parts[0] = rec; parts[1] = edge;
    }
}

// grad[b,v,:] = g * ( sign(x_hat - x) / n  +  w / (B F) * d edge / d x_hat[b,v,:] )
// vnbr lists, per vertex and in corner order, the other end of every incident face edge (two per corner, vptr counts
// corners): the gradient is a gather without the corner -> face -> vertex indirection, four neighbours in flight.
__global__ __launch_bounds__(256) void recon_bwd_kernel(const float* __restrict__ xh, const float* __restrict__ x, const int* __restrict__ vptr,
                                                        const int* __restrict__ vnbr, int B, int N1, int F, float w,
                                                        const float* __restrict__ gscale, float* __restrict__ grad) {
    const float g0 = gscale[0];
    const float s1 = g0 / ((float)B * (float)N1 * 3.f), s2 = g0 * w / ((float)B * (float)F);
    const long n = (long)B * N1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long bb = i / N1;
        const int v = (int)(i - bb * N1);
        const float* ph = xh + bb * N1 * 3;
        const float* pg = x + bb * N1 * 3;
        const int p0 = 2 * vptr[v], p1 = 2 * vptr[v + 1];
        const P3 hv = ld3(ph, v), gv = ld3(pg, v);
        float g[3] = {0.f, 0.f, 0.f};
        for (int e = p0; e < p1; e += 4) {
            int o[4];
            P3 ho[4], go[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = vnbr[e + k < p1 ? e + k : p1 - 1];
#pragma unroll
            for (int k = 0; k < 4; ++k) { ho[k] = ld3(ph, o[k]); go[k] = ld3(pg, o[k]); }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (e + k < p1) edge_grad3(hv, gv, ho[k], go[k], g);
        }
        const float df[3] = {hv.x - gv.x, hv.y - gv.y, hv.z - gv.z};
        P3 o3;
        o3.x = s2 * g[0] + (df[0] > 0.f ? s1 : (df[0] < 0.f ? -s1 : 0.f));
        o3.y = s2 * g[1] + (df[1] > 0.f ? s1 : (df[1] < 0.f ? -s1 : 0.f));
        o3.z = s2 * g[2] + (df[2] > 0.f ? s1 : (df[2] < 0.f ? -s1 : 0.f));
        *reinterpret_cast<P3*>(grad + i * 3) = o3;
    }
}

inline int grid_for(long n, int per_block) {
    long g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return (int)(g < 2048 ? g : 2048);
}

}  // namespace

extern "C" {

int sh_spmm(const int32_t* rowptr, const int32_t* col, const float* val, const float* x, int64_t x_sv, int64_t x_sb, float* y,
            int64_t y_sv, int64_t y_sb, const float* yprev, int64_t yp_sv, int64_t yp_sb, int act_prev, int zero_row, int B,
            int rows, int C, sh_stream_t stream) {
    return sh_spmm_p3(rowptr, col, val, x, x_sv, x_sb, y, y_sv, y_sb, nullptr, yprev, yp_sv, yp_sb, act_prev, zero_row, B, rows, C, stream);
}

int sh_spmm_p3(const int32_t* rowptr, const int32_t* col, const float* val, const float* x, int64_t x_sv, int64_t x_sb, float* y,
               int64_t y_sv, int64_t y_sb, void* y_planes, const float* yprev, int64_t yp_sv, int64_t yp_sb, int act_prev, int zero_row,
               int B, int rows, int C, sh_stream_t stream) {
    SH_REQUIRE(rowptr && col && val && x && (y || y_planes), SH_ERR_INVALID_ARG, "sh_spmm: null pointer");
    SH_REQUIRE(B > 0 && rows > 0 && C > 0, SH_ERR_INVALID_ARG, "sh_spmm: non-positive size");
    SH_REQUIRE(act_prev >= SH_ACT_IDENTITY && act_prev <= SH_ACT_TANH, SH_ERR_INVALID_ARG, "sh_spmm: unknown activation %d", act_prev);
    const bool vec = (C % 4 == 0) && (x_sv % 4 == 0) && (x_sb % 4 == 0) && (y_sv % 4 == 0) && (y_sb % 4 == 0) &&
                     ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) % 16 == 0) &&
                     (!yprev || (yp_sv % 4 == 0 && yp_sb % 4 == 0 && reinterpret_cast<uintptr_t>(yprev) % 16 == 0));
    static const int grid_cap = sh_env_int("SH_SPMM_GRID", 4096, 64, 1 << 20);
    const long items = (long)rows * (((long)B * (vec ? C / 4 : C) + 255) / 256);
    const int grid = (int)(items < grid_cap ? items : grid_cap);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (y_planes) {
        SH_REQUIRE(vec && y_sb == C && y_sv == (int64_t)B * C && sh_p3_bytes(1, B, C) && (reinterpret_cast<uintptr_t>(y_planes) & 15) == 0, SH_ERR_UNSUPPORTED,
                   "sh_spmm_p3: B=%d C=%d has no plane image (vertex-major y; B %% 16 == 0; C == 16 or C %% 32 == 0; 16-byte aligned tensors)", B, C);
        const long bgb = C == 16 ? 1536 : (long)(C / 32) * 3072;
        ShProfScope ps(st, "spmm_kernel<true, p3>|rows=%d B=%d C=%d f32=%d", rows, B, C, y ? 1 : 0);
        // measured (profiles/r05_spmm_x8.txt): the eight-channel form wins on wide rows (C = 128: 396 -> 286 us at batch 1024, 19.1 -> 18.1
        // at 64; C = 64 at batch 1024: 156 -> 138) and loses 1-7 % on the 32-channel levels: 0 = never, 1 = that rule, 2 = always
        static const int x8_mode = sh_env_int("SH_SPMM_P3X8", 1, 0, 2);
        const bool x8 = x8_mode == 2 || (x8_mode == 1 && (C >= 128 || (C == 64 && B >= 256)));
        if (x8) {
            const long items8 = (long)rows * (((long)B * (C / 8) + 255) / 256);
            const int grid8 = (int)(items8 < grid_cap ? items8 : grid_cap);
            SH_LAUNCH_PS(ps, spmm_p3x8_kernel, dim3(grid8), dim3(256), 0, st, rowptr, col, val, x, x_sv, x_sb, y, y_sv, y_sb, yprev, yp_sv, yp_sb,
                         act_prev, zero_row, B, rows, C, static_cast<char*>(y_planes), bgb * (B / 16), bgb);
        } else
        SH_LAUNCH_PS(ps, (spmm_kernel<true, true>), dim3(grid), dim3(256), 0, st, rowptr, col, val, x, x_sv, x_sb, y, y_sv, y_sb,
                     yprev, yp_sv, yp_sb, act_prev, zero_row, B, rows, C, static_cast<char*>(y_planes), bgb * (B / 16), bgb);
        SH_CHECK_LAUNCH("spmm");
        return SH_OK;
    }
    ShProfScope ps(st, "spmm_kernel<%s>|rows=%d B=%d C=%d", vec ? "true" : "false", rows, B, C);
    if (vec)
        SH_LAUNCH_PS(ps, (spmm_kernel<true, false>), dim3(grid), dim3(256), 0, st, rowptr, col, val, x, x_sv, x_sb, y, y_sv, y_sb,
                           yprev, yp_sv, yp_sb, act_prev, zero_row, B, rows, C, nullptr, 0L, 0L);
    else
        SH_LAUNCH_PS(ps, (spmm_kernel<false, false>), dim3(grid), dim3(256), 0, st, rowptr, col, val, x, x_sv, x_sb, y, y_sv, y_sb,
                           yprev, yp_sv, yp_sb, act_prev, zero_row, B, rows, C, nullptr, 0L, 0L);
    SH_CHECK_LAUNCH("spmm");
    return SH_OK;
}

size_t sh_reduce_workspace(void) { return (size_t)RED_BLOCKS * sizeof(float); }

int sh_l1_loss_fwd(const float* a, const float* b, int64_t n, float* loss, void* workspace, sh_stream_t stream) {
    SH_REQUIRE(a && b && loss && workspace && n > 0, SH_ERR_INVALID_ARG, "sh_l1_loss_fwd: bad argument");
    SH_REQUIRE((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) % 16 == 0, SH_ERR_INVALID_ARG,
               "sh_l1_loss_fwd: inputs must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(workspace);
    const int nb = grid_for(n / 4 + 1, 1024) < RED_BLOCKS ? grid_for(n / 4 + 1, 1024) : RED_BLOCKS;
    hipLaunchKernelGGL(l1_partial_kernel, dim3(nb), dim3(256), 0, st, a, b, (long)n, part);
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, st, part, nb, 1.0 / (double)n, loss);
    SH_CHECK_LAUNCH("l1_loss_fwd");
    return SH_OK;
}

int sh_l1_loss_bwd(const float* a, const float* b, int64_t n, const float* gscale, float* grad_b, sh_stream_t stream) {
    SH_REQUIRE(a && b && gscale && grad_b && n > 0, SH_ERR_INVALID_ARG, "sh_l1_loss_bwd: bad argument");
    hipLaunchKernelGGL(l1_bwd_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, static_cast<hipStream_t>(stream), a, b, (long)n, gscale,
                       grad_b);
    SH_CHECK_LAUNCH("l1_loss_bwd");
    return SH_OK;
}

int sh_vertex_l2(const float* a, const float* b, int B, int N1, int N, float scale, float* out, void* workspace,
                 sh_stream_t stream) {
    SH_REQUIRE(a && b && out && workspace && B > 0 && N > 0 && N1 >= N, SH_ERR_INVALID_ARG, "sh_vertex_l2: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(workspace);
    const long n = (long)B * N;
    const int nb = grid_for(n, 1024) < RED_BLOCKS ? grid_for(n, 1024) : RED_BLOCKS;
    hipLaunchKernelGGL(vertex_l2_partial_kernel, dim3(nb), dim3(256), 0, st, a, b, B, N1, N, scale, part);
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, st, part, nb, 1.0 / (double)n, out);
    SH_CHECK_LAUNCH("vertex_l2");
    return SH_OK;
}

int sh_edge_ratio_loss_fwd(const float* x_hat, const float* x, const int32_t* faces, int B, int N1, int F, float* loss,
                           void* workspace, sh_stream_t stream) {
    SH_REQUIRE(x_hat && x && faces && loss && workspace && B > 0 && N1 > 0 && F > 0, SH_ERR_INVALID_ARG,
               "sh_edge_ratio_loss_fwd: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(workspace);
    const long n = (long)B * F;
    const int nb = grid_for(n, 1024) < RED_BLOCKS ? grid_for(n, 1024) : RED_BLOCKS;
    hipLaunchKernelGGL(edge_loss_partial_kernel, dim3(nb), dim3(256), 0, st, x_hat, x, faces, B, N1, F, part);
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, st, part, nb, 1.0 / (double)n, loss);
    SH_CHECK_LAUNCH("edge_ratio_loss_fwd");
    return SH_OK;
}

size_t sh_recon_loss_workspace(void) { return (size_t)2 * RED_BLOCKS * sizeof(float); }

int sh_recon_loss_fwd(const float* x_hat, const float* x, const int32_t* faces, int B, int N1, int F, float edge_w, float* total,
                      float* parts, void* workspace, sh_stream_t stream) {
    SH_REQUIRE(x_hat && x && faces && total && parts && workspace && B > 0 && N1 > 0 && F > 0, SH_ERR_INVALID_ARG,
               "sh_recon_loss_fwd: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(workspace);
    const long n = (long)B * N1 * 3;
    const int nb = grid_for(n, 1024) < RED_BLOCKS ? grid_for(n, 1024) : RED_BLOCKS;
    hipLaunchKernelGGL(recon_partial_kernel, dim3(nb), dim3(256), 0, st, x_hat, x, faces, B, N1, F, part);
    hipLaunchKernelGGL(recon_final_kernel, dim3(1), dim3(256), 0, st, part, nb, 1.0 / (double)n, 1.0 / ((double)B * F), edge_w, total, parts);
    SH_CHECK_LAUNCH("recon_loss_fwd");
    return SH_OK;
}

int sh_recon_loss_bwd(const float* x_hat, const float* x, const int32_t* vptr, const int32_t* vnbr, int B, int N1, int F, float edge_w,
                      const float* gscale, float* grad, sh_stream_t stream) {
    SH_REQUIRE(x_hat && x && vptr && vnbr && gscale && grad && B > 0 && N1 > 0 && F > 0, SH_ERR_INVALID_ARG,
               "sh_recon_loss_bwd: bad argument");
    const long n = (long)B * N1;
    hipLaunchKernelGGL(recon_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x_hat, x, vptr, vnbr, B,
                       N1, F, edge_w, gscale, grad);
    SH_CHECK_LAUNCH("recon_loss_bwd");
    return SH_OK;
}

int sh_edge_ratio_loss_bwd(const float* x_hat, const float* x, const int32_t* faces, const int32_t* vptr, const int32_t* vcorner,
                           int B, int N1, int F, const float* gscale, float* grad, sh_stream_t stream) {
    SH_REQUIRE(x_hat && x && faces && vptr && vcorner && gscale && grad && B > 0 && N1 > 0 && F > 0, SH_ERR_INVALID_ARG,
               "sh_edge_ratio_loss_bwd: bad argument");
    const long n = (long)B * N1;
    hipLaunchKernelGGL(edge_loss_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x_hat, x, faces,
                       vptr, vcorner, B, N1, F, gscale, grad);
    SH_CHECK_LAUNCH("edge_ratio_loss_bwd");
    return SH_OK;
}

}  // extern "C"
