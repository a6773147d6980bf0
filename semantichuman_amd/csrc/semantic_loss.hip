// The small loss terms of the semantic training loop as kernels instead of chains of tensor ops:
//
//   joint regression + joint L1     kps = J x[:, :-1] (train_funcs.py:131,161,230,296,336), F.l1_loss(kps[:, keep], target) (:231,:342)
//   part volume ratio               cal_volloss (train_funcs.py:56-71) averaged over the batch (:323-330)
//   latent-norm regulariser         zpartreg (train_funcs.py:145-152)
//   joints <-> bones                kps2skl / skl2kps (utils_SH.py:26-84), bone directions of the pair loss (:449-452)
//   the weighted sum of the terms   `loss = loss + w * term` (train_funcs.py:141-389)
//
// Each is a forward launch pair (partial results, fixed-order final reduction) and ONE backward launch; no atomics, so the
// gradients are bitwise reproducible.  They are launch-bound (a few hundred KB each): what they buy is launches - the
// tensor-op forms cost 6-12 launches forward and as many backward per term, three to four terms per iteration.
#include "sh_common.h"

namespace {

__device__ __forceinline__ float block_sum256(float v, float* red) {       // valid in thread 0; red: >= 4 floats
    v = sh_wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return threadIdx.x == 0 ? (red[0] + red[1]) + (red[2] + red[3]) : 0.f;
}

// kps[b][j][:] = sum_v J[j][v] * x[b][v][:]      one workgroup per (b, j); x rows of 3 floats, batch stride x_bs
__global__ __launch_bounds__(256) void joint_regress_kernel(const float* __restrict__ x, long x_bs, const float* __restrict__ J, int N, int K,
                                                            float* __restrict__ kps) {
    __shared__ float red[4];
    const int b = blockIdx.x / K, j = blockIdx.x - b * K;
    const float* xb = x + (long)b * x_bs;
    const float* Jj = J + (long)j * N;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int v = threadIdx.x; v < N; v += 256) {
        const float w = Jj[v];
        s0 = fmaf(w, xb[3 * v], s0); s1 = fmaf(w, xb[3 * v + 1], s1); s2 = fmaf(w, xb[3 * v + 2], s2);
    }
    const float t0 = block_sum256(s0, red), t1 = block_sum256(s1, red), t2 = block_sum256(s2, red);
    if (threadIdx.x == 0) { float* o = kps + ((long)b * K + j) * 3; o[0] = t0; o[1] = t1; o[2] = t2; }
}

// out[0] = mean_{b, kk, c} | kps[b][keep[kk]][c] - target[b][kk][c] |
__global__ __launch_bounds__(256) void joint_l1_final_kernel(const float* __restrict__ kps, const int* __restrict__ keep, const float* __restrict__ target,
                                                             int B, int K, int Kk, float* __restrict__ out) {
    __shared__ float red[4];
    const int n = B * Kk * 3;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int c = i % 3, kk = (i / 3) % Kk, b = i / (3 * Kk);
        s += fabsf(kps[((long)b * K + keep[kk]) * 3 + c] - target[i]);
    }
    const float t = block_sum256(s, red);
    if (threadIdx.x == 0) out[0] = t / (float)n;
}

// grad[b][v][c] = gscale / n * sum_kk J[keep[kk]][v] * sign(kps[b][keep[kk]][c] - target[b][kk][c]); rows v >= N (dummy) get 0
__global__ __launch_bounds__(256) void joint_l1_bwd_kernel(const float* __restrict__ kps, const int* __restrict__ keep, const float* __restrict__ target,
                                                           const float* __restrict__ J, int B, int N1, int N, int K, int Kk,
                                                           const float* __restrict__ gscale, float* __restrict__ grad) {
    extern __shared__ float sg[];                      // [Kk][3] signs of this batch entry
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < Kk * 3; i += 256) {
        const int kk = i / 3, c = i - 3 * kk;
        const float d = kps[((long)b * K + keep[kk]) * 3 + c] - target[((long)b * Kk + kk) * 3 + c];
        sg[i] = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    }
    __syncthreads();
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= N1) return;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (v < N) {
        for (int kk = 0; kk < Kk; ++kk) {
            const float w = J[(long)keep[kk] * N + v];
            g0 = fmaf(w, sg[3 * kk], g0); g1 = fmaf(w, sg[3 * kk + 1], g1); g2 = fmaf(w, sg[3 * kk + 2], g2);
        }
    }
    const float sc = gscale[0] / (float)(B * Kk * 3);
    float* o = grad + ((long)b * N1 + v) * 3;
    o[0] = g0 * sc; o[1] = g1 * sc; o[2] = g2 * sc;
}

// vol[b][p] = sum over the faces of part p of (a x b) . c      one workgroup per (b, p); pair = 0 rec, 1 gt
__global__ __launch_bounds__(256) void part_volume_kernel(const float* __restrict__ xr, const float* __restrict__ xg, long x_bs,
                                                          const int* __restrict__ faces, const int* __restrict__ pf_ptr,
                                                          const int* __restrict__ pf, int P, float* __restrict__ vol /* [2][B][P] */, int B) {
    __shared__ float red[4];
    const int b = blockIdx.x / P, p = blockIdx.x - b * P;
    const float* r = xr + (long)b * x_bs;
    const float* g = xg + (long)b * x_bs;
    float sr = 0.f, sgv = 0.f;
    for (int e = pf_ptr[p] + threadIdx.x; e < pf_ptr[p + 1]; e += 256) {
        const int f = pf[e];
        const int i0 = 3 * faces[3 * f], i1 = 3 * faces[3 * f + 1], i2 = 3 * faces[3 * f + 2];
        {
            const float ax = r[i0], ay = r[i0 + 1], az = r[i0 + 2], bx = r[i1], by = r[i1 + 1], bz = r[i1 + 2];
            sr += (ay * bz - az * by) * r[i2] + (az * bx - ax * bz) * r[i2 + 1] + (ax * by - ay * bx) * r[i2 + 2];
        }
        {
            const float ax = g[i0], ay = g[i0 + 1], az = g[i0 + 2], bx = g[i1], by = g[i1 + 1], bz = g[i1 + 2];
            sgv += (ay * bz - az * by) * g[i2] + (az * bx - ax * bz) * g[i2 + 1] + (ax * by - ay * bx) * g[i2 + 2];
        }
    }
    const float tr = block_sum256(sr, red), tg = block_sum256(sgv, red);
    if (threadIdx.x == 0) { vol[(long)b * P + p] = tr; vol[(long)(B + b) * P + p] = tg; }
}
// out[0] = (1/P) sum_p mean_b | |vr / vg| - 1 |
__global__ __launch_bounds__(256) void part_volume_final_kernel(const float* __restrict__ vol, int B, int P, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < B * P; i += 256) s += fabsf(fabsf(vol[i] / vol[(long)B * P + i]) - 1.f);
    const float t = block_sum256(s, red);
    if (threadIdx.x == 0) out[0] = t / (float)(B * P);
}
// grad[b][v][:] = sum over corners of v in part faces of coef[b][part] * d vol / d corner;   coef = g/(B P) sign(|q|-1) sign(q) / vg, q = vr / vg
__global__ __launch_bounds__(256) void part_volume_bwd_kernel(const float* __restrict__ xr, long x_bs, const int* __restrict__ faces,
                                                              const int* __restrict__ face_slot /* [F] index into the part list or -1 */,
                                                              const int* __restrict__ vptr, const int* __restrict__ vcorner,
                                                              const float* __restrict__ vol, int B, int P, int N1,
                                                              const float* __restrict__ gscale, float* __restrict__ grad) {
    const int b = blockIdx.y, v = blockIdx.x * 256 + threadIdx.x;
    if (v >= N1) return;
    const float* r = xr + (long)b * x_bs;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    for (int e = vptr[v]; e < vptr[v + 1]; ++e) {
        const int corner = vcorner[e], f = corner / 3, k = corner - 3 * f;
        const int p = face_slot[f];
        if (p < 0) continue;
        const float vr = vol[(long)b * P + p], vg = vol[(long)(B + b) * P + p];
        const float q = vr / vg, aq = fabsf(q);
        const float s1 = aq > 1.f ? 1.f : (aq < 1.f ? -1.f : 0.f), s2 = q > 0.f ? 1.f : (q < 0.f ? -1.f : 0.f);
        const float coef = s1 * s2 / vg;
        // vol_f = (a x b) . c is cyclic: d/da = b x c, d/db = c x a, d/dc = a x b
        const int iu = 3 * faces[3 * f + (k + 1) % 3], iw = 3 * faces[3 * f + (k + 2) % 3];
        const float ux = r[iu], uy = r[iu + 1], uz = r[iu + 2], wx = r[iw], wy = r[iw + 1], wz = r[iw + 2];
        g0 = fmaf(coef, uy * wz - uz * wy, g0); g1 = fmaf(coef, uz * wx - ux * wz, g1); g2 = fmaf(coef, ux * wy - uy * wx, g2);
    }
    const float sc = gscale[0] / (float)(B * P);
    float* o = grad + ((long)b * N1 + v) * 3;
    o[0] = g0 * sc; o[1] = g1 * sc; o[2] = g2 * sc;
}

// zpartreg: loss = mean_{b, i} | |z[b][pi[i]]| / m[b][mi[i]] - 1 |  (relat) or | |z| - m |; one workgroup; also writes d loss / d z
__global__ __launch_bounds__(256) void zpart_kernel(const float* __restrict__ z, const float* __restrict__ m, const int* __restrict__ pi,
                                                    const int* __restrict__ mi, int B, int P, int L, int M, int n, int relat,
                                                    float* __restrict__ out, float* __restrict__ dz /* [B][P][L] or null */,
                                                    const float* __restrict__ gscale) {
    __shared__ float red[4];
    float s = 0.f;
    if (dz)
        for (int i = threadIdx.x; i < B * P * L; i += 256) dz[i] = 0.f;
    __syncthreads();
    for (int i = threadIdx.x; i < B * n; i += 256) {
        const int b = i / n, k = i - b * n;
        const float* zz = z + ((long)b * P + pi[k]) * L;
        float q = 0.f;
        for (int l = 0; l < L; ++l) q = fmaf(zz[l], zz[l], q);
        const float nz = sqrtf(q), mm = m[(long)b * M + mi[k]];
        const float d = relat ? nz / mm - 1.f : nz - mm;
        s += fabsf(d);
        if (dz) {
            const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            const float c = gscale[0] / (float)(B * n) * sg * (relat ? 1.f / mm : 1.f) / nz;      // d|z|/dz = z / |z|
            float* o = dz + ((long)b * P + pi[k]) * L;
            for (int l = 0; l < L; ++l) o[l] = c * zz[l];
        }
    }
    if (out) {
        const float t = block_sum256(s, red);
        if (threadIdx.x == 0) out[0] = t / (float)(B * n);
    }
}

// ---- skeleton bookkeeping of the loop (utils_SH.py:26-84, :449-452) ------------------------------------------------------
// joints -> bones: vec = joint a - (joint b + joint c) / 2 (c == b for two-joint bones: (b + b) / 2 == b exactly), n = |vec|;
// mode 0: (vec / n, n)  1: (vec, n)  2: vec  3: n.  One thread per (batch entry, bone); the arithmetic of the tensor-op form,
// operation for operation (no contraction), so the results are the same bits.
__global__ __launch_bounds__(256) void kps2skl_kernel(const float* __restrict__ kps, int J, const int* __restrict__ i0, const int* __restrict__ i1,
                                                      const int* __restrict__ i2, int B, int nb, int mode, float* __restrict__ out) {
#pragma clang fp contract(off)
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= B * nb) return;
    const int b = t / nb, k = t - b * nb;
    const float* kb = kps + (long)b * J * 3;
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = kb[3 * i0[k] + c] - (kb[3 * i1[k] + c] + kb[3 * i2[k] + c]) / 2.f;
    const float n = sqrtf((v[0] * v[0] + v[2] * v[2]) + v[1] * v[1]);      // the order torch.sum reduces a 3-element row in (measured)
    const int w = mode == 0 || mode == 1 ? 4 : mode == 2 ? 3 : 1;
    float* o = out + (long)t * w;
    if (mode == 0) { o[0] = v[0] / n; o[1] = v[1] / n; o[2] = v[2] / n; o[3] = n; }
    else if (mode == 1) { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = n; }
    else if (mode == 2) { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; }
    else o[0] = n;
}

// bones -> joints, from the root outwards in list order: joint tail[k] = joint head[k] - bone k, joints not assigned yet are the
// origin (utils_SH.py:77-83); then the kept joints are gathered.  The chain is sequential per batch entry; one WAVE per batch
// entry: the entry's bone rows and the head / tail lists are staged in LDS with coalesced loads first (one thread per entry
// walking them in global memory, with the joints in scratch, was a chain of ~70 dependent round trips: 49 us for 16 entries),
// then lanes 0..2 walk the chain, one coordinate each, with the joints in LDS.  The arithmetic is unchanged (same bits).
// mode 0: bone = skl[:3] * skl[3]   1: skl[:3] of 4   2: skl[:3] of 3
constexpr int SKL_MAX_J = 64;
__global__ __launch_bounds__(64) void skl2kps_kernel(const float* __restrict__ skl, const int* __restrict__ head, const int* __restrict__ tail,
                                                     int B, int nb, int n_j, int mode, const int* __restrict__ keep, int n_keep,
                                                     float* __restrict__ out) {
#pragma clang fp contract(off)
    __shared__ float row[SKL_MAX_J * 4];
    __shared__ float kp[SKL_MAX_J * 3];
    __shared__ int hd[SKL_MAX_J], tl[SKL_MAX_J];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int w = mode == 2 ? 3 : 4;
    const float* sb = skl + (long)b * nb * w;
    for (int i = lane; i < nb * w; i += 64) row[i] = sb[i];
    for (int i = lane; i < nb; i += 64) { hd[i] = head[i]; tl[i] = tail[i]; }
    for (int i = lane; i < n_j * 3; i += 64) kp[i] = 0.f;
    __syncthreads();
    if (lane < 3) {
        const int c = lane;
        for (int k = 0; k < nb; ++k) {
            const float bone = mode == 0 ? row[k * w + c] * row[k * w + 3] : row[k * w + c];
            kp[tl[k] * 3 + c] = kp[hd[k] * 3 + c] - bone;
        }
    }
    __syncthreads();
    float* o = out + (long)b * n_keep * 3;
    for (int i = lane; i < n_keep * 3; i += 64) o[i] = kp[keep[i / 3] * 3 + i % 3];
}

// total = t0 * w0 (w0 == 1: t0 itself) + w1 * t1 + ... in sequence - the loop's `loss = loss + w * term` chain as one launch;
// backward: grads[i] = gscale * w_i
constexpr int WS_MAX = 16;
struct WSumArgs { const float* t[WS_MAX]; float w[WS_MAX]; int n; };
__global__ void weighted_sum_kernel(const WSumArgs a, float* __restrict__ out) {
#pragma clang fp contract(off)
    float s = a.w[0] == 1.f ? a.t[0][0] : a.w[0] * a.t[0][0];
    for (int i = 1; i < a.n; ++i) s = s + a.w[i] * a.t[i][0];
    out[0] = s;
}
__global__ void weighted_sum_bwd_kernel(const WSumArgs a, const float* __restrict__ gscale, float* __restrict__ grads) {
    const int i = threadIdx.x;
    if (i < a.n) grads[i] = a.w[i] == 1.f ? gscale[0] : gscale[0] * a.w[i];
}

}  // namespace

extern "C" {

int sh_kps2skl(const float* kps, int B, int J, const int32_t* i0, const int32_t* i1, const int32_t* i2, int n_bones, int mode, float* out,
               sh_stream_t stream) {
    SH_REQUIRE(kps && i0 && i1 && i2 && out && B > 0 && J > 0 && n_bones > 0, SH_ERR_INVALID_ARG, "sh_kps2skl: bad argument");
    SH_REQUIRE(mode >= 0 && mode <= 3, SH_ERR_INVALID_ARG, "sh_kps2skl: unknown mode %d", mode);
    hipLaunchKernelGGL(kps2skl_kernel, dim3((unsigned)sh_cdiv(B * n_bones, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), kps, J, i0, i1, i2,
                       B, n_bones, mode, out);
    SH_CHECK_LAUNCH("kps2skl");
    return SH_OK;
}

int sh_skl2kps(const float* skl, int B, int n_bones, int mode, const int32_t* head, const int32_t* tail, int n_joints, const int32_t* keep,
               int n_keep, float* out, sh_stream_t stream) {
    SH_REQUIRE(skl && head && tail && keep && out && B > 0 && n_bones > 0 && n_keep > 0, SH_ERR_INVALID_ARG, "sh_skl2kps: bad argument");
    SH_REQUIRE(mode >= 0 && mode <= 2, SH_ERR_INVALID_ARG, "sh_skl2kps: unknown mode %d", mode);
    SH_REQUIRE(n_joints > 0 && n_joints <= SKL_MAX_J, SH_ERR_UNSUPPORTED, "sh_skl2kps: %d joints (at most %d)", n_joints, SKL_MAX_J);
    SH_REQUIRE(n_bones <= SKL_MAX_J, SH_ERR_UNSUPPORTED, "sh_skl2kps: %d bones (at most %d)", n_bones, SKL_MAX_J);
    hipLaunchKernelGGL(skl2kps_kernel, dim3((unsigned)B), dim3(64), 0, static_cast<hipStream_t>(stream), skl, head, tail, B, n_bones,
                       n_joints, mode, keep, n_keep, out);
    SH_CHECK_LAUNCH("skl2kps");
    return SH_OK;
}

int sh_weighted_sum(int n, const float* const* terms, const float* weights, float* out, const float* gscale, float* grads, sh_stream_t stream) {
    SH_REQUIRE(n > 0 && n <= WS_MAX && weights, SH_ERR_INVALID_ARG, "sh_weighted_sum: 1..%d terms", WS_MAX);
    SH_REQUIRE((terms && out) || (gscale && grads), SH_ERR_INVALID_ARG, "sh_weighted_sum: nothing to do");
    WSumArgs a{};
    a.n = n;
    for (int i = 0; i < n; ++i) { a.w[i] = weights[i]; a.t[i] = terms ? terms[i] : nullptr; SH_REQUIRE(!out || a.t[i], SH_ERR_INVALID_ARG, "sh_weighted_sum: null term %d", i); }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (out) hipLaunchKernelGGL(weighted_sum_kernel, dim3(1), dim3(1), 0, st, a, out);
    if (grads) hipLaunchKernelGGL(weighted_sum_bwd_kernel, dim3(1), dim3(64), 0, st, a, gscale, grads);
    SH_CHECK_LAUNCH("weighted_sum");
    return SH_OK;
}

int sh_joint_regress(const float* x, int64_t x_bs, const float* J, int B, int N, int K, float* kps, sh_stream_t stream) {
    SH_REQUIRE(x && J && kps && B > 0 && N > 0 && K > 0, SH_ERR_INVALID_ARG, "sh_joint_regress: bad argument");
    hipLaunchKernelGGL(joint_regress_kernel, dim3(B * K), dim3(256), 0, static_cast<hipStream_t>(stream), x, (long)x_bs, J, N, K, kps);
    SH_CHECK_LAUNCH("joint_regress");
    return SH_OK;
}

int sh_joint_l1_loss_fwd(const float* x, int64_t x_bs, const float* J, const int32_t* keep, const float* target, int B, int N, int K,
                         int Kk, float* kps, float* loss, sh_stream_t stream) {
    SH_REQUIRE(x && J && keep && target && kps && loss && B > 0 && N > 0 && K > 0 && Kk > 0, SH_ERR_INVALID_ARG, "sh_joint_l1_loss_fwd: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(joint_regress_kernel, dim3(B * K), dim3(256), 0, st, x, (long)x_bs, J, N, K, kps);
    hipLaunchKernelGGL(joint_l1_final_kernel, dim3(1), dim3(256), 0, st, kps, keep, target, B, K, Kk, loss);
    SH_CHECK_LAUNCH("joint_l1_loss_fwd");
    return SH_OK;
}

int sh_joint_l1_loss_bwd(const float* kps, const int32_t* keep, const float* target, const float* J, int B, int N1, int N, int K, int Kk,
                         const float* gscale, float* grad, sh_stream_t stream) {
    SH_REQUIRE(kps && keep && target && J && gscale && grad && B > 0 && N1 >= N && N > 0 && K > 0 && Kk > 0 && Kk <= 1024, SH_ERR_INVALID_ARG,
               "sh_joint_l1_loss_bwd: bad argument");
    hipLaunchKernelGGL(joint_l1_bwd_kernel, dim3((N1 + 255) / 256, B), dim3(256), (size_t)Kk * 3 * sizeof(float), static_cast<hipStream_t>(stream),
                       kps, keep, target, J, B, N1, N, K, Kk, gscale, grad);
    SH_CHECK_LAUNCH("joint_l1_loss_bwd");
    return SH_OK;
}

int sh_part_volume_loss_fwd(const float* x_rec, const float* x_gt, int64_t x_bs, const int32_t* faces, const int32_t* pf_ptr,
                            const int32_t* pf, int B, int P, float* vol, float* loss, sh_stream_t stream) {
    SH_REQUIRE(x_rec && x_gt && faces && pf_ptr && pf && vol && loss && B > 0 && P > 0, SH_ERR_INVALID_ARG, "sh_part_volume_loss_fwd: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(part_volume_kernel, dim3(B * P), dim3(256), 0, st, x_rec, x_gt, (long)x_bs, faces, pf_ptr, pf, P, vol, B);
    hipLaunchKernelGGL(part_volume_final_kernel, dim3(1), dim3(256), 0, st, vol, B, P, loss);
    SH_CHECK_LAUNCH("part_volume_loss_fwd");
    return SH_OK;
}

int sh_part_volume_loss_bwd(const float* x_rec, int64_t x_bs, const int32_t* faces, const int32_t* face_slot, const int32_t* vptr,
                            const int32_t* vcorner, const float* vol, int B, int P, int N1, const float* gscale, float* grad,
                            sh_stream_t stream) {
    SH_REQUIRE(x_rec && faces && face_slot && vptr && vcorner && vol && gscale && grad && B > 0 && P > 0 && N1 > 0, SH_ERR_INVALID_ARG,
               "sh_part_volume_loss_bwd: bad argument");
    hipLaunchKernelGGL(part_volume_bwd_kernel, dim3((N1 + 255) / 256, B), dim3(256), 0, static_cast<hipStream_t>(stream), x_rec, (long)x_bs, faces,
                       face_slot, vptr, vcorner, vol, B, P, N1, gscale, grad);
    SH_CHECK_LAUNCH("part_volume_loss_bwd");
    return SH_OK;
}

int sh_zpart_reg(const float* z, const float* measure, const int32_t* part_idx, const int32_t* measure_idx, int B, int P, int L, int M,
                 int n, int relat, float* loss, float* dz, const float* gscale, sh_stream_t stream) {
    SH_REQUIRE(z && measure && part_idx && measure_idx && B > 0 && P > 0 && L > 0 && M > 0 && n > 0 && (loss || (dz && gscale)), SH_ERR_INVALID_ARG,
               "sh_zpart_reg: bad argument");
    hipLaunchKernelGGL(zpart_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), z, measure, part_idx, measure_idx, B, P, L, M, n, relat,
                       loss, dz, gscale);
    SH_CHECK_LAUNCH("zpart_reg");
    return SH_OK;
}

}  // extern "C"
