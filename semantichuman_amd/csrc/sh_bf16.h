// bf16 helpers shared by the reduced-precision kernels (BASELINE config 3: bf16 activations and working weights,
// fp32 accumulation, fp32 master weights / gradients / Adam state).
#pragma once
#include "sh_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// fp32 -> bf16, round to nearest even (a plain cast: hipcc emits v_cvt_pk_bf16_f32, which keeps NaN a NaN)
__device__ __forceinline__ bf16x4 sh_to_bf16x4(f32x4 v) {
    return (bf16x4){(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
}
__device__ __forceinline__ f32x4 sh_from_bf16x4(bf16x4 v) { return (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]}; }

// EXACT split of eight fp32 numbers into three bf16 terms each, x = h + m + l (h = rne_bf16(x), m = rne_bf16(x - h),
// l = x - h - m: 8 + 8 + 8 significand bits; both subtractions are exact in fp32), packed as MFMA operand pieces - the
// bf16x3 form of the fp32 matrix products (include/sh_kernels.h, enum sh_mma_mode: an argument of every entry point whose kernel choice depends on it).  ~44 VALU operations.
__device__ __forceinline__ void sh_split3(const f32x4 a, const f32x4 b, u32x4& h, u32x4& m, u32x4& l) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x0 = i < 2 ? a[2 * i] : b[2 * i - 4], x1 = i < 2 ? a[2 * i + 1] : b[2 * i - 3];
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        const bf16x2_t hp = {(__bf16)x0, (__bf16)x1};
        const unsigned hu = __builtin_bit_cast(unsigned, hp);
        const float r0 = x0 - __builtin_bit_cast(float, hu << 16), r1 = x1 - __builtin_bit_cast(float, hu & 0xFFFF0000u);
        const bf16x2_t mp = {(__bf16)r0, (__bf16)r1};
        const unsigned mu = __builtin_bit_cast(unsigned, mp);
        const float t0 = r0 - __builtin_bit_cast(float, mu << 16), t1 = r1 - __builtin_bit_cast(float, mu & 0xFFFF0000u);
        const bf16x2_t lp = {(__bf16)t0, (__bf16)t1};
        h[i] = hu; m[i] = mu; l[i] = __builtin_bit_cast(unsigned, lp);
    }
}

// the same for one quad: four fp32 numbers -> three pieces of four bf16 (8 bytes each), ~22 VALU operations
__device__ __forceinline__ void sh_split3_quad(const f32x4 a, u32x2& h, u32x2& m, u32x2& l) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float x0 = a[2 * i], x1 = a[2 * i + 1];
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        const bf16x2_t hp = {(__bf16)x0, (__bf16)x1};
        const unsigned hu = __builtin_bit_cast(unsigned, hp);
        const float r0 = x0 - __builtin_bit_cast(float, hu << 16), r1 = x1 - __builtin_bit_cast(float, hu & 0xFFFF0000u);
        const bf16x2_t mp = {(__bf16)r0, (__bf16)r1};
        const unsigned mu = __builtin_bit_cast(unsigned, mp);
        const float t0 = r0 - __builtin_bit_cast(float, mu << 16), t1 = r1 - __builtin_bit_cast(float, mu & 0xFFFF0000u);
        const bf16x2_t lp = {(__bf16)t0, (__bf16)t1};
        h[i] = hu; m[i] = mu; l[i] = __builtin_bit_cast(unsigned, lp);
    }
}

// Geometry of a fragment-ordered bf16 weight (the A operand of v_mfma_f32_16x16x32_bf16, one 1-KiB fragment per
// (k-step, 16-row tile)):  frag[ks][nt][lane][j] = W'[16 nt + (lane & 15)][32 ks + 8 (lane >> 4) + j], zero outside W'.
//   rows of W' = output channels, columns k = s * Cg + c   (Cg % 8 == 0)
//                                          k = 4 s + c      (Cg == 3: every neighbour padded to a quad)
struct ShFragGeom { int kp, nks, nt_tot; };
static inline ShFragGeom sh_frag_geom(int S, int Cg, int Nout) {
    ShFragGeom g;
    const int k = Cg == 3 ? 4 * S : S * Cg;
    g.kp = (k + 31) / 32 * 32;
    g.nks = g.kp / 32;
    const int nt = (Nout + 15) / 16;
    g.nt_tot = nt <= 1 ? 1 : nt <= 2 ? 2 : nt <= 4 ? 4 : (nt + 7) / 8 * 8;
    return g;
}
