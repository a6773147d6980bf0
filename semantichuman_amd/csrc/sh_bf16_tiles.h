// LDS tile images and MFMA fragment reads shared by the bf16 GEMM-shaped kernels whose reduction index is NOT the
// contiguous one of an operand (dense-layer backward passes, the spiral-conv weight gradient).
//
// v_mfma_f32_16x16x32_bf16 wants, per lane (i = lane & 15, g = lane >> 4), EIGHT reduction elements of row i of an
// operand in one register group.  Two kinds of operand tiles, both 64 (outer index) x 64 (reduction r) bf16 per stage:
//
//   natural     memory [idx][r], r contiguous.  LDS image [idx][64 r] (128-byte rows), 16-byte chunk c of row idx stored at
//               chunk c ^ ((idx >> 1) & 7): staging writes and fragment reads are conflict-free.
//   transposed  memory [r][idx], idx contiguous.  LDS image [k-step = r / 32][idx / 16][r % 32][16 idx] (32-byte rows, the
//               four 1-KiB blocks of a k-step 1056 bytes apart so that the 16-byte staging writes of one memory row spread
//               over all banks), read with ds_read_b64_tr_b16 - the hardware transpose: a 16-lane group hands over a
//               4 (r) x 16 (idx) block and each lane receives the 4 r-values of ITS idx.
//
// The order in which a k-step's 32 reduction elements sit in the lanes is free as long as both operands agree.  Lane
// group g takes r = {4g..4g+3} u {16+4g..16+4g+3}: with that choice the two transposed reads of a 32-lane half cover
// 256 contiguous bytes (all 64 banks once), where r = 8g..8g+7 would be a 2-way conflict.  Natural operands deliver the
// same order with two ds_read_b64.
#pragma once
#include "sh_bf16.h"

constexpr int TG_NAT_BYTES = 64 * 128;                 // natural image of a 64 x 64 tile
constexpr int TG_TR_BLK = 1056;                        // one [32 r][16 idx] block + 32 bytes of bank skew
constexpr int TG_TR_KS = 4 * TG_TR_BLK;                // one k-step (four idx blocks)
constexpr int TG_TR_BYTES = 2 * TG_TR_KS;              // transposed image of a 64 (r) x 64 (idx) tile
constexpr int TG_IMG_BYTES = TG_TR_BYTES;              // >= both; every image slot is this large (16-byte multiple)

// byte offset of 16-byte piece (row, c8) of a staged tile inside its image; row = slow memory index, c8 = 8-element column
__device__ __forceinline__ int tg_nat_piece(int idx, int c8) { return idx * 128 + ((c8 ^ ((idx >> 1) & 7)) << 4); }
__device__ __forceinline__ int tg_tr_piece(int r, int c8) {
    return (r >> 5) * TG_TR_KS + (c8 >> 1) * TG_TR_BLK + (r & 31) * 32 + (c8 & 1) * 16;
}

// fragment of 16-row tile `it` (rows 16 it .. 16 it + 15 of the outer index), k-step kk (0 / 1) of the staged 64 r
__device__ __forceinline__ bf16x8 tg_nat_frag(const char* img, int it, int kk, int lane) {
    const int idx = 16 * it + (lane & 15), g = lane >> 4;
    const int sw = (idx >> 1) & 7;
    const char* row = img + idx * 128 + (g & 1) * 8;
    const int c = 4 * kk + (g >> 1);
    const u32x2 lo = *reinterpret_cast<const u32x2*>(row + ((c ^ sw) << 4));
    const u32x2 hi = *reinterpret_cast<const u32x2*>(row + (((c + 2) ^ sw) << 4));
    const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
    return *reinterpret_cast<const bf16x8*>(&v);
}
__device__ __forceinline__ bf16x8 tg_tr_frag(const char* img, int it, int kk, int lane) {
    const int g = lane >> 4, l16 = lane & 15, q = l16 >> 2, p = l16 & 3;
    const char* a = img + kk * TG_TR_KS + it * TG_TR_BLK + (4 * g + q) * 32 + p * 8;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 16 * 32));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return *reinterpret_cast<const bf16x8*>(&v);
}

// 8 consecutive elements starting at element offset `off` of a bf16 or fp32 array, as one bf16 piece
template <bool F32>
__device__ __forceinline__ u32x4 tg_load8(const void* base, long off) {
    if constexpr (F32) {
        const float* s = static_cast<const float*>(base) + off;
        const f32x4 a = *reinterpret_cast<const f32x4*>(s), b = *reinterpret_cast<const f32x4*>(s + 4);
        const bf16x8 o = {(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3], (__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
        return *reinterpret_cast<const u32x4*>(&o);
    } else {
        return *reinterpret_cast<const u32x4*>(static_cast<const unsigned short*>(base) + off);
    }
}
