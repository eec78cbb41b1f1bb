// The three-way bf16 split of fp32 operands shared by the persistent image-conditioner kernels (conv_chain.hip) and the
// large-batch per-layer kernels (conv_bulk.hip): vector typedefs, the LDS layouts of weight images and activation frames, the split
// itself and the six-product accumulation.
#pragma once
#include "nf_conv_core.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define NF_CC_WSLOT 128                              // floats per weight slot: 32 rows x 8 bf16
#define NF_CC_WSLOTS 36                              // (tap, octet) slots of a 32-channel 3 x 3 layer (a ragged first chunk has 9 or 27
                                                     // live slots and zeroes the one behind them; the 1 x 1 layers use <= 24)
#define NF_CC_FP(CS) (16 * (CS))                     // floats per frame plane: 4 octets x CS positions x 16 B

// x = h + m + l with three bf16 values (round to nearest each): pairs, so that the conversions are v_cvt_pk_bf16_f32
__device__ __forceinline__ void nf_cc_split2(f32x2 x, bf16x2& h, bf16x2& m, bf16x2& l) {
    h = __builtin_convertvector(x, bf16x2);
    const f32x2 r1 = x - __builtin_convertvector(h, f32x2);
    m = __builtin_convertvector(r1, bf16x2);
    const f32x2 r2 = r1 - __builtin_convertvector(m, f32x2);
    l = __builtin_convertvector(r2, bf16x2);
}
__device__ __forceinline__ void nf_cc_split1(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

// the eight channels of octet o at frame position f: one 16-byte store per plane
__device__ __forceinline__ void nf_cc_frame_store8(float* F, int CS, int o, int f, const float (&v)[8]) {
    bf16x8 h, m, l;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        bf16x2 h2, m2, l2;
        nf_cc_split2(f32x2{v[j], v[j + 1]}, h2, m2, l2);
        h[j] = h2[0]; h[j + 1] = h2[1];
        m[j] = m2[0]; m[j + 1] = m2[1];
        l[j] = l2[0]; l[j + 1] = l2[1];
    }
    float* p = F + o * 4 * CS + 4 * f;
    *(bf16x8*)(p) = h;
    *(bf16x8*)(p + NF_CC_FP(CS)) = m;
    *(bf16x8*)(p + 2 * NF_CC_FP(CS)) = l;
}

#define NF_CC_MFMA6(AH, AM, AL, BH, BM, BL)                                              \
    do {                                                                                 \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH, BL, acc, 0, 0, 0);             \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AL, BH, acc, 0, 0, 0);             \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AM, BM, acc, 0, 0, 0);             \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH, BM, acc, 0, 0, 0);             \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AM, BH, acc, 0, 0, 0);             \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH, BH, acc, 0, 0, 0);             \
    } while (0)

// eight K values of one weight row, split into the three planes: one 16-byte store per plane
__device__ __forceinline__ void nf_cc_w_put8(float* W8, int WPs, int slot, int row, const float (&v)[8]) {
    bf16x8 h, m, l;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        bf16x2 h2, m2, l2;
        nf_cc_split2(f32x2{v[j], v[j + 1]}, h2, m2, l2);
        h[j] = h2[0]; h[j + 1] = h2[1];
        m[j] = m2[0]; m[j + 1] = m2[1];
        l[j] = l2[0]; l[j + 1] = l2[1];
    }
    float* p = W8 + slot * NF_CC_WSLOT + 4 * row;
    *(bf16x8*)(p) = h;
    *(bf16x8*)(p + WPs) = m;
    *(bf16x8*)(p + 2 * WPs) = l;
}
