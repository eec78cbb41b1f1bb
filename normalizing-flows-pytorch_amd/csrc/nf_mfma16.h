// fp32 matrix-core helpers shared by the register-chained conditioner kernels (flowpp_cond.hip, mlp_chain.hip):
// v_mfma_f32_16x16x4_f32 over 16-row tiles with activations in the row-per-lane layout "R":
//   lane (row = l & 15, g = l >> 4) holds features 16 b + 4 g + r (b = 0..1, r = 0..3) of its row, 8 registers per
//   32-wide vector.  R is the B-operand layout (B[k][j = row], k-group g) AND the C/D layout of out^T = W act^T
//   (D[i = out feature][j = row]: lane col = row, rows 4 g + r), so linears chain register to register.
#pragma once
#include "nf_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define NF_FP_ST 36          // LDS row stride of 32-wide matrices and tiles: 16-byte aligned rows, conflict-free b128 reads

// the 8 entries of a 32-vector in LDS that this lane's registers correspond to (features 16 b + 4 g + r)
__device__ __forceinline__ void nf_fp_ldvec(const float* v, int g, float (&o)[8]) {
    const float4 a = *(const float4*)(v + 4 * g), b = *(const float4*)(v + 16 + 4 * g);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}

__device__ __forceinline__ f32x4 nf_fp_zero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }

// out^T = W act^T:  acc[ob][r] += sum_k W[16 ob + (4 g + r)][colofs + k] act[row][k]   (A = weights, B = activations in R)
template <int NOB>
__device__ __forceinline__ void nf_fp_gemm(const float* W, int st, int colofs, const float (&act)[8], f32x4 (&acc)[NOB], int c16,
                                           int g) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        float4 wv[NOB];
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) wv[ob] = *(const float4*)(W + (16 * ob + c16) * st + colofs + 16 * b + 4 * g);
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
            acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[ob].x, act[4 * b + 0], acc[ob], 0, 0, 0);
            acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[ob].y, act[4 * b + 1], acc[ob], 0, 0, 0);
            acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[ob].z, act[4 * b + 2], acc[ob], 0, 0, 0);
            acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[ob].w, act[4 * b + 3], acc[ob], 0, 0, 0);
        }
    }
}

// data gradient  g_in^T = W^T g^T:  acc[ib][r] += sum_o W[o][colofs + 16 ib + (4 g + r)] gv[row][o],  o over NB 16-blocks
template <int NB>
__device__ __forceinline__ void nf_fp_gemm_d(const float* W, int st, int colofs, const float (&gv)[4 * NB], f32x4 (&acc)[2],
                                             int c16, int g) {
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* wr = W + (16 * b + 4 * g + r) * st + colofs + c16;
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[0], gv[4 * b + r], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[16], gv[4 * b + r], acc[1], 0, 0, 0);
        }
}

// sum over the 32 features of each row (8 in this lane, the rest in the three other groups)
__device__ __forceinline__ float nf_fp_rowsum(float s) {
    s += __shfl_xor(s, 16, NF_WAVE);
    s += __shfl_xor(s, 32, NF_WAVE);
    return s;
}

// A wave's staging tiles are private to it and a wave's LDS operations execute in order: a wave-scope fence (compiler
// ordering) is all a write -> cross-lane read needs.  No block barrier inside the tile loop.
__device__ __forceinline__ void nf_fp_wsync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// row-major store of an R-layout vector into a 16 x 32 tile
__device__ __forceinline__ void nf_fp_store_rows(const float (&v)[8], float* tile, int c16, int g) {
    *(float4*)(tile + c16 * NF_FP_ST + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)(tile + c16 * NF_FP_ST + 16 + 4 * g) = make_float4(v[4], v[5], v[6], v[7]);
}

// column walk of a tile: element [row 4 s + g][column 16 cb + c16], the A / B fragment of a weight-gradient product
template <int NCB>
__device__ __forceinline__ void nf_fp_load_cols(const float* tile, float (&o)[NCB][4], int c16, int g) {
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int s = 0; s < 4; ++s) o[cb][s] = tile[(4 * s + g) * NF_FP_ST + 16 * cb + c16];
}
