// Flow++ conditioner for density (2-D / small-D) data, fused end to end on the fp32 matrix cores:
//   Linear(I0,32) -> GatedLinear -> LayerNorm -> GatedAttn (ONE position: the attention softmax is identically 1, the
//   block reduces to two linears + a sigmoid gate) -> LayerNorm -> Linear(32, O)
//   flows/coupling.py:142-149, flows/modules.py:500-518 (GatedLinear), :541-578 (GatedAttn), nn.LayerNorm.
// Everything is per-sample independent (LayerNorm, not BatchNorm), so the whole network is ONE launch forward and ONE
// launch backward (the backward recomputes the forward from the 4-byte input instead of saving eight activations).
//
// Work decomposition: a wave owns 32-row tiles.  Activations live in registers in the MFMA *A layout* (lane = row,
// half-wave = feature half, 16 features per lane), every 32x32 GEMM is 16 issues of v_mfma_f32_32x32x2_f32 with the
// B fragment read from LDS-resident weights (row stride +1: conflict-free for both W and W^T walks), and the C/D result
// is turned back into the A layout through a padded LDS tile.  Element-wise work (concat-ELU, gates, LayerNorm) is done
// in the A layout where a row's statistics are an in-lane sum + one cross-half shuffle.
// Replaces ~40 framework kernels forward / ~80 backward per coupling layer (0.5 ms / 1.7 ms at B = 65536).
#include "nf_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define NF_FP_WAVES 4
#define NF_FP_TS 33
#define NF_FP_LNEPS 1.0e-5f
#define NF_FP_BWD_TILES 7   // per wave: conversion, gradient, five stashed activations

struct NfFppW {   // device pointers (forward operands)
    const float *x, *W0, *b0, *Wg, *bg, *ln1g, *ln1b, *pos, *Wq, *bq, *W2, *b2, *ln2g, *ln2b, *W5, *b5;
    float* out;
};

// LDS layout (floats)
struct NfFppL {
    int Wg, Wq, W2, W5, W0, b0, bg, ln1g, ln1b, pos, bq, b2, ln2g, ln2b, b5, tiles, total;
};
__host__ __device__ inline NfFppL nf_fpp_layout(int I0, int tiles_per_wave) {
    NfFppL L;
    int o = 0;
    L.Wg = o; o += 32 * 65;
    L.Wq = o; o += 32 * 33;
    L.W2 = o; o += 64 * 33;
    L.W5 = o; o += 64 * 33;
    L.W0 = o; o += 32 * 4;
    L.b0 = o; o += 32;
    L.bg = o; o += 32;
    L.ln1g = o; o += 32;
    L.ln1b = o; o += 32;
    L.pos = o; o += 32;
    L.bq = o; o += 32;
    L.b2 = o; o += 64;
    L.ln2g = o; o += 32;
    L.ln2b = o; o += 32;
    L.b5 = o; o += 64;
    L.tiles = o; o += NF_FP_WAVES * tiles_per_wave * 32 * NF_FP_TS;
    L.total = o;
    (void)I0;
    return L;
}

__device__ __forceinline__ int nf_fp_cdrow(int r, int hs) { return (r & 3) + 8 * (r >> 2) + 4 * hs; }
__device__ __forceinline__ float nf_elu(float x) { return x > 0.f ? x : expf(x) - 1.f; }          // F.elu, alpha = 1
__device__ __forceinline__ float nf_elu_grad(float x) { return x > 0.f ? 1.f : expf(x); }
__device__ __forceinline__ float nf_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ void nf_fpp_stage(const NfFppW& w, float* sm, const NfFppL& L, int I0, int O) {
    for (int i = threadIdx.x; i < 32 * 64; i += blockDim.x) sm[L.Wg + (i >> 6) * 65 + (i & 63)] = w.Wg[i];
    for (int i = threadIdx.x; i < 32 * 32; i += blockDim.x) sm[L.Wq + (i >> 5) * 33 + (i & 31)] = w.Wq[i];
    for (int i = threadIdx.x; i < 64 * 32; i += blockDim.x) sm[L.W2 + (i >> 5) * 33 + (i & 31)] = w.W2[i];
    for (int i = threadIdx.x; i < 64 * 32; i += blockDim.x) sm[L.W5 + (i >> 5) * 33 + (i & 31)] = (i >> 5) < O ? w.W5[i] : 0.f;
    for (int i = threadIdx.x; i < 32 * 4; i += blockDim.x) sm[L.W0 + i] = ((i & 3) < I0) ? w.W0[(i >> 2) * I0 + (i & 3)] : 0.f;
    for (int i = threadIdx.x; i < 32; i += blockDim.x) {
        sm[L.b0 + i] = w.b0[i]; sm[L.bg + i] = w.bg[i]; sm[L.ln1g + i] = w.ln1g[i]; sm[L.ln1b + i] = w.ln1b[i];
        sm[L.pos + i] = w.pos[i]; sm[L.bq + i] = w.bq[i]; sm[L.ln2g + i] = w.ln2g[i]; sm[L.ln2b + i] = w.ln2b[i];
    }
    for (int i = threadIdx.x; i < 64; i += blockDim.x) {
        sm[L.b2 + i] = w.b2[i];
        sm[L.b5 + i] = i < O ? w.b5[i] : 0.f;
    }
    __syncthreads();
}

// acc += A(a, this lane's 16 features) x W[o][koff + k]^T, W in LDS with row stride `st`; output columns o = ooff + (l&31)
__device__ __forceinline__ f32x16 nf_fp_gemm(f32x16 acc, const float (&a)[16], const float* W, int st, int ooff, int koff,
                                             int c32, int hs) {
    const float* wr = W + (ooff + c32) * st + koff + hs * 16;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], wr[kk], acc, 0, 0, 0);
    return acc;
}
// data-gradient GEMM: acc += A(g) x W[koff + k][ioff + i]  (i.e. g W), B fragment = column walk of W
__device__ __forceinline__ f32x16 nf_fp_gemm_t(f32x16 acc, const float (&a)[16], const float* W, int st, int koff, int ioff,
                                               int c32, int hs) {
    const float* wc = W + (koff + hs * 16) * st + ioff + c32;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], wc[kk * st], acc, 0, 0, 0);
    return acc;
}

// The staging tiles are private to one wave, and a wave's LDS operations execute in order: a wave-scope fence (compiler
// ordering) is all a write -> cross-lane read needs.  No block barrier, so waves run their tiles independently.
__device__ __forceinline__ void nf_fp_wsync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ f32x16 nf_fp_zero() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

// C/D layout (+ per-column bias) -> A layout through a padded (wave-private) tile
__device__ __forceinline__ void nf_fp_cd_to_a(const f32x16& acc, const float* bias, float* tile, float (&a)[16], int c32,
                                              int hs) {
    const float bv = bias != nullptr ? bias[c32] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[nf_fp_cdrow(r, hs) * NF_FP_TS + c32] = acc[r] + bv;
    nf_fp_wsync();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) a[kk] = tile[c32 * NF_FP_TS + hs * 16 + kk];
    nf_fp_wsync();
}

// row statistics of an A-layout vector: mean and 1/sqrt(biased var + eps)
__device__ __forceinline__ void nf_fp_rowstats(const float (&v)[16], float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) s += v[kk];
    s += __shfl_xor(s, 32, NF_WAVE);
    mean = s * (1.f / 32.f);
    float q = 0.f;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) { const float d = v[kk] - mean; q = fmaf(d, d, q); }
    q += __shfl_xor(q, 32, NF_WAVE);
    rstd = 1.f / sqrtf(q * (1.f / 32.f) + NF_FP_LNEPS);
}

// the whole forward of one 32-row tile in the A layout; keeps what the backward needs in the caller's registers
struct NfFppFwd {
    float h0[16], u[16], h2[16], q[16], y2[16], a2[16], h4[16];
    float xh1[16], xh2[16];     // LayerNorm-normalised values
    float rstd1, rstd2;
};

// STASH: also leave the five activations the weight-gradient GEMMs multiply with (elu(h0), elu(-h0), t, q, h4) as
// row-major tiles stash[0..4] so the backward does not carry them in registers.
template <bool STASH>
__device__ __forceinline__ void nf_fpp_forward_tile(const float* sm, const NfFppL& L, const float* __restrict__ x, int I0,
                                                    int64_t row, bool rv, float* tile, float* stash, int c32, int hs,
                                                    NfFppFwd& f) {
    float xin[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < I0; ++i) xin[i] = rv ? x[row * I0 + i] : 0.f;
    float c0[16], c1[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        const int k = hs * 16 + kk;
        float h = sm[L.b0 + k];
#pragma unroll
        for (int i = 0; i < 4; ++i) h = fmaf(sm[L.W0 + k * 4 + i], xin[i], h);
        f.h0[kk] = h;
        c0[kk] = nf_elu(h);                                            // concat-ELU (modules.py:509)
        c1[kk] = nf_elu(-h);
    }
    if (STASH) {
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            stash[0 * 32 * NF_FP_TS + c32 * NF_FP_TS + hs * 16 + kk] = c0[kk];
            stash[1 * 32 * NF_FP_TS + c32 * NF_FP_TS + hs * 16 + kk] = c1[kk];
        }
    }
    f32x16 acc = nf_fp_zero();
    acc = nf_fp_gemm(acc, c0, sm + L.Wg, 65, 0, 0, c32, hs);
    acc = nf_fp_gemm(acc, c1, sm + L.Wg, 65, 0, 32, c32, hs);
    nf_fp_cd_to_a(acc, sm + L.bg, tile, f.u, c32, hs);
    float h1[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) h1[kk] = f.h0[kk] + nf_elu(f.u[kk]) * nf_sigmoid(nf_elu(-f.u[kk]));   // modules.py:513-518
    float mean;
    nf_fp_rowstats(h1, mean, f.rstd1);
    float t[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        const int k = hs * 16 + kk;
        f.xh1[kk] = (h1[kk] - mean) * f.rstd1;
        f.h2[kk] = f.xh1[kk] * sm[L.ln1g + k] + sm[L.ln1b + k];
        t[kk] = f.h2[kk] + sm[L.pos + k];                              // modules.py:569
        if (STASH) stash[2 * 32 * NF_FP_TS + c32 * NF_FP_TS + hs * 16 + kk] = t[kk];
    }
    acc = nf_fp_gemm(nf_fp_zero(), t, sm + L.Wq, 33, 0, 0, c32, hs);
    nf_fp_cd_to_a(acc, sm + L.bq, tile, f.q, c32, hs);
    if (STASH) {
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) stash[3 * 32 * NF_FP_TS + c32 * NF_FP_TS + hs * 16 + kk] = f.q[kk];
    }
    acc = nf_fp_gemm(nf_fp_zero(), f.q, sm + L.W2, 33, 0, 0, c32, hs);
    nf_fp_cd_to_a(acc, sm + L.b2, tile, f.y2, c32, hs);
    acc = nf_fp_gemm(nf_fp_zero(), f.q, sm + L.W2, 33, 32, 0, c32, hs);
    nf_fp_cd_to_a(acc, sm + L.b2 + 32, tile, f.a2, c32, hs);
    float h3[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) h3[kk] = f.h2[kk] + f.y2[kk] * nf_sigmoid(f.a2[kk]);                 // modules.py:574-578
    nf_fp_rowstats(h3, mean, f.rstd2);
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        const int k = hs * 16 + kk;
        f.xh2[kk] = (h3[kk] - mean) * f.rstd2;
        f.h4[kk] = f.xh2[kk] * sm[L.ln2g + k] + sm[L.ln2b + k];
        if (STASH) stash[4 * 32 * NF_FP_TS + c32 * NF_FP_TS + hs * 16 + kk] = f.h4[kk];
    }
}

__global__ void __launch_bounds__(NF_FP_WAVES * NF_WAVE) k_flowpp_cond_fwd(NfFppW w, int64_t N, int I0, int O, int64_t tiles,
                                                                           int iters) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const NfFppL L = nf_fpp_layout(I0, 1);
    nf_fpp_stage(w, sm, L, I0, O);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, c32 = lane & 31, hs = lane >> 5;
    float* tile = sm + L.tiles + wid * 32 * NF_FP_TS;
    for (int it = 0; it < iters; ++it) {
        const int64_t t = ((int64_t)it * gridDim.x + blockIdx.x) * NF_FP_WAVES + wid;      // may be >= tiles: row guards
        const int64_t row0 = t * 32;
        const int64_t row = row0 + c32;
        NfFppFwd f;
        nf_fpp_forward_tile<false>(sm, L, w.x, I0, row, row < N && t < tiles, tile, nullptr, c32, hs, f);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            if (nt * 32 < O) {                                                             // block-uniform
                f32x16 acc = nf_fp_gemm(nf_fp_zero(), f.h4, sm + L.W5, 33, nt * 32, 0, c32, hs);
                const int o = nt * 32 + c32;
                if (o < O && t < tiles) {
                    const float bv = sm[L.b5 + o];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t gr = row0 + nf_fp_cdrow(r, hs);
                        if (gr < N) w.out[gr * O + o] = acc[r] + bv;
                    }
                }
            }
        }
    }
}

extern "C" int nf_flowpp_cond_fwd(const float* x, const float* W0, const float* b0, const float* Wg, const float* bg,
                                  const float* ln1_g, const float* ln1_b, const float* pos, const float* Wq, const float* bq,
                                  const float* W2, const float* b2, const float* ln2_g, const float* ln2_b, const float* W5,
                                  const float* b5, float* out, int64_t N, int I0, int O, nf_stream_t stream) {
    if (I0 < 1 || I0 > 4 || O < 1 || O > 64) return NF_E_BADARG;
    if (N <= 0) return N == 0 ? 0 : NF_E_BADARG;
    NfFppW w{x, W0, b0, Wg, bg, ln1_g, ln1_b, pos, Wq, bq, W2, b2, ln2_g, ln2_b, W5, b5, out};
    const int64_t tiles = (N + 31) / 32;
    int64_t g = (tiles + NF_FP_WAVES - 1) / NF_FP_WAVES;
    if (g > 1024) g = 1024;
    const int iters = (int)((tiles + g * NF_FP_WAVES - 1) / (g * NF_FP_WAVES));
    const NfFppL L = nf_fpp_layout(I0, 1);
    hipLaunchKernelGGL(k_flowpp_cond_fwd, dim3((unsigned)g), dim3(NF_FP_WAVES * NF_WAVE), (size_t)L.total * sizeof(float),
                       (hipStream_t)stream, w, N, I0, O, tiles, iters);
    NF_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// backward: recompute the forward of the tile, then back-propagate in the A layout.
//   data gradients : g W  via nf_fp_gemm_t (B fragment = column walk of the LDS-resident weight)
//   weight gradients: G^T Act over the tile's 32 rows (A = column walk of G's row-major LDS tile, B = column walk of
//                     Act's tile), accumulated in registers across the wave's tiles, reduced over the block at the end
//   vector gradients (biases, LayerNorm gamma/beta, pos_emb, W0/b0): column sums of row-major LDS tiles
// ---------------------------------------------------------------------------------------------------------------
struct NfFppG {   // gradient destinations, all ACCUMULATED (+=): zero-filled temporaries or .grad buffers
    const float* g_out;          // (N, O)
    float *g_x;                  // (N, I0) written, nullable
    float *g_W0, *g_b0, *g_Wg, *g_bg, *g_ln1g, *g_ln1b, *g_pos, *g_Wq, *g_bq, *g_W2, *g_b2, *g_ln2g, *g_ln2b, *g_W5, *g_b5;
};

// row-major store of an A-layout vector into a tile (no barrier)
__device__ __forceinline__ void nf_fp_store_rows(const float (&v)[16], float* tile, int c32, int hs) {
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) tile[c32 * NF_FP_TS + hs * 16 + kk] = v[kk];
}
// column walk: element [row = hs*16 + kk][col = c32]
__device__ __forceinline__ void nf_fp_load_cols(const float* tile, float (&v)[16], int c32, int hs) {
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) v[kk] = tile[(hs * 16 + kk) * NF_FP_TS + c32];
}
__device__ __forceinline__ float nf_fp_sum16(const float (&v)[16]) {
    float s = 0.f;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) s += v[kk];
    return s;
}
// per-lane partial column sum (column c32, rows of this half) of an A-layout vector
__device__ __forceinline__ float nf_fp_colsum(const float (&v)[16], float* tile, int c32, int hs) {
    nf_fp_store_rows(v, tile, c32, hs);
    nf_fp_wsync();
    float t[16];
    nf_fp_load_cols(tile, t, c32, hs);
    const float s = nf_fp_sum16(t);
    nf_fp_wsync();
    return s;
}
// weight-gradient tile: acc[o][i] += sum_rows G[row][o] Act[row][i]; returns this lane's partial bias sum (column o = c32)
__device__ __forceinline__ float nf_fp_wgrad(f32x16& acc, const float* Gt, const float* At, int c32, int hs) {
    float gs = 0.f;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        const float ga = Gt[(hs * 16 + kk) * NF_FP_TS + c32];
        const float av = At[(hs * 16 + kk) * NF_FP_TS + c32];
        gs += ga;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga, av, acc, 0, 0, 0);
    }
    return gs;
}
// LayerNorm backward in the A layout: g_in = rstd * (g_xh - mean(g_xh) - xh * mean(g_xh * xh)),  g_xh = g_out * gamma
__device__ __forceinline__ void nf_fp_ln_bwd(const float (&g_out)[16], const float (&xh)[16], const float* gamma, float rstd,
                                             float (&g_in)[16], int hs) {
    float gx[16], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        gx[kk] = g_out[kk] * gamma[hs * 16 + kk];
        s1 += gx[kk];
        s2 = fmaf(gx[kk], xh[kk], s2);
    }
    s1 += __shfl_xor(s1, 32, NF_WAVE);
    s2 += __shfl_xor(s2, 32, NF_WAVE);
    s1 *= (1.f / 32.f);
    s2 *= (1.f / 32.f);
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) g_in[kk] = rstd * (gx[kk] - s1 - xh[kk] * s2);
}

__global__ void __launch_bounds__(NF_FP_WAVES * NF_WAVE) k_flowpp_cond_bwd(NfFppW w, NfFppG g, int64_t N, int I0, int O,
                                                                           int64_t tiles) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const NfFppL L = nf_fpp_layout(I0, NF_FP_BWD_TILES);
    nf_fpp_stage(w, sm, L, I0, O);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, c32 = lane & 31, hs = lane >> 5;
    float* T0 = sm + L.tiles + (wid * NF_FP_BWD_TILES + 0) * 32 * NF_FP_TS;   // conversion tile
    float* TG = sm + L.tiles + (wid * NF_FP_BWD_TILES + 1) * 32 * NF_FP_TS;   // gradient tile (A operand of the weight GEMMs)
    float* ST = sm + L.tiles + (wid * NF_FP_BWD_TILES + 2) * 32 * NF_FP_TS;   // stash: elu(h0), elu(-h0), t, q, h4
    const int TSZ = 32 * NF_FP_TS;
    const bool two = O > 32;

    f32x16 aW5a = nf_fp_zero(), aW5b = nf_fp_zero(), aW2a = nf_fp_zero(), aW2b = nf_fp_zero(), aWq = nf_fp_zero(),
           aWg0 = nf_fp_zero(), aWg1 = nf_fp_zero();
    float vb5a = 0.f, vb5b = 0.f, vb2a = 0.f, vb2b = 0.f, vbq = 0.f, vbg = 0.f, vg2 = 0.f, vbt2 = 0.f, vg1 = 0.f, vbt1 = 0.f,
          vpos = 0.f, vb0 = 0.f, vW0[4] = {0.f, 0.f, 0.f, 0.f};

    for (int64_t t = (int64_t)blockIdx.x * NF_FP_WAVES + wid; t < tiles; t += (int64_t)gridDim.x * NF_FP_WAVES) {
        const int64_t row = t * 32 + c32;
        const bool rv = row < N;
        NfFppFwd f;
        nf_fpp_forward_tile<true>(sm, L, w.x, I0, row, rv, T0, ST, c32, hs, f);
        float xin[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < I0; ++i) xin[i] = rv ? w.x[row * I0 + i] : 0.f;

        // ---- out = W5 h4 + b5 ----------------------------------------------------------------------------------
        float g_h4[16];
        {
            float ga[16], gb[16];
            const float* gp = g.g_out + (rv ? row : 0) * O;      // unconditional (clamped) loads + select: no branches
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int o0 = hs * 16 + kk, o1 = 32 + o0;
                const float v0 = gp[o0 < O ? o0 : O - 1], v1 = gp[o1 < O ? o1 : O - 1];
                ga[kk] = (rv && o0 < O) ? v0 : 0.f;
                gb[kk] = (rv && o1 < O) ? v1 : 0.f;
            }
            f32x16 acc = nf_fp_gemm_t(nf_fp_zero(), ga, sm + L.W5, 33, 0, 0, c32, hs);
            if (two) acc = nf_fp_gemm_t(acc, gb, sm + L.W5, 33, 32, 0, c32, hs);
            nf_fp_store_rows(ga, TG, c32, hs);
            nf_fp_wsync();
            vb5a += nf_fp_wgrad(aW5a, TG, ST + 4 * TSZ, c32, hs);
            nf_fp_wsync();
            if (two) {
                nf_fp_store_rows(gb, TG, c32, hs);
                nf_fp_wsync();
                vb5b += nf_fp_wgrad(aW5b, TG, ST + 4 * TSZ, c32, hs);
                nf_fp_wsync();
            }
            nf_fp_cd_to_a(acc, nullptr, T0, g_h4, c32, hs);     // rows beyond N carry g_out = 0 -> exact zeros
        }
        // ---- LayerNorm 2 ---------------------------------------------------------------------------------------
        float g_h3[16];
        {
            float tmp[16];
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) tmp[kk] = g_h4[kk] * f.xh2[kk];
            vg2 += nf_fp_colsum(tmp, TG, c32, hs);
            vbt2 += nf_fp_colsum(g_h4, TG, c32, hs);
            nf_fp_ln_bwd(g_h4, f.xh2, sm + L.ln2g, f.rstd2, g_h3, hs);
        }
        // ---- gate 2: h3 = h2 + y2 sigmoid(a2);  [y2, a2] = W2 q + b2 ------------------------------------------------
        float g_q[16];
        {
            float g_y2[16], g_a2[16];
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const float s2 = nf_sigmoid(f.a2[kk]);
                g_y2[kk] = g_h3[kk] * s2;
                g_a2[kk] = g_h3[kk] * f.y2[kk] * s2 * (1.f - s2);
            }
            f32x16 acc = nf_fp_gemm_t(nf_fp_zero(), g_y2, sm + L.W2, 33, 0, 0, c32, hs);
            acc = nf_fp_gemm_t(acc, g_a2, sm + L.W2, 33, 32, 0, c32, hs);
            nf_fp_store_rows(g_y2, TG, c32, hs);
            nf_fp_wsync();
            vb2a += nf_fp_wgrad(aW2a, TG, ST + 3 * TSZ, c32, hs);
            nf_fp_wsync();
            nf_fp_store_rows(g_a2, TG, c32, hs);
            nf_fp_wsync();
            vb2b += nf_fp_wgrad(aW2b, TG, ST + 3 * TSZ, c32, hs);
            nf_fp_wsync();
            nf_fp_cd_to_a(acc, nullptr, T0, g_q, c32, hs);
        }
        // ---- q = Wq (h2 + pos) + bq ------------------------------------------------------------------------------------
        float g_h2[16];
        {
            f32x16 acc = nf_fp_gemm_t(nf_fp_zero(), g_q, sm + L.Wq, 33, 0, 0, c32, hs);
            nf_fp_store_rows(g_q, TG, c32, hs);
            nf_fp_wsync();
            vbq += nf_fp_wgrad(aWq, TG, ST + 2 * TSZ, c32, hs);
            nf_fp_wsync();
            float g_t[16];
            nf_fp_cd_to_a(acc, nullptr, T0, g_t, c32, hs);
            vpos += nf_fp_colsum(g_t, TG, c32, hs);
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) g_h2[kk] = g_h3[kk] + g_t[kk];
        }
        // ---- LayerNorm 1 ---------------------------------------------------------------------------------------
        float g_h1[16];
        {
            float tmp[16];
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) tmp[kk] = g_h2[kk] * f.xh1[kk];
            vg1 += nf_fp_colsum(tmp, TG, c32, hs);
            vbt1 += nf_fp_colsum(g_h2, TG, c32, hs);
            nf_fp_ln_bwd(g_h2, f.xh1, sm + L.ln1g, f.rstd1, g_h1, hs);
        }
        // ---- gate 1: h1 = h0 + elu(u) sigmoid(elu(-u));  u = Wg [elu(h0), elu(-h0)] + bg ------------------------------
        float g_h0[16];
        {
            float g_u[16];
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const float uu = f.u[kk];
                const float y = nf_elu(uu), a = nf_elu(-uu), sa = nf_sigmoid(a);
                const float g_y = g_h1[kk] * sa, g_a = g_h1[kk] * y * sa * (1.f - sa);
                g_u[kk] = g_y * nf_elu_grad(uu) - g_a * nf_elu_grad(-uu);
            }
            f32x16 acc0 = nf_fp_gemm_t(nf_fp_zero(), g_u, sm + L.Wg, 65, 0, 0, c32, hs);
            f32x16 acc1 = nf_fp_gemm_t(nf_fp_zero(), g_u, sm + L.Wg, 65, 0, 32, c32, hs);
            nf_fp_store_rows(g_u, TG, c32, hs);
            nf_fp_wsync();
            vbg += nf_fp_wgrad(aWg0, TG, ST + 0 * TSZ, c32, hs);
            (void)nf_fp_wgrad(aWg1, TG, ST + 1 * TSZ, c32, hs);
            nf_fp_wsync();
            float g_c0[16], g_c1[16];
            nf_fp_cd_to_a(acc0, nullptr, T0, g_c0, c32, hs);
            nf_fp_cd_to_a(acc1, nullptr, T0, g_c1, c32, hs);
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
                g_h0[kk] = g_h1[kk] + g_c0[kk] * nf_elu_grad(f.h0[kk]) - g_c1[kk] * nf_elu_grad(-f.h0[kk]);
        }
        // ---- h0 = W0 x + b0 --------------------------------------------------------------------------------------------
        nf_fp_store_rows(g_h0, TG, c32, hs);
        nf_fp_wsync();
        {
            // column sums of g_h0 and of g_h0 * x_i: rows hs*16 .. hs*16+15, column c32 (x of those rows via shuffles)
            float cs = 0.f, cw[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const float gv = TG[(hs * 16 + kk) * NF_FP_TS + c32];
                cs += gv;
#pragma unroll
                for (int i = 0; i < 4; ++i) cw[i] = fmaf(gv, __shfl(xin[i], hs * 16 + kk, NF_WAVE), cw[i]);
            }
            vb0 += cs;
#pragma unroll
            for (int i = 0; i < 4; ++i) vW0[i] += cw[i];
        }
        nf_fp_wsync();
        if (g.g_x != nullptr) {
            for (int i = 0; i < I0; ++i) {
                float s = 0.f;
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) s = fmaf(g_h0[kk], sm[L.W0 + (hs * 16 + kk) * 4 + i], s);
                s += __shfl_xor(s, 32, NF_WAVE);
                if (hs == 0 && rv) g.g_x[row * I0 + i] = s;
            }
        }
    }

    // ---- block reduction and accumulation into the destinations -----------------------------------------------------
    __syncthreads();
    float* red = sm + L.tiles;                                   // [waves][32][32] floats (fits in the tile region)
    auto flush_tile = [&](const f32x16& a, float* dst, int row_off, int n_rows, int ld, int col_off) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wid * 32 + nf_fp_cdrow(r, hs)) * 32 + c32] = a[r];
        __syncthreads();
        for (int e = threadIdx.x; e < 32 * 32; e += blockDim.x) {
            const int oo = e >> 5, ii = e & 31;
            if (oo < n_rows) {
                float s = 0.f;
#pragma unroll
                for (int wv = 0; wv < NF_FP_WAVES; ++wv) s += red[(wv * 32 + oo) * 32 + ii];
                atomicAdd(dst + (row_off + oo) * ld + col_off + ii, s);
            }
        }
        __syncthreads();
    };
    flush_tile(aW5a, g.g_W5, 0, O < 32 ? O : 32, 32, 0);
    if (two) flush_tile(aW5b, g.g_W5, 32, O - 32, 32, 0);
    flush_tile(aW2a, g.g_W2, 0, 32, 32, 0);
    flush_tile(aW2b, g.g_W2, 32, 32, 32, 0);
    flush_tile(aWq, g.g_Wq, 0, 32, 32, 0);
    flush_tile(aWg0, g.g_Wg, 0, 32, 64, 0);
    flush_tile(aWg1, g.g_Wg, 0, 32, 64, 32);
    auto flush_vec = [&](float v, float* dst, int n) {
        v += __shfl_xor(v, 32, NF_WAVE);
        if (hs == 0) red[wid * 32 + c32] = v;
        __syncthreads();
        if (wid == 0 && hs == 0 && c32 < n) {
            float s = 0.f;
#pragma unroll
            for (int wv = 0; wv < NF_FP_WAVES; ++wv) s += red[wv * 32 + c32];
            atomicAdd(dst + c32, s);
        }
        __syncthreads();
    };
    flush_vec(vb5a, g.g_b5, O < 32 ? O : 32);
    if (two) flush_vec(vb5b, g.g_b5 + 32, O - 32);
    flush_vec(vb2a, g.g_b2, 32);
    flush_vec(vb2b, g.g_b2 + 32, 32);
    flush_vec(vbq, g.g_bq, 32);
    flush_vec(vbg, g.g_bg, 32);
    flush_vec(vg2, g.g_ln2g, 32);
    flush_vec(vbt2, g.g_ln2b, 32);
    flush_vec(vg1, g.g_ln1g, 32);
    flush_vec(vbt1, g.g_ln1b, 32);
    flush_vec(vpos, g.g_pos, 32);
    flush_vec(vb0, g.g_b0, 32);
    for (int i = 0; i < I0; ++i) {                               // g_W0 is (32, I0): column i, stride I0
        float v = vW0[i];
        v += __shfl_xor(v, 32, NF_WAVE);
        if (hs == 0) red[wid * 32 + c32] = v;
        __syncthreads();
        if (wid == 0 && hs == 0) {
            float s = 0.f;
#pragma unroll
            for (int wv = 0; wv < NF_FP_WAVES; ++wv) s += red[wv * 32 + c32];
            atomicAdd(g.g_W0 + c32 * I0 + i, s);
        }
        __syncthreads();
    }
}

extern "C" int nf_flowpp_cond_bwd(const float* x, const float* W0, const float* b0, const float* Wg, const float* bg,
                                  const float* ln1_g, const float* ln1_b, const float* pos, const float* Wq, const float* bq,
                                  const float* W2, const float* b2, const float* ln2_g, const float* ln2_b, const float* W5,
                                  const float* b5, const float* g_out, float* g_x, float* g_W0, float* g_b0, float* g_Wg,
                                  float* g_bg, float* g_ln1_g, float* g_ln1_b, float* g_pos, float* g_Wq, float* g_bq,
                                  float* g_W2, float* g_b2, float* g_ln2_g, float* g_ln2_b, float* g_W5, float* g_b5,
                                  int64_t N, int I0, int O, nf_stream_t stream) {
    if (I0 < 1 || I0 > 4 || O < 1 || O > 64) return NF_E_BADARG;
    if (N <= 0) return N == 0 ? 0 : NF_E_BADARG;
    NfFppW w{x, W0, b0, Wg, bg, ln1_g, ln1_b, pos, Wq, bq, W2, b2, ln2_g, ln2_b, W5, b5, nullptr};
    NfFppG g{g_out, g_x, g_W0, g_b0, g_Wg, g_bg, g_ln1_g, g_ln1_b, g_pos, g_Wq, g_bq, g_W2, g_b2, g_ln2_g, g_ln2_b, g_W5, g_b5};
    const int64_t tiles = (N + 31) / 32;
    int64_t gx = (tiles + NF_FP_WAVES - 1) / NF_FP_WAVES;
    if (gx > 256) gx = 256;                                      // one pass of atomics per block: keep the block count low
    const NfFppL L = nf_fpp_layout(I0, NF_FP_BWD_TILES);
    const size_t lds = (size_t)L.total * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_flowpp_cond_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_flowpp_cond_bwd, dim3((unsigned)gx), dim3(NF_FP_WAVES * NF_WAVE), lds, (hipStream_t)stream, w, g, N, I0,
                       O, tiles);
    NF_CHECK_LAUNCH();
    return 0;
}
