// Flow++ conditioner for density (2-D / small-D) data, fused end to end on the fp32 matrix cores:
//   Linear(I0,32) -> GatedLinear -> LayerNorm -> GatedAttn (ONE position: the attention softmax is identically 1, the
//   block reduces to two linears + a sigmoid gate) -> LayerNorm -> Linear(32, O)
//   flows/coupling.py:142-149, flows/modules.py:500-518 (GatedLinear), :541-578 (GatedAttn), nn.LayerNorm.
// Everything is per-sample independent (LayerNorm, not BatchNorm), so the whole network is ONE launch forward and ONE
// launch backward (the backward recomputes the forward from the 4-byte input instead of saving eight activations).
//
// Work decomposition: a wave owns 16-row tiles and v_mfma_f32_16x16x4_f32.  Activations live in registers in a
// row-per-lane layout "R": lane (row = l & 15, g = l >> 4) holds features 16 b + 4 g + r (b = 0..1, r = 0..3) of its
// row, 8 registers per 32-wide vector.  R is at the same time
//   * the B-operand layout  (B[k][j = row], k-group g)   and
//   * the C/D layout of a product whose A operand is the weight matrix (D[i = out feature][j = row]: lane col = row,
//     rows 4 g + r), i.e. of  out^T = W act^T,
// so a chain of linears runs register to register: every GEMM takes the previous one's accumulators as its B operand and
// reads only its weight fragment from LDS.  No layout conversion, no barrier; element-wise work (concat-ELU, gates,
// LayerNorm) happens in R where a row's statistics are an in-lane sum and two cross-group shuffles.
// Only the weight gradients (K = rows) need the transposed view: the two operands go through a wave-private LDS tile,
// the 16x16 products accumulate in registers across the wave's tiles (LDS float atomics measured ~160 cycles per wave
// instruction on gfx950 -- 140 us per launch -- so there are none), and each block leaves ONE partial-sum slab that a
// small second kernel folds into the destinations (same-address global atomics from 256 blocks cost 45 us).
#include <type_traits>

#include "nf_common.h"
#include "nf_det.h"

NF_DET_STATE(nf_fpc)
NF_DET_HOST_API(nf_fpc)
#include "nf_mfma16.h"
#include "nf_mixlog_oct.h"

#define NF_FP_STG 68         // same for the 64-wide GatedLinear weight
#define NF_FP_FWD_WAVES 8
#define NF_FP_BWD_WAVES 8
#define NF_FP_MAX_BLOCKS 256   // backward grid cap = number of partial-sum slabs in the workspace
#define NF_FP_WAVE_LDS (2 * 3 * 16 * NF_FP_ST)   // per backward wave: two sets of {gradient-side, activation-side, spare} tiles
#define NF_FP_LNEPS 1.0e-5f

static_assert(NF_FP_FWD_WAVES * NF_WAVE == 512 && NF_FP_BWD_WAVES * NF_WAVE == 512, "nf_fpp_stage assumes 512 threads");

struct NfFppW {   // device pointers (forward operands)
    const float *x, *W0, *b0, *Wg, *bg, *ln1g, *ln1b, *pos, *Wq, *bq, *W2, *b2, *ln2g, *ln2b, *W5, *b5;
    float* out;
    int64_t xrs; int xcs;                                 // x[row][i] lives at x[row * xrs + i * xcs] (a strided view of z is fine)
};

struct NfFppG {   // gradient destinations, all ACCUMULATED (+=): zero-filled temporaries or .grad buffers
    const float* g_out;          // (N, O)
    float* g_x;                  // (N, I0) written, nullable
    float *g_W0, *g_b0, *g_Wg, *g_bg, *g_ln1g, *g_ln1b, *g_pos, *g_Wq, *g_bq, *g_W2, *g_b2, *g_ln2g, *g_ln2b, *g_W5, *g_b5;
    int64_t gxrs; int gxcs, gx_acc;                       // g_x[row][i] at g_x[row * gxrs + i * gxcs]; gx_acc: += instead of =
};

// the coupling itself, fused into the conditioner's backward (k_flowpp_cond_bwd<NB, true>): on (N, 2) data with K <= 8 mixture
// components the gradient of the conditioner's output is COMPUTED from the saved parameters instead of being read, see the kernel
struct NfFppMix {
    const float* z; const float* params; const float* gy; const float* gld;    // (N, 2), (N, O), (N, 2), (N,)
    const float* pA; const float* pC;                                          // coupling scale / shift scalars
    const float* nls; const float* nb;                                         // the next step's ActNorm (nullable)
    float* gz;                                                                 // (N, 2)
    float *g_scale, *g_bias, *g_nls, *g_nb;                                    // += (g_nls / g_nb nullable with nls)
    int odd, K;
    float eps;
};

// LDS layout (floats): the weights, then (backward) the per-wave tiles
struct NfFppL {
    int Wg, Wq, W2, W5, W0, b0, bg, ln1g, ln1b, pos, bq, b2, ln2g, ln2b, b5, wend;
    int tiles, total;
};
__host__ __device__ inline NfFppL nf_fpp_layout(int bwd_waves) {
    NfFppL L;
    int o = 0;
    L.Wg = o; o += 32 * NF_FP_STG;
    L.Wq = o; o += 32 * NF_FP_ST;
    L.W2 = o; o += 64 * NF_FP_ST;
    L.W5 = o; o += 64 * NF_FP_ST;
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
    L.wend = o;
    L.tiles = o; o += bwd_waves * NF_FP_WAVE_LDS;
    L.total = o;
    return L;
}

#ifndef NF_FEXP_DEFINED
#define NF_FEXP_DEFINED
__device__ __forceinline__ float nf_fexp(float x) { return __expf(x); }
#endif
__device__ __forceinline__ float nf_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + nf_fexp(-x)); }
// concat-ELU pair and its derivatives from ONE exponential: e = exp(-|h|)
//   h > 0: elu(h) = h, elu(-h) = e - 1, elu'(h) = 1, elu'(-h) = e;   h <= 0: elu(h) = e - 1, elu(-h) = -h, elu'(h) = e, elu'(-h) = 1
__device__ __forceinline__ void nf_celu(float h, float& c0, float& c1) {
    const float e = nf_fexp(-fabsf(h)) - 1.f;
    c0 = h > 0.f ? h : e;
    c1 = h > 0.f ? e : -h;
}
__device__ __forceinline__ void nf_celu_grad(float h, float& d0, float& d1) {
    const float e = nf_fexp(-fabsf(h));
    d0 = h > 0.f ? 1.f : e;
    d1 = h > 0.f ? e : 1.f;
}

// weights -> LDS.  Every global load is issued before the first LDS store, so the block pays ONE memory latency (the
// obvious loop-per-tensor form pays ten in a row, ~10 us per launch).  512 threads; float4 loads when `vec` (all four
// matrices 16-byte aligned), scalar loops otherwise.
__device__ __forceinline__ void nf_fpp_stage(const NfFppW& w, float* sm, const NfFppL& L, int I0, int O, bool vec) {
    const int tid = threadIdx.x;
    if (vec) {
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 vg = ((const float4*)w.Wg)[tid];                                  // 32 x 64 = 512 float4
        const float4 v2 = ((const float4*)w.W2)[tid];                                  // 64 x 32
        const float4 vq = tid < 256 ? ((const float4*)w.Wq)[tid] : z4;                 // 32 x 32
        const float4 v5 = (tid >> 3) < O ? ((const float4*)w.W5)[tid] : z4;            // O x 32, zero rows beyond O
        float v0 = 0.f, vv[8], vb2 = 0.f, vb5 = 0.f;
        if (tid < 128) v0 = ((tid & 3) < I0) ? w.W0[(tid >> 2) * I0 + (tid & 3)] : 0.f;
        if (tid < 32) {
            vv[0] = w.b0[tid]; vv[1] = w.bg[tid]; vv[2] = w.ln1g[tid]; vv[3] = w.ln1b[tid];
            vv[4] = w.pos[tid]; vv[5] = w.bq[tid]; vv[6] = w.ln2g[tid]; vv[7] = w.ln2b[tid];
        }
        if (tid >= 64 && tid < 128) {
            vb2 = w.b2[tid - 64];
            vb5 = (tid - 64) < O ? w.b5[tid - 64] : 0.f;
        }
        *(float4*)(sm + L.Wg + (tid >> 4) * NF_FP_STG + 4 * (tid & 15)) = vg;
        *(float4*)(sm + L.W2 + (tid >> 3) * NF_FP_ST + 4 * (tid & 7)) = v2;
        *(float4*)(sm + L.W5 + (tid >> 3) * NF_FP_ST + 4 * (tid & 7)) = v5;
        if (tid < 256) *(float4*)(sm + L.Wq + (tid >> 3) * NF_FP_ST + 4 * (tid & 7)) = vq;
        if (tid < 128) sm[L.W0 + tid] = v0;
        if (tid < 32) {
            sm[L.b0 + tid] = vv[0]; sm[L.bg + tid] = vv[1]; sm[L.ln1g + tid] = vv[2]; sm[L.ln1b + tid] = vv[3];
            sm[L.pos + tid] = vv[4]; sm[L.bq + tid] = vv[5]; sm[L.ln2g + tid] = vv[6]; sm[L.ln2b + tid] = vv[7];
        }
        if (tid >= 64 && tid < 128) {
            sm[L.b2 + tid - 64] = vb2;
            sm[L.b5 + tid - 64] = vb5;
        }
        return;
    }
    for (int i = tid; i < 32 * 64; i += blockDim.x) sm[L.Wg + (i >> 6) * NF_FP_STG + (i & 63)] = w.Wg[i];
    for (int i = tid; i < 32 * 32; i += blockDim.x) sm[L.Wq + (i >> 5) * NF_FP_ST + (i & 31)] = w.Wq[i];
    for (int i = tid; i < 64 * 32; i += blockDim.x) sm[L.W2 + (i >> 5) * NF_FP_ST + (i & 31)] = w.W2[i];
    for (int i = tid; i < 64 * 32; i += blockDim.x)
        sm[L.W5 + (i >> 5) * NF_FP_ST + (i & 31)] = (i >> 5) < O ? w.W5[i] : 0.f;
    for (int i = tid; i < 32 * 4; i += blockDim.x) sm[L.W0 + i] = ((i & 3) < I0) ? w.W0[(i >> 2) * I0 + (i & 3)] : 0.f;
    for (int i = tid; i < 32; i += blockDim.x) {
        sm[L.b0 + i] = w.b0[i]; sm[L.bg + i] = w.bg[i]; sm[L.ln1g + i] = w.ln1g[i]; sm[L.ln1b + i] = w.ln1b[i];
        sm[L.pos + i] = w.pos[i]; sm[L.bq + i] = w.bq[i]; sm[L.ln2g + i] = w.ln2g[i]; sm[L.ln2b + i] = w.ln2b[i];
    }
    for (int i = tid; i < 64; i += blockDim.x) {
        sm[L.b2 + i] = w.b2[i];
        sm[L.b5 + i] = i < O ? w.b5[i] : 0.f;
    }
}
__host__ inline bool nf_fpp_vec_ok(const NfFppW& w) {
    return ((((uintptr_t)w.Wg) | ((uintptr_t)w.Wq) | ((uintptr_t)w.W2) | ((uintptr_t)w.W5)) & 15) == 0;
}

__device__ __forceinline__ void nf_fp_layernorm(const float (&v)[8], const float* gamma, const float* beta, int g, float (&xh)[8],
                                                float& rstd, float (&y)[8]) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
    const float mean = nf_fp_rowsum(s) * (1.f / 32.f);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = v[j] - mean; q = fmaf(d, d, q); }
    rstd = 1.f / sqrtf(nf_fp_rowsum(q) * (1.f / 32.f) + NF_FP_LNEPS);
    float ga[8], be[8];
    nf_fp_ldvec(gamma, g, ga);
    nf_fp_ldvec(beta, g, be);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        xh[j] = (v[j] - mean) * rstd;
        y[j] = fmaf(xh[j], ga[j], be[j]);
    }
}

// the whole forward of one 16-row tile in the R layout; keeps what the backward needs
struct NfFppFwd {
    float h0[8], u[8], q[8], y2[8], a2[8];
    float xh1[8], xh2[8];       // LayerNorm-normalised values
    float h2[8], h4[8];         // LayerNorm outputs (dead in the backward except as GEMM operands)
    float rstd1, rstd2;
};

__device__ __forceinline__ void nf_fpp_forward_tile(const float* sm, const NfFppL& L, const float (&xin)[4], int c16, int g,
                                                    NfFppFwd& f) {
    float c0[8], c1[8], tv[8];
    nf_fp_ldvec(sm + L.b0, g, tv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 16 * (j >> 2) + 4 * g + (j & 3);
        const float4 w0 = *(const float4*)(sm + L.W0 + 4 * k);
        float h = tv[j];
        h = fmaf(w0.x, xin[0], h); h = fmaf(w0.y, xin[1], h); h = fmaf(w0.z, xin[2], h); h = fmaf(w0.w, xin[3], h);
        f.h0[j] = h;
        nf_celu(h, c0[j], c1[j]);                                        // concat-ELU (modules.py:509)
    }
    f32x4 acc[2] = {nf_fp_zero4(), nf_fp_zero4()};
    nf_fp_gemm<2>(sm + L.Wg, NF_FP_STG, 0, c0, acc, c16, g);
    nf_fp_gemm<2>(sm + L.Wg, NF_FP_STG, 32, c1, acc, c16, g);
    nf_fp_ldvec(sm + L.bg, g, tv);
    float h1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        f.u[j] = acc[j >> 2][j & 3] + tv[j];
        float y, a;
        nf_celu(f.u[j], y, a);
        h1[j] = f.h0[j] + y * nf_sigmoid(a);                              // modules.py:513-518
    }
    nf_fp_layernorm(h1, sm + L.ln1g, sm + L.ln1b, g, f.xh1, f.rstd1, f.h2);
    nf_fp_ldvec(sm + L.pos, g, tv);
#pragma unroll
    for (int j = 0; j < 8; ++j) tv[j] += f.h2[j];                         // modules.py:569
    acc[0] = nf_fp_zero4(); acc[1] = nf_fp_zero4();
    nf_fp_gemm<2>(sm + L.Wq, NF_FP_ST, 0, tv, acc, c16, g);
    nf_fp_ldvec(sm + L.bq, g, tv);
#pragma unroll
    for (int j = 0; j < 8; ++j) f.q[j] = acc[j >> 2][j & 3] + tv[j];
    f32x4 acc4[4] = {nf_fp_zero4(), nf_fp_zero4(), nf_fp_zero4(), nf_fp_zero4()};
    nf_fp_gemm<4>(sm + L.W2, NF_FP_ST, 0, f.q, acc4, c16, g);
    nf_fp_ldvec(sm + L.b2, g, tv);
#pragma unroll
    for (int j = 0; j < 8; ++j) f.y2[j] = acc4[j >> 2][j & 3] + tv[j];
    nf_fp_ldvec(sm + L.b2 + 32, g, tv);
    float h3[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        f.a2[j] = acc4[2 + (j >> 2)][j & 3] + tv[j];
        h3[j] = f.h2[j] + f.y2[j] * nf_sigmoid(f.a2[j]);                  // modules.py:574-578
    }
    nf_fp_layernorm(h3, sm + L.ln2g, sm + L.ln2b, g, f.xh2, f.rstd2, f.h4);
}

// the coupling itself in the forward direction (k_flowpp_cond_fwd<NB, true>): what k_mixlog_oct_fwd does, on the tile the
// conditioner just produced
struct NfFppMixF {
    const float* z;                                      // (N, 2) the step's input
    const float* pA; const float* pC;                    // coupling scale / shift scalars
    const float* nls; const float* nb;                   // the next step's ActNorm (nullable)
    float* y; float* ld;                                 // (N, 2), (N,) +=
    int odd, K;
    float eps;
};

// MIX: the conditioner's output tile also goes to a per-wave LDS tile, and the wave runs the mixture coupling on it one
// component per lane (two passes of eight rows x eight lanes): y and the log-det come out of the same launch.
template <int NB, bool MIX>   // NB = ceil(O / 16)
__global__ void __launch_bounds__(NF_FP_FWD_WAVES * NF_WAVE, 4) k_flowpp_cond_fwd(NfFppW w, int64_t N, int I0, int O,
                                                                               int64_t tiles, int vec, NfFppMixF mx) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const NfFppL L = nf_fpp_layout(0);
    nf_fpp_stage(w, sm, L, I0, O, vec != 0);
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, c16 = lane & 15, g = lane >> 4;
    float* PT = sm + L.wend + wid * 16 * NF_FP_ST;       // MIX: this wave's (16 rows x 32 features) parameter tile
    for (int64_t t = (int64_t)blockIdx.x * NF_FP_FWD_WAVES + wid; t < tiles; t += (int64_t)gridDim.x * NF_FP_FWD_WAVES) {
        asm volatile("" ::: "memory");   // keep the weight fragments in LDS: hoisted out of the loop they cost 256 registers
        const int64_t row0 = t * 16, row = row0 + c16;
        float xin[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < I0 && row < N) xin[i] = w.x[row * w.xrs + i * w.xcs];
        NfFppFwd f;
        nf_fpp_forward_tile(sm, L, xin, c16, g, f);
        // out = h4 W5^T + b5 with the activations as the A operand: D col = output feature -> coalesced row stores
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) {
            f32x4 acc = nf_fp_zero4();
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float4 wv = *(const float4*)(sm + L.W5 + (16 * ob + c16) * NF_FP_ST + 16 * b + 4 * g);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f.h4[4 * b + 0], wv.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f.h4[4 * b + 1], wv.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f.h4[4 * b + 2], wv.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f.h4[4 * b + 3], wv.w, acc, 0, 0, 0);
            }
            const int o = 16 * ob + c16;
            if (o < O) {
                const float bv = sm[L.b5 + o];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t gr = row0 + 4 * g + r;
                    if (gr < N) w.out[gr * O + o] = acc[r] + bv;
                    if (MIX) PT[(4 * g + r) * NF_FP_ST + o] = acc[r] + bv;
                }
            }
        }
        if (MIX) {   // ---- the mixture coupling on this tile (mixlog.hip: k_mixlog_oct_fwd) ----
            nf_fp_wsync();
            const bool post = mx.nls != nullptr;
            const int o0 = mx.odd, o1 = 1 ^ mx.odd, K = mx.K, kk = lane & 7;
            const float A = mx.pA[0], Cb = mx.pC[0];
            const float D0 = post ? expf(mx.nls[o0]) : 1.f, D1 = post ? expf(mx.nls[o1]) : 1.f;
            const float S0 = post ? mx.nb[o0] : 0.f, S1 = post ? mx.nb[o1] : 0.f;
            const float ldn = post ? -(mx.nls[0] + mx.nls[1]) : 0.f;
#pragma unroll 1
            for (int p = 0; p < 2; ++p) {
                const int rr = 8 * p + (lane >> 3);
                const int64_t b = row0 + rr;
                const bool live = b < N;
                const int64_t bb = live ? b : N - 1;
                NfOct m;
                nf_oct_load(PT + rr * NF_FP_ST, 1, K, kk, m);
                const float x = mx.z[bb * 2 + o0], zi = mx.z[bb * 2 + o1];
                float lcdf, lpdf, u, l;
                nf_oct_eval(m, x, lcdf, lpdf, u, l);
                const float F = nf_fexp(lcdf);                                  // modules.py:194
                const float xc = fminf(fmaxf(F, mx.eps), 1.f - mx.eps);         // modules.py:147
                const float la = nf_flog(xc), lb = nf_flog(1.f - xc);           // logit and its log-det share the two logs
                const float a = nf_ftanh(m.a_raw) * A + Cb;                     // coupling.py:178
                if (live && kk == 0) {
                    const float yt = (la - lb) * nf_fexp(a) + m.b;              // coupling.py:187
                    mx.y[b * 2 + o0] = post ? (yt - S0) / D0 : yt;
                    mx.y[b * 2 + o1] = post ? (zi - S1) / D1 : zi;
                    mx.ld[b] += ldn + lpdf - (la + lb) + a;                     // coupling.py:184-188
                }
            }
            nf_fp_wsync();                                                      // the tile is rewritten by the next trip
        }
    }
}

template <bool MIX>
static int nf_fpp_launch_fwd(const NfFppW& w, int64_t N, int I0, int O, hipStream_t stream, const NfFppMixF& mx) {
    const int64_t tiles = (N + 15) / 16;
    int64_t gx = (tiles + NF_FP_FWD_WAVES - 1) / NF_FP_FWD_WAVES;
    if (gx > 512) gx = 512;                                     // two 8-wave blocks per CU, weights staged once per block
    const NfFppL L = nf_fpp_layout(0);
    const size_t lds = ((size_t)L.wend + (MIX ? NF_FP_FWD_WAVES * 16 * NF_FP_ST : 0)) * sizeof(float);
    const dim3 grid((unsigned)gx), block(NF_FP_FWD_WAVES * NF_WAVE);
    const int vec = nf_fpp_vec_ok(w) ? 1 : 0;
    switch ((O + 15) / 16) {
        case 1: hipLaunchKernelGGL((k_flowpp_cond_fwd<1, MIX>), grid, block, lds, stream, w, N, I0, O, tiles, vec, mx); break;
        case 2: hipLaunchKernelGGL((k_flowpp_cond_fwd<2, MIX>), grid, block, lds, stream, w, N, I0, O, tiles, vec, mx); break;
        case 3: hipLaunchKernelGGL((k_flowpp_cond_fwd<3, false>), grid, block, lds, stream, w, N, I0, O, tiles, vec, mx); break;
        default: hipLaunchKernelGGL((k_flowpp_cond_fwd<4, false>), grid, block, lds, stream, w, N, I0, O, tiles, vec, mx); break;
    }
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_flowpp_cond_fwd(const float* x, const float* W0, const float* b0, const float* Wg, const float* bg,
                                  const float* ln1_g, const float* ln1_b, const float* pos, const float* Wq, const float* bq,
                                  const float* W2, const float* b2, const float* ln2_g, const float* ln2_b, const float* W5,
                                  const float* b5, float* out, int64_t x_row_stride, int x_col_stride, int64_t N, int I0, int O,
                                  nf_stream_t stream) {
    if (I0 < 1 || I0 > 4 || O < 1 || O > 64) return NF_E_BADARG;
    if (N <= 0) return N == 0 ? 0 : NF_E_BADARG;
    NfFppW w{x, W0, b0, Wg, bg, ln1_g, ln1_b, pos, Wq, bq, W2, b2, ln2_g, ln2_b, W5, b5, out, x_row_stride, x_col_stride};
    return nf_fpp_launch_fwd<false>(w, N, I0, O, (hipStream_t)stream, NfFppMixF{});
}

// the whole forward of a Flow++ density flow step on (N, 2) data: conditioner + mixture coupling (+ the next step's ActNorm),
// one launch (nfhip.h)
extern "C" int nf_flowpp_vec_step_fwd(const float* z, const float* W0, const float* b0, const float* Wg, const float* bg,
                                      const float* ln1_g, const float* ln1_b, const float* pos, const float* Wq, const float* bq,
                                      const float* W2, const float* b2, const float* ln2_g, const float* ln2_b, const float* W5,
                                      const float* b5, const float* a_log_scale, const float* a_bias, const float* next_log_scale,
                                      const float* next_bias, float* params, float* y, float* ld, int K, float logit_eps, int odd,
                                      int64_t N, nf_stream_t stream) {
    const int O = 2 + 3 * K;
    if (K < 1 || K > 8 || z == nullptr || params == nullptr || y == nullptr || ld == nullptr || a_log_scale == nullptr ||
        a_bias == nullptr || (next_log_scale == nullptr) != (next_bias == nullptr))
        return NF_E_BADARG;
    if (N <= 0) return N == 0 ? 0 : NF_E_BADARG;
    const int sel1 = odd ? 0 : 1;
    NfFppW w{z + sel1, W0, b0, Wg, bg, ln1_g, ln1_b, pos, Wq, bq, W2, b2, ln2_g, ln2_b, W5, b5, params, 2, 2};
    NfFppMixF mx{z, a_log_scale, a_bias, next_log_scale, next_bias, y, ld, odd ? 1 : 0, K, logit_eps};
    return nf_fpp_launch_fwd<true>(w, N, 1, O, (hipStream_t)stream, mx);
}

// ---------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------
// per-block partial sums ("slab", floats), dense row-major like the parameters
enum {
    NF_S_W5 = 0, NF_S_W2 = 2048, NF_S_WQ = 4096, NF_S_WG = 5120, NF_S_W0 = 7168, NF_S_B5 = 7296, NF_S_B2 = 7360,
    NF_S_BQ = 7424, NF_S_BG = 7456, NF_S_LN2G = 7488, NF_S_LN2B = 7520, NF_S_LN1G = 7552, NF_S_LN1B = 7584, NF_S_POS = 7616,
    NF_S_B0 = 7648, NF_S_MIX = 7680, NF_S_END = 7744     // NF_S_MIX: the fused coupling's seven scalar sums (k_flowpp_cond_bwd<NB, true>)
};

// register relief: park an R-layout vector in LDS (each lane reads back exactly what it wrote)
__device__ __forceinline__ void nf_fp_park(const float (&v)[8], float* slot, int lane) {
    *(float4*)(slot + 4 * lane) = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)(slot + 256 + 4 * lane) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void nf_fp_unpark(const float* slot, float (&v)[8], int lane) {
    const float4 a = *(const float4*)(slot + 4 * lane), b = *(const float4*)(slot + 256 + 4 * lane);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
// acc[OBO + ob][IBO + ib] += sum_rows G[row][16 ob + .] Act[row][16 ib + .]  (D: lane col = 16 ib + c16, rows 4 g + r);
// vb (nullable) += this lane's share of the column sums of G (the bias gradient)
template <int NOB, int NIB, int TOB, int TIB, int OBO, int IBO>
__device__ __forceinline__ void nf_fp_wgrad(const float (&ga)[NOB][4], const float (&av)[NIB][4], f32x4 (&acc)[TOB][TIB],
                                            float* vb) {
    if (vb != nullptr) {
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) vb[ob] += (ga[ob][0] + ga[ob][1]) + (ga[ob][2] + ga[ob][3]);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
            for (int ib = 0; ib < NIB; ++ib)
                acc[OBO + ob][IBO + ib] =
                    __builtin_amdgcn_mfma_f32_16x16x4f32(ga[ob][s], av[ib][s], acc[OBO + ob][IBO + ib], 0, 0, 0);
}
// this lane's share (rows g, g + 4, ..; columns c16 and 16 + c16) of the column sums of an R-layout vector
__device__ __forceinline__ void nf_fp_colsum(const float (&v)[8], float* tile, float (&acc)[2], int c16, int g) {
    nf_fp_store_rows(v, tile, c16, g);
    nf_fp_wsync();
    float c[2][4];
    nf_fp_load_cols<2>(tile, c, c16, g);
    acc[0] += (c[0][0] + c[0][1]) + (c[0][2] + c[0][3]);
    acc[1] += (c[1][0] + c[1][1]) + (c[1][2] + c[1][3]);
    nf_fp_wsync();
}
// LayerNorm backward in R: g_in = rstd * (g_xh - mean(g_xh) - xh * mean(g_xh * xh)),  g_xh = g_out * gamma
__device__ __forceinline__ void nf_fp_ln_bwd(const float (&g_out)[8], const float (&xh)[8], const float* gamma, float rstd,
                                             float (&g_in)[8], int g) {
    float ga[8], gx[8], s1 = 0.f, s2 = 0.f;
    nf_fp_ldvec(gamma, g, ga);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        gx[j] = g_out[j] * ga[j];
        s1 += gx[j];
        s2 = fmaf(gx[j], xh[j], s2);
    }
    s1 = nf_fp_rowsum(s1) * (1.f / 32.f);
    s2 = nf_fp_rowsum(s2) * (1.f / 32.f);
#pragma unroll
    for (int j = 0; j < 8; ++j) g_in[j] = rstd * (gx[j] - s1 - xh[j] * s2);
}

template <bool FIRST>
__device__ __forceinline__ void nf_fp_put(float* R, int idx, float v) {
    R[idx] = FIRST ? v : R[idx] + v;
}

// One output block (16 x 16) of a weight gradient over the rows of ALL the workgroup's waves:
//   acc += sum_w sum_rows G_w[row][gcol + c16'] Act_w[row][acol + .]   (tiles of wave w at Gt + w * tstride, At + w * tstride)
// returns this lane's share of the column sums of G (the bias gradient of outputs gcol ..)
__device__ __forceinline__ float nf_fp_coop_block(const float* Gt, const float* At, int tstride, int gcol, int acol, f32x4& acc,
                                                  int c16, int g) {
    float bs = 0.f;
#pragma unroll 2
    for (int wv = 0; wv < NF_FP_BWD_WAVES; ++wv) {
        const float* gp = Gt + wv * tstride + gcol + c16;
        const float* ap = At + wv * tstride + acol + c16;
        float ga[4], av[4];
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) { ga[s2] = gp[(4 * s2 + g) * NF_FP_ST]; av[s2] = ap[(4 * s2 + g) * NF_FP_ST]; }
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            bs += ga[s2];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s2], av[s2], acc, 0, 0, 0);
        }
    }
    return bs;
}

// Weight gradients are formed per workgroup: in each of five phases every wave parks the gradient-side and the
// activation-side tile of its 16 rows in LDS, a barrier, then wave w owns ONE 16 x 16 output block of that matrix over the
// rows of all eight waves (32 MFMAs), a barrier.  Five accumulators per wave instead of thirty: nothing spills (the first
// version moved 286 MB of scratch per launch, profiles/r01_pmc_summary.txt), and the block results go to the slab as they are.
// MIX: the Flow++ coupling step on two features (flows/coupling.py:172-210, K <= 8).  Per 16-row tile the wave first runs the
// coupling's backward one mixture component per lane (two passes of eight rows x eight lanes, nf_mixlog_oct.h) from the SAVED
// conditioner output, writes the gradient of the coupling's input, and leaves the gradient of the conditioner's output in its
// LDS tile -- the 6.8 MB (N, 26) tensor that k_mixlog_oct_bwd wrote and this kernel read back twice never exists, and the
// coupling's backward launch (22 us per flow step at C3) is gone.  Its seven scalar sums (coupling scale / shift, the next
// step's ActNorm) ride the slab.
template <int NB, bool MIX>
__global__ void __launch_bounds__(NF_FP_BWD_WAVES * NF_WAVE) k_flowpp_cond_bwd(NfFppW w, NfFppG gr, float* __restrict__ slabs,
                                                                               int64_t N, int I0, int O, int64_t tiles, int iters,
                                                                               int vec, NfFppMix mx) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const NfFppL L = nf_fpp_layout(NF_FP_BWD_WAVES);
    nf_fpp_stage(w, sm, L, I0, O, vec != 0);
    __syncthreads();
    const int lane0 = threadIdx.x & 63, wid = threadIdx.x >> 6;
    constexpr int TSZ = 16 * NF_FP_ST;
    // two sets of [waves] x {gradient-side tile (A fragments of the weight products), activation-side tile (B fragments),
    // spare: second half of a 64-wide side, private scratch otherwise}.  Phases alternate between the sets, so ONE barrier per
    // phase suffices: a set is overwritten two phases later, and every wave has passed the barrier in between only after it
    // finished its share of the products that read it.
    constexpr int SETSZ = 3 * NF_FP_BWD_WAVES * TSZ;

    f32x4 a5 = nf_fp_zero4(), a2 = nf_fp_zero4(), aq = nf_fp_zero4(), ag = nf_fp_zero4(), a0 = nf_fp_zero4();
    float b5 = 0.f, b2 = 0.f, bq = 0.f, bg = 0.f, b0 = 0.f;
    float vln2g[2] = {0.f, 0.f}, vln2b[2] = {0.f, 0.f}, vln1g[2] = {0.f, 0.f}, vln1b[2] = {0.f, 0.f}, vpos[2] = {0.f, 0.f};
    float macc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // MIX: g_a tanh, g_a | next ActNorm: sum g_h0, g_h0 h0, g_h1, g_h1 h1, g_ld

    for (int it = 0; it < iters; ++it) {                  // uniform trip count: the phases are workgroup barriers
        // the lane index is laundered once per trip: otherwise ~100 loop-invariant LDS / global addresses derived from it are
        // hoisted into registers for the whole loop and the kernel spills (seen: 256 VGPRs + 240 B scratch)
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int c16 = lane & 15, g = lane >> 4;
#define NF_FPP_SET(k)                                                                   \
    float* TGs = sm + L.tiles + (((it + (k)) & 1) ? SETSZ : 0);                         \
    float* TAs = TGs + NF_FP_BWD_WAVES * TSZ;                                           \
    float* TXs = TAs + NF_FP_BWD_WAVES * TSZ;                                           \
    float* TG = TGs + wid * TSZ;                                                        \
    float* TA = TAs + wid * TSZ;                                                        \
    float* TX = TXs + wid * TSZ;                                                        \
    (void)TG; (void)TA; (void)TX; (void)TXs
        const int64_t tile0 = ((int64_t)it * gridDim.x + blockIdx.x) * NF_FP_BWD_WAVES;
        const int64_t row0 = (tile0 + wid) * 16, row = row0 + c16;
        const bool rv = row < N;
        float xin[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < I0 && rv) xin[i] = w.x[row * w.xrs + i * w.xcs];
        if (MIX) {   // ---- the coupling's backward: g_out of the conditioner -> this wave's gradient tile of the phase-1 set ----
            NF_FPP_SET(0);
            const bool post = mx.nls != nullptr;
            const int o0 = mx.odd, o1 = 1 ^ mx.odd;                // squeeze.py:68-69: transformed / conditioning feature
            const float A = mx.pA[0], Cb = mx.pC[0];
            const float D0 = post ? __expf(mx.nls[o0]) : 1.f, D1 = post ? __expf(mx.nls[o1]) : 1.f;
            const float S0 = post ? mx.nb[o0] : 0.f, S1 = post ? mx.nb[o1] : 0.f;
            const int K = mx.K, kk = lane & 7;
            const bool on = kk < K;
            for (int e = lane; e < 16 * NF_FP_ST; e += NF_WAVE) TG[e] = 0.f;     // features >= O and the rows past N stay zero
            nf_fp_wsync();
#pragma unroll 1
            for (int p = 0; p < 2; ++p) {
                const int rr = 8 * p + (lane >> 3);
                const int64_t b = row0 + rr;
                const bool live = b < N;
                const int64_t bb = live ? b : N - 1;
                NfOct m;
                nf_oct_load(mx.params + bb * O, 1, K, kk, m);
                const float x = mx.z[bb * 2 + o0], zi = mx.z[bb * 2 + o1];
                const float g_h0 = mx.gy[bb * 2 + o0], g_h1 = mx.gy[bb * 2 + o1];
                const float g_ld = mx.gld[bb];
                const float g_y = post ? g_h0 / D0 : g_h0;
                float lcdf, lpdf, u, l;
                nf_oct_eval(m, x, lcdf, lpdf, u, l);
                const float F = nf_fexp(lcdf), fd = nf_fexp(lpdf);
                const bool inside = (F >= mx.eps) && (F <= 1.f - mx.eps);    // torch.clamp passes the gradient on [min, max]
                const float xc = fminf(fmaxf(F, mx.eps), 1.f - mx.eps);
                const float y1 = nf_flog(xc) - nf_flog(1.f - xc);
                const float th = nf_ftanh(m.a_raw);
                const float ea = nf_fexp(th * A + Cb);
                const float g_y1 = g_y * ea;                                 // y = y1 * exp(a) + b ; ld += a
                const float g_a = g_y * y1 * ea + g_ld;
                const float gF = inside ? (g_y1 - g_ld * (1.f - 2.f * xc)) / (xc * (1.f - xc)) : 0.f;   // logit + its log-det
                const float tot = gF * F + g_ld;                             // sum_j g_logpi_j
                const float r = on ? nf_fexp(m.lp + (u - m.s - 2.f * (fmaxf(u, 0.f) + l)) - lpdf) : 0.f;   // responsibilities
                const float omt = -nf_ftanh(0.5f * u);                       // 1 - 2 sigmoid(u)
                const float wv_ = g_ld * r * omt * m.es;
                const float gx = gF * fd + nf_oct_sum(wv_);
                if (live) {
                    float* Tr = TG + rr * NF_FP_ST;
                    if (on) {
                        Tr[2 + K + kk] = -gF * fd * r - wv_;                                            // g_mu_k
                        Tr[2 + 2 * K + kk] = -gF * fd * r * (x - m.mu) + g_ld * r * (-omt * u - 1.f);   // g_s_k
                        const float g_logpi = gF * nf_fexp(m.lp + (fminf(u, 0.f) - l)) + g_ld * r;
                        Tr[2 + kk] = g_logpi - nf_fexp(m.lp) * tot;                                     // through log_softmax
                    }
                    if (kk == 0) {
                        Tr[0] = g_a * A * (1.f - th * th);
                        Tr[1] = g_y;
                        mx.gz[b * 2 + o0] = gx;
                        mx.gz[b * 2 + o1] = post ? g_h1 / D1 : g_h1;
                        macc[0] += g_a * th;
                        macc[1] += g_a;
                        if (post) {
                            const float h0 = (y1 * ea + m.b - S0) / D0, h1 = (zi - S1) / D1;
                            macc[2] += g_h0;
                            macc[3] += g_h0 * h0;
                            macc[4] += g_h1;
                            macc[5] += g_h1 * h1;
                            macc[6] += g_ld;
                        }
                    }
                }
            }
            nf_fp_wsync();
        }
        NfFppFwd f;
        nf_fpp_forward_tile(sm, L, xin, c16, g, f);

        // ---- phase 1: out = W5 h4 + b5 -----------------------------------------------------------------------------
        float g_h4[8], g_h3[8];
        {
        NF_FPP_SET(0);
        nf_fp_store_rows(f.h4, TA, c16, g);
        {
            float go[4 * NB];                                    // g_out in R: features 16 b + 4 g + r of this lane's row
            if (MIX) {
#pragma unroll
                for (int j = 0; j < 4 * NB; ++j) go[j] = TG[c16 * NF_FP_ST + 16 * (j >> 2) + 4 * g + (j & 3)];
            } else {
            const float* gp = gr.g_out + (rv ? row : 0) * O;     // unconditional (clamped) loads + select: no branches
#pragma unroll
            for (int j = 0; j < 4 * NB; ++j) {
                const int o = 16 * (j >> 2) + 4 * g + (j & 3);
                const float v = gp[o < O ? o : O - 1];
                go[j] = (rv && o < O) ? v : 0.f;
            }
            }
            f32x4 acc[2] = {nf_fp_zero4(), nf_fp_zero4()};
            nf_fp_gemm_d<NB>(sm + L.W5, NF_FP_ST, 0, go, acc, c16, g);
#pragma unroll
            for (int j = 0; j < 8; ++j) g_h4[j] = acc[j >> 2][j & 3];
        }
        {   // LayerNorm 2 (its column sums use the private scratch tile)
            float tmp[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) tmp[j] = g_h4[j] * f.xh2[j];
            nf_fp_colsum(tmp, TX, vln2g, c16, g);
            nf_fp_colsum(g_h4, TX, vln2b, c16, g);
            nf_fp_ln_bwd(g_h4, f.xh2, sm + L.ln2g, f.rstd2, g_h3, g);
        }
        __syncthreads();
        if (wid < 2 * NB) {   // owner of block (ob, ib) of g_W5: the G fragment comes straight from global (coalesced 64-byte rows)
            const int ob = wid >> 1, ib = wid & 1;
            const int o = 16 * ob + c16;
#pragma unroll 2
            for (int wv = 0; wv < NF_FP_BWD_WAVES; ++wv) {
                const int64_t r0 = (tile0 + wv) * 16;
                float ga[4], av[4];
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    const int64_t r2 = r0 + 4 * s2 + g;
                    if (MIX) {
                        ga[s2] = TGs[wv * TSZ + (4 * s2 + g) * NF_FP_ST + o];      // zero past O and past N by construction
                    } else {
                        const float v = gr.g_out[(r2 < N ? r2 : N - 1) * O + (o < O ? o : O - 1)];
                        ga[s2] = (r2 < N && o < O) ? v : 0.f;
                    }
                    av[s2] = TAs[wv * TSZ + (4 * s2 + g) * NF_FP_ST + 16 * ib + c16];
                }
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    b5 += ga[s2];
                    a5 = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s2], av[s2], a5, 0, 0, 0);
                }
            }
        }
        }
        // ---- phase 2: gate 2: h3 = h2 + y2 sigmoid(a2);  [y2, a2] = W2 q + b2 ----------------------------------------------
        float g_q[8];
        {
            NF_FPP_SET(1);
            float gcat[16];                                      // [g_y2 | g_a2]: the 64 outputs of conv2
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float s2 = nf_sigmoid(f.a2[j]);
                gcat[j] = g_h3[j] * s2;
                gcat[8 + j] = g_h3[j] * f.y2[j] * s2 * (1.f - s2);
            }
            f32x4 acc[2] = {nf_fp_zero4(), nf_fp_zero4()};
            nf_fp_gemm_d<4>(sm + L.W2, NF_FP_ST, 0, gcat, acc, c16, g);
#pragma unroll
            for (int j = 0; j < 8; ++j) g_q[j] = acc[j >> 2][j & 3];
            float half[8];
            nf_fp_store_rows(f.q, TA, c16, g);
#pragma unroll
            for (int j = 0; j < 8; ++j) half[j] = gcat[j];
            nf_fp_store_rows(half, TG, c16, g);
#pragma unroll
            for (int j = 0; j < 8; ++j) half[j] = gcat[8 + j];
            nf_fp_store_rows(half, TX, c16, g);
            __syncthreads();
            // g_W2 is 64 x 32: eight blocks, one per wave
            const int ob = wid >> 1, ib = wid & 1;
            b2 += nf_fp_coop_block(ob < 2 ? TGs : TXs, TAs, TSZ, 16 * (ob & 1), 16 * ib, a2, c16, g);
        }
        // ---- phase 3: q = Wq (h2 + pos) + bq ---------------------------------------------------------------------------
        float g_h2[8], g_h1[8];
        {
        NF_FPP_SET(2);
        {
            f32x4 acc[2] = {nf_fp_zero4(), nf_fp_zero4()};
            nf_fp_gemm_d<2>(sm + L.Wq, NF_FP_ST, 0, g_q, acc, c16, g);
            float tv[8], g_t[8];
            nf_fp_ldvec(sm + L.pos, g, tv);
#pragma unroll
            for (int j = 0; j < 8; ++j) { tv[j] += f.h2[j]; g_t[j] = acc[j >> 2][j & 3]; }
            nf_fp_store_rows(tv, TA, c16, g);
            nf_fp_store_rows(g_q, TG, c16, g);
            nf_fp_colsum(g_t, TX, vpos, c16, g);
#pragma unroll
            for (int j = 0; j < 8; ++j) g_h2[j] = g_h3[j] + g_t[j];
        }
        {   // LayerNorm 1
            float tmp[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) tmp[j] = g_h2[j] * f.xh1[j];
            nf_fp_colsum(tmp, TX, vln1g, c16, g);
            nf_fp_colsum(g_h2, TX, vln1b, c16, g);
            nf_fp_ln_bwd(g_h2, f.xh1, sm + L.ln1g, f.rstd1, g_h1, g);
        }
        __syncthreads();
        if (wid < 4) bq += nf_fp_coop_block(TGs, TAs, TSZ, 16 * (wid >> 1), 16 * (wid & 1), aq, c16, g);   // g_Wq: four blocks
        }
        // ---- phase 4: gate 1: h1 = h0 + elu(u) sigmoid(elu(-u));  u = Wg [elu(h0), elu(-h0)] + bg --------------------------
        float g_h0[8];
        {
            NF_FPP_SET(3);
            float g_u[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float y, a, dy, da;
                nf_celu(f.u[j], y, a);
                nf_celu_grad(f.u[j], dy, da);
                const float sa = nf_sigmoid(a);
                g_u[j] = g_h1[j] * sa * (dy - y * (1.f - sa) * da);
            }
            f32x4 acc0[2] = {nf_fp_zero4(), nf_fp_zero4()}, acc1[2] = {nf_fp_zero4(), nf_fp_zero4()};
            nf_fp_gemm_d<2>(sm + L.Wg, NF_FP_STG, 0, g_u, acc0, c16, g);
            nf_fp_gemm_d<2>(sm + L.Wg, NF_FP_STG, 32, g_u, acc1, c16, g);
            float c0[8], c1[8], d0[8], d1[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                nf_celu(f.h0[j], c0[j], c1[j]);
                nf_celu_grad(f.h0[j], d0[j], d1[j]);
                g_h0[j] = g_h1[j] + acc0[j >> 2][j & 3] * d0[j] - acc1[j >> 2][j & 3] * d1[j];
            }
            nf_fp_store_rows(g_u, TG, c16, g);
            nf_fp_store_rows(c0, TA, c16, g);
            nf_fp_store_rows(c1, TX, c16, g);
            __syncthreads();
            // g_Wg is 32 x 64: eight blocks, one per wave
            const int ob = wid >> 2, ib = wid & 3;
            const float bs = nf_fp_coop_block(TGs, ib < 2 ? TAs : TXs, TSZ, 16 * ob, 16 * (ib & 1), ag, c16, g);
            if (ib == 0) bg += bs;
        }
        // ---- phase 5: h0 = W0 x + b0 ---------------------------------------------------------------------------------------
        {
        NF_FPP_SET(4);
        nf_fp_store_rows(g_h0, TG, c16, g);
        if (gr.g_x != nullptr) {
            float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 16 * (j >> 2) + 4 * g + (j & 3);
                const float4 w0 = *(const float4*)(sm + L.W0 + 4 * k);
                s4[0] = fmaf(g_h0[j], w0.x, s4[0]); s4[1] = fmaf(g_h0[j], w0.y, s4[1]);
                s4[2] = fmaf(g_h0[j], w0.z, s4[2]); s4[3] = fmaf(g_h0[j], w0.w, s4[3]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float sv = nf_fp_rowsum(s4[i]);
                if (i < I0 && g == 0 && rv) {
                    float* dst = gr.g_x + row * gr.gxrs + i * gr.gxcs;
                    *dst = gr.gx_acc ? *dst + sv : sv;
                }
            }
        }
        __syncthreads();
        if (wid < 2) {   // g_W0 (32 x I0): two blocks; B fragment = x straight from global, zero beyond I0 / N
#pragma unroll 2
            for (int wv = 0; wv < NF_FP_BWD_WAVES; ++wv) {
                const int64_t r0 = (tile0 + wv) * 16;
                float ga[4], av[4];
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    const int64_t r2 = r0 + 4 * s2 + g;
                    const float v = w.x[(r2 < N ? r2 : N - 1) * w.xrs + (c16 < I0 ? c16 : 0) * w.xcs];
                    av[s2] = (r2 < N && c16 < I0) ? v : 0.f;
                    ga[s2] = TGs[wv * TSZ + (4 * s2 + g) * NF_FP_ST + 16 * wid + c16];
                }
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    b0 += ga[s2];
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s2], av[s2], a0, 0, 0, 0);
                }
            }
        }
        }
    }
#undef NF_FPP_SET
    __syncthreads();                                      // the tiles are re-used for the final reduction

    // ---- this workgroup's slab: every weight block has exactly one owner; the LayerNorm / pos sums are folded over the waves ---
    float* slab = slabs + (size_t)blockIdx.x * NF_S_END;
    const int c16 = lane0 & 15, g = lane0 >> 4;
    {
        const int ob = wid >> 1, ib = wid & 1;
        if (wid < 2 * NB) {
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[NF_S_W5 + (16 * ob + 4 * g + r) * 32 + 16 * ib + c16] = a5[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[NF_S_W2 + (16 * ob + 4 * g + r) * 32 + 16 * ib + c16] = a2[r];
        if (wid < 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[NF_S_WQ + (16 * ob + 4 * g + r) * 32 + 16 * ib + c16] = aq[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[NF_S_WG + (16 * (wid >> 2) + 4 * g + r) * 64 + 16 * (wid & 3) + c16] = ag[r];
        if (wid < 2 && c16 < 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[NF_S_W0 + (16 * wid + 4 * g + r) * 4 + c16] = a0[r];
        }
        b5 = nf_fp_rowsum(b5); b2 = nf_fp_rowsum(b2); bq = nf_fp_rowsum(bq); bg = nf_fp_rowsum(bg); b0 = nf_fp_rowsum(b0);
        if (g == 0) {
            if (wid < 2 * NB && ib == 0) slab[NF_S_B5 + 16 * ob + c16] = b5;
            if (ib == 0) slab[NF_S_B2 + 16 * ob + c16] = b2;
            if (wid < 4 && ib == 0) slab[NF_S_BQ + 16 * ob + c16] = bq;
            if ((wid & 3) == 0) slab[NF_S_BG + 16 * (wid >> 2) + c16] = bg;
            if (wid < 2) slab[NF_S_B0 + 16 * wid + c16] = b0;
        }
    }
    float* red = sm + L.tiles;                            // [waves][5][32]; the tiles are idle after the last barrier
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        vln2g[a] = nf_fp_rowsum(vln2g[a]); vln2b[a] = nf_fp_rowsum(vln2b[a]); vln1g[a] = nf_fp_rowsum(vln1g[a]);
        vln1b[a] = nf_fp_rowsum(vln1b[a]); vpos[a] = nf_fp_rowsum(vpos[a]);
        if (g == 0) {
            float* rw = red + wid * 160 + 16 * a + c16;
            rw[0] = vln2g[a]; rw[32] = vln2b[a]; rw[64] = vln1g[a]; rw[96] = vln1b[a]; rw[128] = vpos[a];
        }
    }
    if (MIX) {
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const float t = nf_wave_sum(macc[q]);
            if (lane0 == 0) red[NF_FP_BWD_WAVES * 160 + wid * 8 + q] = t;
        }
    }
    __syncthreads();
    if (threadIdx.x < 160) {
        float t = 0.f;
#pragma unroll
        for (int wv = 0; wv < NF_FP_BWD_WAVES; ++wv) t += red[wv * 160 + threadIdx.x];
        const int v = threadIdx.x >> 5, k = threadIdx.x & 31;
        const int base = v == 0 ? NF_S_LN2G : v == 1 ? NF_S_LN2B : v == 2 ? NF_S_LN1G : v == 3 ? NF_S_LN1B : NF_S_POS;
        slab[base + k] = t;
    } else if (threadIdx.x < 160 + 64) {                  // the slab is NF_S_END wide whether or not the coupling rides along
        const int q = threadIdx.x - 160;
        float t = 0.f;
        if (MIX && q < 7)
#pragma unroll
            for (int wv = 0; wv < NF_FP_BWD_WAVES; ++wv) t += red[NF_FP_BWD_WAVES * 160 + wv * 8 + q];
        slab[NF_S_MIX + q] = t;
    }
}

// dst += sum over the blocks' slabs.  grid (NF_S_END / 64, 4): 64 slab entries x 4 slab groups per block, the y index picks
// a quarter of the slabs (<= 16 independent loads per thread); the four partial sums meet in the destination by atomics.
__device__ __forceinline__ void nf_fpp_finalize_body(const float* __restrict__ slabs, int nblk, const NfFppG& gr, int I0, int O,
                                                     const NfFppMix& mx) {
    __shared__ float red[4][64];
    const int el = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + el;
    float p[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int b = blockIdx.y * 64 + k * 4 + grp;
        p[k] = b < nblk ? slabs[(size_t)b * NF_S_END + e] : 0.f;
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += p[k];
    red[grp][el] = s;
    __syncthreads();
    if (grp != 0) return;                                 // wave 0 adds for the workgroup: one lane per slab entry (deterministic mode: the
    NF_DET_COL_CHAIN();                                   // the <= 4 workgroups of the slab quarters of (entries blockIdx.x, step blockIdx.z): a chain
    NF_DET_ENTER_WAVE_K(nf_fpc);                          // of their own (round 6; one grid-wide chain of 3 840 workgroups before)
    const bool live = blockIdx.y * 64 < nblk;
    s = (red[0][el] + red[1][el]) + (red[2][el] + red[3][el]);
    if (!live) {}
    else if (e < NF_S_W2) { if ((e >> 5) < O) atomicAdd(gr.g_W5 + e, s); }
    else if (e < NF_S_WQ) atomicAdd(gr.g_W2 + e - NF_S_W2, s);
    else if (e < NF_S_WG) atomicAdd(gr.g_Wq + e - NF_S_WQ, s);
    else if (e < NF_S_W0) atomicAdd(gr.g_Wg + e - NF_S_WG, s);
    else if (e < NF_S_B5) { const int k = (e - NF_S_W0) >> 2, i = (e - NF_S_W0) & 3; if (i < I0) atomicAdd(gr.g_W0 + k * I0 + i, s); }
    else if (e < NF_S_B2) { if (e - NF_S_B5 < O) atomicAdd(gr.g_b5 + e - NF_S_B5, s); }
    else if (e < NF_S_BQ) atomicAdd(gr.g_b2 + e - NF_S_B2, s);
    else if (e < NF_S_BG) atomicAdd(gr.g_bq + e - NF_S_BQ, s);
    else if (e < NF_S_LN2G) atomicAdd(gr.g_bg + e - NF_S_BG, s);
    else if (e < NF_S_LN2B) atomicAdd(gr.g_ln2g + e - NF_S_LN2G, s);
    else if (e < NF_S_LN1G) atomicAdd(gr.g_ln2b + e - NF_S_LN2B, s);
    else if (e < NF_S_LN1B) atomicAdd(gr.g_ln1g + e - NF_S_LN1G, s);
    else if (e < NF_S_POS) atomicAdd(gr.g_ln1b + e - NF_S_LN1B, s);
    else if (e < NF_S_B0) atomicAdd(gr.g_pos + e - NF_S_POS, s);
    else if (e < NF_S_MIX) atomicAdd(gr.g_b0 + e - NF_S_B0, s);
    else if (mx.g_scale != nullptr) {                     // the fused coupling's sums (mixlog.hip: k_mixlog_oct_bwd's tail)
        const int q = e - NF_S_MIX, o0 = mx.odd, o1 = 1 ^ mx.odd;
        if (q == 0) atomicAdd(mx.g_scale, s);
        else if (q == 1) atomicAdd(mx.g_bias, s);
        else if (mx.nls != nullptr && q < 7) {            // next ActNorm: g_log_scale_c = -sum g_h h - sum g_ld, g_bias_c = -sum g_h / e^ls
            if (q == 2) atomicAdd(mx.g_nb + o0, -s / expf(mx.nls[o0]));
            else if (q == 3) atomicAdd(mx.g_nls + o0, -s);
            else if (q == 4) atomicAdd(mx.g_nb + o1, -s / expf(mx.nls[o1]));
            else if (q == 5) atomicAdd(mx.g_nls + o1, -s);
            else { atomicAdd(mx.g_nls + o0, -s); atomicAdd(mx.g_nls + o1, -s); }
        }
    }
    NF_DET_LEAVE_WAVE_K(nf_fpc);
}

__global__ void __launch_bounds__(256) k_flowpp_cond_finalize(const float* __restrict__ slabs, int nblk, NfFppG gr, int I0,
                                                              int O, NfFppMix mx) {
    nf_fpp_finalize_body(slabs, nblk, gr, I0, O, mx);
}

// the finalizes of up to NF_FP_FIN_STEPS flow steps in one launch (blockIdx.z = step): nothing downstream of a step's backward
// waits for its parameter gradients, so the per-step finalize launches (7 us + a launch gap each, 32 per C3 backward pass) are
// deferred -- every step keeps its own slab workspace -- and run together after the last step (nf_flowpp_vec_step_finalize)
#define NF_FP_FIN_STEPS 8
struct NfFppFinStep { const float* slabs; NfFppG g; NfFppMix mx; };
struct NfFppFinArgs { NfFppFinStep st[NF_FP_FIN_STEPS]; };
static_assert(sizeof(NfFppFinArgs) <= 3584, "finalize descriptors travel in the kernel argument segment");

__global__ void __launch_bounds__(256) k_flowpp_cond_finalize_multi(NfFppFinArgs a, int nblk, int I0, int O) {
    const NfFppFinStep& st = a.st[blockIdx.z];
    nf_fpp_finalize_body(st.slabs, nblk, st.g, I0, O, st.mx);
}

static_assert(NF_FP_MAX_BLOCKS * NF_S_END == NF_FLOWPP_BWD_WS_FLOATS, "workspace size in include/nfhip.h");

// phase: 0 = both kernels, 1 = the backward kernel only, 2 = the slab finalize only (so that the caller can put the finalize
// on a second stream, where it overlaps the next flow step's backward kernel; it then owns the workspace until it is done)
template <int NB, bool MIX = false>
static int nf_fpp_launch_bwd(const NfFppW& w, const NfFppG& g, float* workspace, int64_t N, int I0, int O, hipStream_t stream,
                             const NfFppMix& mx = NfFppMix{}, int phase = 0) {
    const int64_t tiles = (N + 15) / 16;
    int64_t gx = (tiles + NF_FP_BWD_WAVES - 1) / NF_FP_BWD_WAVES;
    if (gx > NF_FP_MAX_BLOCKS) gx = NF_FP_MAX_BLOCKS;            // one 8-wave block per CU
    const NfFppL L = nf_fpp_layout(NF_FP_BWD_WAVES);
    const size_t lds = (size_t)L.total * sizeof(float);
    static bool attr_set = false;                                // > 64 KB of dynamic LDS needs the opt-in, once per kernel
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_flowpp_cond_bwd<NB, MIX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int iters = (int)((tiles + gx * NF_FP_BWD_WAVES - 1) / (gx * NF_FP_BWD_WAVES));
    if (phase != 2) {
        hipLaunchKernelGGL((k_flowpp_cond_bwd<NB, MIX>), dim3((unsigned)gx), dim3(NF_FP_BWD_WAVES * NF_WAVE), lds, stream, w, g,
                           workspace, N, I0, O, tiles, iters, nf_fpp_vec_ok(w) ? 1 : 0, mx);
        NF_CHECK_LAUNCH();
    }
    if (phase == 1) return 0;
    hipLaunchKernelGGL(k_flowpp_cond_finalize, dim3(NF_S_END / 64, (unsigned)((gx + 63) / 64)), dim3(256), 0, stream, (const float*)workspace, (int)gx, g,
                       I0, O, mx);
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_flowpp_cond_bwd(const float* x, const float* W0, const float* b0, const float* Wg, const float* bg,
                                  const float* ln1_g, const float* ln1_b, const float* pos, const float* Wq, const float* bq,
                                  const float* W2, const float* b2, const float* ln2_g, const float* ln2_b, const float* W5,
                                  const float* b5, const float* g_out, float* g_x, float* g_W0, float* g_b0, float* g_Wg,
                                  float* g_bg, float* g_ln1_g, float* g_ln1_b, float* g_pos, float* g_Wq, float* g_bq,
                                  float* g_W2, float* g_b2, float* g_ln2_g, float* g_ln2_b, float* g_W5, float* g_b5,
                                  float* workspace, int64_t x_row_stride, int x_col_stride, int64_t gx_row_stride, int gx_col_stride,
                                  int gx_accumulate, int64_t N, int I0, int O, nf_stream_t stream) {
    if (I0 < 1 || I0 > 4 || O < 1 || O > 64 || workspace == nullptr) return NF_E_BADARG;
    if (N <= 0) return N == 0 ? 0 : NF_E_BADARG;
    NfFppW w{x, W0, b0, Wg, bg, ln1_g, ln1_b, pos, Wq, bq, W2, b2, ln2_g, ln2_b, W5, b5, nullptr, x_row_stride, x_col_stride};
    NfFppG g{g_out, g_x, g_W0, g_b0, g_Wg, g_bg, g_ln1_g, g_ln1_b, g_pos, g_Wq, g_bq, g_W2, g_b2, g_ln2_g, g_ln2_b, g_W5, g_b5,
             gx_row_stride, gx_col_stride, gx_accumulate};
    switch ((O + 15) / 16) {
        case 1: return nf_fpp_launch_bwd<1>(w, g, workspace, N, I0, O, (hipStream_t)stream);
        case 2: return nf_fpp_launch_bwd<2>(w, g, workspace, N, I0, O, (hipStream_t)stream);
        case 3: return nf_fpp_launch_bwd<3>(w, g, workspace, N, I0, O, (hipStream_t)stream);
        default: return nf_fpp_launch_bwd<4>(w, g, workspace, N, I0, O, (hipStream_t)stream);
    }
}

// the whole backward of a Flow++ density flow step on (N, 2) data: coupling (flows/coupling.py:172-210) + conditioner + optionally
// the next step's ActNorm, two launches (nfhip.h)
extern "C" int nf_flowpp_vec_step_bwd(const float* g_h, const float* g_ld, const float* z, const float* params, const float* W0,
                                      const float* b0, const float* Wg, const float* bg, const float* ln1_g, const float* ln1_b,
                                      const float* pos, const float* Wq, const float* bq, const float* W2, const float* b2,
                                      const float* ln2_g, const float* ln2_b, const float* W5, const float* b5,
                                      const float* a_log_scale, const float* a_bias, const float* next_log_scale,
                                      const float* next_bias, float* g_z, float* g_W0, float* g_b0, float* g_Wg, float* g_bg,
                                      float* g_ln1_g, float* g_ln1_b, float* g_pos, float* g_Wq, float* g_bq, float* g_W2,
                                      float* g_b2, float* g_ln2_g, float* g_ln2_b, float* g_W5, float* g_b5, float* g_scale,
                                      float* g_bias, float* g_next_log_scale, float* g_next_bias, float* workspace, int K,
                                      float logit_eps, int odd, int64_t N, int phase, nf_stream_t stream) {
    const int O = 2 + 3 * K;
    if (phase < 0 || phase > 2) return NF_E_BADARG;
    if (K < 1 || K > 8 || workspace == nullptr || g_h == nullptr || g_ld == nullptr || z == nullptr || params == nullptr ||
        g_z == nullptr || g_scale == nullptr || g_bias == nullptr || a_log_scale == nullptr || a_bias == nullptr)
        return NF_E_BADARG;
    if ((next_log_scale == nullptr) != (next_bias == nullptr)) return NF_E_BADARG;
    if (next_log_scale != nullptr && (g_next_log_scale == nullptr || g_next_bias == nullptr)) return NF_E_BADARG;
    if (N <= 0) return N == 0 ? 0 : NF_E_BADARG;
    const int sel1 = odd ? 0 : 1;                                            // the conditioning feature (squeeze.py:68-69)
    NfFppW w{z + sel1, W0, b0, Wg, bg, ln1_g, ln1_b, pos, Wq, bq, W2, b2, ln2_g, ln2_b, W5, b5, nullptr, 2, 2};
    NfFppG g{nullptr, g_z + sel1, g_W0, g_b0, g_Wg, g_bg, g_ln1_g, g_ln1_b, g_pos, g_Wq, g_bq, g_W2, g_b2, g_ln2_g, g_ln2_b, g_W5, g_b5,
             2, 2, 1};
    NfFppMix mx{z, params, g_h, g_ld, a_log_scale, a_bias, next_log_scale, next_bias, g_z, g_scale, g_bias, g_next_log_scale,
                g_next_bias, odd ? 1 : 0, K, logit_eps};
    if (O <= 16) return nf_fpp_launch_bwd<1, true>(w, g, workspace, N, 1, O, (hipStream_t)stream, mx, phase);
    return nf_fpp_launch_bwd<2, true>(w, g, workspace, N, 1, O, (hipStream_t)stream, mx, phase);
}


extern "C" int nf_flowpp_vec_step_finalize(const nf_flowpp_fin_desc* descs, int n, int K, int64_t N, nf_stream_t stream) {
    if (descs == nullptr || n < 0 || K < 1 || K > 8) return NF_E_BADARG;
    if (n == 0 || N <= 0) return (n == 0 || N == 0) ? 0 : NF_E_BADARG;
    const int O = 2 + 3 * K;
    const int64_t tiles = (N + 15) / 16;
    int64_t gx = (tiles + NF_FP_BWD_WAVES - 1) / NF_FP_BWD_WAVES;
    if (gx > NF_FP_MAX_BLOCKS) gx = NF_FP_MAX_BLOCKS;            // = the grid of the backward kernel that wrote the slabs
    for (int s0 = 0; s0 < n; s0 += NF_FP_FIN_STEPS) {
        const int ns = n - s0 < NF_FP_FIN_STEPS ? n - s0 : NF_FP_FIN_STEPS;
        NfFppFinArgs a{};
        for (int k = 0; k < ns; ++k) {
            const nf_flowpp_fin_desc& d = descs[s0 + k];
            if (d.workspace == nullptr || d.g_scale == nullptr || d.g_bias == nullptr) return NF_E_BADARG;
            if (d.next_log_scale != nullptr && (d.g_next_log_scale == nullptr || d.g_next_bias == nullptr)) return NF_E_BADARG;
            a.st[k].slabs = d.workspace;
            a.st[k].g = NfFppG{nullptr, nullptr, d.g_W0, d.g_b0, d.g_Wg, d.g_bg, d.g_ln1_g, d.g_ln1_b, d.g_pos, d.g_Wq, d.g_bq, d.g_W2,
                               d.g_b2, d.g_ln2_g, d.g_ln2_b, d.g_W5, d.g_b5, 2, 2, 1};
            a.st[k].mx = NfFppMix{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, d.next_log_scale, nullptr, nullptr, d.g_scale,
                                  d.g_bias, d.g_next_log_scale, d.g_next_bias, d.odd ? 1 : 0, K, 0.f};
        }
        hipLaunchKernelGGL(k_flowpp_cond_finalize_multi, dim3(NF_S_END / 64, (unsigned)((gx + 63) / 64), (unsigned)ns), dim3(256), 0,
                           (hipStream_t)stream, a, (int)gx, 1, O);
    }
    NF_CHECK_LAUNCH();
    return 0;
}
