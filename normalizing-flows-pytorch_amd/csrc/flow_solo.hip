// The RealNVP density flow (flows/realnvp.py:49-53: S x [BatchNorm(affine=False), AffineCoupling around the MLP conditioner of
// flows/modules.py:391-413]) for batches of N <= 256 rows with D = 2 -- config 1 of BASELINE.json -- as ONE workgroup per direction.
//
// Why: at 256 rows the grid kernels of mlp_chain.hip are two workgroups that meet in global memory twelve times per flow step (six
// training-mode BatchNorm reductions forward, six backward): 1.5 us per meeting, 0.58 of the 1.49 ms of a train step.  One workgroup
// needs no global meeting at all -- if the whole batch fits its registers.  It does in the TRANSPOSED form:
//   * four waves, one per SIMD (512 registers each); wave w owns batch columns 64 w .. 64 w + 63 as two blocks of 32;
//   * an activation is features x batch: lane (c32, hs) of block b holds column 64 w + 32 b + c32, its sixteen registers the features
//     fm(r, hs) = (r & 3) + 8 (r >> 2) + 4 hs -- the C / D layout of v_mfma_f32_32x32x2_f32;
//   * a linear layer is D = W x act with the WEIGHTS as the A operand: K step r contracts the features fm(r, 0), fm(r, 1), so the B
//     operand of K step r is register r of the previous layer's result, as it lies -- layers chain register to register, forward
//     (A = v[o][fm]) and for the data gradient (A = v[fm][i]); no LDS between two layers;
//   * sums over the batch (BatchNorm statistics, BatchNorm-backward sums) are a 16-shuffle butterfly per wave (nf_cv_butterfly16) and
//     one LDS meeting of the four waves: two barriers, no global traffic; statistics are merged as (count, mean, M2) with a per-wave
//     centre, so one pass is as exact as two;
//   * weight gradients contract over the batch: both operands are transposed through a wave-private LDS tile (32 writes + 8 reads of
//     16 bytes per tensor), every wave leaves its own partial product in the slab format of mlp_chain.hip (wave w = workgroup w >> 1,
//     row group w & 1 there), so the deferred fold (k_glow_fold_all) and the weight-norm backward are shared unchanged.
// Numerics are those of mlp_chain.hip / linear_bn.hip: fp32 MFMA products, biased variance for normalisation, unbiased for the running
// estimate, weight-norm as a scale of the activation column (weight_norm.py:40), flow-BatchNorm statistics as buffers (no gradient
// through them, modules.py:285-296).
#include <cstdlib>

#include "nf_common.h"

#include "nf_conv_core.h"

#include "nf_flow_rec.h"

#define NF_SO_THREADS 256
#define NF_SO_ST 33                                   // row stride of a staged 32 x 32 weight matrix (odd: conflict-free both ways)
#define NF_SO_TS 68                                   // row stride of a transposition tile (16-byte rows, conflict-free b128 reads)
typedef float f32x4s __attribute__((ext_vector_type(4)));

// LDS map (floats)
#define SO_W 0                                        // [4][32 * 33] v of the hidden linears 1 .. 4
#define SO_V0 (SO_W + 4 * 32 * NF_SO_ST)              // [32] v_0[o][0]
#define SO_V5 (SO_V0 + 32)                            // [2][32] v_5
#define SO_WS (SO_V5 + 64)                            // [6][32] weight-norm column scales g_i / (||v[:, i]|| + eps)
#define SO_G (SO_WS + 192)                            // [6][32] gains
#define SO_B (SO_G + 192)                             // [6][32] biases
#define SO_GA (SO_B + 192)                            // [5][32] BatchNorm gamma
#define SO_BE (SO_GA + 160)                           // [5][32] BatchNorm beta
#define SO_BNC (SO_BE + 160)                          // [5][4][32] scale, shift, mean, invstd
#define SO_HD (SO_BNC + 640)                          // head: [0..1] exp(log_gamma) [2..3] beta [4..5] mean [6..7] sqrt(var) [8] log-det [9] a [10] c [12..15] raw ls, bs
#define SO_RED (SO_HD + 32)                           // [4 waves][3][32] partial sums of a meeting (sum, squares / second sum, centre)
#define SO_TOT (SO_RED + 4 * 3 * 32)                  // [3][32] totals of a meeting
#define SO_REC (SO_TOT + 96)                          // [2][record words as floats x 2]
#define SO_REC_WORDS ((int)((sizeof(NfGlowFlowStep) + 7) / 8))
#define SO_TILES (SO_REC + 2 * 2 * SO_REC_WORDS)      // backward: [4 waves][2][32 * 68] transposition tiles (activation | gradient)
static_assert(SO_REC_WORDS <= NF_SO_THREADS, "one 8-byte word of a step record per thread");
static inline size_t nf_so_lds_bytes(bool bwd) { return sizeof(float) * (size_t)(SO_TILES + (bwd ? 4 * 2 * 32 * NF_SO_TS : 0)); }

__device__ __forceinline__ int so_fm(int r, int hs) { return (r & 3) + 8 * (r >> 2) + 4 * hs; }
// the lane's sixteen features of a 32-vector in LDS: four 16-byte reads
__device__ __forceinline__ void so_ldvec(const float* base, int hs, float (&out)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4s v = *(const f32x4s*)(base + 8 * q + 4 * hs);
#pragma unroll
        for (int k = 0; k < 4; ++k) out[4 * q + k] = v[k];
    }
}
// A operands from the staged matrix of hidden linear l (1 .. 4): forward v[o = c32][fm(r, hs)], data gradient v[fm(r, hs)][i = c32]
__device__ __forceinline__ void so_ld_afwd(const float* sm, int l, int c32, int hs, float (&a)[16]) {
    const float* W = sm + SO_W + (l - 1) * 32 * NF_SO_ST + c32 * NF_SO_ST;
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = W[so_fm(r, hs)];
}
__device__ __forceinline__ void so_ld_abwd(const float* sm, int l, int c32, int hs, float (&a)[16]) {
    const float* W = sm + SO_W + (l - 1) * 32 * NF_SO_ST + c32;
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = W[so_fm(r, hs) * NF_SO_ST];
}
// acc[b] = A x B[b] over the sixteen K steps; B is the lane's own registers
__device__ __forceinline__ void so_gemm(const float (&a)[16], const float (&bv)[2][16], f32x16 (&acc)[2]) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], bv[b][r], acc[b], 0, 0, 0);
}
// The workgroup's meetings order LDS traffic only (nothing in a step is handed from wave to wave through global memory): a barrier
// that waits for the LDS counter alone.  __syncthreads() also drains the vector-memory counter, i.e. it waits for the parameter
// prefetch of the next step and for every statistics / slab store in flight -- one global round trip per meeting (measured: 21 us
// per forward step with it).
__device__ __forceinline__ void so_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// sum over the 32 lanes of a wave half (every lane ends with the total)
__device__ __forceinline__ float so_half_sum(float v) {
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, NF_WAVE);
    return v;
}

// ---- what a thread prefetches of a step's parameters (all loads of a step in flight at once, one step ahead) ----------------------
struct SoParams {
    float w[4][4];                                    // v_1 .. v_4, elements t + 256 e (consecutive lanes, consecutive addresses: no alignment demand)
    float g, b;                                       // t < 192: gain / bias of linear t >> 5, index t & 31
    float ga, be, rm, rv;                             // t < 160: BatchNorm t >> 5
    float x0, x1;                                     // 192 <= t < 224: v_0[t - 192]; 224 <= t: v_5[0 .. 1][t - 224]; 160 <= t < 192: head scalars
    long long nbt;                                    // t = 32 j: num_batches_tracked of BatchNorm j
};
__device__ __forceinline__ void so_prefetch(SoParams& P, const NfGlowFlowStep& st, int t) {
    const int k = t & 31;
#pragma unroll
    for (int l = 0; l < 4; ++l)
#pragma unroll
        for (int e = 0; e < 4; ++e) P.w[l][e] = st.p.v[l + 1][t + 256 * e];
    P.g = 0.f; P.b = 0.f; P.ga = 0.f; P.be = 0.f; P.rm = 0.f; P.rv = 0.f; P.x0 = 0.f; P.x1 = 0.f; P.nbt = 0;
    if (t < 192) {
        const int l = t >> 5;
        const int I = l == 0 ? 1 : 32, O = l == 5 ? 2 : 32;
        if (k < I) P.g = st.p.g[l][k];
        if (k < O) P.b = st.p.b[l][k];
    }
    if (t < 160) {
        const int j = t >> 5;
        P.ga = st.p.gamma[j][k]; P.be = st.p.beta[j][k]; P.rm = st.p.rmean[j][k]; P.rv = st.p.rvar[j][k];
        if (k == 0 && st.p.nbt[j] != nullptr) P.nbt = st.p.nbt[j][0];
    } else if (t < 192) {
        const int q = t - 160;
        if (q < 2) P.x0 = st.h.ls[q];
        else if (q < 4) P.x0 = st.h.bs[q - 2];
        else if (q == 4) P.x0 = st.h.a[0];
        else if (q == 5) P.x0 = st.h.c[0];
        else if (q < 8) P.x0 = st.h.rmean[q - 6];
        else if (q < 10) P.x0 = st.h.rvar[q - 8];
    } else if (t < 224) {
        P.x0 = st.p.v[0][t - 192];
    } else {
        P.x0 = st.p.v[5][t - 224]; P.x1 = st.p.v[5][32 + t - 224];
    }
}
// registers -> LDS tables; the caller brackets it with barriers.  Returns nothing: rm / rv / the head's running statistics stay in P.
__device__ __forceinline__ void so_stage(float* sm, const SoParams& P, int t) {
    const int k = t & 31;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = t + 256 * e;
            sm[SO_W + l * 32 * NF_SO_ST + (idx >> 5) * NF_SO_ST + (idx & 31)] = P.w[l][e];
        }
    }
    if (t < 192) { sm[SO_G + t] = P.g; sm[SO_B + t] = P.b; }
    if (t < 160) { sm[SO_GA + t] = P.ga; sm[SO_BE + t] = P.be; }
    else if (t < 192) {
        const int q = t - 160;
        if (q < 2) { sm[SO_HD + q] = expf(P.x0); sm[SO_HD + 12 + q] = P.x0; }
        else if (q < 4) sm[SO_HD + q] = P.x0;
        else if (q == 4) sm[SO_HD + 9] = P.x0;
        else if (q == 5) sm[SO_HD + 10] = P.x0;
    } else if (t < 224) sm[SO_V0 + k] = P.x0;
    else { sm[SO_V5 + k] = P.x0; sm[SO_V5 + 32 + k] = P.x1; }
}
// the weight-norm column scales of the six linears (threads 0 .. 191, after the staging barrier)
__device__ __forceinline__ void so_weight_norm(float* sm, int t, float wn_eps) {
    if (t >= 192) return;
    const int l = t >> 5, k = t & 31;
    float ss = 0.f;
    if (l == 0) {
#pragma unroll 8
        for (int o = 0; o < 32; ++o) ss = fmaf(sm[SO_V0 + o], sm[SO_V0 + o], ss);
    } else if (l == 5) {
        ss = sm[SO_V5 + k] * sm[SO_V5 + k] + sm[SO_V5 + 32 + k] * sm[SO_V5 + 32 + k];
    } else {
        const float* W = sm + SO_W + (l - 1) * 32 * NF_SO_ST + k;
#pragma unroll 8
        for (int o = 0; o < 32; ++o) ss = fmaf(W[o * NF_SO_ST], W[o * NF_SO_ST], ss);
    }
    const bool live = l != 0 || k == 0;
    sm[SO_WS + t] = live ? sm[SO_G + t] / (sqrtf(ss) + wn_eps) : 0.f;
}

// ---- meetings of the four waves --------------------------------------------------------------------------------------------------
// NQ per-lane feature arrays (already summed over the lane's two blocks) -> SO_TOT[q][feature]: butterfly, LDS, two barriers
template <int NQ>
__device__ __forceinline__ void so_colsums(float* sm, const float (&p)[NQ][16], int c32, int hs, int wid) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const float t = nf_cv_butterfly16(p[q], c32);
        if ((c32 & 1) == 0) sm[SO_RED + (wid * 3 + q) * 32 + so_fm(c32 >> 1, hs)] = t;
    }
    so_barrier();
    if (threadIdx.x < NQ * 32) {
        const int q = threadIdx.x >> 5, f = threadIdx.x & 31;
        sm[SO_TOT + q * 32 + f] = (sm[SO_RED + (0 * 3 + q) * 32 + f] + sm[SO_RED + (1 * 3 + q) * 32 + f]) +
                                  (sm[SO_RED + (2 * 3 + q) * 32 + f] + sm[SO_RED + (3 * 3 + q) * 32 + f]);
    }
    so_barrier();
}
// columns of wave w inside the batch
__device__ __forceinline__ int so_wave_rows(int N, int w) { return min(max(N - 64 * w, 0), 64); }

// training-mode BatchNorm j over a (= the producing linear's output incl. bias and residual): statistics of the whole batch, constants
// -> SO_BNC, bookkeeping (running statistics, saved mean / invstd) by the finalising threads 32 j .. 32 j + 31, which hold the old
// running statistics in P
__device__ __forceinline__ void so_batchnorm_train(float* sm, int j, const float (&a)[2][16], const bool (&cv)[2], int N, float eps, float mom,
                                                   float rm_old, float rv_old, long long nbt_old, const NfGlowFlowStep& st, float* save, int c32, int hs,
                                                   int wid) {
    // per-wave centre: the wave's first column (exists whenever the wave has any column of the batch)
    float c[16], p[2][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        c[r] = __shfl(a[0][r], hs * 32, NF_WAVE);
        const float d0 = cv[0] ? a[0][r] - c[r] : 0.f, d1 = cv[1] ? a[1][r] - c[r] : 0.f;
        p[0][r] = d0 + d1;
        p[1][r] = fmaf(d0, d0, d1 * d1);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float t = nf_cv_butterfly16(p[q], c32);
        if ((c32 & 1) == 0) sm[SO_RED + (wid * 3 + q) * 32 + so_fm(c32 >> 1, hs)] = t;
    }
    if (c32 == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sm[SO_RED + (wid * 3 + 2) * 32 + so_fm(r, hs)] = c[r];
    }
    so_barrier();
    const int t = threadIdx.x;
    if ((t >> 5) == j) {                              // one half wave finalises: (count, mean, M2) of the four waves merged in order
        const int f = t & 31;
        float n = 0.f, mean = 0.f, M2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float nw = (float)so_wave_rows(N, w);
            if (nw > 0.f) {
                const float s1 = sm[SO_RED + (w * 3 + 0) * 32 + f], s2 = sm[SO_RED + (w * 3 + 1) * 32 + f], cw = sm[SO_RED + (w * 3 + 2) * 32 + f];
                const float mw = cw + s1 / nw, m2w = fmaxf(s2 - s1 * s1 / nw, 0.f);
                const float tot = n + nw, dl = mw - mean;
                M2 = M2 + m2w + dl * dl * (n * nw / tot);
                mean = mean + dl * (nw / tot);
                n = tot;
            }
        }
        const float var = M2 / (float)N;               // biased, as BatchNorm normalises
        const float invstd = 1.f / sqrtf(var + eps);
        const float sc = sm[SO_GA + j * 32 + f] * invstd;
        sm[SO_BNC + (4 * j + 0) * 32 + f] = sc;
        sm[SO_BNC + (4 * j + 1) * 32 + f] = sm[SO_BE + j * 32 + f] - mean * sc;
        sm[SO_BNC + (4 * j + 2) * 32 + f] = mean;
        sm[SO_BNC + (4 * j + 3) * 32 + f] = invstd;
        const float unb = N > 1 ? var * ((float)N / (float)(N - 1)) : var;
        st.p.rmean[j][f] = (1.f - mom) * rm_old + mom * mean;
        st.p.rvar[j][f] = (1.f - mom) * rv_old + mom * unb;
        save[(2 * j + 0) * 32 + f] = mean;
        save[(2 * j + 1) * 32 + f] = invstd;
        if (f == 0 && st.p.nbt[j] != nullptr) st.p.nbt[j][0] = nbt_old + 1;      // (no load here: it would wait for the prefetch in flight)
    }
    so_barrier();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// forward, training mode
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(NF_SO_THREADS) k_solo_fwd(const NfGlowFlowStep* __restrict__ steps, int S, const float* __restrict__ z0,
                                                            float* __restrict__ ys, float* __restrict__ ld, float* __restrict__ saves,
                                                            int save_stride, int N, float eps, float mom, float wn_eps) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int t = threadIdx.x, lane = t & 63, wid = __builtin_amdgcn_readfirstlane(t >> 6), c32 = lane & 31, hs = lane >> 5;
    unsigned long long (*rec)[SO_REC_WORDS] = reinterpret_cast<unsigned long long (*)[SO_REC_WORDS]>(sm + SO_REC);
    const bool rt = t < SO_REC_WORDS;
    if (rt) {
        rec[0][t] = reinterpret_cast<const unsigned long long*>(steps)[t];
        if (S > 1) rec[1][t] = reinterpret_cast<const unsigned long long*>(steps + 1)[t];
    }
    int col[2];
    bool cv[2];
    float z[2][2], ldv[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        col[b] = 64 * wid + 32 * b + c32;
        cv[b] = col[b] < N;
        const int cc = cv[b] ? col[b] : 0;
        z[b][0] = cv[b] ? z0[2 * cc] : 0.f;
        z[b][1] = cv[b] ? z0[2 * cc + 1] : 0.f;
        ldv[b] = cv[b] ? ld[cc] : 0.f;
    }
    so_barrier();
    SoParams P;
    so_prefetch(P, *reinterpret_cast<const NfGlowFlowStep*>(rec[0]), t);
#pragma unroll 1
    for (int s = 0; s < S; ++s) {
        const NfGlowFlowStep& st = *reinterpret_cast<const NfGlowFlowStep*>(rec[s & 1]);
        float* save = saves + (size_t)s * save_stride;
        so_stage(sm, P, t);
        const float rm_old = P.rm, rv_old = P.rv, x0_old = P.x0;
        const long long nbt_old = P.nbt;   // (the old running statistics of this step: the finalising threads need them)
        const float frv_all = __shfl(P.x0, 40 + (t & 1), NF_WAVE);   // wave 2: lanes 38, 39 take the head's running variance of channel 0, 1
        // flow-BatchNorm partial sums of z around the wave's first column, in the same meeting as the staging barrier
        {
            float q4[4];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float cen = __shfl(z[0][c], 0, NF_WAVE);
                const float d0 = cv[0] ? z[0][c] - cen : 0.f, d1 = cv[1] ? z[1][c] - cen : 0.f;
                q4[c] = so_half_sum(d0 + d1);
                q4[2 + c] = so_half_sum(fmaf(d0, d0, d1 * d1));
                if (lane == 0) sm[SO_RED + (wid * 3 + 2) * 32 + c] = cen;
            }
            if (lane == 0) {
                sm[SO_RED + (wid * 3 + 0) * 32 + 0] = q4[0]; sm[SO_RED + (wid * 3 + 0) * 32 + 1] = q4[1];
                sm[SO_RED + (wid * 3 + 1) * 32 + 0] = q4[2]; sm[SO_RED + (wid * 3 + 1) * 32 + 1] = q4[3];
            }
        }
        so_barrier();
        unsigned long long nxt = 0;
        if (rt && s + 2 < S) nxt = reinterpret_cast<const unsigned long long*>(steps + s + 2)[t];
        if (s + 1 < S) so_prefetch(P, *reinterpret_cast<const NfGlowFlowStep*>(rec[(s + 1) & 1]), t);
        so_weight_norm(sm, t, wn_eps);
        if (t == 166 || t == 167) {                   // flow BatchNorm (modules.py:283-296): batch statistics are buffers; these two threads
            const int c = t - 166;                     // hold the old running mean of channel c (their neighbours' lanes the variance)
            float n = 0.f, mean = 0.f, M2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float nw = (float)so_wave_rows(N, w);
                if (nw > 0.f) {
                    const float s1 = sm[SO_RED + (w * 3 + 0) * 32 + c], s2 = sm[SO_RED + (w * 3 + 1) * 32 + c], cw = sm[SO_RED + (w * 3 + 2) * 32 + c];
                    const float mw = cw + s1 / nw, m2w = fmaxf(s2 - s1 * s1 / nw, 0.f);
                    const float tot = n + nw, dl = mw - mean;
                    M2 = M2 + m2w + dl * dl * (n * nw / tot);
                    mean = mean + dl * (nw / tot);
                    n = tot;
                }
            }
            const float var = M2 / (float)N + st.h.fbn_eps;                      // biased, eps inside (modules.py:287)
            sm[SO_HD + 4 + c] = mean;
            sm[SO_HD + 6 + c] = sqrtf(var);
            sm[SO_HD + 16 + c] = sm[SO_HD + 12 + c] - 0.5f * logf(var);           // this channel's log-det
            st.h.bmean[c] = mean; st.h.bvar[c] = var;
            st.h.rmean[c] = x0_old * (1.f - st.h.fbn_mom) + mean * st.h.fbn_mom;   // modules.py:291-294
            st.h.rvar[c] = frv_all * (1.f - st.h.fbn_mom) + var * st.h.fbn_mom;
            save[2 * NF_MC_NB * 32 + c] = mean; save[2 * NF_MC_NB * 32 + 4 + c] = var;
        }
        so_barrier();
        // ---- head: h = exp(log_gamma) (z - mean) / sqrt(var) + beta; x = the conditioning feature ----
        const int sel = st.h.odd ? 1 : 0;             // the transformed feature; the other one conditions (squeeze.py:68-69)
        const float dld = sm[SO_HD + 16] + sm[SO_HD + 17];
        float h[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float zn = (z[b][c] - sm[SO_HD + 4 + c]) / sm[SO_HD + 6 + c];
                h[b][c] = fmaf(sm[SO_HD + c], zn, 0.f) + sm[SO_HD + 2 + c];
            }
        // ---- the conditioner ----
        float a[5][2][16];
        {
            float w0[16], b0[16];
            so_ldvec(sm + SO_V0, hs, w0);
            so_ldvec(sm + SO_B, hs, b0);
            const float ws0 = sm[SO_WS];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float x = (sel ? h[b][0] : h[b][1]) * ws0;
#pragma unroll
                for (int r = 0; r < 16; ++r) a[0][b][r] = fmaf(w0[r], x, 0.f) + b0[r];
            }
        }
        so_batchnorm_train(sm, 0, a[0], cv, N, eps, mom, rm_old, rv_old, nbt_old, st, save, c32, hs, wid);
#pragma unroll
        for (int l = 1; l < 5; ++l) {
            float sc[16], sh[16], ws[16], bias[16], A[16], act[2][16];
            so_ldvec(sm + SO_BNC + (4 * (l - 1) + 0) * 32, hs, sc);
            so_ldvec(sm + SO_BNC + (4 * (l - 1) + 1) * 32, hs, sh);
            so_ldvec(sm + SO_WS + l * 32, hs, ws);
            so_ldvec(sm + SO_B + l * 32, hs, bias);
            so_ld_afwd(sm, l, c32, hs, A);
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) act[b][r] = fmaxf(fmaf(a[l - 1][b][r], sc[r], sh[r]), 0.f) * ws[r];
            f32x16 acc[2];
            so_gemm(A, act, acc);
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[b][r];
                    if ((l & 1) == 0) v += a[l - 2][b][r];
                    a[l][b][r] = v + bias[r];
                }
            so_batchnorm_train(sm, l, a[l], cv, N, eps, mom, rm_old, rv_old, nbt_old, st, save, c32, hs, wid);
        }
        // ---- linear 5 (32 -> t | s_raw) on the vector ALU, the affine coupling (coupling.py:104-113) ----
        {
            float sc[16], sh[16], ws[16], v50[16], v51[16];
            so_ldvec(sm + SO_BNC + (4 * 4 + 0) * 32, hs, sc);
            so_ldvec(sm + SO_BNC + (4 * 4 + 1) * 32, hs, sh);
            so_ldvec(sm + SO_WS + 5 * 32, hs, ws);
            so_ldvec(sm + SO_V5, hs, v50);
            so_ldvec(sm + SO_V5 + 32, hs, v51);
            const float b50 = sm[SO_B + 5 * 32], b51 = sm[SO_B + 5 * 32 + 1], ca = sm[SO_HD + 9], cc = sm[SO_HD + 10];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float tp = 0.f, sp = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float av = fmaxf(fmaf(a[4][b][r], sc[r], sh[r]), 0.f) * ws[r];
                    tp = fmaf(v50[r], av, tp);
                    sp = fmaf(v51[r], av, sp);
                }
                tp += __shfl_xor(tp, 32, NF_WAVE);
                sp += __shfl_xor(sp, 32, NF_WAVE);
                const float tt = tp + b50, sraw = sp + b51;
                const float sv = tanhf(sraw) * ca + cc;
                const float h0 = sel ? h[b][1] : h[b][0], h1 = sel ? h[b][0] : h[b][1];
                const float y0 = h0 * expf(sv) + tt;
                z[b][sel] = y0;
                z[b][1 - sel] = h1;
                ldv[b] += sv + dld;
                if (cv[b] && hs == 0) {
                    float* yr = ys + ((size_t)s * N + col[b]) * 2;
                    yr[0] = z[b][0]; yr[1] = z[b][1];
                }
            }
        }
        so_barrier();                               // the tables are restaged by the next step; every reader of record s is through
        if (rt) rec[s & 1][t] = nxt;                   // its slot takes step s + 2 (read behind the next step's first barrier)
    }
#pragma unroll
    for (int b = 0; b < 2; ++b)
        if (cv[b] && hs == 0) ld[col[b]] = ldv[b];
}

// ---------------------------------------------------------------------------------------------------------------------------------
// backward (training mode, deferred fold): forward recomputed from the step's input and the saved statistics, then the chain of
//   G_J -> weight / bias gradient partials of linear J (per wave, slab) -> data gradient -> ReLU mask -> two batch sums -> BatchNorm backward
// ---------------------------------------------------------------------------------------------------------------------------------
// transposes G and act (both [2][16] per lane) through the wave's tiles and leaves the wave's partial of g_Weff[l] ([i][o]) + bias sums
__device__ __forceinline__ void so_wgrad(float* sm, float* slab_l, const float (&act)[2][16], const float (&G)[2][16], int c32, int hs, int wid) {
    float* TA = sm + SO_TILES + (wid * 2 + 0) * 32 * NF_SO_TS;
    float* TG = sm + SO_TILES + (wid * 2 + 1) * 32 * NF_SO_TS;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            TA[so_fm(r, hs) * NF_SO_TS + 32 * b + c32] = act[b][r];
            TG[so_fm(r, hs) * NF_SO_TS + 32 * b + c32] = G[b][r];
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bsum = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const f32x4s av = *(const f32x4s*)(TA + c32 * NF_SO_TS + 32 * hs + 4 * k);
        const f32x4s gv = *(const f32x4s*)(TG + c32 * NF_SO_TS + 32 * hs + 4 * k);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bsum += gv[e];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], gv[e], acc, 0, 0, 0);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();                   // (the tiles are rewritten by the next layer)
#pragma unroll
    for (int r = 0; r < 16; ++r) slab_l[so_fm(r, hs) * 32 + c32] = acc[r];       // D[i = fm][o = c32]
    bsum += __shfl_xor(bsum, 32, NF_WAVE);
    if (hs == 0) slab_l[1024 + c32] = bsum;
}

__global__ void __launch_bounds__(NF_SO_THREADS) k_solo_bwd(const NfGlowFlowStep* __restrict__ steps, int S, const float* __restrict__ z0,
                                                            const float* __restrict__ ys, const float* __restrict__ g_y,
                                                            const float* __restrict__ g_ld, float* __restrict__ gzs,
                                                            const float* __restrict__ saves, int save_stride, int accumulate,
                                                            float* __restrict__ slabs, float* __restrict__ head_rec, int N, float wn_eps) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int t = threadIdx.x, lane = t & 63, wid = __builtin_amdgcn_readfirstlane(t >> 6), c32 = lane & 31, hs = lane >> 5;
    unsigned long long (*rec)[SO_REC_WORDS] = reinterpret_cast<unsigned long long (*)[SO_REC_WORDS]>(sm + SO_REC);
    const bool rt = t < SO_REC_WORDS;
    const int blocks = (N + NF_MLP_ROWS_PER_BLOCK - 1) / NF_MLP_ROWS_PER_BLOCK;      // regions per step of the slab / head-sum formats
    if (rt) {
        rec[(S - 1) & 1][t] = reinterpret_cast<const unsigned long long*>(steps + S - 1)[t];
        if (S > 1) rec[(S - 2) & 1][t] = reinterpret_cast<const unsigned long long*>(steps + S - 2)[t];
    }
    int col[2];
    bool cv[2];
    float gy[2][2], gld[2], zin[2][2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        col[b] = 64 * wid + 32 * b + c32;
        cv[b] = col[b] < N;
        const int cc = cv[b] ? col[b] : 0;
        gy[b][0] = cv[b] ? g_y[2 * cc] : 0.f;
        gy[b][1] = cv[b] ? g_y[2 * cc + 1] : 0.f;
        gld[b] = (cv[b] && g_ld != nullptr) ? g_ld[cc] : 0.f;
        const float* zr = S == 1 ? z0 : ys + (size_t)(S - 2) * N * 2;
        zin[b][0] = cv[b] ? zr[2 * cc] : 0.f;
        zin[b][1] = cv[b] ? zr[2 * cc + 1] : 0.f;
    }
    so_barrier();
    SoParams P;
    so_prefetch(P, *reinterpret_cast<const NfGlowFlowStep*>(rec[(S - 1) & 1]), t);
    float sv_mean = 0.f, sv_inv = 0.f, sv_h = 0.f;     // saved statistics of the step, prefetched with its parameters
    float gb_old = 0.f, gg_old = 0.f;                  // threads 32 J + f: the BatchNorm gradient sinks' old values (accumulate)
    {
        const float* save = saves + (size_t)(S - 1) * save_stride;
        const NfGlowFlowStep& st0 = *reinterpret_cast<const NfGlowFlowStep*>(rec[(S - 1) & 1]);
        if (t < 160 && accumulate) { gb_old = st0.g.beta[t >> 5][t & 31]; gg_old = st0.g.gamma[t >> 5][t & 31]; }
        if (t < 160) { sv_mean = save[(2 * (t >> 5)) * 32 + (t & 31)]; sv_inv = save[(2 * (t >> 5) + 1) * 32 + (t & 31)]; }
        else if (t < 164) sv_h = save[2 * NF_MC_NB * 32 + (t - 160 < 2 ? t - 160 : 4 + t - 162)];
    }
    const float invN = 1.f / (float)N;
#pragma unroll 1
    for (int s = S - 1; s >= 0; --s) {
        const NfGlowFlowStep& st = *reinterpret_cast<const NfGlowFlowStep*>(rec[s & 1]);
        so_stage(sm, P, t);
        if (t < 160) {                                 // BatchNorm constants from the saved statistics
            const int j = t >> 5, f = t & 31;
            const float sc = P.ga * sv_inv;
            sm[SO_BNC + (4 * j + 0) * 32 + f] = sc;
            sm[SO_BNC + (4 * j + 1) * 32 + f] = P.be - sv_mean * sc;
            sm[SO_BNC + (4 * j + 2) * 32 + f] = sv_mean;
            sm[SO_BNC + (4 * j + 3) * 32 + f] = sv_inv;
        } else if (t < 162) sm[SO_HD + 4 + (t - 160)] = sv_h;            // flow BatchNorm: mean
        else if (t < 164) sm[SO_HD + 6 + (t - 162)] = sqrtf(sv_h);       // sqrt(var)
        so_barrier();
        // next step's record, parameters, statistics and input rows: in flight under this step
        unsigned long long nxt = 0;
        if (rt && s >= 2) nxt = reinterpret_cast<const unsigned long long*>(steps + s - 2)[t];
        float zn_[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
        const float gb_cur = gb_old, gg_cur = gg_old;
        if (s >= 1) {
            const NfGlowFlowStep& stn = *reinterpret_cast<const NfGlowFlowStep*>(rec[(s - 1) & 1]);
            if (t < 160 && accumulate) { gb_old = stn.g.beta[t >> 5][t & 31]; gg_old = stn.g.gamma[t >> 5][t & 31]; }
            so_prefetch(P, *reinterpret_cast<const NfGlowFlowStep*>(rec[(s - 1) & 1]), t);
            const float* save = saves + (size_t)(s - 1) * save_stride;
            if (t < 160) { sv_mean = save[(2 * (t >> 5)) * 32 + (t & 31)]; sv_inv = save[(2 * (t >> 5) + 1) * 32 + (t & 31)]; }
            else if (t < 164) sv_h = save[2 * NF_MC_NB * 32 + (t - 160 < 2 ? t - 160 : 4 + t - 162)];
            const float* zr = s == 1 ? z0 : ys + (size_t)(s - 2) * N * 2;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int cc = cv[b] ? col[b] : 0;
                zn_[b][0] = cv[b] ? zr[2 * cc] : 0.f;
                zn_[b][1] = cv[b] ? zr[2 * cc + 1] : 0.f;
            }
        }
        so_weight_norm(sm, t, wn_eps);
        so_barrier();
        float* slab = slabs + ((size_t)s * blocks + (wid >> 1)) * NF_MC_SLAB + (wid & 1) * NF_MC_SLAB_Q;   // + l * NF_MC_SLAB_L
        const bool slab_ok = (wid >> 1) < blocks;      // (N <= 128: the waves 2, 3 have no column and no region)
        // ---- head and conditioner, recomputed ----
        const int sel = st.h.odd ? 1 : 0;
        float h[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float zn = (zin[b][c] - sm[SO_HD + 4 + c]) / sm[SO_HD + 6 + c];
                h[b][c] = fmaf(sm[SO_HD + c], zn, 0.f) + sm[SO_HD + 2 + c];
            }
        float a[5][2][16], xw[2];
        {
            float w0[16], b0[16];
            so_ldvec(sm + SO_V0, hs, w0);
            so_ldvec(sm + SO_B, hs, b0);
            const float ws0 = sm[SO_WS];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                xw[b] = sel ? h[b][0] : h[b][1];
                const float x = xw[b] * ws0;
#pragma unroll
                for (int r = 0; r < 16; ++r) a[0][b][r] = fmaf(w0[r], x, 0.f) + b0[r];
            }
        }
#pragma unroll
        for (int l = 1; l < 5; ++l) {
            float sc[16], sh[16], ws[16], bias[16], A[16], act[2][16];
            so_ldvec(sm + SO_BNC + (4 * (l - 1) + 0) * 32, hs, sc);
            so_ldvec(sm + SO_BNC + (4 * (l - 1) + 1) * 32, hs, sh);
            so_ldvec(sm + SO_WS + l * 32, hs, ws);
            so_ldvec(sm + SO_B + l * 32, hs, bias);
            so_ld_afwd(sm, l, c32, hs, A);
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) act[b][r] = fmaxf(fmaf(a[l - 1][b][r], sc[r], sh[r]), 0.f) * ws[r];
            f32x16 acc[2];
            so_gemm(A, act, acc);
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[b][r];
                    if ((l & 1) == 0) v += a[l - 2][b][r];
                    a[l][b][r] = v + bias[r];
                }
        }
        // ---- linear 5 and the coupling's backward (coupling.py:104-113) -> G of the conditioner output, gradient of h ----
        float tg[2][16];                               // gradient of what the current linear multiplied with (ReLU output), per layer
        float Gh[2][2], hsum[2] = {0.f, 0.f};          // head sums: sum g_s, sum g_s tanh(s_raw)
        {
            float sc[16], sh[16], ws[16], v50[16], v51[16];
            so_ldvec(sm + SO_BNC + (4 * 4 + 0) * 32, hs, sc);
            so_ldvec(sm + SO_BNC + (4 * 4 + 1) * 32, hs, sh);
            so_ldvec(sm + SO_WS + 5 * 32, hs, ws);
            so_ldvec(sm + SO_V5, hs, v50);
            so_ldvec(sm + SO_V5 + 32, hs, v51);
            const float b50 = sm[SO_B + 5 * 32], b51 = sm[SO_B + 5 * 32 + 1], ca = sm[SO_HD + 9], cc = sm[SO_HD + 10];
            float p5[2][16], bs5[2] = {0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 16; ++r) { p5[0][r] = 0.f; p5[1][r] = 0.f; }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float pre[16], tp = 0.f, sp = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    pre[r] = fmaxf(fmaf(a[4][b][r], sc[r], sh[r]), 0.f);
                    const float av = pre[r] * ws[r];
                    tp = fmaf(v50[r], av, tp);
                    sp = fmaf(v51[r], av, sp);
                }
                tp += __shfl_xor(tp, 32, NF_WAVE);
                sp += __shfl_xor(sp, 32, NF_WAVE);
                const float sraw = sp + b51;
                (void)tp; (void)b50;
                const float th = tanhf(sraw);
                const float ev = expf(th * ca + cc);
                const float h0 = sel ? h[b][1] : h[b][0];
                const float gy0 = cv[b] ? (sel ? gy[b][1] : gy[b][0]) : 0.f, gy1 = cv[b] ? (sel ? gy[b][0] : gy[b][1]) : 0.f;
                const float gsv = cv[b] ? gy0 * h0 * ev + gld[b] : 0.f;               // ld += s: the log-det gradient enters here
                const float gsraw = gsv * ca * (1.f - th * th);
                hsum[0] += gsv;
                hsum[1] += gsv * th;
                Gh[b][sel] = gy0 * ev;
                Gh[b][1 - sel] = gy1;
                bs5[0] += gy0; bs5[1] += gsraw;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    tg[b][r] = (v50[r] * gy0 + v51[r] * gsraw) * ws[r];
                    p5[0][r] = fmaf(gy0, pre[r], p5[0][r]);
                    p5[1][r] = fmaf(gsraw, pre[r], p5[1][r]);
                }
            }
            // linear 5's partial: rows i of [i][o], o = 0 (t), 1 (s_raw); the other 30 columns are zero
            const float w50 = nf_cv_butterfly16(p5[0], c32), w51 = nf_cv_butterfly16(p5[1], c32);
            const float bt0 = so_half_sum(bs5[0]), bt1 = so_half_sum(bs5[1]);
            if (slab_ok) {
                float* sl = slab + 5 * NF_MC_SLAB_L;
                if ((c32 & 1) == 0) {
                    float* row = sl + so_fm(c32 >> 1, hs) * 32;
                    *(f32x4s*)(row) = f32x4s{w50, w51, 0.f, 0.f};
#pragma unroll
                    for (int q = 1; q < 8; ++q) *(f32x4s*)(row + 4 * q) = f32x4s{0.f, 0.f, 0.f, 0.f};
                }
                if (lane < 32) sl[1024 + lane] = lane == 0 ? bt0 : (lane == 1 ? bt1 : 0.f);
            }
        }
        // ---- BatchNorm J backward, then linear J: J = 4 .. 0.  ONE gradient array: tg holds, in turn, the gradient of the ReLU output,
        // its masked form gn, and G_J (the gradient of a_J) -- the five activations already take 160 of a lane's registers ----
        float Gs[2][16];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) Gs[b][r] = 0.f;
        float gx[2] = {0.f, 0.f};
#pragma unroll
        for (int J = 4; J >= 0; --J) {
            {
                float sc[16], sh[16], mean[16], invstd[16], p[2][16];
                so_ldvec(sm + SO_BNC + (4 * J + 0) * 32, hs, sc);
                so_ldvec(sm + SO_BNC + (4 * J + 1) * 32, hs, sh);
                so_ldvec(sm + SO_BNC + (4 * J + 2) * 32, hs, mean);
                so_ldvec(sm + SO_BNC + (4 * J + 3) * 32, hs, invstd);
#pragma unroll
                for (int r = 0; r < 16; ++r) { p[0][r] = 0.f; p[1][r] = 0.f; }
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float gn = (cv[b] && fmaf(a[J][b][r], sc[r], sh[r]) > 0.f) ? tg[b][r] : 0.f;
                        tg[b][r] = gn;
                        p[0][r] += gn;
                        p[1][r] = fmaf(gn, (a[J][b][r] - mean[r]) * invstd[r], p[1][r]);
                    }
                // the meeting: sum gn, sum gn xhat (+ the head sums of the step on its first round)
                if (J == 4) {
                    const float h0 = so_half_sum(hsum[0]), h1 = so_half_sum(hsum[1]);
                    if (lane == 0) { sm[SO_RED + (wid * 3 + 2) * 32 + 0] = h0; sm[SO_RED + (wid * 3 + 2) * 32 + 1] = h1; }
                }
                so_colsums<2>(sm, p, c32, hs, wid);
            }
            if ((t >> 5) == J) {                       // BatchNorm parameter gradients (g_beta = sum g, g_gamma = sum g xhat): the half wave
                const int f = t & 31;                  // that holds the sinks' old values stores (a load here would be a global round trip)
                st.g.beta[J][f] = gb_cur + sm[SO_TOT + f];
                st.g.gamma[J][f] = gg_cur + sm[SO_TOT + 32 + f];
            } else if (J == 4 && t >= 64 && t < 128) {  // the step's head sums in the format of k_glow_fold_all: region 0 carries them
                const int q = t - 64;
                float v = 0.f;
                if (q == 24 || q == 26) {
                    const int e = q == 24 ? 0 : 1;
                    v = (sm[SO_RED + (0 * 3 + 2) * 32 + e] + sm[SO_RED + (1 * 3 + 2) * 32 + e]) +
                        (sm[SO_RED + (2 * 3 + 2) * 32 + e] + sm[SO_RED + (3 * 3 + 2) * 32 + e]);
                }
                head_rec[((size_t)s * blocks) * 64 + q] = v;
                if (blocks > 1) head_rec[((size_t)s * blocks + 1) * 64 + q] = 0.f;
            }
            {
                float sc[16], mean[16], invstd[16], mg[16], mgx[16];
                so_ldvec(sm + SO_BNC + (4 * J + 0) * 32, hs, sc);
                so_ldvec(sm + SO_BNC + (4 * J + 2) * 32, hs, mean);
                so_ldvec(sm + SO_BNC + (4 * J + 3) * 32, hs, invstd);
                so_ldvec(sm + SO_TOT, hs, mg);
                so_ldvec(sm + SO_TOT + 32, hs, mgx);
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float xh = (a[J][b][r] - mean[r]) * invstd[r];
                        float v = sc[r] * (tg[b][r] - mg[r] * invN - xh * (mgx[r] * invN));
                        if (J == 0 || J == 2) v += Gs[b][r];  // a_J also feeds the residual connection two linears on
                        v = cv[b] ? v : 0.f;
                        tg[b][r] = v;                  // = G_J
                        if (J == 2 || J == 4) Gs[b][r] = v;
                    }
            }
            if (J >= 1) {
                // linear J: what it multiplied with (before the weight-norm scale), its weight-gradient partial, its data gradient
                if (slab_ok) {
                    float scp[16], shp[16], act[2][16];
                    so_ldvec(sm + SO_BNC + (4 * (J - 1) + 0) * 32, hs, scp);
                    so_ldvec(sm + SO_BNC + (4 * (J - 1) + 1) * 32, hs, shp);
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) act[b][r] = fmaxf(fmaf(a[J - 1][b][r], scp[r], shp[r]), 0.f);
                    so_wgrad(sm, slab + J * NF_MC_SLAB_L, act, tg, c32, hs, wid);
                }
                float ws[16], A[16];
                so_ld_abwd(sm, J, c32, hs, A);
                so_ldvec(sm + SO_WS + J * 32, hs, ws);
                f32x16 acc[2];
                so_gemm(A, tg, acc);
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) tg[b][r] = acc[b][r] * ws[r];
            } else {
                // linear 0 (1 -> 32): column i = 0 of [i][o] and the bias sums by butterfly, g_x on the vector ALU
                float w0[16], q0[2][16];
                so_ldvec(sm + SO_V0, hs, w0);
                const float ws0 = sm[SO_WS];
#pragma unroll
                for (int r = 0; r < 16; ++r) { q0[0][r] = 0.f; q0[1][r] = 0.f; }
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    float gp = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        gp = fmaf(tg[b][r], w0[r], gp);
                        q0[0][r] = fmaf(tg[b][r], xw[b], q0[0][r]);
                        q0[1][r] += tg[b][r];
                    }
                    gp += __shfl_xor(gp, 32, NF_WAVE);
                    gx[b] = gp * ws0;
                }
                const float gw = nf_cv_butterfly16(q0[0], c32), gb0 = nf_cv_butterfly16(q0[1], c32);
                if (slab_ok && (c32 & 1) == 0) {
                    float* sl = slab + 0 * NF_MC_SLAB_L;
                    sl[so_fm(c32 >> 1, hs)] = gw;                    // [i = 0][o]
                    sl[1024 + so_fm(c32 >> 1, hs)] = gb0;
                }
            }
        }
        // ---- head backward: the statistics are buffers (modules.py:285-296): g_z = exp(log_gamma) g_h / sqrt(var) ----
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            Gh[b][1 - sel] += gx[b];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float gzn = fmaf(sm[SO_HD + c], Gh[b][c], 0.f);
                gy[b][c] = cv[b] ? gzn / sm[SO_HD + 6 + c] : 0.f;
            }
            if (cv[b] && hs == 0) {
                float* gr = gzs + ((size_t)s * N + col[b]) * 2;
                gr[0] = gy[b][0]; gr[1] = gy[b][1];
            }
            zin[b][0] = zn_[b][0]; zin[b][1] = zn_[b][1];
        }
        so_barrier();
        if (rt) rec[s & 1][t] = nxt;                   // (slot of step s - 2: read behind the next step's first barrier)
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
static int nf_so_on = -1;                             // the environment (NF_FLOW_SOLO=0: off) at first use, nf_flow_solo_config afterwards
static int nf_so_enabled() {
    if (nf_so_on < 0) { const char* e = getenv("NF_FLOW_SOLO"); nf_so_on = (e == nullptr || e[0] != '0') ? 1 : 0; }
    return nf_so_on;
}
extern "C" int nf_flow_solo_config(int on) {
    nf_so_enabled();
    if (on >= 0) nf_so_on = on ? 1 : 0;
    return 0;
}
int nf_solo_plan(int64_t N, int D) { return (nf_so_enabled() && D == 2 && N >= 1 && N <= 256) ? 1 : 0; }
int nf_solo_fwd(const void* steps_dev, int S, const float* z0, float* ys, float* ld, float* saves, int save_stride, int64_t N, float bn_eps,
                float bn_momentum, float wn_eps, hipStream_t stream) {
    const size_t lds = nf_so_lds_bytes(false);
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)k_solo_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    hipLaunchKernelGGL(k_solo_fwd, dim3(1), dim3(NF_SO_THREADS), lds, stream, (const NfGlowFlowStep*)steps_dev, S, z0, ys, ld, saves, save_stride,
                       (int)N, bn_eps, bn_momentum, wn_eps);
    NF_CHECK_LAUNCH();
    return 0;
}
int nf_solo_bwd(const void* steps_dev, int S, const float* z0, const float* ys, const float* g_y, const float* g_ld, float* gzs,
                const float* saves, int save_stride, int accumulate, float* slabs_all, float* head_rec, int64_t N, float wn_eps,
                hipStream_t stream) {
    const size_t lds = nf_so_lds_bytes(true);
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)k_solo_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    hipLaunchKernelGGL(k_solo_bwd, dim3(1), dim3(NF_SO_THREADS), lds, stream, (const NfGlowFlowStep*)steps_dev, S, z0, ys, g_y, g_ld, gzs, saves,
                       save_stride, accumulate, slabs_all, head_rec, (int)N, wn_eps);
    NF_CHECK_LAUNCH();
    return 0;
}
