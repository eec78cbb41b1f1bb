// Invertible residual block (Residual Flow) forward / inverse + log-det on the GPU for D <= 4 features:
//   g(x) = W3 lipswish(W2 lipswish(W1 x + b1) + b2) + b3      (flows/iresblock.py:258-278, flows/modules.py:215-222)
// One thread per sample, the three (spectrally normalised) weight matrices broadcast from LDS.  The per-sample Jacobian
// J = W3 D2 W2 D1 W1 (D x D) is formed in forward mode, so every log-det estimator of the reference reduces to a few
// D x D operations per sample instead of nested autograd VJP sweeps:
//   exact  : log |det(I + J)|                                                    (iresblock.py:17-32)
//   series : mean_s sum_k coef[s][k] * v_s^T (J^T)^k v_s   with host-drawn noise v and Russian-roulette lengths
//            (fixed: coef = (-1)^(k+1)/k, iresblock.py:35-56; unbias: / P(N >= k), iresblock.py:59-81)
// Spectral normalisation (flows/spectral_norm.py:26-43) is a separate one-block-per-matrix launch, and the fixed-point
// inverse (iresblock.py:236-255) is a chain of single-iteration launches gated by a device flag that reproduces the
// reference's batch-global exit without a host round trip.
#include "nf_common.h"
#include "nf_det.h"

NF_DET_STATE(nf_rsm)
NF_DET_HOST_API(nf_rsm)

#define NF_RES_H 32
#define NF_RES_MAXD 4
#define NF_RES_MAXS 4
#define NF_RES_MAXK 64

__device__ __forceinline__ float nf_lipswish(float x, float beta, float& dx) {
    const float s = 1.f / (1.f + expf(-beta * x));
    dx = (s + beta * x * s * (1.f - s)) * (1.f / 1.1f);     // d/dx [ x sigmoid(beta x) / 1.1 ]
    return x * s * (1.f / 1.1f);
}

struct NfResW {
    const float *W1, *b1, *W2, *b2, *W3, *b3, *beta1, *beta2;
};

// weights -> LDS: W1 (H x D), W2 (H x H, row stride H+1), W3 (D x H), biases, betas
template <int D>
__device__ __forceinline__ void nf_res_stage(const NfResW& w, float* sm) {
    float* W1 = sm;
    float* b1 = W1 + NF_RES_H * D;
    float* W2 = b1 + NF_RES_H;
    float* b2 = W2 + NF_RES_H * (NF_RES_H + 1);
    float* W3 = b2 + NF_RES_H;
    float* b3 = W3 + D * NF_RES_H;
    for (int i = threadIdx.x; i < NF_RES_H * D; i += blockDim.x) W1[i] = w.W1[i];
    for (int i = threadIdx.x; i < NF_RES_H * NF_RES_H; i += blockDim.x) W2[(i / NF_RES_H) * (NF_RES_H + 1) + (i % NF_RES_H)] = w.W2[i];
    for (int i = threadIdx.x; i < D * NF_RES_H; i += blockDim.x) W3[i] = w.W3[i];
    for (int i = threadIdx.x; i < NF_RES_H; i += blockDim.x) { b1[i] = w.b1[i]; b2[i] = w.b2[i]; }
    for (int i = threadIdx.x; i < D; i += blockDim.x) b3[i] = w.b3[i];
    __syncthreads();
}
#define NF_RES_LDS(D) ((NF_RES_H * (D) + NF_RES_H + NF_RES_H * (NF_RES_H + 1) + NF_RES_H + (D) * NF_RES_H + (D)) * sizeof(float))

// g(x) and, if JAC, the Jacobian columns J[:, d] = W3 (D2 (W2 (D1 W1[:, d])))
template <int D, bool JAC>
__device__ __forceinline__ void nf_res_eval(const float* sm, float beta1, float beta2, const float (&x)[D], float (&g)[D],
                                            float (&J)[D][D]) {
    const float* W1 = sm;
    const float* b1 = W1 + NF_RES_H * D;
    const float* W2 = b1 + NF_RES_H;
    const float* b2 = W2 + NF_RES_H * (NF_RES_H + 1);
    const float* W3 = b2 + NF_RES_H;
    const float* b3 = W3 + D * NF_RES_H;
    float a1[NF_RES_H], t1[JAC ? D : 1][NF_RES_H];
#pragma unroll
    for (int o = 0; o < NF_RES_H; ++o) {
        float h = b1[o];
#pragma unroll
        for (int d = 0; d < D; ++d) h = fmaf(W1[o * D + d], x[d], h);
        float dh;
        a1[o] = nf_lipswish(h, beta1, dh);
        if (JAC) {
#pragma unroll
            for (int d = 0; d < D; ++d) t1[d][o] = dh * W1[o * D + d];
        }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
        g[d] = b3[d];
        if (JAC) {
#pragma unroll
            for (int e = 0; e < D; ++e) J[d][e] = 0.f;
        }
    }
    for (int o = 0; o < NF_RES_H; ++o) {
        float h = b2[o];
        float jt[JAC ? D : 1];
        if (JAC) {
#pragma unroll
            for (int d = 0; d < D; ++d) jt[d] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < NF_RES_H; ++i) {
            const float w = W2[o * (NF_RES_H + 1) + i];
            h = fmaf(w, a1[i], h);
            if (JAC) {
#pragma unroll
                for (int d = 0; d < D; ++d) jt[d] = fmaf(w, t1[d][i], jt[d]);
            }
        }
        float dh;
        const float a2 = nf_lipswish(h, beta2, dh);
#pragma unroll
        for (int r = 0; r < D; ++r) {
            const float w3 = W3[r * NF_RES_H + o];
            g[r] = fmaf(w3, a2, g[r]);
            if (JAC) {
#pragma unroll
                for (int d = 0; d < D; ++d) J[r][d] = fmaf(w3, dh * jt[d], J[r][d]);
            }
        }
    }
}

template <int D>
__device__ __forceinline__ float nf_det_I_plus(const float (&J)[D][D]) {
    float A[D][D];
#pragma unroll
    for (int r = 0; r < D; ++r)
#pragma unroll
        for (int c = 0; c < D; ++c) A[r][c] = J[r][c] + (r == c ? 1.f : 0.f);
    float det = 1.f;                          // Gaussian elimination without pivoting: I + J is near identity (Lip(g) < 1)
#pragma unroll
    for (int k = 0; k < D; ++k) {
        det *= A[k][k];
#pragma unroll
        for (int r = k + 1; r < D; ++r) {
            const float f = A[r][k] / A[k][k];
#pragma unroll
            for (int c = k + 1; c < D; ++c) A[r][c] -= f * A[k][c];
        }
    }
    return det;
}

// mode 0: y = x + g only.  mode 1: + exact log-det.  mode 2: + series estimator with noise v (B, S, D)
template <int D>
__global__ void __launch_bounds__(NF_BLOCK) k_resmlp_fwd(NfResW w, const float* __restrict__ x, float* __restrict__ y,
                                                         float* __restrict__ ld, float ld_sign, int mode,
                                                         const float* __restrict__ v, const float* __restrict__ coef,
                                                         const int* __restrict__ n_terms, int S, int64_t B) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    nf_res_stage<D>(w, sm);
    const float beta1 = w.beta1[0], beta2 = w.beta2[0];
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += (int64_t)gridDim.x * blockDim.x) {
        float xv[D], g[D], J[D][D];
#pragma unroll
        for (int d = 0; d < D; ++d) xv[d] = x[b * D + d];
        if (mode == 0) nf_res_eval<D, false>(sm, beta1, beta2, xv, g, J);
        else nf_res_eval<D, true>(sm, beta1, beta2, xv, g, J);
        if (y != nullptr) {
#pragma unroll
            for (int d = 0; d < D; ++d) y[b * D + d] = xv[d] + g[d];
        }
        if (mode == 1) {
            ld[b] += ld_sign * logf(fabsf(nf_det_I_plus<D>(J)));
        } else if (mode == 2) {
            float total = 0.f;
            for (int s = 0; s < S; ++s) {
                float vv[D], wv[D];
#pragma unroll
                for (int d = 0; d < D; ++d) { vv[d] = v[(b * S + s) * D + d]; wv[d] = vv[d]; }
                const int n = n_terms[s];
                for (int k = 1; k <= n; ++k) {           // w <- J^T w  (one vector-Jacobian product), tr = w . v
                    float nw[D];
#pragma unroll
                    for (int c = 0; c < D; ++c) {
                        float a = 0.f;
#pragma unroll
                        for (int r = 0; r < D; ++r) a = fmaf(J[r][c], wv[r], a);
                        nw[c] = a;
                    }
                    float tr = 0.f;
#pragma unroll
                    for (int d = 0; d < D; ++d) { wv[d] = nw[d]; tr = fmaf(nw[d], vv[d], tr); }
                    total = fmaf(coef[s * NF_RES_MAXK + (k - 1)], tr, total);
                }
            }
            ld[b] += ld_sign * total / (float)S;
        }
    }
}

// one fixed-point iteration x <- z - g(x); runs only while the previous iteration left some |dx| >= ftol
template <int D>
__global__ void __launch_bounds__(NF_BLOCK) k_resmlp_fixed_point(NfResW w, const float* __restrict__ z, float* __restrict__ x,
                                                                 int* __restrict__ flags, int it, float ftol, int64_t B) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    if (it > 0 && flags[it - 1] == 0) return;                // converged (or skipped) before: the reference left its loop
    nf_res_stage<D>(w, sm);
    const float beta1 = w.beta1[0], beta2 = w.beta2[0];
    bool moving = false;
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += (int64_t)gridDim.x * blockDim.x) {
        float xv[D], g[D], J[D][D];
#pragma unroll
        for (int d = 0; d < D; ++d) xv[d] = x[b * D + d];
        nf_res_eval<D, false>(sm, beta1, beta2, xv, g, J);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float nx = z[b * D + d] - g[d];
            moving |= !(fabsf(nx - xv[d]) < ftol);             // iresblock.py:248
            x[b * D + d] = nx;
        }
    }
    if (__any(moving) && (threadIdx.x & (NF_WAVE - 1)) == 0) atomicOr(flags + it, 1);
}

// spectral normalisation of up to 3 matrices (one block each): one power iteration, u / v updated in place,
// W_eff = W_bar * min(coeff / (sigma + eps), 1).  Skipped like the fixed-point step when `flags` says "converged".
struct NfSnArgs {
    const float* Wbar[3];
    float* u[3];
    float* v[3];
    float* Weff[3];
    int h[3];
    int w[3];
};
__global__ void __launch_bounds__(NF_BLOCK) k_spectral_weights(NfSnArgs a, float coeff, float eps, const int* __restrict__ flags,
                                                               int it) {
    __shared__ float su[64], sv[64], scratch[NF_BLOCK / NF_WAVE];
    __shared__ float s_scale;
    if (flags != nullptr && it > 0 && flags[it - 1] == 0) return;
    const int m = blockIdx.x, H = a.h[m], Wd = a.w[m];
    const float* W = a.Wbar[m];
    if ((int)threadIdx.x < H) su[threadIdx.x] = a.u[m][threadIdx.x];
    __syncthreads();
    float t = 0.f;                                            // v = l2normalize(W^T u)
    if ((int)threadIdx.x < Wd)
        for (int r = 0; r < H; ++r) t = fmaf(W[r * Wd + threadIdx.x], su[r], t);
    float n2 = nf_block_sum((int)threadIdx.x < Wd ? t * t : 0.f, scratch);
    if (threadIdx.x == 0) s_scale = sqrtf(n2) + 1e-12f;
    __syncthreads();
    if ((int)threadIdx.x < Wd) { sv[threadIdx.x] = t / s_scale; a.v[m][threadIdx.x] = sv[threadIdx.x]; }
    __syncthreads();
    float q = 0.f;                                            // u = l2normalize(W v)
    if ((int)threadIdx.x < H)
        for (int c = 0; c < Wd; ++c) q = fmaf(W[threadIdx.x * Wd + c], sv[c], q);
    n2 = nf_block_sum((int)threadIdx.x < H ? q * q : 0.f, scratch);
    if (threadIdx.x == 0) s_scale = sqrtf(n2) + 1e-12f;
    __syncthreads();
    const float un = (int)threadIdx.x < H ? q / s_scale : 0.f;
    if ((int)threadIdx.x < H) a.u[m][threadIdx.x] = un;
    const float sigma = nf_block_sum(un * q, scratch);        // u . (W v)
    if (threadIdx.x == 0) s_scale = fminf(coeff / (sigma + eps), 1.f);
    __syncthreads();
    for (int i = threadIdx.x; i < H * Wd; i += blockDim.x) a.Weff[m][i] = W[i] * s_scale;
}

// ---------------------------------------------------------------------------------------------------------------
#define NF_RES_DISPATCH(D, CALL) \
    switch (D) { case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break; case 4: CALL(4); break; default: return NF_E_UNSUPPORTED; }

extern "C" int nf_resmlp_fwd(const float* x, const float* W1, const float* b1, const float* W2, const float* b2,
                             const float* W3, const float* b3, const float* beta1, const float* beta2, float* y, float* ld,
                             float ld_sign, int mode, const float* noise, const float* coef, const int* n_terms, int S,
                             int64_t B, int D, nf_stream_t stream) {
    if (D < 1 || mode < 0 || mode > 2 || (mode > 0 && ld == nullptr)) return NF_E_BADARG;
    if (mode == 2 && (S < 1 || S > NF_RES_MAXS || noise == nullptr || coef == nullptr || n_terms == nullptr)) return NF_E_BADARG;
    if (B == 0) return 0;
    NfResW w{W1, b1, W2, b2, W3, b3, beta1, beta2};
    unsigned g = nf_grid_for(B, 64);
    if (g > 2048) g = 2048;
#define CALL(DT) hipLaunchKernelGGL(k_resmlp_fwd<DT>, dim3(g), dim3(64), NF_RES_LDS(DT), (hipStream_t)stream, w, x, y, ld, ld_sign, mode, noise, coef, n_terms, S, B)
    NF_RES_DISPATCH(D, CALL)
#undef CALL
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_resmlp_fixed_point_step(const float* z, float* x, const float* W1, const float* b1, const float* W2,
                                          const float* b2, const float* W3, const float* b3, const float* beta1,
                                          const float* beta2, int* flags, int iteration, float ftol, int64_t B, int D,
                                          nf_stream_t stream) {
    if (D < 1 || iteration < 0 || flags == nullptr) return NF_E_BADARG;
    if (B == 0) return 0;
    NfResW w{W1, b1, W2, b2, W3, b3, beta1, beta2};
    unsigned g = nf_grid_for(B, 64);
    if (g > 2048) g = 2048;
#define CALL(DT) hipLaunchKernelGGL(k_resmlp_fixed_point<DT>, dim3(g), dim3(64), NF_RES_LDS(DT), (hipStream_t)stream, w, z, x, flags, iteration, ftol, B)
    NF_RES_DISPATCH(D, CALL)
#undef CALL
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_spectral_weights(const float* const* W_bar, float* const* u, float* const* v, float* const* W_eff,
                                   const int* rows, const int* cols, int n_mats, float coeff, float eps, const int* flags,
                                   int iteration, nf_stream_t stream) {
    if (n_mats < 1 || n_mats > 3) return NF_E_BADARG;
    NfSnArgs a;
    for (int i = 0; i < n_mats; ++i) {
        if (rows[i] < 1 || cols[i] < 1 || rows[i] > 64 || cols[i] > 64) return NF_E_BADARG;
        a.Wbar[i] = W_bar[i]; a.u[i] = u[i]; a.v[i] = v[i]; a.Weff[i] = W_eff[i]; a.h[i] = rows[i]; a.w[i] = cols[i];
    }
    hipLaunchKernelGGL(k_spectral_weights, dim3((unsigned)n_mats), dim3(NF_BLOCK), 0, (hipStream_t)stream, a, coeff, eps, flags,
                       iteration);
    NF_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// TRAINING backward of the block (iresblock.py:84-109 Neumann gradient estimator, :112-185 memory-saving autograd Function).
// The reference differentiates, per block and step,
//     L  =  < dL/dg , g(x) >  +  c * S(x, theta),      S = s^T J(x, theta) v,   s = v + sum_k coef_k (J^T)^k v  (held constant),
// where c is the upstream gradient of the log-det taken from the FIRST sample (iresblock.py:166) and v the Hutchinson noise:
// S needs second derivatives of g, which nested autograd sweeps provide there.  For the 2 -> 32 -> 32 -> 2 LipSwish network they
// are closed form.  With h1 = W1 x + b1, a1 = phi(h1), h2 = W2 a1 + b2, a = W1 v, p1 = phi'(h1) a, q = W2 p1, p2 = phi'(h2) q:
//     S = (W3^T s) . p2,   and with r3 = W3^T s, g3 = W3^T dL/dg the combined signals are
//     GQ  = c r3 phi'(h2)                                  (gradient of q)
//     GH2 = g3 phi'(h2) + c r3 phi''(h2) q                 (gradient of h2)
//     GP1 = W2^T GQ,  GA1 = W2^T GH2,  GH1 = GA1 phi'(h1) + GP1 phi''(h1) a   (gradient of h1)
//     dW3 = dL/dg a2^T + c s p2^T,  dW2 = GQ p1^T + GH2 a1^T,  db2 = GH2,  dW1 = GH1 x^T + (GP1 phi'(h1)) v^T,  db1 = GH1,  db3 = dL/dg,
//     dx = W1^T GH1  (the residual connection's identity term is added by the caller), and the two LipSwish slopes beta get
//     dbeta2 = sum g3 dphi/dbeta(h2) + c r3 q dphi'/dbeta(h2),   dbeta1 = sum GA1 dphi/dbeta(h1) + GP1 a dphi'/dbeta(h1).
// Work decomposition: lane = (sample slot, hidden unit): a wave works on TWO samples at a time, every per-unit quantity is one
// register, matrix-vector products read the other units' values as LDS broadcasts; a lane accumulates ITS row of dW2 (32
// registers) and its entries of the other gradients over all the samples of its wave, and the block adds its totals to the
// (zeroed) output with one atomic per entry.
struct NfLipD { float f, d1, d2, db, d1b; };              // phi, phi', phi'', dphi/dbeta, dphi'/dbeta
__device__ __forceinline__ NfLipD nf_lipswish_all(float h, float beta) {
    const float u = beta * h;
    const float s = 1.f / (1.f + expf(-u));
    const float sp = s * (1.f - s), spp = sp * (1.f - 2.f * s);
    const float k = 1.f / 1.1f;
    NfLipD r;
    r.f = h * s * k;
    r.d1 = (s + u * sp) * k;
    r.d2 = (2.f * beta * sp + beta * u * spp) * k;
    r.db = h * h * sp * k;
    r.d1b = h * (2.f * sp + u * spp) * k;
    return r;
}
__device__ __forceinline__ float nf_half_allsum(float v) {     // sum over the 32 lanes of a wave half, result in every lane
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, NF_WAVE);
    return v;
}

#define NF_RT_WAVES 4
#define NF_RT_THREADS (NF_RT_WAVES * NF_WAVE)
// g_out layout: W1 (32 x D) | b1 (32) | W2 (32 x 32) | b2 (32) | W3 (D x 32) | b3 (D) | beta1 | beta2
template <int D>
__global__ void __launch_bounds__(NF_RT_THREADS) k_resmlp_train_bwd(NfResW w, const float* __restrict__ x, const float* __restrict__ vn,
                                                                    const float* __restrict__ coef, int n_terms,
                                                                    const float* __restrict__ d_g, const float* __restrict__ d_ld,
                                                                    float* __restrict__ d_x, float* __restrict__ g_out, int64_t B) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    nf_res_stage<D>(w, sm);
    const float* W1 = sm;
    const float* b1 = W1 + NF_RES_H * D;
    const float* W2 = b1 + NF_RES_H;
    const float* b2 = W2 + NF_RES_H * (NF_RES_H + 1);
    const float* W3 = b2 + NF_RES_H;
    float* xb = sm + NF_RES_LDS(D) / sizeof(float);          // per (wave, slot): a1[32] | p1[32] | t1[D][32] | GQ[32] | GH2[32]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, slot = lane >> 5, u = lane & 31;
    constexpr int PER = (4 + D) * NF_RES_H;
    float* my = xb + (wid * 2 + slot) * PER;
    float* A1 = my;
    float* P1 = my + NF_RES_H;
    float* T1 = my + 2 * NF_RES_H;
    float* GQl = my + (2 + D) * NF_RES_H;
    float* GHl = my + (3 + D) * NF_RES_H;
    const float beta1 = w.beta1[0], beta2 = w.beta2[0], cs = d_ld[0];

    float accW2[NF_RES_H], accW1[D], accW3[D], acc_b1 = 0.f, acc_b2 = 0.f, acc_be1 = 0.f, acc_be2 = 0.f, acc_b3 = 0.f;
#pragma unroll
    for (int i = 0; i < NF_RES_H; ++i) accW2[i] = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) { accW1[d] = 0.f; accW3[d] = 0.f; }

    const int64_t pairs = (B + 1) / 2;
    for (int64_t pr = (int64_t)blockIdx.x * NF_RT_WAVES + wid; pr < pairs; pr += (int64_t)gridDim.x * NF_RT_WAVES) {
        const int64_t b = 2 * pr + slot;
        const bool ok = b < B;
        float xv[D], vv[D], dg[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            xv[d] = ok ? x[b * D + d] : 0.f;
            vv[d] = ok ? vn[b * D + d] : 0.f;
            dg[d] = ok ? d_g[b * D + d] : 0.f;
        }
        // ---- layer 1 (lane = unit u) --------------------------------------------------------------------------------------
        float h1 = b1[u], aV = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) { h1 = fmaf(W1[u * D + d], xv[d], h1); aV = fmaf(W1[u * D + d], vv[d], aV); }
        const NfLipD l1 = nf_lipswish_all(h1, beta1);
        const float p1 = l1.d1 * aV;
        A1[u] = l1.f;
        P1[u] = p1;
#pragma unroll
        for (int d = 0; d < D; ++d) T1[d * NF_RES_H + u] = l1.d1 * W1[u * D + d];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- layer 2 (lane = unit u): h2, q, Jacobian row pieces -----------------------------------------------------------------
        float h2 = b2[u], q = 0.f, jt[D];
#pragma unroll
        for (int d = 0; d < D; ++d) jt[d] = 0.f;
#pragma unroll 8
        for (int i = 0; i < NF_RES_H; ++i) {
            const float wv = W2[u * (NF_RES_H + 1) + i];
            h2 = fmaf(wv, A1[i], h2);
            q = fmaf(wv, P1[i], q);
#pragma unroll
            for (int d = 0; d < D; ++d) jt[d] = fmaf(wv, T1[d * NF_RES_H + i], jt[d]);
        }
        const NfLipD l2 = nf_lipswish_all(h2, beta2);
        float J[D][D];
#pragma unroll
        for (int r = 0; r < D; ++r)
#pragma unroll
            for (int d = 0; d < D; ++d) J[r][d] = nf_half_allsum(W3[r * NF_RES_H + u] * l2.d1 * jt[d]);
        // ---- s = v + sum_k coef_k (J^T)^k v  (every lane, D x D) ----------------------------------------------------------------------
        float sv[D], wv2[D];
#pragma unroll
        for (int d = 0; d < D; ++d) { sv[d] = vv[d]; wv2[d] = vv[d]; }
        for (int k = 1; k <= n_terms; ++k) {
            float nw[D];
#pragma unroll
            for (int c = 0; c < D; ++c) {
                float a = 0.f;
#pragma unroll
                for (int r = 0; r < D; ++r) a = fmaf(J[r][c], wv2[r], a);
                nw[c] = a;
            }
            const float ck = coef[k - 1];
#pragma unroll
            for (int d = 0; d < D; ++d) { wv2[d] = nw[d]; sv[d] = fmaf(ck, nw[d], sv[d]); }
        }
        // ---- signals of layer 2 ------------------------------------------------------------------------------------------------------
        float r3 = 0.f, g3 = 0.f;
#pragma unroll
        for (int r = 0; r < D; ++r) { r3 = fmaf(W3[r * NF_RES_H + u], sv[r], r3); g3 = fmaf(W3[r * NF_RES_H + u], dg[r], g3); }
        const float p2 = l2.d1 * q;
        const float GQ = ok ? cs * r3 * l2.d1 : 0.f;
        const float GH2 = ok ? fmaf(cs * r3 * l2.d2, q, g3 * l2.d1) : 0.f;
        if (ok) {
#pragma unroll
            for (int r = 0; r < D; ++r) accW3[r] += fmaf(cs * sv[r], p2, dg[r] * l2.f);
            acc_b2 += GH2;
            acc_be2 += fmaf(cs * r3 * q, l2.d1b, g3 * l2.db);
#pragma unroll
            for (int d = 0; d < D; ++d)
                if (u == d) acc_b3 += dg[d];
        }
#pragma unroll 8
        for (int i = 0; i < NF_RES_H; ++i) accW2[i] += fmaf(GQ, P1[i], GH2 * A1[i]);
        GQl[u] = GQ;
        GHl[u] = GH2;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- back to layer 1 (lane = unit u): W2^T products walk a COLUMN of W2 (row stride 33: conflict-free) --------------------------
        float gp1 = 0.f, ga1 = 0.f;
#pragma unroll 8
        for (int o = 0; o < NF_RES_H; ++o) {
            const float wv = W2[o * (NF_RES_H + 1) + u];
            gp1 = fmaf(wv, GQl[o], gp1);
            ga1 = fmaf(wv, GHl[o], ga1);
        }
        const float GH1 = fmaf(gp1 * l1.d2, aV, ga1 * l1.d1);
        if (ok) {
            acc_b1 += GH1;
            acc_be1 += fmaf(gp1 * aV, l1.d1b, ga1 * l1.db);
#pragma unroll
            for (int d = 0; d < D; ++d) accW1[d] += fmaf(gp1 * l1.d1, vv[d], GH1 * xv[d]);
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float dx = nf_half_allsum(W1[u * D + d] * GH1);
            if (ok && u == 0) d_x[b * D + d] = dx;
        }
        __builtin_amdgcn_wave_barrier();                  // the slot buffers are rewritten by the next pair
    }
    // ---- totals: the two slots of a wave, then one atomic per entry and wave ------------------------------------------------------------
    float* gW1 = g_out;
    float* gb1 = gW1 + NF_RES_H * D;
    float* gW2 = gb1 + NF_RES_H;
    float* gb2 = gW2 + NF_RES_H * NF_RES_H;
    float* gW3 = gb2 + NF_RES_H;
    float* gb3 = gW3 + D * NF_RES_H;
    float* gbe = gb3 + D;
    // every wave of every workgroup adds to the same entries: deterministic mode takes the workgroups in block order and, inside one,
    // the waves in wave order (within a wave no two lanes share an address)
    NF_DET_ENTER_ALL(nf_rsm);
    nf_det_waves(nf_det_, NF_RT_WAVES, wid, [&] {
#pragma unroll
        for (int i = 0; i < NF_RES_H; ++i) {
            const float t = accW2[i] + __shfl_xor(accW2[i], 32, NF_WAVE);
            if (slot == 0) atomicAdd(gW2 + u * NF_RES_H + i, t);
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float t1 = accW1[d] + __shfl_xor(accW1[d], 32, NF_WAVE);
            const float t3 = accW3[d] + __shfl_xor(accW3[d], 32, NF_WAVE);
            if (slot == 0) { atomicAdd(gW1 + u * D + d, t1); atomicAdd(gW3 + d * NF_RES_H + u, t3); }
        }
        {
            const float tb1 = acc_b1 + __shfl_xor(acc_b1, 32, NF_WAVE);
            const float tb2 = acc_b2 + __shfl_xor(acc_b2, 32, NF_WAVE);
            const float tb3 = acc_b3 + __shfl_xor(acc_b3, 32, NF_WAVE);
            if (slot == 0) { atomicAdd(gb1 + u, tb1); atomicAdd(gb2 + u, tb2); if (u < D) atomicAdd(gb3 + u, tb3); }
            float e1 = nf_half_allsum(acc_be1);
            float e2 = nf_half_allsum(acc_be2);
            e1 += __shfl_xor(e1, 32, NF_WAVE);
            e2 += __shfl_xor(e2, 32, NF_WAVE);
            if (lane == 0) { atomicAdd(gbe, e1); atomicAdd(gbe + 1, e2); }
        }
    });
    NF_DET_LEAVE_ALL(nf_rsm);
}

extern "C" int nf_resmlp_train_bwd(const float* x, const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                                   const float* b3, const float* beta1, const float* beta2, const float* noise, const float* coef,
                                   int n_terms, const float* d_g, const float* d_ld, float* d_x, float* g_params, int64_t B, int D,
                                   nf_stream_t stream) {
    if (D < 1 || D > NF_RES_MAXD || n_terms < 0 || n_terms > NF_RES_MAXK || B < 0) return NF_E_BADARG;
    if (B == 0) return 0;
    NfResW w = {W1, b1, W2, b2, W3, b3, beta1, beta2};
    int64_t pairs = (B + 1) / 2;
    unsigned grid = (unsigned)((pairs + NF_RT_WAVES - 1) / NF_RT_WAVES);
    if (grid > 256) grid = 256;
    hipStream_t st = (hipStream_t)stream;
#define NF_L(D_)                                                                                                              \
    hipLaunchKernelGGL(k_resmlp_train_bwd<D_>, dim3(grid), dim3(NF_RT_THREADS),                                               \
                       NF_RES_LDS(D_) + NF_RT_WAVES * 2 * (4 + D_) * NF_RES_H * sizeof(float), st, w, x, noise, coef, n_terms, d_g, \
                       d_ld, d_x, g_params, B)
    switch (D) {
        case 1: NF_L(1); break;
        case 2: NF_L(2); break;
        case 3: NF_L(3); break;
        default: NF_L(4); break;
    }
#undef NF_L
    NF_CHECK_LAUNCH();
    return 0;
}

// autograd of nf_spectral_weights (spectral_norm.py:36-43): W_eff = W_bar * min(coeff / (sigma + eps), 1), sigma = u^T W_bar v with the
// power-iteration vectors held constant.  g_W_bar += g_W_eff * scale  -  [scale < 1] <g_W_eff, W_bar> coeff / (sigma + eps)^2 * u v^T
struct NfSnBwdArgs {
    const float* Wbar[3];
    const float* u[3];
    const float* v[3];
    const float* gWeff[3];
    float* gWbar[3];
    int h[3];
    int w[3];
};
__global__ void __launch_bounds__(NF_BLOCK) k_spectral_bwd(NfSnBwdArgs a, float coeff, float eps) {
    __shared__ float scratch[NF_BLOCK / NF_WAVE];
    __shared__ float bc[2];
    const int m = blockIdx.x;
    const float* W = a.Wbar[m];
    const float* u = a.u[m];
    const float* v = a.v[m];
    const float* g = a.gWeff[m];
    float* out = a.gWbar[m];
    const int C = a.w[m], n = a.h[m] * C;
    float ps = 0.f, pd = 0.f;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const int r = e / C, c = e - r * C;
        ps = fmaf(u[r] * v[c], W[e], ps);
        pd = fmaf(g[e], W[e], pd);
    }
    const float sigma = nf_block_sum(ps, scratch);
    const float dot = nf_block_sum(pd, scratch);
    if (threadIdx.x == 0) { bc[0] = sigma; bc[1] = dot; }
    __syncthreads();
    const float sg = bc[0], scale = coeff / (sg + eps);
    const bool active = scale < 1.f;
    const float k2 = active ? -bc[1] * coeff / ((sg + eps) * (sg + eps)) : 0.f;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const int r = e / C, c = e - r * C;
        out[e] += (active ? g[e] * scale : g[e]) + k2 * u[r] * v[c];
    }
}
extern "C" int nf_spectral_weights_bwd(const float* const* W_bar, const float* const* u, const float* const* v,
                                       const float* const* g_W_eff, float* const* g_W_bar, const int* rows, const int* cols, int n_mats,
                                       float coeff, float eps, nf_stream_t stream) {
    if (n_mats < 1 || n_mats > 3) return NF_E_BADARG;
    NfSnBwdArgs a;
    for (int i = 0; i < n_mats; ++i) {
        if (rows[i] < 1 || cols[i] < 1 || rows[i] > 64 || cols[i] > 64) return NF_E_BADARG;
        a.Wbar[i] = W_bar[i]; a.u[i] = u[i]; a.v[i] = v[i]; a.gWeff[i] = g_W_eff[i]; a.gWbar[i] = g_W_bar[i]; a.h[i] = rows[i]; a.w[i] = cols[i];
    }
    hipLaunchKernelGGL(k_spectral_bwd, dim3((unsigned)n_mats), dim3(NF_BLOCK), 0, (hipStream_t)stream, a, coeff, eps);
    NF_CHECK_LAUNCH();
    return 0;
}
