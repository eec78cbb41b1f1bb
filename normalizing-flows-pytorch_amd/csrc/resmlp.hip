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
