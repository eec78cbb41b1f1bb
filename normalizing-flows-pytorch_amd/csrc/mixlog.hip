// Flow++ mixture-of-logistics coupling, fully fused per transformed element:
//   forward : log_softmax(pi) -> mixture CDF (log space) -> Logit(eps) -> * exp(a) + b, with the three log-det terms
//             (log pdf, logit, a) reduced per sample into ld;            coupling.py:172-190, modules.py:64-97,190-194
//   inverse : affine^-1 -> sigmoid -> bisection of the CDF (bracket +-1e3, 25 or 100 iterations by the reference's
//             batch-global exit rule, reproduced with a device flag) -> - log pdf;   coupling.py:192-210, modules.py:196-212
//   backward: analytic gradients in the responsibilities form r_k = pi_k pdf_k / f, which
//             stays finite in the tails where f underflows.
// Transcendental-heavy (about 6 exp/log per mixture component) but still read-dominated: (4+3K)*4 B per element.
//
// Parameter addressing: params is (B, (2+3K)*Ch, h, w); for half-element e = (m*h+i)*w+j of sample b
//   a_raw = P[e], b = P[nh+e], logit pi_k = P[(2+k)*nh+e], mu_k = P[(2+K+k)*nh+e], s_k = P[(2+2K+k)*nh+e],  nh = Ch*h*w,
// i.e. every k-plane is contiguous along e: coalesced for images; for 2-D data (nh = 1) a sample's 2+3K values are one
// contiguous row.
#include "nf_common.h"

#define NF_MX_ROWS_MAX 16
#define NF_MX_SLAB 1024

struct NfLse {  // streaming log-sum-exp
    float m, s;
    __device__ __forceinline__ void init() { m = -INFINITY; s = 0.f; }
    __device__ __forceinline__ void add(float v) {
        if (v > m) { s = s * expf(m - v) + 1.f; m = v; }
        else s += expf(v - m);
    }
    __device__ __forceinline__ float value() const { return m + logf(s); }
};

__device__ __forceinline__ float nf_lse_logits(const float* __restrict__ P, int K, int64_t nh) {
    NfLse l; l.init();
    for (int k = 0; k < K; ++k) l.add(P[(2 + k) * nh]);
    return l.value();
}

// log CDF and log PDF of the mixture at x (P already offset to the element)
__device__ __forceinline__ void nf_mix_eval(const float* __restrict__ P, int K, int64_t nh, float lse_pi, float x,
                                            float& lcdf, float& lpdf) {
    NfLse c, d; c.init(); d.init();
    for (int k = 0; k < K; ++k) {
        const float lp = P[(2 + k) * nh] - lse_pi;
        const float mu = P[(2 + K + k) * nh], s = P[(2 + 2 * K + k) * nh];
        const float u = (x - mu) * expf(-s);                       // modules.py:66, :72
        c.add(lp + nf_logsigmoid(u));                              // modules.py:73, :97
        d.add(lp + (u - s - 2.f * nf_softplus(u)));                // modules.py:67, :85
    }
    lcdf = c.value();
    lpdf = d.value();
}

__device__ __forceinline__ float nf_mix_cdf(const float* __restrict__ P, int K, int64_t nh, float lse_pi, float x) {
    NfLse c; c.init();
    for (int k = 0; k < K; ++k) {
        const float u = (x - P[(2 + K + k) * nh]) * expf(-P[(2 + 2 * K + k) * nh]);
        c.add(P[(2 + k) * nh] - lse_pi + nf_logsigmoid(u));
    }
    return expf(c.value());
}

// ---- forward element ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float nf_mixlog_fwd_elem(const float* __restrict__ P, int K, int64_t nh, float x, float A,
                                                    float Cb, float eps, float& acc) {
    const float lse_pi = nf_lse_logits(P, K, nh);
    float lcdf, lpdf;
    nf_mix_eval(P, K, nh, lse_pi, x, lcdf, lpdf);
    const float F = expf(lcdf);                                   // modules.py:194
    const float xc = fminf(fmaxf(F, eps), 1.f - eps);             // modules.py:147
    const float xi = fminf(fmaxf(xc, 1.0e-8f), 1.f - 1.0e-8f);    // modules.py:31
    const float yi = logf(xi / (1.f - xi));
    const float y1 = logf(xc / (1.f - xc));
    const float a = tanhf(P[0]) * A + Cb;                         // coupling.py:178
    acc += lpdf + (-(yi - 2.f * nf_softplus(yi))) + a;            // coupling.py:184-188
    return y1 * expf(a) + P[nh];                                  // coupling.py:187
}

// ---- backward element (appendix B6, responsibilities form) ----------------------------------------------------------
__device__ __forceinline__ float nf_mixlog_bwd_elem(const float* __restrict__ P, float* __restrict__ GP, int K,
                                                    int64_t nh, float x, float gy, float gld, float A, float Cb,
                                                    float eps, float& acc_A, float& acc_C) {
    const float lse_pi = nf_lse_logits(P, K, nh);
    float lcdf, lpdf;
    nf_mix_eval(P, K, nh, lse_pi, x, lcdf, lpdf);
    const float F = expf(lcdf), f = expf(lpdf);
    const bool inside = (F >= eps) && (F <= 1.f - eps);          // torch.clamp passes the gradient on [min, max]
    const float xc = fminf(fmaxf(F, eps), 1.f - eps);
    const float y1 = logf(xc / (1.f - xc));
    const float th = tanhf(P[0]);
    const float ea = expf(th * A + Cb);
    // affine: y = y1 * exp(a) + b ; ld += a
    const float g_y1 = gy * ea;
    const float g_a = gy * y1 * ea + gld;
    GP[0] = g_a * A * (1.f - th * th);
    GP[nh] = gy;
    acc_A += g_a * th;
    acc_C += g_a;
    // logit: y1 = log(F/(1-F)), ld += -log(F (1-F))
    const float gF = inside ? (g_y1 - gld * (1.f - 2.f * xc)) / (xc * (1.f - xc)) : 0.f;
    // mixture
    const float tot = gF * F + gld;                              // sum_j g_logpi_j
    float gx = gF * f;
    for (int k = 0; k < K; ++k) {
        const float lp = P[(2 + k) * nh] - lse_pi;
        const float mu = P[(2 + K + k) * nh], s = P[(2 + 2 * K + k) * nh];
        const float es = expf(-s);
        const float u = (x - mu) * es;
        const float r = expf(lp + (u - s - 2.f * nf_softplus(u)) - lpdf);   // pi_k pdf_k / f
        const float omt = -tanhf(0.5f * u);                                 // 1 - 2 sigmoid(u)
        const float w = gld * r * omt * es;
        gx += w;
        GP[(2 + K + k) * nh] = -gF * f * r - w;                                          // g_mu_k
        GP[(2 + 2 * K + k) * nh] = -gF * f * r * (x - mu) + gld * r * (-omt * u - 1.f);  // g_s_k
        const float g_logpi = gF * expf(lp + nf_logsigmoid(u)) + gld * r;
        GP[(2 + k) * nh] = g_logpi - expf(lp) * tot;                                     // through log_softmax
    }
    return gx;
}

// ---------------------------------------------------------------------------------------------------------------
// forward kernels
__global__ void __launch_bounds__(NF_BLOCK) k_mixlog_rows_fwd(const float* __restrict__ z, const float* __restrict__ prm,
                                                              const float* __restrict__ pA, const float* __restrict__ pC,
                                                              float* __restrict__ y, float* __restrict__ ld, NfSplit s,
                                                              int K, float eps, int64_t B) {
    const float A = pA[0], Cb = pC[0];
    const int64_t PS = (int64_t)(2 + 3 * K) * s.n_half;
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += (int64_t)gridDim.x * blockDim.x) {
        const float* zb = z + b * s.n_full;
        float* yb = y + b * s.n_full;
        float acc = 0.f;
        for (int e = 0; e < s.n_half; ++e) {
            const int o0 = nf_half_to_full(s, 0, e), o1 = nf_half_to_full(s, 1, e);
            yb[o0] = nf_mixlog_fwd_elem(prm + b * PS + e, K, s.n_half, zb[o0], A, Cb, eps, acc);
            yb[o1] = zb[o1];
        }
        ld[b] += acc;
    }
}

__global__ void __launch_bounds__(NF_BLOCK) k_mixlog_slab_fwd(const float* __restrict__ z, const float* __restrict__ prm,
                                                              const float* __restrict__ pA, const float* __restrict__ pC,
                                                              float* __restrict__ y, float* __restrict__ ld, NfSplit s,
                                                              int K, float eps) {
    __shared__ float scratch[NF_BLOCK / NF_WAVE];
    const float A = pA[0], Cb = pC[0];
    const int64_t b = blockIdx.x;
    const int64_t PS = (int64_t)(2 + 3 * K) * s.n_half;
    const int e0 = blockIdx.y * NF_MX_SLAB, e1 = min(e0 + NF_MX_SLAB, s.n_half);
    const float* zb = z + b * s.n_full;
    float* yb = y + b * s.n_full;
    float acc = 0.f;
    for (int e = e0 + threadIdx.x; e < e1; e += NF_BLOCK) {
        const int o0 = nf_half_to_full(s, 0, e), o1 = nf_half_to_full(s, 1, e);
        yb[o0] = nf_mixlog_fwd_elem(prm + b * PS + e, K, s.n_half, zb[o0], A, Cb, eps, acc);
        yb[o1] = zb[o1];
    }
    const float tot = nf_block_sum(acc, scratch);
    if (threadIdx.x == 0) {
        if (gridDim.y == 1) ld[b] += tot;
        else atomicAdd(ld + b, tot);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward kernels
__global__ void __launch_bounds__(NF_BLOCK) k_mixlog_bwd(const float* __restrict__ gy, const float* __restrict__ gld,
                                                         const float* __restrict__ z, const float* __restrict__ prm,
                                                         const float* __restrict__ pA, const float* __restrict__ pC,
                                                         float* __restrict__ gz, float* __restrict__ gprm,
                                                         float* __restrict__ g_scale, float* __restrict__ g_bias,
                                                         NfSplit s, int K, float eps, int64_t total) {
    __shared__ float scratch[NF_BLOCK / NF_WAVE];
    const float A = pA[0], Cb = pC[0];
    const int64_t PS = (int64_t)(2 + 3 * K) * s.n_half;
    float acc_A = 0.f, acc_C = 0.f;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / s.n_half;
        const int e = (int)(t - b * s.n_half);
        const int64_t fb = b * s.n_full;
        const int o0 = nf_half_to_full(s, 0, e), o1 = nf_half_to_full(s, 1, e);
        gz[fb + o0] = nf_mixlog_bwd_elem(prm + b * PS + e, gprm + b * PS + e, K, s.n_half, z[fb + o0], gy[fb + o0],
                                         gld[b], A, Cb, eps, acc_A, acc_C);
        gz[fb + o1] = gy[fb + o1];
    }
    const float ta = nf_block_sum(acc_A, scratch);
    const float tc = nf_block_sum(acc_C, scratch);
    if (threadIdx.x == 0) {
        atomicAdd(g_scale, ta);
        atomicAdd(g_bias, tc);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// inverse: phase 1 = affine^-1, sigmoid, 25 bisection steps, bracket stored, "stuck" flag raised;
//          phase 2 = 75 more steps iff the flag is set (the reference's batch-global rule), then - log pdf.
__global__ void __launch_bounds__(NF_BLOCK) k_mixlog_inv_phase1(const float* __restrict__ yin, const float* __restrict__ prm,
                                                                const float* __restrict__ pA, const float* __restrict__ pC,
                                                                float* __restrict__ ld, float* __restrict__ lohi,
                                                                int* __restrict__ flag, NfSplit s, int K, int64_t total) {
    const float A = pA[0], Cb = pC[0];
    const int64_t PS = (int64_t)(2 + 3 * K) * s.n_half;
    bool stuck = false;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / s.n_half;
        const int e = (int)(t - b * s.n_half);
        const float* P = prm + b * PS + e;
        const float a = tanhf(P[0]) * A + Cb;
        const float v = expf(-a) * (yin[b * s.n_full + nf_half_to_full(s, 0, e)] - P[s.n_half]);   // coupling.py:204
        const float target = 1.f / (1.f + expf(-v));                                               // modules.py:155
        const float dl = -a + (v - 2.f * nf_softplus(v));                                          // coupling.py:205, modules.py:153
        if (s.n_half == 1) ld[b] += dl;
        else atomicAdd(ld + b, dl);
        const float lse_pi = nf_lse_logits(P, K, s.n_half);
        float lo = -1.0e3f, hi = 1.0e3f;                                                           // modules.py:197-198
        for (int it = 0; it < 25; ++it) {
            const float mid = (lo + hi) * 0.5f;
            const float val = nf_mix_cdf(P, K, s.n_half, lse_pi, mid);
            lo = val < target ? mid : lo;                                                          // modules.py:202-203
            hi = val > target ? mid : hi;
        }
        lohi[2 * t] = lo;
        lohi[2 * t + 1] = hi;
        lohi[2 * total + t] = target;
        stuck |= !(fabsf(hi - lo) < 1.0e-4f);                                                      // modules.py:205
    }
    if (__any(stuck) && (threadIdx.x & (NF_WAVE - 1)) == 0) atomicOr(flag, 1);
}

__global__ void __launch_bounds__(NF_BLOCK) k_mixlog_inv_phase2(const float* __restrict__ yin, const float* __restrict__ prm,
                                                                float* __restrict__ y, float* __restrict__ ld,
                                                                const float* __restrict__ lohi, const int* __restrict__ flag,
                                                                NfSplit s, int K, int64_t total) {
    const int64_t PS = (int64_t)(2 + 3 * K) * s.n_half;
    const int more = flag[0] ? 75 : 0;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / s.n_half;
        const int e = (int)(t - b * s.n_half);
        const float* P = prm + b * PS + e;
        const float lse_pi = nf_lse_logits(P, K, s.n_half);
        float lo = lohi[2 * t], hi = lohi[2 * t + 1];
        const float target = lohi[2 * total + t];
        for (int it = 0; it < more; ++it) {
            const float mid = (lo + hi) * 0.5f;
            const float val = nf_mix_cdf(P, K, s.n_half, lse_pi, mid);
            lo = val < target ? mid : lo;
            hi = val > target ? mid : hi;
        }
        const float x = (lo + hi) * 0.5f;                                                          // modules.py:208
        float lcdf, lpdf;
        nf_mix_eval(P, K, s.n_half, lse_pi, x, lcdf, lpdf);
        if (s.n_half == 1) ld[b] -= lpdf;                                                          // modules.py:209-212
        else atomicAdd(ld + b, -lpdf);
        const int64_t fb = b * s.n_full;
        const int o1 = nf_half_to_full(s, 1, e);
        y[fb + nf_half_to_full(s, 0, e)] = x;
        y[fb + o1] = yin[fb + o1];
    }
}

// ---------------------------------------------------------------------------------------------------------------
static inline bool nf_mixlog_args(NfSplit& s, int mode, int odd, int C, int H, int W, int K) {
    return nf_make_split(s, mode, odd, C, H, W) && mode != NF_SPLIT_NONE && K >= 1 && K <= 64;
}

extern "C" int nf_mixlog_coupling_fwd(const float* z, const float* params, const float* a_log_scale,
                                      const float* a_bias, float* y, float* ld, int K, float logit_eps, int mode, int odd,
                                      int64_t B, int C, int H, int W, nf_stream_t stream) {
    NfSplit s;
    if (!nf_mixlog_args(s, mode, odd, C, H, W, K)) return NF_E_BADARG;
    if (B == 0 || s.n_half == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (s.n_half <= NF_MX_ROWS_MAX) {
        hipLaunchKernelGGL(k_mixlog_rows_fwd, dim3(nf_grid_for(B, 64)), dim3(64), 0, st, z, params, a_log_scale, a_bias, y,
                           ld, s, K, logit_eps, B);
    } else {
        if (B > 0x7fffffffLL) return NF_E_BADARG;
        dim3 grid((unsigned)B, (unsigned)((s.n_half + NF_MX_SLAB - 1) / NF_MX_SLAB));
        hipLaunchKernelGGL(k_mixlog_slab_fwd, grid, dim3(NF_BLOCK), 0, st, z, params, a_log_scale, a_bias, y, ld, s, K,
                           logit_eps);
    }
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_mixlog_coupling_bwd(const float* g_y, const float* g_ld, const float* z, const float* params,
                                      const float* a_log_scale, const float* a_bias, float* g_z, float* g_params,
                                      float* g_scale, float* g_bias, int K, float logit_eps, int mode, int odd, int64_t B,
                                      int C, int H, int W, nf_stream_t stream) {
    NfSplit s;
    if (!nf_mixlog_args(s, mode, odd, C, H, W, K)) return NF_E_BADARG;
    const int64_t total = B * s.n_half;
    if (total == 0) return 0;
    hipLaunchKernelGGL(k_mixlog_bwd, dim3(nf_grid_for(total, 64)), dim3(64), 0, (hipStream_t)stream, g_y, g_ld, z, params,
                       a_log_scale, a_bias, g_z, g_params, g_scale, g_bias, s, K, logit_eps, total);
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_mixlog_coupling_inv(const float* z, const float* params, const float* a_log_scale,
                                      const float* a_bias, float* y, float* ld, float* scratch, int* stuck_flag, int K,
                                      int mode, int odd, int64_t B, int C, int H, int W, nf_stream_t stream) {
    NfSplit s;
    if (!nf_mixlog_args(s, mode, odd, C, H, W, K)) return NF_E_BADARG;
    const int64_t total = B * s.n_half;
    if (total == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(stuck_flag, 0, sizeof(int), st);
    if (e != hipSuccess) return (int)e;
    const dim3 grid(nf_grid_for(total, 64)), block(64);
    hipLaunchKernelGGL(k_mixlog_inv_phase1, grid, block, 0, st, z, params, a_log_scale, a_bias, ld, scratch, stuck_flag, s,
                       K, total);
    NF_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_mixlog_inv_phase2, grid, block, 0, st, z, params, y, ld, scratch, stuck_flag, s, K, total);
    NF_CHECK_LAUNCH();
    return 0;
}
