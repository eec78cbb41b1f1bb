// Flow++ mixture-of-logistics coupling, fully fused per transformed element:
//   forward : log_softmax(pi) -> mixture CDF (log space) -> Logit(eps) -> * exp(a) + b, with the three log-det terms
//             (log pdf, logit, a) reduced per sample into ld;            coupling.py:172-190, modules.py:64-97,190-194
//   inverse : affine^-1 -> sigmoid -> bisection of the CDF (bracket +-1e3, 25 or 100 iterations by the reference's
//             batch-global exit rule, reproduced with a device flag) -> - log pdf;   coupling.py:192-210, modules.py:196-212
//   backward: analytic gradients in the responsibilities form r_k = pi_k pdf_k / f, which
//             stays finite in the tails where f underflows.
// Bytes: (4+3K)*4 B per transformed element (112 B at K = 8) + 8 B pass-through; ~7 transcendentals per component forward, ~13
// backward, all in their hardware forms (v_exp_f32 / v_log_f32 / v_rcp_f32 through nf_mixlog_oct.h: the libm forms are 8 .. 30
// instructions each and made these kernels instruction-bound at a tenth of the transcendental rate).
//
// Every element's 2+3K parameters are loaded ONCE into registers (template KT >= K, unused slots masked with
// log pi = -inf), so the 25..100 bisection steps and the backward's second sweep never go back to memory.
// Parameter addressing: params is (B, (2+3K)*Ch, h, w); for half-element e = (m*h+i)*w+j of sample b
//   a_raw = P[e], b = P[nh+e], logit pi_k = P[(2+k)*nh+e], mu_k = P[(2+K+k)*nh+e], s_k = P[(2+2K+k)*nh+e],  nh = Ch*h*w:
// every k-plane is contiguous along e (coalesced for images).  For 2-D data (nh = 1) a sample's 2+3K values are one
// contiguous row: the block stages its rows through LDS (row stride 2+3K+1 words) so that global traffic is coalesced
// both for the parameters and for their gradients.
#include "nf_common.h"
#include "nf_det.h"

NF_DET_STATE(nf_ml)
NF_DET_HOST_API(nf_ml)
#include "nf_mixlog_oct.h"

#define NF_MX_ROWS_MAX 16
#define NF_MX_SLAB 256      // elements per workgroup of the image forward: 256 = one per thread (1024: four sequential 26-load round trips per thread, 20 us against 12 at Flowpp CIFAR shape B = 64)
#define NF_MX_GRID 1024

template <int KT>
struct NfMix {
    float a_raw, b;
    float lp[KT], mu[KT], s[KT], es[KT];   // log pi (normalised), mu, s, exp(-s)
};

template <int KT>
__device__ __forceinline__ void nf_mix_load(const float* __restrict__ P, int64_t nh, int K, NfMix<KT>& m) {
    m.a_raw = P[0];
    m.b = P[nh];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        m.lp[k] = k < K ? P[(2 + k) * nh] : -INFINITY;
        m.mu[k] = k < K ? P[(2 + K + k) * nh] : 0.f;
        m.s[k] = k < K ? P[(2 + 2 * K + k) * nh] : 0.f;
        mx = fmaxf(mx, m.lp[k]);
    }
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < KT; ++k) se += nf_fexp(m.lp[k] - mx);
    const float lse = mx + nf_flog(se);                                 // F.log_softmax over the mixture axis (coupling.py:180)
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        m.lp[k] -= lse;
        m.es[k] = nf_fexp(-m.s[k]);
    }
}

// log CDF and log PDF of the mixture at x (modules.py:64-97): logsigmoid(u) = min(u,0) - l, softplus(u) = max(u,0) + l
// with the shared l = log1p(exp(-|u|))
template <int KT>
__device__ __forceinline__ void nf_mix_eval(const NfMix<KT>& m, float x, float& lcdf, float& lpdf) {
    float c[KT], d[KT], cm = -INFINITY, dm = -INFINITY;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        const float u = (x - m.mu[k]) * m.es[k];
        const float l = nf_flog(1.f + nf_fexp(-fabsf(u)));
        c[k] = m.lp[k] + (fminf(u, 0.f) - l);
        d[k] = m.lp[k] + (u - m.s[k] - 2.f * (fmaxf(u, 0.f) + l));
        cm = fmaxf(cm, c[k]);
        dm = fmaxf(dm, d[k]);
    }
    float sc = 0.f, sd = 0.f;
#pragma unroll
    for (int k = 0; k < KT; ++k) { sc += nf_fexp(c[k] - cm); sd += nf_fexp(d[k] - dm); }
    lcdf = cm + nf_flog(sc);
    lpdf = dm + nf_flog(sd);
}

// mixture CDF in linear space for the bisection: sum_k pi_k sigmoid(u_k) == exp(logsumexp(log pi + logsigmoid)) up to
// rounding; a third of the transcendentals, and the bracket decisions are insensitive to that rounding
template <int KT>
__device__ __forceinline__ float nf_mix_cdf(const NfMix<KT>& m, const float (&pi)[KT], float x) {
    float F = 0.f;
#pragma unroll
    for (int k = 0; k < KT; ++k) F = fmaf(pi[k], __builtin_amdgcn_rcpf(1.f + nf_fexp(-(x - m.mu[k]) * m.es[k])), F);
    return F;
}

template <int KT>
__device__ __forceinline__ float nf_mixlog_fwd_elem(const NfMix<KT>& m, float x, float A, float Cb, float eps, float& acc) {
    float lcdf, lpdf;
    nf_mix_eval<KT>(m, x, lcdf, lpdf);
    const float F = nf_fexp(lcdf);                                // modules.py:194
    const float xc = fminf(fmaxf(F, eps), 1.f - eps);             // modules.py:147
    const float la = nf_flog(xc), lb = nf_flog(1.f - xc);               // logit and its log-det share the two logs
    const float a = nf_ftanh(m.a_raw) * A + Cb;                   // coupling.py:178
    acc += lpdf - (la + lb) + a;                                  // coupling.py:184-188
    return (la - lb) * nf_fexp(a) + m.b;                             // coupling.py:187
}

// backward element (appendix B6, responsibilities form); writes the 2+3K parameter gradients through GP (stride gnh)
template <int KT>
__device__ __forceinline__ float nf_mixlog_bwd_elem(const NfMix<KT>& m, float* __restrict__ GP, int64_t gnh, int K, float x,
                                                    float gy, float gld, float A, float Cb, float eps, float& acc_A,
                                                    float& acc_C) {
    float lcdf, lpdf;
    nf_mix_eval<KT>(m, x, lcdf, lpdf);
    const float F = nf_fexp(lcdf), f = nf_fexp(lpdf);
    const bool inside = (F >= eps) && (F <= 1.f - eps);          // torch.clamp passes the gradient on [min, max]
    const float xc = fminf(fmaxf(F, eps), 1.f - eps);
    const float y1 = nf_flog(xc) - nf_flog(1.f - xc);
    const float th = nf_ftanh(m.a_raw);
    const float ea = nf_fexp(th * A + Cb);
    const float g_y1 = gy * ea;                                  // y = y1 * exp(a) + b ; ld += a
    const float g_a = gy * y1 * ea + gld;
    GP[0] = g_a * A * (1.f - th * th);
    GP[gnh] = gy;
    acc_A += g_a * th;
    acc_C += g_a;
    const float gF = inside ? (g_y1 - gld * (1.f - 2.f * xc)) * __builtin_amdgcn_rcpf(xc * (1.f - xc)) : 0.f;   // logit + its log-det
    const float tot = gF * F + gld;                              // sum_j g_logpi_j
    float gx = gF * f;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        if (k < K) {
            const float u = (x - m.mu[k]) * m.es[k];
            const float eu = nf_fexp(-fabsf(u)), ru = __builtin_amdgcn_rcpf(1.f + eu);
            const float l = nf_flog(1.f + eu);
            const float r = nf_fexp(m.lp[k] + (u - m.s[k] - 2.f * (fmaxf(u, 0.f) + l)) - lpdf);   // pi_k pdf_k / f
            const float omt = copysignf((1.f - eu) * ru, -u);                              // 1 - 2 sigmoid(u) = -tanh(u / 2)
            const float w = gld * r * omt * m.es[k];
            gx += w;
            GP[(2 + K + k) * gnh] = -gF * f * r - w;                                            // g_mu_k
            GP[(2 + 2 * K + k) * gnh] = -gF * f * r * (x - m.mu[k]) + gld * r * (-omt * u - 1.f);   // g_s_k
            const float g_logpi = gF * nf_fexp(m.lp[k] + (fminf(u, 0.f) - l)) + gld * r;
            GP[(2 + k) * gnh] = g_logpi - nf_fexp(m.lp[k]) * tot;                                  // through log_softmax
        }
    }
    return gx;
}

// ---- the shared-transcendental form of one element (round 6; the comment in front of the row kernels below says why) ----------------
#define NF_MR_TINY 1.0e-30f
// the row's parameters and everything of them that does not depend on x
template <int KT>
struct NfMixLin {
    float a_raw, b, inv_se;
    float w[KT], mu[KT], es[KT];       // exp(lp - max lp) (0 beyond K), mu, exp(-s)
};
template <int KT>
__device__ __forceinline__ void nf_mr_load(const float* __restrict__ row, int64_t nh, int K, NfMixLin<KT>& m, float (&lp)[KT], float (&sv)[KT]) {
    m.a_raw = row[0];
    m.b = row[nh];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        lp[k] = k < K ? row[(2 + k) * nh] : -INFINITY;
        m.mu[k] = k < K ? row[(2 + K + k) * nh] : 0.f;
        sv[k] = k < K ? row[(2 + 2 * K + k) * nh] : 0.f;
        mx = fmaxf(mx, lp[k]);
    }
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        m.w[k] = nf_fexp(lp[k] - mx);
        m.es[k] = nf_fexp(-sv[k]);
        se += m.w[k];
    }
    m.inv_se = __builtin_amdgcn_rcpf(se);
}
// Fs = sum w sigma, fs = sum w es sigma (1 - sigma) (both still to be divided by sum w); t, r of every component kept for the backward
template <int KT>
__device__ __forceinline__ void nf_mr_eval(const NfMixLin<KT>& m, float x, float& Fs, float& fs, float (&u)[KT], float (&t)[KT], float (&r)[KT]) {
    Fs = 0.f; fs = 0.f;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        u[k] = (x - m.mu[k]) * m.es[k];
        t[k] = nf_fexp(-fabsf(u[k]));
        r[k] = __builtin_amdgcn_rcpf(1.f + t[k]);
        const float sig = (u[k] >= 0.f ? 1.f : t[k]) * r[k];
        Fs = fmaf(m.w[k], sig, Fs);
        fs = fmaf(m.w[k] * m.es[k], t[k] * r[k] * r[k], fs);
    }
}
// the log-space record of the same row for the rare rows whose linear-space density underflows
template <int KT>
__device__ __forceinline__ void nf_mr_logspace(const NfMixLin<KT>& m, const float (&lp)[KT], const float (&sv)[KT], NfMix<KT>& q) {
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < KT; ++k) mx = fmaxf(mx, lp[k]);
    const float lse = mx - nf_flog(m.inv_se);
    q.a_raw = m.a_raw; q.b = m.b;
#pragma unroll
    for (int k = 0; k < KT; ++k) { q.lp[k] = lp[k] - lse; q.mu[k] = m.mu[k]; q.s[k] = sv[k]; q.es[k] = m.es[k]; }
}


// one transformed element, forward and backward, in the shared-transcendental form (parameters at stride nh / gradients at stride gnh)
template <int KT>
__device__ __forceinline__ float nf_mr_fwd_elem(const float* __restrict__ P, int64_t nh, int K, float x, float A, float Cb, float eps, float& acc) {
    NfMixLin<KT> m;
    float lp[KT], sv[KT], u[KT], t[KT], r[KT], Fs, fs;
    nf_mr_load<KT>(P, nh, K, m, lp, sv);
    nf_mr_eval<KT>(m, x, Fs, fs, u, t, r);
    float F = Fs * m.inv_se, lpdf;
    if (fs > NF_MR_TINY) {
        lpdf = nf_flog(fs * m.inv_se);
    } else {
        NfMix<KT> q;
        nf_mr_logspace<KT>(m, lp, sv, q);
        float lcdf;
        nf_mix_eval<KT>(q, x, lcdf, lpdf);
        F = nf_fexp(lcdf);
    }
    const float xc = fminf(fmaxf(F, eps), 1.f - eps);                  // modules.py:147
    const float la = nf_flog(xc), lb = nf_flog(1.f - xc);
    const float a = nf_ftanh(m.a_raw) * A + Cb;                        // coupling.py:178
    acc += lpdf - (la + lb) + a;                                       // coupling.py:184-188
    return (la - lb) * nf_fexp(a) + m.b;                               // coupling.py:187
}
template <int KT>
__device__ __forceinline__ float nf_mr_bwd_elem(const float* P, int64_t nh, float* GP, int64_t gnh,   /* (P may BE GP: every parameter is loaded before the first store) */ int K, float x, float g_y,
                                                float g_ld, float A, float Cb, float eps, float& acc_A, float& acc_C) {
    NfMixLin<KT> m;
    float lp[KT], sv[KT], u[KT], t[KT], r[KT], Fs, fs;
    nf_mr_load<KT>(P, nh, K, m, lp, sv);
    nf_mr_eval<KT>(m, x, Fs, fs, u, t, r);
    if (!(fs > NF_MR_TINY)) {
        NfMix<KT> q;
        nf_mr_logspace<KT>(m, lp, sv, q);
        return nf_mixlog_bwd_elem<KT>(q, GP, gnh, K, x, g_y, g_ld, A, Cb, eps, acc_A, acc_C);
    }
    const float F = Fs * m.inv_se, f = fs * m.inv_se;
    const bool inside = (F >= eps) && (F <= 1.f - eps);                 // torch.clamp passes the gradient on [min, max]
    const float xc = fminf(fmaxf(F, eps), 1.f - eps);
    const float y1 = nf_flog(xc) - nf_flog(1.f - xc);
    const float th = nf_ftanh(m.a_raw);
    const float ea = nf_fexp(th * A + Cb);
    const float g_y1 = g_y * ea;                                        // y = y1 * exp(a) + b ; ld += a
    const float g_a = g_y * y1 * ea + g_ld;
    const float gF = inside ? (g_y1 - g_ld * (1.f - 2.f * xc)) * __builtin_amdgcn_rcpf(xc * (1.f - xc)) : 0.f;
    const float tot = gF * F + g_ld;                                    // sum_j g_logpi_j
    const float inv_fs = __builtin_amdgcn_rcpf(fs);
    const float gFf = gF * f;
    float gx = gFf;
    GP[0] = g_a * A * (1.f - th * th);
    GP[gnh] = g_y;
    acc_A += g_a * th;
    acc_C += g_a;
#pragma unroll
    for (int k = 0; k < KT; ++k)
        if (k < K) {
            const float wes = m.w[k] * m.es[k];
            const float rk = wes * t[k] * r[k] * r[k] * inv_fs;                       // responsibility pi_k pdf_k / f
            const float omt = copysignf((1.f - t[k]) * r[k], -u[k]);                  // 1 - 2 sigmoid(u)
            const float wk = g_ld * rk * omt * m.es[k];
            gx += wk;
            const float pik = m.w[k] * m.inv_se;
            const float sig = (u[k] >= 0.f ? 1.f : t[k]) * r[k];
            GP[(2 + K + k) * gnh] = -gFf * rk - wk;                                           // g_mu_k
            GP[(2 + 2 * K + k) * gnh] = -gFf * rk * (x - m.mu[k]) + g_ld * rk * (-omt * u[k] - 1.f);   // g_s_k
            GP[(2 + k) * gnh] = gF * pik * sig + g_ld * rk - pik * tot;                       // g_logit_k through log_softmax
        }
    return gx;
}


// ---- LDS staging of parameter rows (2-D data: nh == 1) ----------------------------------------------------------
__device__ __forceinline__ void nf_stage_rows_in(const float* __restrict__ g, float* __restrict__ tile, int64_t row0,
                                                 int64_t B, int PS) {
    const int rows = (int)min((int64_t)blockDim.x, B - row0);
    const int total = rows * PS, RS = PS + 1;
    const float* src = g + row0 * PS;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int r = i / PS;
        tile[r * RS + (i - r * PS)] = src[i];
    }
    __syncthreads();
}
__device__ __forceinline__ void nf_stage_rows_out(float* __restrict__ g, const float* __restrict__ tile, int64_t row0,
                                                  int64_t B, int PS) {
    __syncthreads();
    const int rows = (int)min((int64_t)blockDim.x, B - row0);
    const int total = rows * PS, RS = PS + 1;
    float* dst = g + row0 * PS;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int r = i / PS;
        dst[i] = tile[r * RS + (i - r * PS)];
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------
// forward
template <int KT>
__global__ void __launch_bounds__(NF_BLOCK) k_mixlog_rows_fwd(const float* __restrict__ z, const float* __restrict__ prm, const float* __restrict__ pA,
                                  const float* __restrict__ pC, float* __restrict__ y, float* __restrict__ ld, NfSplit s,
                                  int K, float eps, int64_t B) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const float A = pA[0], Cb = pC[0];
    const int PS1 = 2 + 3 * K;
    const int64_t PS = (int64_t)PS1 * s.n_half;
    const bool staged = s.n_half == 1;
    for (int64_t row0 = (int64_t)blockIdx.x * blockDim.x; row0 < B; row0 += (int64_t)gridDim.x * blockDim.x) {
        if (staged) nf_stage_rows_in(prm, tile, row0, B, PS1);
        const int64_t b = row0 + threadIdx.x;
        if (b < B) {
            const float* zb = z + b * s.n_full;
            float* yb = y + b * s.n_full;
            float acc = 0.f;
            for (int e = 0; e < s.n_half; ++e) {
                NfMix<KT> m;
                nf_mix_load<KT>(staged ? tile + threadIdx.x * (PS1 + 1) : prm + b * PS + e, s.n_half, K, m);
                const int o0 = nf_half_to_full(s, 0, e), o1 = nf_half_to_full(s, 1, e);
                yb[o0] = nf_mixlog_fwd_elem<KT>(m, zb[o0], A, Cb, eps, acc);
                yb[o1] = zb[o1];
            }
            ld[b] += acc;
        }
        if (staged) __syncthreads();
    }
}

template <int KT>
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
        yb[o0] = nf_mr_fwd_elem<KT>(prm + b * PS + e, s.n_half, K, zb[o0], A, Cb, eps, acc);
        yb[o1] = zb[o1];
    }
    const float tot = nf_block_sum(acc, scratch);
    if (threadIdx.x == 0) {
        if (gridDim.y == 1) ld[b] += tot;
        else { NF_DET_ENTER_COL(nf_ml); atomicAdd(ld + b, tot); NF_DET_LEAVE_COL(nf_ml); }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward: one thread per transformed element, persistent blocks (one pair of atomics per block)
template <int KT>
__global__ void __launch_bounds__(NF_BLOCK) k_mixlog_bwd(const float* __restrict__ gy, const float* __restrict__ gld, const float* __restrict__ z,
                             const float* __restrict__ prm, const float* __restrict__ pA, const float* __restrict__ pC,
                             float* __restrict__ gz, float* __restrict__ gprm, float* __restrict__ g_scale,
                             float* __restrict__ g_bias, NfSplit s, int K, float eps, int64_t total, float* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    __shared__ float scratch[NF_BLOCK / NF_WAVE];
    const float A = pA[0], Cb = pC[0];
    const int PS1 = 2 + 3 * K;
    const int64_t PS = (int64_t)PS1 * s.n_half;
    const bool staged = s.n_half == 1;
    float acc_A = 0.f, acc_C = 0.f;
    for (int64_t t0 = (int64_t)blockIdx.x * blockDim.x; t0 < total; t0 += (int64_t)gridDim.x * blockDim.x) {
        if (staged) nf_stage_rows_in(prm, tile, t0, total, PS1);
        const int64_t t = t0 + threadIdx.x;
        if (t < total) {
            const int64_t b = t / s.n_half;
            const int e = (int)(t - b * s.n_half);
            const int64_t fb = b * s.n_full;
            float* rowp = tile + threadIdx.x * (PS1 + 1);
            const int o0 = nf_half_to_full(s, 0, e), o1 = nf_half_to_full(s, 1, e);
            // the gradient row overwrites this thread's own parameter row in the tile (every parameter is in registers before the first store)
            gz[fb + o0] = nf_mr_bwd_elem<KT>(staged ? rowp : prm + b * PS + e, s.n_half, staged ? rowp : gprm + b * PS + e, s.n_half, K,
                                              z[fb + o0], gy[fb + o0], gld[b], A, Cb, eps, acc_A, acc_C);
            gz[fb + o1] = gy[fb + o1];
        }
        if (staged) nf_stage_rows_out(gprm, tile, t0, total, PS1);
    }
    const float ta = nf_block_sum(acc_A, scratch);
    const float tc = nf_block_sum(acc_C, scratch);
    if (threadIdx.x == 0) {
        if (partials != nullptr) {         // the caller folds them (nf_slab_sum): same-address float atomics retire at ~10 ns apiece, 384
            partials[blockIdx.x] = ta;     // workgroups of a CIFAR-shape launch spent ~4 of their 18.7 us queueing for two addresses
            partials[gridDim.x + blockIdx.x] = tc;
        } else {
            NF_DET_ADD2(nf_ml, g_scale, ta, g_bias, tc);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// inverse.  phase 1: affine^-1, sigmoid, 25 bisection steps; the "stuck" flag is raised when some bracket is still
// >= 1e-4 (the reference then runs all 100 iterations for the whole batch, modules.py:205);
// phase 2 (same launch shape): 75 more steps iff the flag is set, then x = mid, ld -= log pdf(x).
template <int KT, int PHASE>
__global__ void __launch_bounds__(NF_BLOCK) k_mixlog_inv(const float* __restrict__ yin, const float* __restrict__ prm, const float* __restrict__ pA,
                             const float* __restrict__ pC, float* __restrict__ y, float* __restrict__ ld,
                             float* __restrict__ lohi, int* __restrict__ flag, NfSplit s, int K, int64_t total) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int PS1 = 2 + 3 * K;
    const int64_t PS = (int64_t)PS1 * s.n_half;
    const bool staged = s.n_half == 1;
    const int steps = PHASE == 1 ? 25 : (flag[0] ? 75 : 0);
    bool stuck = false;
    // image data adds per-wave log-det sums into ld[b] by atomics: in deterministic mode the workgroup holds the turn over its whole walk
    // and its waves (and, in a wave that straddles two samples, its lanes) add one after the other
    const bool det_img = s.n_half != 1 && nf_det_on(nf_ml_det);
    if (det_img) {
        if (threadIdx.x == 0) nf_det_wait(nf_ml_det);
        __syncthreads();
    }
    for (int64_t t0 = (int64_t)blockIdx.x * blockDim.x; t0 < total; t0 += (int64_t)gridDim.x * blockDim.x) {
        if (staged) nf_stage_rows_in(prm, tile, t0, total, PS1);
        const int64_t t = t0 + threadIdx.x;
        const int64_t b = (t < total ? t : total - 1) / s.n_half;
        float dld = 0.f;                                   // this element's log-det term (slab data: summed per wave below)
        if (t < total) {
            const int e = (int)(t - b * s.n_half);
            const int64_t fb = b * s.n_full;
            NfMix<KT> m;
            nf_mix_load<KT>(staged ? tile + threadIdx.x * (PS1 + 1) : prm + b * PS + e, s.n_half, K, m);
            float pi[KT];
#pragma unroll
            for (int k = 0; k < KT; ++k) pi[k] = expf(m.lp[k]);
            float lo, hi, target;
            if (PHASE == 1) {
                const float a = tanhf(m.a_raw) * pA[0] + pC[0];
                const float v = expf(-a) * (yin[fb + nf_half_to_full(s, 0, e)] - m.b);              // coupling.py:204
                target = 1.f / (1.f + expf(-v));                                                    // modules.py:155
                const float dl = -a + (v - 2.f * nf_softplus(v));                                   // coupling.py:205, modules.py:153
                if (s.n_half == 1) ld[b] += dl;
                else dld = dl;
                lo = -1.0e3f;                                                                       // modules.py:197-198
                hi = 1.0e3f;
            } else {
                lo = lohi[t];
                hi = lohi[total + t];
                target = lohi[2 * total + t];
            }
            for (int it = 0; it < steps; ++it) {
                const float mid = (lo + hi) * 0.5f;
                const float val = nf_mix_cdf<KT>(m, pi, mid);
                if (PHASE == 2 && (mid == lo || mid == hi || val == target)) break;                 // collapsed: the rest are no-ops
                lo = val < target ? mid : lo;                                                       // modules.py:202-203
                hi = val > target ? mid : hi;
            }
            if (PHASE == 1) {
                lohi[t] = lo;
                lohi[total + t] = hi;
                lohi[2 * total + t] = target;
                stuck |= !(fabsf(hi - lo) < 1.0e-4f);                                               // modules.py:205
            } else {
                const float x = (lo + hi) * 0.5f;                                                   // modules.py:208
                float lcdf, lpdf;
                nf_mix_eval<KT>(m, x, lcdf, lpdf);
                if (s.n_half == 1) ld[b] -= lpdf;                                                   // modules.py:209-212
                else dld = -lpdf;
                const int o1 = nf_half_to_full(s, 1, e);
                y[fb + nf_half_to_full(s, 0, e)] = x;
                y[fb + o1] = yin[fb + o1];
            }
        }
        if (s.n_half != 1) {
            // image data: a sample's n_half elements all add into ld[b].  One atomic per element was 98 k atomics on 64 addresses per
            // launch (650 us of a 16 ms sampling pass of the CIFAR-shape Flow++); a wave whose lanes share the sample adds ONE sum.
            const int64_t bw = __shfl(b, 0, NF_WAVE);
            const bool one = __all(b == bw);
            float v = dld;
            if (one) {
#pragma unroll
                for (int off = NF_WAVE / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, NF_WAVE);
            }
            if (!det_img) {
                if (one) {
                    if ((threadIdx.x & (NF_WAVE - 1)) == 0) atomicAdd(ld + bw, v);
                } else if (t < total) atomicAdd(ld + b, dld);
            } else {
                for (int w_ = 0; w_ < (int)(blockDim.x >> 6); ++w_) {
                    if ((int)(threadIdx.x >> 6) == w_) {
                        if (one) {
                            if ((threadIdx.x & (NF_WAVE - 1)) == 0) atomicAdd(ld + bw, v);
                        } else {
                            for (int l_ = 0; l_ < NF_WAVE; ++l_)
                                if ((int)(threadIdx.x & (NF_WAVE - 1)) == l_ && t < total) { atomicAdd(ld + b, dld); __threadfence(); }
                        }
                        __threadfence();
                    }
                    __syncthreads();
                }
            }
        }
        if (staged) __syncthreads();
    }
    if (det_img) {
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) nf_det_pass(nf_ml_det);
    }
    if (PHASE == 1 && __any(stuck) && (threadIdx.x & (NF_WAVE - 1)) == 0) atomicOr(flag, 1);
}

// ---------------------------------------------------------------------------------------------------------------
static inline bool nf_mixlog_args(NfSplit& s, int mode, int odd, int C, int H, int W, int K) {
    return nf_make_split(s, mode, odd, C, H, W) && mode != NF_SPLIT_NONE && K >= 1 && K <= 32;
}
// ---------------------------------------------------------------------------------------------------------------
// K <= 8 on vector data (rows mode): ONE MIXTURE COMPONENT PER LANE, eight lanes ("octet") share an element.
// At the reference's Flow++ density batch (65 536 elements) one thread per element is one wave per SIMD walking a
// ~3000-instruction latency chain (13 us forward, 22 us backward, nothing to overlap with).  Split by component the
// chain is ~8x shorter and there are 8x the waves; the log-sum-exp / softmax reductions over the components are three
// DPP steps each (quad_perm xor 1, xor 2, row_half_mirror) -- VALU-speed, no LDS.
// ---------------------------------------------------------------------------------------------------------------
// POST: the ActNorm of the NEXT flow step (flows/flowpp.py:60-66 alternates ActNorm and coupling on density data) is applied
// to the coupling's output before it is stored -- h = (y - bias) / exp(log_scale), ld -= sum log_scale (modules.py:246-249) --
// so the pair costs one pass over the rows instead of two.  n_half == 1 (two features) only.
template <bool POST>
__global__ void __launch_bounds__(NF_BLOCK) k_mixlog_oct_fwd(const float* __restrict__ z, const float* __restrict__ prm,
                                                             const float* __restrict__ pA, const float* __restrict__ pC,
                                                             const float* __restrict__ nls, const float* __restrict__ nb,
                                                             float* __restrict__ y, float* __restrict__ ld, NfSplit s, int K,
                                                             float eps, int64_t B) {
    const float A = pA[0], Cb = pC[0];
    const int kk = threadIdx.x & 7;
    const int64_t PS = (int64_t)(2 + 3 * K) * s.n_half;
    const int64_t per = (int64_t)gridDim.x * (blockDim.x >> 3);
    float D0 = 1.f, D1 = 1.f, S0 = 0.f, S1 = 0.f, ldn = 0.f;             // next ActNorm on features o0 (transformed), o1
    if (POST) {
        const int o0 = nf_half_to_full(s, 0, 0), o1 = nf_half_to_full(s, 1, 0);
        D0 = expf(nls[o0]); D1 = expf(nls[o1]); S0 = nb[o0]; S1 = nb[o1];
        ldn = -(nls[0] + nls[1]);
    }
    for (int64_t b0 = (int64_t)blockIdx.x * (blockDim.x >> 3); b0 < B; b0 += per) {      // uniform trip count per wave (DPP)
        const int64_t b = b0 + (threadIdx.x >> 3);
        const bool live = b < B;
        const int64_t bb = live ? b : B - 1;
        const float* zb = z + bb * s.n_full;
        float* yb = y + bb * s.n_full;
        float acc = ldn;
        for (int e = 0; e < s.n_half; ++e) {
            NfOct m;
            nf_oct_load(prm + bb * PS + e, s.n_half, K, kk, m);
            const int o0 = nf_half_to_full(s, 0, e), o1 = nf_half_to_full(s, 1, e);
            const float x = zb[o0];
            float lcdf, lpdf, u, l;
            nf_oct_eval(m, x, lcdf, lpdf, u, l);
            const float F = nf_fexp(lcdf);                                   // modules.py:194
            const float xc = fminf(fmaxf(F, eps), 1.f - eps);             // modules.py:147
            const float la = nf_flog(xc), lb = nf_flog(1.f - xc);               // logit and its log-det share the two logs
            const float a = nf_ftanh(m.a_raw) * A + Cb;                      // coupling.py:178
            acc += lpdf - (la + lb) + a;                                  // coupling.py:184-188
            if (live && kk == 0) {
                const float yt = (la - lb) * nf_fexp(a) + m.b;               // coupling.py:187
                yb[o0] = POST ? (yt - S0) / D0 : yt;
                yb[o1] = POST ? (zb[o1] - S1) / D1 : zb[o1];
            }
        }
        if (live && kk == 0) ld[b] += acc;
    }
}

// The inverse with one mixture component per lane (density data, K <= 8; see k_mixlog_oct_fwd): the bisection's mixture CDF is one
// sigmoid per lane and three DPP adds (the same sum on all eight lanes: the butterfly adds commute), eight times the waves of the
// element-per-thread form in flight.  PHASE 1 runs the 25 steps every batch needs and WRITES THE RESULT -- x = mid, ld += the
// affine / sigmoid terms - log pdf(x) -- and raises the flag when some bracket is still >= 1e-4; PHASE 2 returns at once unless
// the flag is set (the reference then runs all 100 iterations for the whole batch, modules.py:205): it takes the saved brackets
// 75 steps further and replaces x and the log-pdf term.
template <int PHASE>
__global__ void __launch_bounds__(NF_BLOCK) k_mixlog_oct_inv(const float* __restrict__ yin, const float* __restrict__ prm,
                                                             const float* __restrict__ pA, const float* __restrict__ pC,
                                                             float* __restrict__ y, float* __restrict__ ld, float* __restrict__ lohi,
                                                             int* __restrict__ flag, NfSplit s, int K, int64_t B) {
    if (PHASE == 2 && flag[0] == 0) return;
    const float A = pA[0], Cb = pC[0];
    const int kk = threadIdx.x & 7;
    const int64_t PS = (int64_t)(2 + 3 * K) * s.n_half;
    const int64_t per = (int64_t)gridDim.x * (blockDim.x >> 3);
    const int64_t total = B * s.n_half;
    bool stuck = false;
    for (int64_t b0 = (int64_t)blockIdx.x * (blockDim.x >> 3); b0 < B; b0 += per) {      // uniform trip count per wave (DPP)
        const int64_t b = b0 + (threadIdx.x >> 3);
        const bool live = b < B;
        const int64_t bb = live ? b : B - 1;
        const float* zb = yin + bb * s.n_full;
        float* yb = y + bb * s.n_full;
        float acc = 0.f;
        for (int e = 0; e < s.n_half; ++e) {
            NfOct m;
            nf_oct_load(prm + bb * PS + e, s.n_half, K, kk, m);
            const float pik = nf_fexp(m.lp);                                 // 0 for the padding lanes (lp = -inf)
            const int o0 = nf_half_to_full(s, 0, e), o1 = nf_half_to_full(s, 1, e);
            const int64_t t = bb * s.n_half + e;
            float lo, hi, target, lpdf_old = 0.f;
            if (PHASE == 1) {
                const float a = tanhf(m.a_raw) * A + Cb;
                const float v = expf(-a) * (zb[o0] - m.b);                    // coupling.py:204
                target = 1.f / (1.f + expf(-v));                             // modules.py:155
                acc += -a + (v - 2.f * nf_softplus(v));                      // coupling.py:205, modules.py:153
                lo = -1.0e3f;                                                // modules.py:197-198
                hi = 1.0e3f;
            } else {
                lo = lohi[t];
                hi = lohi[total + t];
                target = lohi[2 * total + t];
                float lcdf, u, l;
                nf_oct_eval(m, (lo + hi) * 0.5f, lcdf, lpdf_old, u, l);      // what phase 1 subtracted
            }
            for (int it = 0; it < (PHASE == 1 ? 25 : 75); ++it) {
                const float mid = (lo + hi) * 0.5f;
                const float val = nf_oct_sum(pik * __builtin_amdgcn_rcpf(1.f + nf_fexp(-(mid - m.mu) * m.es)));
                // a bracket that has collapsed to neighbouring floats (mid is one of its ends) or sits on val == target cannot
                // change any more: once that holds for the whole wave the remaining iterations are no-ops
                if (PHASE == 2 && __all(mid == lo || mid == hi || val == target)) break;
                lo = val < target ? mid : lo;                                // modules.py:202-203
                hi = val > target ? mid : hi;
            }
            const float x = (lo + hi) * 0.5f;                                // modules.py:208
            float lcdf, lpdf, u, l;
            nf_oct_eval(m, x, lcdf, lpdf, u, l);
            acc += PHASE == 1 ? -lpdf : lpdf_old - lpdf;                     // modules.py:209-212
            if (PHASE == 1) stuck |= live && !(fabsf(hi - lo) < 1.0e-4f);    // modules.py:205
            if (live && kk == 0) {
                if (PHASE == 1) {
                    lohi[t] = lo;
                    lohi[total + t] = hi;
                    lohi[2 * total + t] = target;
                    yb[o1] = zb[o1];
                }
                yb[o0] = x;
            }
        }
        if (live && kk == 0) ld[b] += acc;
    }
    if (PHASE == 1 && __any(stuck) && (threadIdx.x & (NF_WAVE - 1)) == 0) atomicOr(flag, 1);
}

// 1024-thread workgroups: every workgroup ends in one pair of same-address atomics, which the L2 serialises at about 20 ns
// each (512 workgroups of 256 threads spent 10 of their 23 us there); 128 x 16 waves keeps the same number of waves in flight
#define NF_OCT_BWD_THREADS 1024
#define NF_OCT_BWD_MAX_BLOCKS 128
// POST: g_y / g_ld arrive for the NEXT step's ActNorm output h; its parameter gradients (g_log_scale_c = -sum g_h h - sum g_ld,
// g_bias_c = -sum g_h / exp(log_scale_c), appendix B2) are reduced here from the recomputed coupling output and the coupling
// continues with g_y = g_h / exp(log_scale).  Seven block totals -> one cross-wave reduction, seven atomics per workgroup.
template <bool POST>
__global__ void __launch_bounds__(NF_OCT_BWD_THREADS) k_mixlog_oct_bwd(const float* __restrict__ gy, const float* __restrict__ gld,
                                                             const float* __restrict__ z, const float* __restrict__ prm,
                                                             const float* __restrict__ pA, const float* __restrict__ pC,
                                                             const float* __restrict__ nls, const float* __restrict__ nb,
                                                             float* __restrict__ gz, float* __restrict__ gprm,
                                                             float* __restrict__ g_scale, float* __restrict__ g_bias,
                                                             float* __restrict__ g_nls, float* __restrict__ g_nb, NfSplit s,
                                                             int K, float eps, int64_t B) {
    __shared__ float scratch[NF_OCT_BWD_THREADS / NF_WAVE][8];
    const float A = pA[0], Cb = pC[0];
    const int kk = threadIdx.x & 7;
    const bool on = kk < K;
    const int64_t PS = (int64_t)(2 + 3 * K) * s.n_half;
    const int64_t per = (int64_t)gridDim.x * (blockDim.x >> 3);
    float D0 = 1.f, D1 = 1.f, S0 = 0.f, S1 = 0.f;
    if (POST) {
        const int o0 = nf_half_to_full(s, 0, 0), o1 = nf_half_to_full(s, 1, 0);
        D0 = expf(nls[o0]); D1 = expf(nls[o1]); S0 = nb[o0]; S1 = nb[o1];
    }
    float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // g_a*th, g_a | sum g_h0, sum g_h0 h0, sum g_h1, sum g_h1 h1, sum g_ld
    for (int64_t b0 = (int64_t)blockIdx.x * (blockDim.x >> 3); b0 < B; b0 += per) {
        const int64_t b = b0 + (threadIdx.x >> 3);
        const bool live = b < B;
        const int64_t bb = live ? b : B - 1;
        const int64_t fb = bb * s.n_full;
        const float g_ld = gld[bb];
        for (int e = 0; e < s.n_half; ++e) {
            NfOct m;
            nf_oct_load(prm + bb * PS + e, s.n_half, K, kk, m);
            float* GP = gprm + bb * PS + e;
            const int64_t gnh = s.n_half;
            const int o0 = nf_half_to_full(s, 0, e), o1 = nf_half_to_full(s, 1, e);
            const float x = z[fb + o0];
            const float g_h0 = gy[fb + o0];
            const float g_y = POST ? g_h0 / D0 : g_h0;
            float lcdf, lpdf, u, l;
            nf_oct_eval(m, x, lcdf, lpdf, u, l);
            const float F = nf_fexp(lcdf), f = nf_fexp(lpdf);
            const bool inside = (F >= eps) && (F <= 1.f - eps);          // torch.clamp passes the gradient on [min, max]
            const float xc = fminf(fmaxf(F, eps), 1.f - eps);
            const float y1 = nf_flog(xc) - nf_flog(1.f - xc);
            const float th = nf_ftanh(m.a_raw);
            const float ea = nf_fexp(th * A + Cb);
            const float g_y1 = g_y * ea;                                 // y = y1 * exp(a) + b ; ld += a
            const float g_a = g_y * y1 * ea + g_ld;
            const float gF = inside ? (g_y1 - g_ld * (1.f - 2.f * xc)) / (xc * (1.f - xc)) : 0.f;   // logit + its log-det
            const float tot = gF * F + g_ld;                             // sum_j g_logpi_j
            // this lane's component (appendix B6, responsibilities form r_k = pi_k pdf_k / f)
            const float r = on ? nf_fexp(m.lp + (u - m.s - 2.f * (fmaxf(u, 0.f) + l)) - lpdf) : 0.f;
            const float omt = -nf_ftanh(0.5f * u);                          // 1 - 2 sigmoid(u)
            const float w = g_ld * r * omt * m.es;
            const float gx = gF * f + nf_oct_sum(w);
            if (live) {
                if (on) {
                    GP[(2 + K + kk) * gnh] = -gF * f * r - w;                                          // g_mu_k
                    GP[(2 + 2 * K + kk) * gnh] = -gF * f * r * (x - m.mu) + g_ld * r * (-omt * u - 1.f);   // g_s_k
                    const float g_logpi = gF * nf_fexp(m.lp + (fminf(u, 0.f) - l)) + g_ld * r;
                    GP[(2 + kk) * gnh] = g_logpi - nf_fexp(m.lp) * tot;                                    // through log_softmax
                }
                if (kk == 0) {
                    const float zi = z[fb + o1], g_h1 = gy[fb + o1];
                    GP[0] = g_a * A * (1.f - th * th);
                    GP[gnh] = g_y;
                    gz[fb + o0] = gx;
                    gz[fb + o1] = POST ? g_h1 / D1 : g_h1;
                    acc[0] += g_a * th;
                    acc[1] += g_a;
                    if (POST) {
                        const float h0 = (y1 * ea + m.b - S0) / D0, h1 = (zi - S1) / D1;
                        acc[2] += g_h0;
                        acc[3] += g_h0 * h0;
                        acc[4] += g_h1;
                        acc[5] += g_h1 * h1;
                        acc[6] += g_ld;
                    }
                }
            }
        }
    }
    const int NA = POST ? 7 : 2;
    const int lane = threadIdx.x & (NF_WAVE - 1), wid = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 7; ++q)
        if (q < NA) {
            const float v = nf_wave_sum(acc[q]);
            if (lane == 0) scratch[wid][q] = v;
        }
    __syncthreads();
    if ((int)threadIdx.x < NA) {
        float t = 0.f;
        const int nw = blockDim.x >> 6;
        for (int w = 0; w < nw; ++w) t += scratch[w][threadIdx.x];
        scratch[0][threadIdx.x] = t;          // only rows read by the same thread were summed: no hazard on row 0
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        NF_DET_ENTER(nf_ml);
        atomicAdd(g_scale, scratch[0][0]);
        atomicAdd(g_bias, scratch[0][1]);
        if (POST) {
            const int o0 = nf_half_to_full(s, 0, 0), o1 = nf_half_to_full(s, 1, 0);
            const float t0 = scratch[0][2], t0h = scratch[0][3], t1 = scratch[0][4], t1h = scratch[0][5], sg = scratch[0][6];
            atomicAdd(g_nls + o0, -t0h - sg);
            atomicAdd(g_nls + o1, -t1h - sg);
            atomicAdd(g_nb + o0, -t0 / D0);
            atomicAdd(g_nb + o1, -t1 / D1);
        }
        NF_DET_LEAVE(nf_ml);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// LARGE BATCHES of density data (two features, K <= 8): ONE ROW PER THREAD, shared transcendentals (round 6).
// The octet kernels above are latency designs for the reference's batch (65 536 rows: one element per eight lanes shortens the chain);
// every lane of an octet repeats the element's scalar work, and the log-space form spends ~15 transcendentals per LANE (120 per row):
// at 2 M rows they ran at 10 % of the transcendental rate and 28 / 18 / 8 % of HBM (profiles/r05_kernel_sweep.txt).  With enough rows for
// >= 4 waves per SIMD of one row per thread the chain length does not matter and the work per row does:
//   * per component only   w_k = exp(lp_k - max lp)   es_k = exp(-s_k)   t_k = exp(-|u_k|)   r_k = 1 / (1 + t_k)   are transcendental;
//     sigma(u_k) = (u_k >= 0 ? 1 : t_k) r_k,  sigma (1 - sigma) = t_k r_k^2,  so
//         F = sum_k w_k sigma_k / sum_k w_k          f = sum_k w_k es_k t_k r_k^2 / sum_k w_k
//     and every gradient term of the backward is a product of these (4 K + 8 transcendental-rate instructions per row, not ~25 K);
//   * F is what the reference forms anyway (modules.py:194: exp(log cdf)); log f is needed to full RELATIVE accuracy where f underflows --
//     a row whose linear-space f falls below NF_MR_TINY takes the log-space path of the element kernels (nf_mix_eval / nf_mixlog_bwd_elem);
//   * a workgroup's 256 parameter rows are staged through LDS with 16-byte coalesced loads (row stride 2 + 3 K + 1 words: conflict-free
//     per-thread reads), the backward writes its 2 + 3 K gradients over its own row and the tile leaves with 16-byte stores.
#define NF_MR_ROWS 256
#define NF_MR_MIN_ROWS (int64_t)(4 * 256 * 256)       // >= 4 waves per SIMD of one row per thread; below: the octet kernels

// flat element e of a tile of rows (PS1 words each) -> LDS word e + e / PS1 (one pad word per row); magic = ceil(2^20 / PS1): exact for
// e < 2^13 and PS1 <= 26 (the quotient's error e / 2^20 < 1 / PS1), and e * magic < 2^31
__device__ __forceinline__ int nf_mr_lds_index(int e, int magic) { return e + (int)(((unsigned)e * (unsigned)magic) >> 20); }

// rows [row0, row0 + NF_MR_ROWS) of a (B, PS1) tensor <-> the LDS tile; 16-byte global accesses when the tile is whole and aligned
template <bool OUT>
__device__ __forceinline__ void nf_mr_stage(float* __restrict__ g, float* __restrict__ tile, int64_t row0, int64_t B, int PS1, int magic) {
    const int rows = (int)min((int64_t)NF_MR_ROWS, B - row0);
    const int total = rows * PS1;
    float* src = g + row0 * PS1;
    const bool vec = (total & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
    if (vec) {
        for (int q = threadIdx.x; q < (total >> 2); q += NF_MR_ROWS) {
            const int e = 4 * q;
            if (!OUT) {
                const float4 v = reinterpret_cast<const float4*>(src)[q];
                tile[nf_mr_lds_index(e, magic)] = v.x; tile[nf_mr_lds_index(e + 1, magic)] = v.y;
                tile[nf_mr_lds_index(e + 2, magic)] = v.z; tile[nf_mr_lds_index(e + 3, magic)] = v.w;
            } else {
                float4 v;
                v.x = tile[nf_mr_lds_index(e, magic)]; v.y = tile[nf_mr_lds_index(e + 1, magic)];
                v.z = tile[nf_mr_lds_index(e + 2, magic)]; v.w = tile[nf_mr_lds_index(e + 3, magic)];
                reinterpret_cast<float4*>(src)[q] = v;
            }
        }
    } else {
        for (int e = threadIdx.x; e < total; e += NF_MR_ROWS) {
            if (!OUT) tile[nf_mr_lds_index(e, magic)] = src[e];
            else src[e] = tile[nf_mr_lds_index(e, magic)];
        }
    }
}

template <int KT>
__global__ void __launch_bounds__(NF_MR_ROWS) k_mixlog_row_fwd(const float* __restrict__ z, const float* __restrict__ prm, const float* __restrict__ pA,
                                                               const float* __restrict__ pC, float* __restrict__ y, float* __restrict__ ld, int odd,
                                                               int K, int magic, float eps, int64_t B) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const float A = pA[0], Cb = pC[0];
    const int PS1 = 2 + 3 * K;
    for (int64_t row0 = (int64_t)blockIdx.x * NF_MR_ROWS; row0 < B; row0 += (int64_t)gridDim.x * NF_MR_ROWS) {
        nf_mr_stage<false>(const_cast<float*>(prm), tile, row0, B, PS1, magic);
        __syncthreads();
        const int64_t b = row0 + threadIdx.x;
        if (b < B) {
            const float2 zz = reinterpret_cast<const float2*>(z)[b];
            const float x = odd ? zz.y : zz.x, pass = odd ? zz.x : zz.y;      // squeeze1d: the transformed feature is feature `odd`
            float acc = 0.f;
            const float yt = nf_mr_fwd_elem<KT>(tile + threadIdx.x * (PS1 + 1), 1, K, x, A, Cb, eps, acc);
            reinterpret_cast<float2*>(y)[b] = odd ? make_float2(pass, yt) : make_float2(yt, pass);
            ld[b] += acc;
        }
        __syncthreads();
    }
}

template <int KT>
__global__ void __launch_bounds__(NF_MR_ROWS) k_mixlog_row_bwd(const float* __restrict__ gy, const float* __restrict__ gld, const float* __restrict__ z,
                                                               const float* __restrict__ prm, const float* __restrict__ pA, const float* __restrict__ pC,
                                                               float* __restrict__ gz, float* __restrict__ gprm, float* __restrict__ g_scale,
                                                               float* __restrict__ g_bias, int odd, int K, int magic, float eps, int64_t B) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    __shared__ float scratch[NF_MR_ROWS / NF_WAVE];
    const float A = pA[0], Cb = pC[0];
    const int PS1 = 2 + 3 * K;
    float acc_A = 0.f, acc_C = 0.f;
    for (int64_t row0 = (int64_t)blockIdx.x * NF_MR_ROWS; row0 < B; row0 += (int64_t)gridDim.x * NF_MR_ROWS) {
        nf_mr_stage<false>(const_cast<float*>(prm), tile, row0, B, PS1, magic);
        __syncthreads();
        const int64_t b = row0 + threadIdx.x;
        if (b < B) {
            float* GP = tile + threadIdx.x * (PS1 + 1);                        // the gradient row overwrites this thread's own parameter row
            const float2 zz = reinterpret_cast<const float2*>(z)[b], gg = reinterpret_cast<const float2*>(gy)[b];
            const float x = odd ? zz.y : zz.x, g_y = odd ? gg.y : gg.x, g_pass = odd ? gg.x : gg.y;
            const float gx = nf_mr_bwd_elem<KT>(GP, 1, GP, 1, K, x, g_y, gld[b], A, Cb, eps, acc_A, acc_C);
            reinterpret_cast<float2*>(gz)[b] = odd ? make_float2(g_pass, gx) : make_float2(gx, g_pass);
        }
        __syncthreads();
        nf_mr_stage<true>(gprm, tile, row0, B, PS1, magic);
        __syncthreads();
    }
    const float ta = nf_block_sum(acc_A, scratch);
    const float tc = nf_block_sum(acc_C, scratch);
    if (threadIdx.x == 0) {
        NF_DET_ADD2(nf_ml, g_scale, ta, g_bias, tc);
    }
}

// inverse: phase 1 = affine^-1, sigmoid, 25 bisection steps, x = mid, ld += the affine / sigmoid terms - log pdf(x), brackets saved and the
// flag raised while some bracket is still >= 1e-4; phase 2 (returns at once unless the flag is set) takes them 75 steps further
template <int KT, int PHASE>
__global__ void __launch_bounds__(NF_MR_ROWS) k_mixlog_row_inv(const float* __restrict__ yin, const float* __restrict__ prm, const float* __restrict__ pA,
                                                               const float* __restrict__ pC, float* __restrict__ y, float* __restrict__ ld,
                                                               float* __restrict__ lohi, int* __restrict__ flag, int odd, int K, int magic, int64_t B) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    if (PHASE == 2 && flag[0] == 0) return;
    const float A = pA[0], Cb = pC[0];
    const int PS1 = 2 + 3 * K;
    bool stuck = false;
    for (int64_t row0 = (int64_t)blockIdx.x * NF_MR_ROWS; row0 < B; row0 += (int64_t)gridDim.x * NF_MR_ROWS) {
        nf_mr_stage<false>(const_cast<float*>(prm), tile, row0, B, PS1, magic);
        __syncthreads();
        const int64_t b = row0 + threadIdx.x;
        if (b < B) {
            NfMixLin<KT> m;
            float lp[KT], sv[KT];
            nf_mr_load<KT>(tile + threadIdx.x * (PS1 + 1), 1, K, m, lp, sv);
            float pi[KT];
#pragma unroll
            for (int k = 0; k < KT; ++k) pi[k] = m.w[k] * m.inv_se;
            NfMix<KT> q;
            nf_mr_logspace<KT>(m, lp, sv, q);
            const float2 zz = reinterpret_cast<const float2*>(yin)[b];
            float lo, hi, target, acc = 0.f, lpdf_old = 0.f;
            if (PHASE == 1) {
                const float a = tanhf(m.a_raw) * A + Cb;
                const float v = expf(-a) * ((odd ? zz.y : zz.x) - m.b);        // coupling.py:204
                target = 1.f / (1.f + expf(-v));                               // modules.py:155
                acc = -a + (v - 2.f * nf_softplus(v));                         // coupling.py:205, modules.py:153
                lo = -1.0e3f;                                                  // modules.py:197-198
                hi = 1.0e3f;
            } else {
                lo = lohi[b]; hi = lohi[B + b]; target = lohi[2 * B + b];
                float lcdf;
                nf_mix_eval<KT>(q, (lo + hi) * 0.5f, lcdf, lpdf_old);          // what phase 1 subtracted
            }
            // sigma((x - mu) es) = 1 / (1 + 2^((mu - x) es log2 e)): the scale is folded into es once, the loop is sub, mul, v_exp, add, v_rcp, fma
            float e2[KT];
#pragma unroll
            for (int k = 0; k < KT; ++k) e2[k] = m.es[k] * 1.4426950408889634f;
            for (int it = 0; it < (PHASE == 1 ? 25 : 75); ++it) {
                const float mid = (lo + hi) * 0.5f;
                float val = 0.f;
#pragma unroll
                for (int k = 0; k < KT; ++k) val = fmaf(pi[k], __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f((m.mu[k] - mid) * e2[k])), val);
                if (PHASE == 2 && (mid == lo || mid == hi || val == target)) break;   // collapsed: the rest are no-ops
                lo = val < target ? mid : lo;                                  // modules.py:202-203
                hi = val > target ? mid : hi;
            }
            const float x = (lo + hi) * 0.5f;                                  // modules.py:208
            float lcdf, lpdf;
            nf_mix_eval<KT>(q, x, lcdf, lpdf);
            acc += PHASE == 1 ? -lpdf : lpdf_old - lpdf;                       // modules.py:209-212
            if (PHASE == 1) {
                stuck |= !(fabsf(hi - lo) < 1.0e-4f);                          // modules.py:205
                lohi[b] = lo; lohi[B + b] = hi; lohi[2 * B + b] = target;
            }
            const float pass = odd ? zz.x : zz.y;
            if (PHASE == 1) reinterpret_cast<float2*>(y)[b] = odd ? make_float2(pass, x) : make_float2(x, pass);
            else y[2 * b + (odd ? 1 : 0)] = x;
            ld[b] += acc;
        }
        __syncthreads();
    }
    if (PHASE == 1 && __any(stuck) && (threadIdx.x & (NF_WAVE - 1)) == 0) atomicOr(flag, 1);
}

// the row kernels take a launch: two features, K <= 8, enough rows to fill the machine with one row per thread
static int nf_mr_min_rows = -1;                          // (tests lower it through nf_mixlog_rows_config to run the row kernels on small batches)
extern "C" int nf_mixlog_rows_config(int64_t min_rows) {
    nf_mr_min_rows = min_rows < 0 ? -1 : (int)(min_rows > 0x7fffffff ? 0x7fffffff : min_rows);
    return 0;
}
static inline bool nf_mr_takes(const NfSplit& s, int K, int64_t B) {
    const int64_t lim = nf_mr_min_rows >= 0 ? (int64_t)nf_mr_min_rows : NF_MR_MIN_ROWS;
    return s.n_half == 1 && s.n_full == 2 && K >= 1 && K <= 8 && B >= lim;
}
static inline int nf_mr_magic(int K) { return ((1 << 20) + (2 + 3 * K) - 1) / (2 + 3 * K); }
static inline size_t nf_mr_lds(int K) { return (size_t)NF_MR_ROWS * (2 + 3 * K + 1) * sizeof(float); }
static inline unsigned nf_mr_grid(int64_t B, int64_t cap = 8192) {
    const int64_t g = (B + NF_MR_ROWS - 1) / NF_MR_ROWS;
    return (unsigned)(g > cap ? cap : g);
}
// the backward ends in two same-address float atomics per workgroup (coupling scale / shift), which the L2 retires one after the other:
// 8 192 workgroups would queue for ~100 us there; 1 024 persistent ones (four per compute unit) keep the tail at ~10 us
#define NF_MR_BWD_GRID 1024

static inline int nf_mx_threads(int K) { return (2 + 3 * K) <= 50 ? 256 : 128; }        // rows tile <= 52 KB of LDS
static inline size_t nf_mx_lds(const NfSplit& s, int K, int threads) {
    return s.n_half == 1 ? (size_t)threads * (2 + 3 * K + 1) * sizeof(float) : 0;
}
#define NF_MX_DISPATCH(K, CALL)                 \
    do {                                        \
        if ((K) <= 4) { CALL(4); }              \
        else if ((K) <= 8) { CALL(8); }         \
        else if ((K) <= 16) { CALL(16); }       \
        else { CALL(32); }                      \
    } while (0)

extern "C" int nf_mixlog_coupling_fwd(const float* z, const float* params, const float* a_log_scale,
                                      const float* a_bias, float* y, float* ld, int K, float logit_eps, int mode, int odd,
                                      int64_t B, int C, int H, int W, nf_stream_t stream) {
    NfSplit s;
    if (!nf_mixlog_args(s, mode, odd, C, H, W, K)) return NF_E_BADARG;
    if (B == 0 || s.n_half == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (nf_mr_takes(s, K, B)) {                            // large density batches: one row per thread, shared transcendentals
        const int od = s.odd;
        if (K <= 4) hipLaunchKernelGGL(k_mixlog_row_fwd<4>, dim3(nf_mr_grid(B)), dim3(NF_MR_ROWS), nf_mr_lds(K), st, z, params, a_log_scale, a_bias, y, ld,
                                       od, K, nf_mr_magic(K), logit_eps, B);
        else hipLaunchKernelGGL(k_mixlog_row_fwd<8>, dim3(nf_mr_grid(B)), dim3(NF_MR_ROWS), nf_mr_lds(K), st, z, params, a_log_scale, a_bias, y, ld, od, K,
                                nf_mr_magic(K), logit_eps, B);
    } else if (s.n_half <= NF_MX_ROWS_MAX && K <= 8) {     // one component per lane (see k_mixlog_oct_fwd)
        unsigned g = nf_grid_for(B * 8, NF_BLOCK);
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(k_mixlog_oct_fwd<false>, dim3(g), dim3(NF_BLOCK), 0, st, z, params, a_log_scale, a_bias, nullptr, nullptr, y, ld, s, K,
                           logit_eps, B);
    } else if (s.n_half <= NF_MX_ROWS_MAX) {
        const int th = nf_mx_threads(K);
        unsigned g = nf_grid_for(B, th);
        if (g > NF_MX_GRID) g = NF_MX_GRID;
#define CALL(KT) hipLaunchKernelGGL(k_mixlog_rows_fwd<KT>, dim3(g), dim3(th), nf_mx_lds(s, K, th), st, z, params, a_log_scale, a_bias, y, ld, s, K, logit_eps, B)
        NF_MX_DISPATCH(K, CALL);
#undef CALL
    } else {
        if (B > 0x7fffffffLL) return NF_E_BADARG;
        dim3 grid((unsigned)B, (unsigned)((s.n_half + NF_MX_SLAB - 1) / NF_MX_SLAB));
#define CALL(KT) hipLaunchKernelGGL(k_mixlog_slab_fwd<KT>, grid, dim3(NF_BLOCK), 0, st, z, params, a_log_scale, a_bias, y, ld, s, K, logit_eps)
        NF_MX_DISPATCH(K, CALL);
#undef CALL
    }
    NF_CHECK_LAUNCH();
    return 0;
}

// workgroups of the element-per-thread backward launch (image data), 0 where another kernel serves the shape
extern "C" int nf_mixlog_bwd_blocks(int K, int mode, int64_t B, int C, int H, int W) {
    NfSplit s;
    if (!nf_mixlog_args(s, mode, 0, C, H, W, K)) return 0;
    const int64_t total = B * s.n_half;
    if (total == 0 || (s.n_half <= NF_MX_ROWS_MAX && K <= 8)) return 0;
    unsigned g = nf_grid_for(total, nf_mx_threads(K));
    return (int)(g > NF_MX_GRID ? NF_MX_GRID : g);
}

static int nf_mixlog_bwd_launch(const float* g_y, const float* g_ld, const float* z, const float* params, const float* a_log_scale,
                                const float* a_bias, float* g_z, float* g_params, float* g_scale, float* g_bias, float* partials, int K,
                                float logit_eps, int mode, int odd, int64_t B, int C, int H, int W, nf_stream_t stream) {
    NfSplit s;
    if (!nf_mixlog_args(s, mode, odd, C, H, W, K)) return NF_E_BADARG;
    const int64_t total = B * s.n_half;
    if (total == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (partials == nullptr && nf_mr_takes(s, K, B)) {     // large density batches: one row per thread
        const int od = s.odd;
        if (K <= 4) hipLaunchKernelGGL(k_mixlog_row_bwd<4>, dim3(nf_mr_grid(B, NF_MR_BWD_GRID)), dim3(NF_MR_ROWS), nf_mr_lds(K), st, g_y, g_ld, z, params, a_log_scale,
                                       a_bias, g_z, g_params, g_scale, g_bias, od, K, nf_mr_magic(K), logit_eps, B);
        else hipLaunchKernelGGL(k_mixlog_row_bwd<8>, dim3(nf_mr_grid(B, NF_MR_BWD_GRID)), dim3(NF_MR_ROWS), nf_mr_lds(K), st, g_y, g_ld, z, params, a_log_scale, a_bias,
                                g_z, g_params, g_scale, g_bias, od, K, nf_mr_magic(K), logit_eps, B);
        NF_CHECK_LAUNCH();
        return 0;
    }
    if (s.n_half <= NF_MX_ROWS_MAX && K <= 8) {            // one component per lane
        if (partials != nullptr) return NF_E_UNSUPPORTED;
        unsigned g2 = nf_grid_for(B * 8, NF_OCT_BWD_THREADS);
        if (g2 > NF_OCT_BWD_MAX_BLOCKS) g2 = NF_OCT_BWD_MAX_BLOCKS;
        hipLaunchKernelGGL(k_mixlog_oct_bwd<false>, dim3(g2), dim3(NF_OCT_BWD_THREADS), 0, st, g_y, g_ld, z, params, a_log_scale,
                           a_bias, nullptr, nullptr, g_z, g_params, g_scale, g_bias, nullptr, nullptr, s, K, logit_eps, B);
        NF_CHECK_LAUNCH();
        return 0;
    }
    const int th = nf_mx_threads(K);
    unsigned g = nf_grid_for(total, th);
    if (g > NF_MX_GRID) g = NF_MX_GRID;
#define CALL(KT) hipLaunchKernelGGL(k_mixlog_bwd<KT>, dim3(g), dim3(th), nf_mx_lds(s, K, th), st, g_y, g_ld, z, params, a_log_scale, a_bias, g_z, g_params, g_scale, g_bias, s, K, logit_eps, total, partials)
    NF_MX_DISPATCH(K, CALL);
#undef CALL
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_mixlog_coupling_bwd(const float* g_y, const float* g_ld, const float* z, const float* params,
                                      const float* a_log_scale, const float* a_bias, float* g_z, float* g_params,
                                      float* g_scale, float* g_bias, int K, float logit_eps, int mode, int odd, int64_t B,
                                      int C, int H, int W, nf_stream_t stream) {
    return nf_mixlog_bwd_launch(g_y, g_ld, z, params, a_log_scale, a_bias, g_z, g_params, g_scale, g_bias, nullptr, K, logit_eps, mode, odd, B,
                                C, H, W, stream);
}

// the same with the gradients of the coupling's scale / shift LEFT as per-workgroup partial sums, partials[0 .. n) | partials[n .. 2 n),
// n = nf_mixlog_bwd_blocks(...) > 0: the caller folds them (nf_slab_sum) with whatever else it folds
extern "C" int nf_mixlog_coupling_bwd_partials(const float* g_y, const float* g_ld, const float* z, const float* params,
                                               const float* a_log_scale, const float* a_bias, float* g_z, float* g_params, float* partials,
                                               int K, float logit_eps, int mode, int odd, int64_t B, int C, int H, int W,
                                               nf_stream_t stream) {
    if (partials == nullptr || nf_mixlog_bwd_blocks(K, mode, B, C, H, W) == 0) return NF_E_BADARG;
    return nf_mixlog_bwd_launch(g_y, g_ld, z, params, a_log_scale, a_bias, g_z, g_params, nullptr, nullptr, partials, K, logit_eps, mode, odd, B,
                                C, H, W, stream);
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
    if (nf_mr_takes(s, K, B)) {                            // large density batches: one row per thread
        const int od = s.odd;
        const dim3 gr(nf_mr_grid(B));
        if (K <= 4) {
            hipLaunchKernelGGL((k_mixlog_row_inv<4, 1>), gr, dim3(NF_MR_ROWS), nf_mr_lds(K), st, z, params, a_log_scale, a_bias, y, ld, scratch, stuck_flag,
                               od, K, nf_mr_magic(K), B);
            hipLaunchKernelGGL((k_mixlog_row_inv<4, 2>), gr, dim3(NF_MR_ROWS), nf_mr_lds(K), st, z, params, a_log_scale, a_bias, y, ld, scratch, stuck_flag,
                               od, K, nf_mr_magic(K), B);
        } else {
            hipLaunchKernelGGL((k_mixlog_row_inv<8, 1>), gr, dim3(NF_MR_ROWS), nf_mr_lds(K), st, z, params, a_log_scale, a_bias, y, ld, scratch, stuck_flag,
                               od, K, nf_mr_magic(K), B);
            hipLaunchKernelGGL((k_mixlog_row_inv<8, 2>), gr, dim3(NF_MR_ROWS), nf_mr_lds(K), st, z, params, a_log_scale, a_bias, y, ld, scratch, stuck_flag,
                               od, K, nf_mr_magic(K), B);
        }
        NF_CHECK_LAUNCH();
        return 0;
    }
    if (s.n_half <= NF_MX_ROWS_MAX && K <= 8) {            // one component per lane (see k_mixlog_oct_inv)
        unsigned go = nf_grid_for(B * 8, NF_BLOCK);
        if (go > 2048) go = 2048;
        hipLaunchKernelGGL(k_mixlog_oct_inv<1>, dim3(go), dim3(NF_BLOCK), 0, st, z, params, a_log_scale, a_bias, y, ld, scratch, stuck_flag,
                           s, K, B);
        hipLaunchKernelGGL(k_mixlog_oct_inv<2>, dim3(go), dim3(NF_BLOCK), 0, st, z, params, a_log_scale, a_bias, y, ld, scratch, stuck_flag,
                           s, K, B);
        NF_CHECK_LAUNCH();
        return 0;
    }
    const int th = nf_mx_threads(K);
    const unsigned g = nf_grid_for(total, th);
    const size_t lds = nf_mx_lds(s, K, th);
#define CALL(KT)                                                                                                              \
    hipLaunchKernelGGL((k_mixlog_inv<KT, 1>), dim3(g), dim3(th), lds, st, z, params, a_log_scale, a_bias, y, ld, scratch, stuck_flag, s, K, total); \
    hipLaunchKernelGGL((k_mixlog_inv<KT, 2>), dim3(g), dim3(th), lds, st, z, params, a_log_scale, a_bias, y, ld, scratch, stuck_flag, s, K, total)
    NF_MX_DISPATCH(K, CALL);
#undef CALL
    NF_CHECK_LAUNCH();
    return 0;
}

// ---- Flow++ density step pair: mixture coupling on two features, then the next step's ActNorm (nfhip.h) -------------------
extern "C" int nf_flowpp_vec_couple_fwd(const float* z, const float* params, const float* a_log_scale, const float* a_bias,
                                        const float* next_log_scale, const float* next_bias, float* y, float* ld, int K,
                                        float logit_eps, int odd, int64_t B, nf_stream_t stream) {
    NfSplit s;
    if (!nf_mixlog_args(s, NF_SPLIT_1D, odd, 2, 1, 1, K) || K > 8 || next_log_scale == nullptr || next_bias == nullptr)
        return NF_E_BADARG;
    if (B == 0) return 0;
    unsigned g = nf_grid_for(B * 8, NF_BLOCK);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_mixlog_oct_fwd<true>, dim3(g), dim3(NF_BLOCK), 0, (hipStream_t)stream, z, params, a_log_scale, a_bias,
                       next_log_scale, next_bias, y, ld, s, K, logit_eps, B);
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_flowpp_vec_couple_bwd(const float* g_h, const float* g_ld, const float* z, const float* params,
                                        const float* a_log_scale, const float* a_bias, const float* next_log_scale,
                                        const float* next_bias, float* g_z, float* g_params, float* g_scale, float* g_bias,
                                        float* g_next_log_scale, float* g_next_bias, int K, float logit_eps, int odd, int64_t B,
                                        nf_stream_t stream) {
    NfSplit s;
    if (!nf_mixlog_args(s, NF_SPLIT_1D, odd, 2, 1, 1, K) || K > 8 || next_log_scale == nullptr || next_bias == nullptr ||
        g_next_log_scale == nullptr || g_next_bias == nullptr)
        return NF_E_BADARG;
    if (B == 0) return 0;
    unsigned g2 = nf_grid_for(B * 8, NF_OCT_BWD_THREADS);
    if (g2 > NF_OCT_BWD_MAX_BLOCKS) g2 = NF_OCT_BWD_MAX_BLOCKS;
    hipLaunchKernelGGL(k_mixlog_oct_bwd<true>, dim3(g2), dim3(NF_OCT_BWD_THREADS), 0, (hipStream_t)stream, g_h, g_ld, z, params,
                       a_log_scale, a_bias, next_log_scale, next_bias, g_z, g_params, g_scale, g_bias, g_next_log_scale,
                       g_next_bias, s, K, logit_eps, B);
    NF_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Standalone MixLogCDF (flows/modules.py:186-212): x (B, n), log_pi / mu / s (B, K, n) with log_pi ALREADY normalised by the
// caller (coupling.py:180) -- the module surface `forward / backward(x, log_pi, mu, s, log_df_dz)` of SURVEY.md section 8(b).
// One thread per element, the 3K parameters of the element in registers; per-sample log-det by one atomic per element
// (n == 1: a plain add).  The bisection compares exp(logsumexp(log pi + logsigmoid)) -- the reference's own expression
// (modules.py:201) -- with the target, so that the batch-global 25-or-100 rule sees the same kind of `val == x` ties.
template <int KT>
__device__ __forceinline__ void nf_cdf_load(const float* __restrict__ lp, const float* __restrict__ mu, const float* __restrict__ s,
                                            int64_t n, int K, NfMix<KT>& m) {
    m.a_raw = 0.f;
    m.b = 0.f;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        m.lp[k] = k < K ? lp[k * n] : -INFINITY;
        m.mu[k] = k < K ? mu[k * n] : 0.f;
        m.s[k] = k < K ? s[k * n] : 0.f;
        m.es[k] = expf(-m.s[k]);
    }
}

// the active lanes of ONE wave add in lane order (the wave walks the lane index in lockstep: one atomic in flight per iteration)
__device__ __forceinline__ void nf_lanes_in_turn_add(float* p, float v) {
    for (int l = 0; l < NF_WAVE; ++l)
        if ((int)(threadIdx.x & (NF_WAVE - 1)) == l) { atomicAdd(p, v); __threadfence(); }
}

template <int KT>
__global__ void __launch_bounds__(NF_BLOCK) k_mixlogcdf_fwd(const float* __restrict__ x, const float* __restrict__ lp, const float* __restrict__ mu,
                                                            const float* __restrict__ s, float* __restrict__ out, float* __restrict__ ld,
                                                            int64_t n, int K, int64_t total, int det) {
    // det (deterministic mode, n > 1): the launch is ONE wave and its lanes add one after the other -- element order
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / n, e = t - b * n, p = b * K * n + e;
        NfMix<KT> m;
        nf_cdf_load<KT>(lp + p, mu + p, s + p, n, K, m);
        float lcdf, lpdf;
        nf_mix_eval<KT>(m, x[t], lcdf, lpdf);
        out[t] = expf(lcdf);                                                                        // modules.py:193-194
        if (n == 1) ld[b] += lpdf;
        else if (!det) atomicAdd(ld + b, lpdf);
        else nf_lanes_in_turn_add(ld + b, lpdf);
    }
}

template <int KT>
__global__ void __launch_bounds__(NF_BLOCK) k_mixlogcdf_bwd(const float* __restrict__ g_out, const float* __restrict__ g_ld, const float* __restrict__ x,
                                                            const float* __restrict__ lp, const float* __restrict__ mu, const float* __restrict__ s,
                                                            float* __restrict__ g_x, float* __restrict__ g_lp, float* __restrict__ g_mu,
                                                            float* __restrict__ g_s, int64_t n, int K, int64_t total) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / n, e = t - b * n, p = b * K * n + e;
        NfMix<KT> m;
        nf_cdf_load<KT>(lp + p, mu + p, s + p, n, K, m);
        const float xv = x[t], gF = g_out[t], gl = g_ld[b];
        float lcdf, lpdf;
        nf_mix_eval<KT>(m, xv, lcdf, lpdf);
        const float f = expf(lpdf);
        float gx = gF * f;                                       // appendix B6, responsibilities form r_k = pi_k pdf_k / f
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            if (k < K) {
                const float u = (xv - m.mu[k]) * m.es[k];
                const float l = log1pf(expf(-fabsf(u)));
                const float r = expf(m.lp[k] + (u - m.s[k] - 2.f * (fmaxf(u, 0.f) + l)) - lpdf);
                const float omt = -tanhf(0.5f * u);              // 1 - 2 sigmoid(u)
                const float w = gl * r * omt * m.es[k];
                gx += w;
                g_mu[p + k * n] = -gF * f * r - w;
                g_s[p + k * n] = -gF * f * r * (xv - m.mu[k]) + gl * r * (-omt * u - 1.f);
                g_lp[p + k * n] = gF * expf(m.lp[k] + (fminf(u, 0.f) - l)) + gl * r;
            }
        }
        g_x[t] = gx;
    }
}

template <int KT, int PHASE>
__global__ void __launch_bounds__(NF_BLOCK) k_mixlogcdf_inv(const float* __restrict__ target, const float* __restrict__ lp, const float* __restrict__ mu,
                                                            const float* __restrict__ s, float* __restrict__ x, float* __restrict__ ld,
                                                            float* __restrict__ lohi, int* __restrict__ flag, int64_t n, int K, int64_t total, int det) {
    const int steps = PHASE == 1 ? 25 : (flag[0] ? 75 : 0);
    bool stuck = false;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / n, e = t - b * n, p = b * K * n + e;
        NfMix<KT> m;
        nf_cdf_load<KT>(lp + p, mu + p, s + p, n, K, m);
        const float tg = target[t];
        float lo = PHASE == 1 ? -1.0e3f : lohi[t];                                                  // modules.py:197-198
        float hi = PHASE == 1 ? 1.0e3f : lohi[total + t];
        for (int it = 0; it < steps; ++it) {
            const float mid = (lo + hi) * 0.5f;
            float lcdf, lpdf;
            nf_mix_eval<KT>(m, mid, lcdf, lpdf);
            const float val = expf(lcdf);                                                           // modules.py:201
            if (PHASE == 2 && (mid == lo || mid == hi || val == tg)) break;                         // collapsed: the rest are no-ops
            lo = val < tg ? mid : lo;                                                               // modules.py:202-203
            hi = val > tg ? mid : hi;
        }
        if (PHASE == 1) {
            lohi[t] = lo;
            lohi[total + t] = hi;
            stuck |= !(fabsf(hi - lo) < 1.0e-4f);                                                   // modules.py:205
        } else {
            const float xv = (lo + hi) * 0.5f;                                                      // modules.py:208
            float lcdf, lpdf;
            nf_mix_eval<KT>(m, xv, lcdf, lpdf);
            x[t] = xv;
            if (n == 1) ld[b] -= lpdf;                                                              // modules.py:209-212
            else if (!det) atomicAdd(ld + b, -lpdf);
            else nf_lanes_in_turn_add(ld + b, -lpdf);
        }
    }
    if (PHASE == 1 && __any(stuck) && (threadIdx.x & (NF_WAVE - 1)) == 0) atomicOr(flag, 1);
}

extern "C" int nf_mixlogcdf_fwd(const float* x, const float* log_pi, const float* mu, const float* s, float* out, float* ld, int K,
                                int64_t B, int64_t n, nf_stream_t stream) {
    if (K < 1 || K > 32 || B < 0 || n < 0) return NF_E_BADARG;
    const int64_t total = B * n;
    if (total == 0) return 0;
    const int det = (n > 1 && nf_ml_det_host) ? 1 : 0;     // per-element atomics into ld[b]: deterministic mode runs ONE wave, lanes in turn
    const unsigned g = det ? 1u : nf_grid_for(total, NF_BLOCK);
#define CALL(KT) hipLaunchKernelGGL(k_mixlogcdf_fwd<KT>, dim3(g), dim3(det ? NF_WAVE : NF_BLOCK), 0, (hipStream_t)stream, x, log_pi, mu, s, out, ld, n, K, total, det)
    NF_MX_DISPATCH(K, CALL);
#undef CALL
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_mixlogcdf_bwd(const float* g_out, const float* g_ld, const float* x, const float* log_pi, const float* mu,
                                const float* s, float* g_x, float* g_log_pi, float* g_mu, float* g_s, int K, int64_t B, int64_t n,
                                nf_stream_t stream) {
    if (K < 1 || K > 32 || B < 0 || n < 0) return NF_E_BADARG;
    const int64_t total = B * n;
    if (total == 0) return 0;
    const unsigned g = nf_grid_for(total, NF_BLOCK);
#define CALL(KT) hipLaunchKernelGGL(k_mixlogcdf_bwd<KT>, dim3(g), dim3(NF_BLOCK), 0, (hipStream_t)stream, g_out, g_ld, x, log_pi, mu, s, g_x, g_log_pi, g_mu, g_s, n, K, total)
    NF_MX_DISPATCH(K, CALL);
#undef CALL
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_mixlogcdf_inv(const float* target, const float* log_pi, const float* mu, const float* s, float* x, float* ld,
                                float* scratch, int* stuck_flag, int K, int64_t B, int64_t n, nf_stream_t stream) {
    if (K < 1 || K > 32 || B < 0 || n < 0) return NF_E_BADARG;
    const int64_t total = B * n;
    if (total == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(stuck_flag, 0, sizeof(int), st);
    if (e != hipSuccess) return (int)e;
    const int det = (n > 1 && nf_ml_det_host) ? 1 : 0;
    const unsigned g = det ? 1u : nf_grid_for(total, NF_BLOCK);
    const unsigned bt = det ? NF_WAVE : NF_BLOCK;
#define CALL(KT)                                                                                                                     \
    hipLaunchKernelGGL((k_mixlogcdf_inv<KT, 1>), dim3(g), dim3(bt), 0, st, target, log_pi, mu, s, x, ld, scratch, stuck_flag, n, K, total, det); \
    hipLaunchKernelGGL((k_mixlogcdf_inv<KT, 2>), dim3(g), dim3(bt), 0, st, target, log_pi, mu, s, x, ld, scratch, stuck_flag, n, K, total, det)
    NF_MX_DISPATCH(K, CALL);
#undef CALL
    NF_CHECK_LAUNCH();
    return 0;
}
