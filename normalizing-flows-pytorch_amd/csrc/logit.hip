// Logit bijector (flows/modules.py:141-156, helpers :19-32) with the per-sample log-det reduction fused in.
// HBM-bound, 8 B/element.  forward: clamp -> log(x/(1-x)), ld += sum -(y - 2 softplus(y)); inverse: sigmoid.
#include "nf_common.h"

#define NF_LG_SLAB 4096

template <bool INVERSE>
__device__ __forceinline__ float nf_logit_elem(float x, float eps, float& ldterm) {
    if (INVERSE) {                                   // modules.py:152-155
        ldterm = x - 2.f * nf_softplus(x);
        return 1.f / (1.f + expf(-x));
    }
    const float xc = fminf(fmaxf(x, eps), 1.f - eps);            // modules.py:147
    // log-det term -(y - 2 softplus(y)) at y = logit(xc) (modules.py:19-32) is exactly -(log xc + log(1 - xc)):
    // softplus(logit(x)) = -log(1 - x).  Two logs serve both outputs (the literal form costs 2 logs + exp + log1p and
    // made this kernel transcendental-bound at 1.5 TB/s); differs from the literal form by ~1 ulp.
    const float la = logf(xc), lb = logf(1.f - xc);
    ldterm = -(la + lb);
    return la - lb;
}

template <bool INVERSE>
__global__ void __launch_bounds__(NF_BLOCK) k_logit_fwd(const float* __restrict__ x, float* __restrict__ y,
                                                        float* __restrict__ ld, float eps, int64_t n) {
    __shared__ float scratch[NF_BLOCK / NF_WAVE];
    const int64_t b = blockIdx.x;
    const int64_t e0 = (int64_t)blockIdx.y * NF_LG_SLAB;
    const int64_t e1 = min(e0 + NF_LG_SLAB, n);
    float acc = 0.f;
    for (int64_t e = e0 + threadIdx.x; e < e1; e += NF_BLOCK) {
        float t;
        y[b * n + e] = nf_logit_elem<INVERSE>(x[b * n + e], eps, t);
        acc += t;
    }
    const float tot = nf_block_sum(acc, scratch);
    if (threadIdx.x == 0) {
        if (gridDim.y == 1) ld[b] += tot;
        else atomicAdd(ld + b, tot);
    }
}

// autograd of the forward direction; zero outside the clamp range (torch.clamp backward)
__global__ void __launch_bounds__(NF_BLOCK) k_logit_bwd(const float* __restrict__ gy, const float* __restrict__ gld,
                                                        const float* __restrict__ x, float* __restrict__ gx, float eps,
                                                        int64_t n, int64_t total) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const float xv = x[t];
        float g = 0.f;
        if (xv >= eps && xv <= 1.f - eps) {
            const float inv = 1.f / (xv * (1.f - xv));
            g = gy[t] * inv - gld[t / n] * (1.f - 2.f * xv) * inv;
        }
        gx[t] = g;
    }
}

extern "C" int nf_logit_fwd(const float* x, float* y, float* ld, float eps, int inverse, int64_t B, int64_t n,
                            nf_stream_t stream) {
    if (n <= 0 || B > 0x7fffffffLL) return NF_E_BADARG;
    if (B == 0) return 0;
    dim3 grid((unsigned)B, (unsigned)((n + NF_LG_SLAB - 1) / NF_LG_SLAB));
    if (inverse) hipLaunchKernelGGL(k_logit_fwd<true>, grid, dim3(NF_BLOCK), 0, (hipStream_t)stream, x, y, ld, eps, n);
    else hipLaunchKernelGGL(k_logit_fwd<false>, grid, dim3(NF_BLOCK), 0, (hipStream_t)stream, x, y, ld, eps, n);
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_logit_bwd(const float* g_y, const float* g_ld, const float* x, float* g_x, float eps, int64_t B,
                            int64_t n, nf_stream_t stream) {
    if (n <= 0) return NF_E_BADARG;
    if (B == 0) return 0;
    hipLaunchKernelGGL(k_logit_bwd, dim3(nf_grid_for(B * n)), dim3(NF_BLOCK), 0, (hipStream_t)stream, g_y, g_ld, x, g_x,
                       eps, n, B * n);
    NF_CHECK_LAUNCH();
    return 0;
}
