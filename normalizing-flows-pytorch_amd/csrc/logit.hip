// Logit bijector (flows/modules.py:141-156, helpers :19-32) with the per-sample log-det reduction fused in.
// HBM-bound, 8 B/element.  forward: clamp -> log(x/(1-x)), ld += sum -(y - 2 softplus(y)); inverse: sigmoid.
#include "nf_common.h"
#include "nf_det.h"

NF_DET_STATE(nf_lg)
NF_DET_HOST_API(nf_lg)

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
        else { NF_DET_ENTER_COL(nf_lg); atomicAdd(ld + b, tot); NF_DET_LEAVE_COL(nf_lg); }
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

// ---------------------------------------------------------------------------------------------------------------
// The other elementwise bijector modules of flows/modules.py:125-183 (no reference model builds them): Sigmoid, Tanh, Arctanh, each
// direction one pass with the per-sample log-det reduction fused in, like Logit.
//   kind 0: y = sigmoid(x),            ld += x - 2 softplus(x)                      (Sigmoid.forward, modules.py:129-132)
//   kind 1: y = logit(clamp(x)),       ld += -(log xc + log(1 - xc))                (Sigmoid.backward, :134-138; clamp [1e-8, 1 - 1e-8]: the
//                                                                                    upper bound IS 1 in fp32, as in the reference's fp32 run)
//   kind 2: y = tanh(x),               ld += log(1 - y^2)                           (Tanh.forward :162-165, Arctanh.backward :181-183)
//   kind 3: y = arctanh(x),            ld += -log(1 - xc^2), xc = clamp(x, +-(1 - 1e-8)) (Tanh.backward :167-170, Arctanh.forward :177-179)
// ---------------------------------------------------------------------------------------------------------------
#define NF_BJ_LO 1.0e-8f
__device__ __forceinline__ float nf_bij_elem(int kind, float x, float& ldterm) {
    if (kind == 0) {
        ldterm = x - 2.f * nf_softplus(x);
        return 1.f / (1.f + expf(-x));
    }
    if (kind == 1) {
        const float xc = fminf(fmaxf(x, NF_BJ_LO), (float)(1.0 - 1.0e-8));
        const float la = logf(xc), lb = logf(1.f - xc);
        ldterm = -(la + lb);
        return la - lb;
    }
    if (kind == 2) {
        const float y = tanhf(x);
        ldterm = logf(1.f - y * y);
        return y;
    }
    const float hi = (float)(1.0 - 1.0e-8);
    const float xc = fminf(fmaxf(x, -hi), hi);
    ldterm = -logf(1.f - xc * xc);
    return atanhf(x);
}
__global__ void __launch_bounds__(NF_BLOCK) k_bijector_fwd(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ ld,
                                                           int kind, int64_t n) {
    __shared__ float scratch[NF_BLOCK / NF_WAVE];
    const int64_t b = blockIdx.x;
    const int64_t e0 = (int64_t)blockIdx.y * NF_LG_SLAB;
    const int64_t e1 = min(e0 + NF_LG_SLAB, n);
    float acc = 0.f;
    for (int64_t e = e0 + threadIdx.x; e < e1; e += NF_BLOCK) {
        float t;
        y[b * n + e] = nf_bij_elem(kind, x[b * n + e], t);
        acc += t;
    }
    const float tot = nf_block_sum(acc, scratch);
    if (threadIdx.x == 0) {
        if (gridDim.y == 1) ld[b] += tot;
        else { NF_DET_ENTER_COL(nf_lg); atomicAdd(ld + b, tot); NF_DET_LEAVE_COL(nf_lg); }
    }
}
// autograd of the modules' forward directions (kinds 0, 2, 3): g_x = g_y dy/dx + g_ld d(ld term)/dx
__global__ void __launch_bounds__(NF_BLOCK) k_bijector_bwd(const float* __restrict__ gy, const float* __restrict__ gld,
                                                           const float* __restrict__ x, float* __restrict__ gx, int kind, int64_t n,
                                                           int64_t total) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const float xv = x[t], g = gy[t], gl = gld[t / n];
        float r;
        if (kind == 0) {
            const float sg = 1.f / (1.f + expf(-xv));
            r = g * sg * (1.f - sg) + gl * (1.f - 2.f * sg);
        } else if (kind == 2) {
            const float y = tanhf(xv);
            r = g * (1.f - y * y) - 2.f * y * gl;
        } else {
            const float hi = (float)(1.0 - 1.0e-8);
            const float inv = 1.f / (1.f - xv * xv);
            r = g * inv + ((xv >= -hi && xv <= hi) ? gl * 2.f * xv * inv : 0.f);
        }
        gx[t] = r;
    }
}
extern "C" int nf_bijector_fwd(const float* x, float* y, float* ld, int kind, int64_t B, int64_t n, nf_stream_t stream) {
    if (n <= 0 || B > 0x7fffffffLL || kind < 0 || kind > 3) return NF_E_BADARG;
    if (B == 0) return 0;
    dim3 grid((unsigned)B, (unsigned)((n + NF_LG_SLAB - 1) / NF_LG_SLAB));
    hipLaunchKernelGGL(k_bijector_fwd, grid, dim3(NF_BLOCK), 0, (hipStream_t)stream, x, y, ld, kind, n);
    NF_CHECK_LAUNCH();
    return 0;
}
extern "C" int nf_bijector_bwd(const float* g_y, const float* g_ld, const float* x, float* g_x, int kind, int64_t B, int64_t n,
                               nf_stream_t stream) {
    if (n <= 0 || (kind != 0 && kind != 2 && kind != 3)) return NF_E_BADARG;
    if (B == 0) return 0;
    hipLaunchKernelGGL(k_bijector_bwd, dim3(nf_grid_for(B * n)), dim3(NF_BLOCK), 0, (hipStream_t)stream, g_y, g_ld, x, g_x, kind, n, B * n);
    NF_CHECK_LAUNCH();
    return 0;
}
