// Fused Adam over flat parameter / gradient / moment buffers: the whole optimizer step is ONE launch (+ a 1-thread
// step-counter bump), instead of the thousands of per-parameter scalar kernels a capturable framework Adam issues for
// the ~1100 small tensors of a 32-step flow.  Semantics = torch.optim.Adam (no amsgrad), which is what the reference
// trains with (main.py:56-64, configs/default.yaml:13-20):
//   g += wd * p ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// `step` and `lr` live in device memory so that a captured hipGraph replays with the right values.
#include "nf_common.h"

__global__ void k_adam_tick(int* __restrict__ step) { step[0] += 1; }

__global__ void __launch_bounds__(NF_BLOCK) k_adam_step(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        const int* __restrict__ step, const float* __restrict__ lr,
                                                        float b1, float b2, float eps, float wd, float grad_scale,
                                                        int64_t n) {
    const float t = (float)step[0];
    const float bc1 = 1.f - powf(b1, t);
    const float bc2_sqrt = sqrtf(1.f - powf(b2, t));
    const float step_size = lr[0] / bc1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gi = g[i] * grad_scale;
        const float pi = p[i];
        if (wd != 0.f) gi = fmaf(wd, pi, gi);
        const float mi = fmaf(b1, m[i], (1.f - b1) * gi);
        const float vi = fmaf(b2, v[i], (1.f - b2) * gi * gi);
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    }
}

extern "C" int nf_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int* step,
                            const float* lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                            int64_t n, nf_stream_t stream) {
    if (n < 0) return NF_E_BADARG;
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_adam_tick, dim3(1), dim3(1), 0, st, step);
    NF_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_adam_step, dim3(nf_grid_for(n)), dim3(NF_BLOCK), 0, st, param, grad, exp_avg, exp_avg_sq, step, lr,
                       beta1, beta2, eps, weight_decay, grad_scale, n);
    NF_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// gather of many small gradient tensors into the flat bucket: one launch per NF_COPY_MAX tensors instead of one
// AccumulateGrad add per parameter (an image Glow has ~2 600 framework-produced parameter gradients: 12 ms of adds)
// ---------------------------------------------------------------------------------------------------------------
struct NfCopyArgs { nf_copy_desc d[NF_COPY_MAX]; };

__global__ void __launch_bounds__(NF_BLOCK) k_multi_copy(NfCopyArgs args) {
    const nf_copy_desc& d = args.d[blockIdx.y];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.n; i += (int64_t)gridDim.x * blockDim.x) d.dst[i] = d.src[i];
}

extern "C" int nf_multi_copy(const nf_copy_desc* descs, int n_tensors, nf_stream_t stream) {
    if (descs == nullptr || n_tensors < 1 || n_tensors > NF_COPY_MAX) return NF_E_BADARG;
    NfCopyArgs args;
    int64_t maxn = 1;
    for (int i = 0; i < n_tensors; ++i) {
        if (descs[i].n < 0 || (descs[i].n > 0 && (descs[i].src == nullptr || descs[i].dst == nullptr))) return NF_E_BADARG;
        args.d[i] = descs[i];
        if (descs[i].n > maxn) maxn = descs[i].n;
    }
    int64_t gx = (maxn + NF_BLOCK - 1) / NF_BLOCK;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(k_multi_copy, dim3((unsigned)gx, (unsigned)n_tensors), dim3(NF_BLOCK), 0, (hipStream_t)stream, args);
    NF_CHECK_LAUNCH();
    return 0;
}
