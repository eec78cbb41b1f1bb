// Weight normalisation of many layers in one launch (flows/weight_norm.py:35-41):  w = v * g / (||v||_dim0 + eps), the norm
// taken over the output index (dim 0) for every column m = (input channel, ky, kx).  The image conditioners hold ~1000
// weight-normed convolutions; as framework ops each is ~12 tiny kernels forward and ~25 backward (2.5 us each: 60 ms of a
// 164 ms Glow-CIFAR train step).  Here up to NF_WN_MAX_LAYERS layers share one launch per direction: blockIdx.y = layer,
// threads walk columns (consecutive threads = consecutive m: coalesced), each thread loops over the O rows of its column.
#include "nf_common.h"

struct NfWnArgs { nf_wn_desc d[NF_WN_MAX_LAYERS]; };

__global__ void __launch_bounds__(NF_BLOCK) k_weight_norm_fwd(NfWnArgs args, float eps) {
    const nf_wn_desc& d = args.d[blockIdx.y];
    for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < d.M; m += gridDim.x * blockDim.x) {
        float ss = 0.f;
        for (int o = 0; o < d.O; ++o) { const float v = d.v[(size_t)o * d.M + m]; ss = fmaf(v, v, ss); }
        const float sc = d.g[m] / (sqrtf(ss) + eps);
        for (int o = 0; o < d.O; ++o) d.w[(size_t)o * d.M + m] = d.v[(size_t)o * d.M + m] * sc;
    }
}

// g_v = g_w * g / den - v * (<g_w, v> g / (den^2 ||v||)),  g_g = <g_w, v> / den,  den = ||v|| + eps   (per column)
__global__ void __launch_bounds__(NF_BLOCK) k_weight_norm_bwd(NfWnArgs args, float eps) {
    const nf_wn_desc& d = args.d[blockIdx.y];
    const bool acc = d.accumulate != 0;
    for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < d.M; m += gridDim.x * blockDim.x) {
        float ss = 0.f, dt = 0.f;
        for (int o = 0; o < d.O; ++o) {
            const float v = d.v[(size_t)o * d.M + m];
            ss = fmaf(v, v, ss);
            dt = fmaf(d.g_w[(size_t)o * d.M + m], v, dt);
        }
        const float nrm = sqrtf(ss), den = nrm + eps, g = d.g[m];
        const float c1 = g / den, c2 = nrm > 0.f ? dt * g / (den * den * nrm) : 0.f;
        for (int o = 0; o < d.O; ++o) {
            const size_t e = (size_t)o * d.M + m;
            const float gv = d.g_w[e] * c1 - d.v[e] * c2;
            d.g_v[e] = (acc ? d.g_v[e] : 0.f) + gv;
        }
        d.g_g[m] = (acc ? d.g_g[m] : 0.f) + dt / den;
    }
}

static int nf_wn_launch(const nf_wn_desc* descs, int n, float eps, bool bwd, hipStream_t st) {
    if (descs == nullptr || n < 1 || n > NF_WN_MAX_LAYERS) return NF_E_BADARG;
    NfWnArgs args;
    int maxM = 1;
    for (int i = 0; i < n; ++i) {
        if (descs[i].O < 1 || descs[i].M < 1) return NF_E_BADARG;
        args.d[i] = descs[i];
        if (descs[i].M > maxM) maxM = descs[i].M;
    }
    const dim3 grid((unsigned)((maxM + NF_BLOCK - 1) / NF_BLOCK), (unsigned)n);
    if (bwd) hipLaunchKernelGGL(k_weight_norm_bwd, grid, dim3(NF_BLOCK), 0, st, args, eps);
    else hipLaunchKernelGGL(k_weight_norm_fwd, grid, dim3(NF_BLOCK), 0, st, args, eps);
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_weight_norm_fwd(const nf_wn_desc* descs, int n_layers, float eps, nf_stream_t stream) {
    return nf_wn_launch(descs, n_layers, eps, false, (hipStream_t)stream);
}
extern "C" int nf_weight_norm_bwd(const nf_wn_desc* descs, int n_layers, float eps, nf_stream_t stream) {
    return nf_wn_launch(descs, n_layers, eps, true, (hipStream_t)stream);
}
