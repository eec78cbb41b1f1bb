// Weight normalisation of many layers in one launch (flows/weight_norm.py:35-41):  w = v * g / (||v||_dim0 + eps), the norm
// taken over the output index (dim 0) for every column m = (input channel, ky, kx).  The image conditioners hold ~1000
// weight-normed convolutions; as framework ops each is ~12 tiny kernels forward and ~25 backward (2.5 us each: 60 ms of a
// 164 ms Glow-CIFAR train step).  Here up to NF_WN_MAX_LAYERS layers share one launch per direction: blockIdx.y = layer,
// a workgroup takes 32 columns (consecutive lanes = consecutive m: coalesced) and splits their O rows over eight lane groups.
#include "nf_common.h"

struct NfWnArgs { nf_wn_desc d[NF_WN_MAX_LAYERS]; };

// A workgroup = 32 columns x 8 row groups: lane (m, r) walks rows r, r + 8, ... of column m (consecutive lanes = consecutive m: 128-byte
// rows), NF_WN_U rows per trip with every load of the trip in flight; the eight partial sums of a column meet in LDS in a fixed order.
// (One thread per column with the O rows as a serial chain was 12 - 44 us per 64-layer launch: a 32 x 32 x 3 x 3 layer is 288 columns,
// two workgroups, 2 x 4 dependent trips of eight loads.)
#define NF_WN_U 4
#define NF_WN_COLS 32
#define NF_WN_RG (NF_BLOCK / NF_WN_COLS)

__global__ void __launch_bounds__(NF_BLOCK) k_weight_norm_fwd(NfWnArgs args, float eps) {
    const nf_wn_desc& d = args.d[blockIdx.y];
    __shared__ float red[NF_WN_RG][NF_WN_COLS];
    const int mm = threadIdx.x & (NF_WN_COLS - 1), rg = threadIdx.x / NF_WN_COLS;
    for (int m0 = blockIdx.x * NF_WN_COLS; m0 < d.M; m0 += gridDim.x * NF_WN_COLS) {        // (block-uniform)
        const int m = min(m0 + mm, d.M - 1);
        float ss = 0.f;
        for (int o0 = rg; o0 < d.O; o0 += NF_WN_RG * NF_WN_U) {
            float v[NF_WN_U];
#pragma unroll
            for (int u = 0; u < NF_WN_U; ++u) v[u] = d.v[(size_t)min(o0 + NF_WN_RG * u, d.O - 1) * d.M + m];
#pragma unroll
            for (int u = 0; u < NF_WN_U; ++u)
                if (o0 + NF_WN_RG * u < d.O) ss = fmaf(v[u], v[u], ss);
        }
        red[rg][mm] = ss;
        __syncthreads();
        ss = 0.f;
#pragma unroll
        for (int r = 0; r < NF_WN_RG; ++r) ss += red[r][mm];
        __syncthreads();
        const float sc = d.g[m] / (sqrtf(ss) + eps);
        if (m0 + mm < d.M) {
            for (int o0 = rg; o0 < d.O; o0 += NF_WN_RG * NF_WN_U) {
                float v[NF_WN_U];
#pragma unroll
                for (int u = 0; u < NF_WN_U; ++u) v[u] = d.v[(size_t)min(o0 + NF_WN_RG * u, d.O - 1) * d.M + m];
#pragma unroll
                for (int u = 0; u < NF_WN_U; ++u)
                    if (o0 + NF_WN_RG * u < d.O) d.w[(size_t)(o0 + NF_WN_RG * u) * d.M + m] = v[u] * sc;
            }
        }
    }
}

// g_v = g_w * g / den - v * (<g_w, v> g / (den^2 ||v||)),  g_g = <g_w, v> / den,  den = ||v|| + eps   (per column)
__global__ void __launch_bounds__(NF_BLOCK) k_weight_norm_bwd(NfWnArgs args, float eps) {
    const nf_wn_desc& d = args.d[blockIdx.y];
    const bool acc = d.accumulate != 0;
    __shared__ float red[2][NF_WN_RG][NF_WN_COLS];
    const int mm = threadIdx.x & (NF_WN_COLS - 1), rg = threadIdx.x / NF_WN_COLS;
    for (int m0 = blockIdx.x * NF_WN_COLS; m0 < d.M; m0 += gridDim.x * NF_WN_COLS) {        // (block-uniform)
        const int m = min(m0 + mm, d.M - 1);
        float ss = 0.f, dt = 0.f;
        for (int o0 = rg; o0 < d.O; o0 += NF_WN_RG * NF_WN_U) {
            float v[NF_WN_U], gw[NF_WN_U];
#pragma unroll
            for (int u = 0; u < NF_WN_U; ++u) {
                const size_t e = (size_t)min(o0 + NF_WN_RG * u, d.O - 1) * d.M + m;
                v[u] = d.v[e];
                gw[u] = d.g_w[e];
            }
#pragma unroll
            for (int u = 0; u < NF_WN_U; ++u)
                if (o0 + NF_WN_RG * u < d.O) {
                    ss = fmaf(v[u], v[u], ss);
                    dt = fmaf(gw[u], v[u], dt);
                }
        }
        red[0][rg][mm] = ss;
        red[1][rg][mm] = dt;
        __syncthreads();
        ss = 0.f; dt = 0.f;
#pragma unroll
        for (int r = 0; r < NF_WN_RG; ++r) { ss += red[0][r][mm]; dt += red[1][r][mm]; }
        __syncthreads();
        const float nrm = sqrtf(ss), den = nrm + eps, g = d.g[m];
        const float c1 = g / den, c2 = nrm > 0.f ? dt * g / (den * den * nrm) : 0.f;
        if (m0 + mm < d.M) {
            for (int o0 = rg; o0 < d.O; o0 += NF_WN_RG * NF_WN_U) {
                float v[NF_WN_U], gw[NF_WN_U], old[NF_WN_U];
#pragma unroll
                for (int u = 0; u < NF_WN_U; ++u) {
                    const size_t e = (size_t)min(o0 + NF_WN_RG * u, d.O - 1) * d.M + m;
                    v[u] = d.v[e];
                    gw[u] = d.g_w[e];
                    old[u] = acc ? d.g_v[e] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < NF_WN_U; ++u)
                    if (o0 + NF_WN_RG * u < d.O) d.g_v[(size_t)(o0 + NF_WN_RG * u) * d.M + m] = old[u] + (gw[u] * c1 - v[u] * c2);
            }
            if (rg == 0) d.g_g[m] = (acc ? d.g_g[m] : 0.f) + dt / den;
        }
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
    const dim3 grid((unsigned)((maxM + NF_WN_COLS - 1) / NF_WN_COLS), (unsigned)n);
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
