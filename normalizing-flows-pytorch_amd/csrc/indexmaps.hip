// Bit-exact index maps: split halves (gather / scatter) and space-to-depth (Squeeze2d / Unsqueeze2d).
// Reference: flows/squeeze.py:5-17, :32-61, :64-83, :86-111, :153-189.  Pure permutations -> HBM-bound, 8 B/element.
#include "nf_common.h"

__global__ void __launch_bounds__(NF_BLOCK) k_half_gather(const float* __restrict__ z, float* __restrict__ half,
                                                          NfSplit s, int which, int64_t total) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / s.n_half;
        const int e = (int)(t - b * s.n_half);
        half[t] = z[b * s.n_full + nf_half_to_full(s, which, e)];
    }
}

// one thread per half-element index e: the element of half `which` is written, its partner in the other half zeroed
__global__ void __launch_bounds__(NF_BLOCK) k_half_scatter(const float* __restrict__ half, float* __restrict__ full,
                                                           NfSplit s, int which, int64_t total) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / s.n_half;
        const int e = (int)(t - b * s.n_half);
        float* base = full + b * s.n_full;
        base[nf_half_to_full(s, which, e)] = half[t];
        base[nf_half_to_full(s, which ^ 1, e)] = 0.f;
    }
}

template <bool TO_DEPTH>
__global__ void __launch_bounds__(NF_BLOCK) k_space_depth(const float* __restrict__ in, float* __restrict__ out,
                                                          int H, int W, int n, int64_t total) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / n;
        const int e = (int)(t - b * n);
        const int64_t f = b * n + nf_squeezed_to_full(e, H, W);
        if (TO_DEPTH) out[t] = in[f];
        else out[f] = in[t];
    }
}

extern "C" int nf_half_gather(const float* z, float* half, int which, int mode, int odd, int64_t B, int C, int H,
                              int W, nf_stream_t stream) {
    NfSplit s;
    if (!nf_make_split(s, mode, odd, C, H, W) || mode == NF_SPLIT_NONE || (which & ~1)) return NF_E_BADARG;
    const int64_t total = B * s.n_half;
    if (total == 0) return 0;
    hipLaunchKernelGGL(k_half_gather, dim3(nf_grid_for(total)), dim3(NF_BLOCK), 0, (hipStream_t)stream, z, half, s,
                       which, total);
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_half_scatter(const float* half, float* full, int which, int mode, int odd, int64_t B, int C, int H,
                               int W, nf_stream_t stream) {
    NfSplit s;
    if (!nf_make_split(s, mode, odd, C, H, W) || mode == NF_SPLIT_NONE || (which & ~1)) return NF_E_BADARG;
    const int64_t total = B * s.n_half;
    if (total == 0) return 0;
    hipLaunchKernelGGL(k_half_scatter, dim3(nf_grid_for(total)), dim3(NF_BLOCK), 0, (hipStream_t)stream, half, full, s,
                       which, total);
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_squeeze2d(const float* z, float* out, int64_t B, int C, int H, int W, nf_stream_t stream) {
    if ((H & 1) || (W & 1)) return NF_E_BADARG;
    const int n = C * H * W;
    const int64_t total = B * n;
    if (total == 0) return 0;
    hipLaunchKernelGGL(k_space_depth<true>, dim3(nf_grid_for(total)), dim3(NF_BLOCK), 0, (hipStream_t)stream, z, out, H,
                       W, n, total);
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_unsqueeze2d(const float* z, float* out, int64_t B, int C, int H, int W, nf_stream_t stream) {
    if ((H & 1) || (W & 1)) return NF_E_BADARG;  // C, H, W: the UNSQUEEZED (output) dims
    const int n = C * H * W;
    const int64_t total = B * n;
    if (total == 0) return 0;
    hipLaunchKernelGGL(k_space_depth<false>, dim3(nf_grid_for(total)), dim3(NF_BLOCK), 0, (hipStream_t)stream, z, out,
                       H, W, n, total);
    NF_CHECK_LAUNCH();
    return 0;
}
