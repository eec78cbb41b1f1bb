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

// ---------------------------------------------------------------------------------------------------------------
// Vectorised image variants (W % 4 == 0, 16-byte aligned bases).  A thread owns ONE float4 of a row (c, y) of the FULL tensor:
// its even-x components belong to squeezed channel k0 = 4 c + 2 (y & 1), the odd-x ones to k0 + 1 (squeeze.py:36-41, :90-92),
// and the two elements of a parity are consecutive columns j = 2 x4, 2 x4 + 1 of the squeezed / half tensor: one float2.
// So the full tensor is always touched in whole 16-byte vectors and the half / squeezed one in 8-byte pairs that consecutive
// threads lay end to end.  Same maps as nf_half_to_full / nf_squeezed_to_full (the bit-exact tests cover both paths).
// ---------------------------------------------------------------------------------------------------------------
#define NF_IVSLAB 1024
__device__ __forceinline__ void nf_ck_sel(const NfSplit& s, int k, int i, int j0, int& which, int& e) {
    const int q = k / s.C;
    const int sel = (q == 1 || q == 2) ? 1 : 0;
    const int m = sel ? k - s.C : (q == 0 ? k : k - 2 * s.C);
    which = sel ^ s.odd;
    e = (m * s.h + i) * s.w + j0;
}

// MODE 0: gather half `which` (rows that hold none of it are not read); 1: scatter half `which`, zeros elsewhere; 2: scatter on top of
// `base` (full tensor): dst = base + scatter(src)
template <int MODE, bool CHECKER>
__global__ void __launch_bounds__(NF_BLOCK) k_half_move_v4(const float* __restrict__ src, float* __restrict__ dst, NfSplit s, int which,
                                                           const float* __restrict__ base = nullptr) {
    const int64_t b = blockIdx.x;
    const int n4 = s.n_full >> 2, W4 = s.W >> 2, h4 = s.n_half >> 2;
    const float* full_r = src + b * s.n_full;            // MODE 0: source is the full tensor
    float* full_w = dst + b * s.n_full;                  // MODE 1: destination is the full tensor
    const float* half_r = src + b * s.n_half;
    float* half_w = dst + b * s.n_half;
    const int v1 = min((int)(blockIdx.y + 1) * NF_IVSLAB, n4);
    for (int v = blockIdx.y * NF_IVSLAB + threadIdx.x; v < v1; v += NF_BLOCK) {
        if (!CHECKER) {
            const int sel = v >= h4 ? 1 : 0;
            const bool mine = (sel ^ s.odd) == which;
            const int e4 = v - sel * h4;
            if (MODE == 0) {
                if (mine) reinterpret_cast<float4*>(half_w)[e4] = reinterpret_cast<const float4*>(full_r)[v];
            } else {
                float4 o = mine ? reinterpret_cast<const float4*>(half_r)[e4] : make_float4(0.f, 0.f, 0.f, 0.f);
                if (MODE == 2) {
                    const float4 bv = reinterpret_cast<const float4*>(base + b * s.n_full)[v];
                    o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w;
                }
                reinterpret_cast<float4*>(full_w)[v] = o;
            }
        } else {
            const int r = v / W4, x4 = v - r * W4;
            const int cc = r / s.H, yy = r - cc * s.H;
            const int k0 = 4 * cc + 2 * (yy & 1);
            int w0, e0, w1, e1;
            nf_ck_sel(s, k0, yy >> 1, 2 * x4, w0, e0);
            nf_ck_sel(s, k0 + 1, yy >> 1, 2 * x4, w1, e1);
            if (MODE == 0) {
                if (w0 == which || w1 == which) {
                    const float4 zv = reinterpret_cast<const float4*>(full_r)[v];
                    if (w0 == which) *reinterpret_cast<float2*>(half_w + e0) = make_float2(zv.x, zv.z);
                    if (w1 == which) *reinterpret_cast<float2*>(half_w + e1) = make_float2(zv.y, zv.w);
                }
            } else {
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                if (w0 == which) { const float2 t = *reinterpret_cast<const float2*>(half_r + e0); o.x = t.x; o.z = t.y; }
                if (w1 == which) { const float2 t = *reinterpret_cast<const float2*>(half_r + e1); o.y = t.x; o.w = t.y; }
                if (MODE == 2) {
                    const float4 bv = reinterpret_cast<const float4*>(base + b * s.n_full)[v];
                    o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w;
                }
                reinterpret_cast<float4*>(full_w)[v] = o;
            }
        }
    }
}

// space-to-depth: `full` is (C, H, W), `deep` is (4 C, H / 2, W / 2) with channel k = 4 c + 2 dy + dx
template <bool TO_DEPTH>
__global__ void __launch_bounds__(NF_BLOCK) k_space_depth_v4(const float* __restrict__ in, float* __restrict__ out, int H, int W, int n) {
    const int64_t b = blockIdx.x;
    const int n4 = n >> 2, W4 = W >> 2, h = H >> 1, w = W >> 1;
    const float* ib = in + b * n;
    float* ob = out + b * n;
    const int v1 = min((int)(blockIdx.y + 1) * NF_IVSLAB, n4);
    for (int v = blockIdx.y * NF_IVSLAB + threadIdx.x; v < v1; v += NF_BLOCK) {
        const int r = v / W4, x4 = v - r * W4;
        const int cc = r / H, yy = r - cc * H;
        const int k0 = 4 * cc + 2 * (yy & 1);
        const int e0 = (k0 * h + (yy >> 1)) * w + 2 * x4, e1 = e0 + h * w;        // channel k0 + 1 is one plane further
        if (TO_DEPTH) {
            const float4 zv = reinterpret_cast<const float4*>(ib)[v];
            *reinterpret_cast<float2*>(ob + e0) = make_float2(zv.x, zv.z);
            *reinterpret_cast<float2*>(ob + e1) = make_float2(zv.y, zv.w);
        } else {
            const float2 a = *reinterpret_cast<const float2*>(ib + e0), c2 = *reinterpret_cast<const float2*>(ib + e1);
            reinterpret_cast<float4*>(ob)[v] = make_float4(a.x, c2.x, a.y, c2.y);
        }
    }
}

static inline bool nf_iv_ok(const void* a, const void* b2, int W, int n_full, int n_half, int64_t B) {
    return (((uintptr_t)a | (uintptr_t)b2) & 15) == 0 && W % 4 == 0 && n_half % 4 == 0 && n_full < (1 << 30) && B <= 0x7fffffffLL;
}

extern "C" int nf_half_gather(const float* z, float* half, int which, int mode, int odd, int64_t B, int C, int H,
                              int W, nf_stream_t stream) {
    NfSplit s;
    if (!nf_make_split(s, mode, odd, C, H, W) || mode == NF_SPLIT_NONE || (which & ~1)) return NF_E_BADARG;
    const int64_t total = B * s.n_half;
    if (total == 0) return 0;
    if ((mode == NF_SPLIT_CHANNEL || mode == NF_SPLIT_CHECKER) && nf_iv_ok(z, half, W, s.n_full, s.n_half, B)) {
        dim3 grid((unsigned)B, (unsigned)((s.n_full / 4 + NF_IVSLAB - 1) / NF_IVSLAB));
        if (mode == NF_SPLIT_CHECKER) hipLaunchKernelGGL((k_half_move_v4<0, true>), grid, dim3(NF_BLOCK), 0, (hipStream_t)stream, z, half, s, which);
        else hipLaunchKernelGGL((k_half_move_v4<0, false>), grid, dim3(NF_BLOCK), 0, (hipStream_t)stream, z, half, s, which);
        NF_CHECK_LAUNCH();
        return 0;
    }
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
    if ((mode == NF_SPLIT_CHANNEL || mode == NF_SPLIT_CHECKER) && nf_iv_ok(half, full, W, s.n_full, s.n_half, B)) {
        dim3 grid((unsigned)B, (unsigned)((s.n_full / 4 + NF_IVSLAB - 1) / NF_IVSLAB));
        if (mode == NF_SPLIT_CHECKER) hipLaunchKernelGGL((k_half_move_v4<1, true>), grid, dim3(NF_BLOCK), 0, (hipStream_t)stream, half, full, s, which);
        else hipLaunchKernelGGL((k_half_move_v4<1, false>), grid, dim3(NF_BLOCK), 0, (hipStream_t)stream, half, full, s, which);
        NF_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(k_half_scatter, dim3(nf_grid_for(total)), dim3(NF_BLOCK), 0, (hipStream_t)stream, half, full, s,
                       which, total);
    NF_CHECK_LAUNCH();
    return 0;
}

// out = base + scatter(half): the gradient of a tensor that was consumed whole AND through a gather of one of its halves, in one pass
__global__ void __launch_bounds__(NF_BLOCK) k_half_scatter_add(const float* __restrict__ half, const float* __restrict__ base, float* __restrict__ out,
                                                               NfSplit s, int which, int64_t total) {
    const int P = s.H * s.W;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / s.n_full;
        const int r = (int)(t - b * s.n_full), c = r / P, p = r - c * P;
        int w, e;
        nf_full_to_half(s, c, p, w, e);
        out[t] = base[t] + (w == which ? half[b * s.n_half + e] : 0.f);
    }
}

extern "C" int nf_half_scatter_add(const float* half, const float* base, float* out, int which, int mode, int odd, int64_t B, int C, int H,
                                   int W, nf_stream_t stream) {
    NfSplit s;
    if (!nf_make_split(s, mode, odd, C, H, W) || (mode != NF_SPLIT_CHANNEL && mode != NF_SPLIT_CHECKER) || (which & ~1)) return NF_E_BADARG;
    if (half == nullptr || base == nullptr || out == nullptr) return NF_E_BADARG;
    const int64_t total = B * s.n_full;
    if (total == 0) return 0;
    if (nf_iv_ok(half, out, W, s.n_full, s.n_half, B) && ((uintptr_t)base & 15) == 0) {
        dim3 grid((unsigned)B, (unsigned)((s.n_full / 4 + NF_IVSLAB - 1) / NF_IVSLAB));
        if (mode == NF_SPLIT_CHECKER) hipLaunchKernelGGL((k_half_move_v4<2, true>), grid, dim3(NF_BLOCK), 0, (hipStream_t)stream, half, out, s, which, base);
        else hipLaunchKernelGGL((k_half_move_v4<2, false>), grid, dim3(NF_BLOCK), 0, (hipStream_t)stream, half, out, s, which, base);
        NF_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(k_half_scatter_add, dim3(nf_grid_for(total)), dim3(NF_BLOCK), 0, (hipStream_t)stream, half, base, out, s, which, total);
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_squeeze2d(const float* z, float* out, int64_t B, int C, int H, int W, nf_stream_t stream) {
    if ((H & 1) || (W & 1)) return NF_E_BADARG;
    const int n = C * H * W;
    const int64_t total = B * n;
    if (total == 0) return 0;
    if (nf_iv_ok(z, out, W, n, 4, B)) {
        dim3 grid((unsigned)B, (unsigned)((n / 4 + NF_IVSLAB - 1) / NF_IVSLAB));
        hipLaunchKernelGGL(k_space_depth_v4<true>, grid, dim3(NF_BLOCK), 0, (hipStream_t)stream, z, out, H, W, n);
        NF_CHECK_LAUNCH();
        return 0;
    }
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
    if (nf_iv_ok(z, out, W, n, 4, B)) {
        dim3 grid((unsigned)B, (unsigned)((n / 4 + NF_IVSLAB - 1) / NF_IVSLAB));
        hipLaunchKernelGGL(k_space_depth_v4<false>, grid, dim3(NF_BLOCK), 0, (hipStream_t)stream, z, out, H, W, n);
        NF_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(k_space_depth<false>, dim3(nf_grid_for(total)), dim3(NF_BLOCK), 0, (hipStream_t)stream, z, out,
                       H, W, n, total);
    NF_CHECK_LAUNCH();
    return 0;
}

// Squeeze1d / Unsqueeze1d as flow layers (flows/squeeze.py:114-151): out = cat(z[:, odd::2], z[:, 1 - odd::2]) and its inverse map --
// a permutation of the D columns (squeeze.py:63-83), one element per thread.
__global__ void __launch_bounds__(NF_BLOCK) k_squeeze1d(const float* __restrict__ in, float* __restrict__ out, int64_t total, int D, int odd,
                                                        int inverse) {
    const int h = D >> 1;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / D;
        const int j = (int)(t - b * D);                        // position in the squeezed (concatenated) row
        const int src = j < h ? 2 * j + odd : 2 * (j - h) + 1 - odd;       // its column in the interleaved row
        if (inverse) out[b * D + src] = in[t];
        else out[t] = in[b * D + src];
    }
}
extern "C" int nf_squeeze1d(const float* in, float* out, int64_t B, int D, int odd, int inverse, nf_stream_t stream) {
    if (D < 2 || (D & 1) || in == nullptr || out == nullptr) return NF_E_BADARG;
    if (B <= 0) return B == 0 ? 0 : NF_E_BADARG;
    hipLaunchKernelGGL(k_squeeze1d, dim3(nf_grid_for(B * D)), dim3(NF_BLOCK), 0, (hipStream_t)stream, in, out, B * D, D, odd ? 1 : 0,
                       inverse ? 1 : 0);
    NF_CHECK_LAUNCH();
    return 0;
}
