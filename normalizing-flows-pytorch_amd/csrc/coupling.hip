// Fused affine coupling: split gather -> tanh*scale+bias -> z0*exp(s)+t -> merge scatter -> per-sample sum(s) into ld.
// Reference: flows/coupling.py:32-43 (split/merge wrapper), :104-122 (transform / inverse), flows/maf.py:101-107.
// HBM-bound: reads z (4 B/elem) + params (4 B/elem of z) and writes y (4 B/elem) = 12 B per element of z.
//
// Two launch shapes:
//   rows : n_half <= NF_ROWS_MAX (2-D data: one transformed feature per sample) -> one thread per sample,
//          no cross-lane reduction for ld at all.
//   slab : images -> grid (sample, slab); 256 threads stride over a slab of the sample's half tensor; the
//          per-sample sum(s) is a wave64 shuffle + LDS reduction and ONE write (or one atomic when slabs > 1).
#include "nf_common.h"

#define NF_ROWS_MAX 16
#define NF_SLAB 2048
#define NF_ATOMIC_GRID 512  // kernels ending in same-address atomics: <= 2 blocks per CU (an atomic costs ~23 ns serialised)

__device__ __forceinline__ float nf_scale_of(float s_raw, float a, float c) { return tanhf(s_raw) * a + c; }

// ---------------------------------------------------------------------------------------------------------------
template <bool INVERSE>
__global__ void __launch_bounds__(NF_BLOCK) k_affine_rows_fwd(const float* __restrict__ z, const float* __restrict__ tp,
                                                              const float* __restrict__ sp, int64_t pbs,
                                                              const float* __restrict__ p_a, const float* __restrict__ p_c,
                                                              float* __restrict__ y, float* __restrict__ ld, NfSplit s,
                                                              int64_t B) {
    const float a = p_a[0], c = p_c[0];
    const bool has_pass = s.mode != NF_SPLIT_NONE;
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += (int64_t)gridDim.x * blockDim.x) {
        const float* zb = z + b * s.n_full;
        float* yb = y + b * s.n_full;
        float acc = 0.f;
        for (int e = 0; e < s.n_half; ++e) {
            const int o0 = nf_half_to_full(s, 0, e);
            const float sv = nf_scale_of(sp[b * pbs + e], a, c);
            const float t = tp[b * pbs + e];
            const float z0 = zb[o0];
            yb[o0] = INVERSE ? expf(-sv) * (z0 - t) : z0 * expf(sv) + t;
            acc += sv;
            if (has_pass) {
                const int o1 = nf_half_to_full(s, 1, e);
                yb[o1] = zb[o1];
            }
        }
        ld[b] += INVERSE ? -acc : acc;
    }
}

template <bool INVERSE>
__global__ void __launch_bounds__(NF_BLOCK) k_affine_slab_fwd(const float* __restrict__ z, const float* __restrict__ tp,
                                                              const float* __restrict__ sp, int64_t pbs,
                                                              const float* __restrict__ p_a, const float* __restrict__ p_c,
                                                              float* __restrict__ y, float* __restrict__ ld, NfSplit s) {
    __shared__ float scratch[NF_BLOCK / NF_WAVE];
    const float a = p_a[0], c = p_c[0];
    const bool has_pass = s.mode != NF_SPLIT_NONE;
    const int64_t b = blockIdx.x;
    const int e0 = blockIdx.y * NF_SLAB;
    const int e1 = min(e0 + NF_SLAB, s.n_half);
    const float* zb = z + b * s.n_full;
    float* yb = y + b * s.n_full;
    const float* tb = tp + b * pbs;
    const float* sb = sp + b * pbs;
    float acc = 0.f;
    for (int e = e0 + threadIdx.x; e < e1; e += NF_BLOCK) {
        const int o0 = nf_half_to_full(s, 0, e);
        const float sv = nf_scale_of(sb[e], a, c);
        const float t = tb[e];
        const float z0 = zb[o0];
        yb[o0] = INVERSE ? expf(-sv) * (z0 - t) : z0 * expf(sv) + t;
        acc += sv;
        if (has_pass) {
            const int o1 = nf_half_to_full(s, 1, e);
            yb[o1] = zb[o1];
        }
    }
    const float tot = nf_block_sum(acc, scratch);
    if (threadIdx.x == 0) {
        const float d = INVERSE ? -tot : tot;
        if (gridDim.y == 1) ld[b] += d;
        else atomicAdd(ld + b, d);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// autograd of the forward direction
__device__ __forceinline__ void nf_affine_bwd_elem(float gy0, float gld, float z0, float sr, float a, float c,
                                                   float& g_z0, float& g_sr, float& acc_a, float& acc_c) {
    const float th = tanhf(sr);
    const float es = expf(th * a + c);
    g_z0 = gy0 * es;
    const float gs = gy0 * z0 * es + gld;
    g_sr = gs * a * (1.f - th * th);
    acc_a += gs * th;
    acc_c += gs;
}

__global__ void __launch_bounds__(NF_BLOCK) k_affine_rows_bwd(const float* __restrict__ gy, const float* __restrict__ gld,
                                                              const float* __restrict__ z, const float* __restrict__ sp,
                                                              int64_t pbs, const float* __restrict__ p_a,
                                                              const float* __restrict__ p_c, float* __restrict__ gz,
                                                              float* __restrict__ gt, float* __restrict__ gs,
                                                              float* __restrict__ g_scale, float* __restrict__ g_bias,
                                                              NfSplit s, int64_t B) {
    __shared__ float scratch[NF_BLOCK / NF_WAVE];
    const float a = p_a[0], c = p_c[0];
    const bool has_pass = s.mode != NF_SPLIT_NONE;
    float acc_a = 0.f, acc_c = 0.f;
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += (int64_t)gridDim.x * blockDim.x) {
        const int64_t fb = b * s.n_full;
        const float gl = gld[b];
        for (int e = 0; e < s.n_half; ++e) {
            const int o0 = nf_half_to_full(s, 0, e);
            const float g0 = gy[fb + o0];
            float g_z0, g_sr;
            nf_affine_bwd_elem(g0, gl, z[fb + o0], sp[b * pbs + e], a, c, g_z0, g_sr, acc_a, acc_c);
            gz[fb + o0] = g_z0;
            gt[b * pbs + e] = g0;
            gs[b * pbs + e] = g_sr;
            if (has_pass) {
                const int o1 = nf_half_to_full(s, 1, e);
                gz[fb + o1] = gy[fb + o1];
            }
        }
    }
    const float ta = nf_block_sum(acc_a, scratch);
    const float tc = nf_block_sum(acc_c, scratch);
    if (threadIdx.x == 0) {
        atomicAdd(g_scale, ta);
        atomicAdd(g_bias, tc);
    }
}

__global__ void __launch_bounds__(NF_BLOCK) k_affine_slab_bwd(const float* __restrict__ gy, const float* __restrict__ gld,
                                                              const float* __restrict__ z, const float* __restrict__ sp,
                                                              int64_t pbs, const float* __restrict__ p_a,
                                                              const float* __restrict__ p_c, float* __restrict__ gz,
                                                              float* __restrict__ gt, float* __restrict__ gs,
                                                              float* __restrict__ g_scale, float* __restrict__ g_bias,
                                                              NfSplit s, int64_t B, int slabs) {
    __shared__ float scratch[NF_BLOCK / NF_WAVE];
    const float a = p_a[0], c = p_c[0];
    const bool has_pass = s.mode != NF_SPLIT_NONE;
    float acc_a = 0.f, acc_c = 0.f;
    const int64_t items = B * slabs;                      // persistent blocks: ONE pair of atomics per block
    for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
        const int64_t b = item / slabs;
        const int e0 = (int)(item - b * slabs) * NF_SLAB;
        const int e1 = min(e0 + NF_SLAB, s.n_half);
        const int64_t fb = b * s.n_full;
        const float gl = gld[b];
        for (int e = e0 + threadIdx.x; e < e1; e += NF_BLOCK) {
            const int o0 = nf_half_to_full(s, 0, e);
            const float g0 = gy[fb + o0];
            float g_z0, g_sr;
            nf_affine_bwd_elem(g0, gl, z[fb + o0], sp[b * pbs + e], a, c, g_z0, g_sr, acc_a, acc_c);
            gz[fb + o0] = g_z0;
            gt[b * pbs + e] = g0;
            gs[b * pbs + e] = g_sr;
            if (has_pass) {
                const int o1 = nf_half_to_full(s, 1, e);
                gz[fb + o1] = gy[fb + o1];
            }
        }
    }
    const float ta = nf_block_sum(acc_a, scratch);
    const float tc = nf_block_sum(acc_c, scratch);
    if (threadIdx.x == 0) {
        atomicAdd(g_scale, ta);
        atomicAdd(g_bias, tc);
    }
}

// ---------------------------------------------------------------------------------------------------------------
extern "C" int nf_affine_coupling_fwd(const float* z, const float* t_ptr, const float* s_ptr, int64_t param_bstride,
                                      const float* s_log_scale, const float* s_bias, float* y, float* ld, int mode,
                                      int odd, int inverse, int64_t B, int C, int H, int W, nf_stream_t stream) {
    NfSplit s;
    if (!nf_make_split(s, mode, odd, C, H, W)) return NF_E_BADARG;
    if (B == 0 || s.n_half == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (s.n_half <= NF_ROWS_MAX) {
        dim3 grid(nf_grid_for(B)), block(NF_BLOCK);
        if (inverse) hipLaunchKernelGGL(k_affine_rows_fwd<true>, grid, block, 0, st, z, t_ptr, s_ptr, param_bstride, s_log_scale, s_bias, y, ld, s, B);
        else hipLaunchKernelGGL(k_affine_rows_fwd<false>, grid, block, 0, st, z, t_ptr, s_ptr, param_bstride, s_log_scale, s_bias, y, ld, s, B);
    } else {
        if (B > 0x7fffffffLL) return NF_E_BADARG;
        dim3 grid((unsigned)B, (unsigned)((s.n_half + NF_SLAB - 1) / NF_SLAB)), block(NF_BLOCK);
        if (inverse) hipLaunchKernelGGL(k_affine_slab_fwd<true>, grid, block, 0, st, z, t_ptr, s_ptr, param_bstride, s_log_scale, s_bias, y, ld, s);
        else hipLaunchKernelGGL(k_affine_slab_fwd<false>, grid, block, 0, st, z, t_ptr, s_ptr, param_bstride, s_log_scale, s_bias, y, ld, s);
    }
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_affine_coupling_bwd(const float* g_y, const float* g_ld, const float* z, const float* t_ptr,
                                      const float* s_ptr, int64_t param_bstride, const float* s_log_scale,
                                      const float* s_bias, float* g_z, float* g_t, float* g_s, float* g_scale,
                                      float* g_bias, int mode, int odd, int64_t B, int C, int H, int W,
                                      nf_stream_t stream) {
    (void)t_ptr;  // the shift does not enter any gradient
    NfSplit s;
    if (!nf_make_split(s, mode, odd, C, H, W)) return NF_E_BADARG;
    if (B == 0 || s.n_half == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (s.n_half <= NF_ROWS_MAX) {
        unsigned g = nf_grid_for(B);
        if (g > NF_ATOMIC_GRID) g = NF_ATOMIC_GRID;
        hipLaunchKernelGGL(k_affine_rows_bwd, dim3(g), dim3(NF_BLOCK), 0, st, g_y, g_ld, z, s_ptr, param_bstride,
                           s_log_scale, s_bias, g_z, g_t, g_s, g_scale, g_bias, s, B);
    } else {
        const int slabs = (s.n_half + NF_SLAB - 1) / NF_SLAB;
        const int64_t items = B * slabs;
        const unsigned g = (unsigned)(items < 2 * NF_ATOMIC_GRID ? items : 2 * NF_ATOMIC_GRID);
        hipLaunchKernelGGL(k_affine_slab_bwd, dim3(g), dim3(NF_BLOCK), 0, st, g_y, g_ld, z, s_ptr, param_bstride,
                           s_log_scale, s_bias, g_z, g_t, g_s, g_scale, g_bias, s, B, slabs);
    }
    NF_CHECK_LAUNCH();
    return 0;
}
