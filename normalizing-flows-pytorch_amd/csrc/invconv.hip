// Glow's invertible 1x1 convolution as a per-pixel small mat-vec.  Reference: flows/modules.py:441-497.
//   apply : y[b,:,p] = M z[b,:,p]    (M = W forward, W^-1 inverse, W^T for the autograd of z), ld += +-P*sum(log_s)
//   wgrad : g_M[r,c] = sum_{b,p} g_y[b,r,p] z[b,c,p]
// HBM-bound (8 B/element; 2C FLOP/element, C <= 48 -> 12 FLOP/B, under the fp32 ridge of ~25 FLOP/B): the matrix is
// read through wave-uniform addresses, so the compiler keeps it on the scalar path (s_load -> SGPR operands of
// v_fmac) and the vector memory pipe carries only z and y, coalesced along the pixel axis of NCHW.
#include "nf_common.h"
#include "nf_det.h"

NF_DET_STATE(nf_ic)
NF_DET_HOST_API(nf_ic)

template <int CT, bool TRANSPOSE>
__global__ void __launch_bounds__(NF_BLOCK) k_invconv_apply(const float* __restrict__ z, const float* __restrict__ M,
                                                            float* __restrict__ y, float* __restrict__ ld,
                                                            const float* __restrict__ log_s, float ld_sign, int64_t B,
                                                            int P) {
    const int64_t npix = B * P;
    const int64_t gstride = (int64_t)gridDim.x * blockDim.x;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t t = gtid; t < npix; t += gstride) {
        const int64_t b = t / P;
        const int64_t base = b * CT * P + (t - b * P);
        float in[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) in[c] = z[base + (int64_t)c * P];
#pragma unroll
        for (int r = 0; r < CT; ++r) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < CT; ++c) acc = fmaf(TRANSPOSE ? M[c * CT + r] : M[r * CT + c], in[c], acc);
            y[base + (int64_t)r * P] = acc;
        }
    }
    if (ld != nullptr) {
        float s = 0.f;
        for (int c = 0; c < CT; ++c) s += log_s[c];
        const float d = ld_sign * (float)P * s;                 // modules.py:479-480, :494-495
        for (int64_t b = gtid; b < B; b += gstride) ld[b] += d;
    }
}

// any C: inputs re-read per output row (L1-resident); correctness fallback for channel counts without a template
template <bool TRANSPOSE>
__global__ void __launch_bounds__(NF_BLOCK) k_invconv_apply_any(const float* __restrict__ z, const float* __restrict__ M,
                                                                float* __restrict__ y, float* __restrict__ ld,
                                                                const float* __restrict__ log_s, float ld_sign,
                                                                int64_t B, int C, int P) {
    const int64_t npix = B * P;
    const int64_t gstride = (int64_t)gridDim.x * blockDim.x;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t t = gtid; t < npix; t += gstride) {
        const int64_t b = t / P;
        const int64_t base = b * C * P + (t - b * P);
        for (int r = 0; r < C; ++r) {
            float acc = 0.f;
            for (int c = 0; c < C; ++c) acc = fmaf(TRANSPOSE ? M[c * C + r] : M[r * C + c], z[base + (int64_t)c * P], acc);
            y[base + (int64_t)r * P] = acc;
        }
    }
    if (ld != nullptr) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += log_s[c];
        const float d = ld_sign * (float)P * s;
        for (int64_t b = gtid; b < B; b += gstride) ld[b] += d;
    }
}

// g_M = sum over pixels of g_y z^T.  A block stages NF_TP pixels x C channels of both operands in LDS (row stride
// NF_TP+1 words: lanes walk the channel axis, +1 makes that conflict-free), every thread owns entries (r,c) of g_M.
#define NF_TP 128
__global__ void __launch_bounds__(NF_BLOCK) k_invconv_wgrad(const float* __restrict__ gy, const float* __restrict__ z,
                                                            float* __restrict__ gM, int64_t B, int C, int P,
                                                            int64_t tiles_per_block) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int RS = NF_TP + 1;
    float* gT = lds;
    float* zT = lds + (size_t)C * RS;
    const int64_t npix = B * P;
    const int n_ent = C * C;
    const int NE = 12;                                   // entries per thread: supports C*C <= 12*256 (C <= 55)
    float acc[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) acc[k] = 0.f;
    const int64_t tile0 = (int64_t)blockIdx.x * tiles_per_block;
    for (int64_t tile = tile0; tile < tile0 + tiles_per_block; ++tile) {
        const int64_t t0 = tile * NF_TP;
        if (t0 >= npix) break;
        const int np = (int)min((int64_t)NF_TP, npix - t0);
        __syncthreads();
        for (int i = threadIdx.x; i < C * NF_TP; i += blockDim.x) {
            const int c = i / NF_TP, q = i - c * NF_TP;
            float a = 0.f, v = 0.f;
            if (q < np) {
                const int64_t t = t0 + q, b = t / P;
                const int64_t addr = (b * C + c) * P + (t - b * P);
                a = gy[addr];
                v = z[addr];
            }
            gT[c * RS + q] = a;
            zT[c * RS + q] = v;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int e = threadIdx.x + k * NF_BLOCK;
            if (e < n_ent) {
                const int r = e / C, c = e - r * C;
                const float* ga = gT + r * RS;
                const float* za = zT + c * RS;
                float s = 0.f;
                for (int q = 0; q < NF_TP; ++q) s = fmaf(ga[q], za[q], s);
                acc[k] += s;
            }
        }
    }
    NF_DET_ENTER_ALL(nf_ic);               // (one thread per entry and workgroup)
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        const int e = threadIdx.x + k * NF_BLOCK;
        if (e < n_ent) atomicAdd(gM + e, acc[k]);
    }
    NF_DET_LEAVE_ALL(nf_ic);
}

// small C (<= 4): every thread keeps the whole C x C partial in registers, one block reduction + C*C atomics per block.
// 256 blocks of 1024 threads (sixteen waves per CU; the kernel ends in same-address atomics), two pixels' loads in flight per trip.
#define NF_IW_BIG 1024
template <int CT>
__global__ void __launch_bounds__(NF_IW_BIG) k_invconv_wgrad_small(const float* __restrict__ gy, const float* __restrict__ z,
                                                                   float* __restrict__ gM, int64_t B, int P) {
    __shared__ float scratch[NF_IW_BIG / NF_WAVE];
    float acc[CT][CT];
#pragma unroll
    for (int r = 0; r < CT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[r][c] = 0.f;
    const int64_t npix = B * P;
    const int64_t gstride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < npix; t += 2 * gstride) {
        const int64_t t2 = t + gstride < npix ? t + gstride : t;
        const float w2 = t + gstride < npix ? 1.f : 0.f;
        const int64_t b = t / P, b2 = t2 / P;
        const int64_t base = b * CT * P + (t - b * P), base2 = b2 * CT * P + (t2 - b2 * P);
        float g[CT], v[CT], g2[CT], v2[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            g[c] = gy[base + (int64_t)c * P]; v[c] = z[base + (int64_t)c * P];
            g2[c] = gy[base2 + (int64_t)c * P] * w2; v2[c] = z[base2 + (int64_t)c * P];
        }
#pragma unroll
        for (int r = 0; r < CT; ++r)
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[r][c] = fmaf(g2[r], v2[c], fmaf(g[r], v[c], acc[r][c]));
    }
#pragma unroll
    for (int r = 0; r < CT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const float t = nf_block_sum(acc[r][c], scratch);
            if (threadIdx.x == 0) { NF_DET_ENTER(nf_ic); atomicAdd(gM + r * CT + c, t); NF_DET_LEAVE(nf_ic); }
        }
}

// ---------------------------------------------------------------------------------------------------------------
// PLU weight assembly (modules.py:471-473) and its autograd, one workgroup (C <= 64: three C x C tiles in LDS).
//   W = P L' U',  L' = L o L_mask + I,  U' = U o U_mask + diag(sign_s exp(log_s))
//   g_L = (P^T g_W U'^T) o L_mask,  g_U = (L'^T P^T g_W) o U_mask,
//   g_log_s[i] = (L'^T P^T g_W)[i][i] sign_s[i] exp(log_s[i]) + pixels * sum_b g_ld[b]      (appendix B3)
#define NF_PLU_MAXC 64
__device__ __forceinline__ void nf_plu_weight_fwd_body(float* sm, const float* __restrict__ Pm, const float* __restrict__ L,
                                                       const float* __restrict__ U, const float* __restrict__ Lmask,
                                                       const float* __restrict__ Umask, const float* __restrict__ sign_s,
                                                       const float* __restrict__ log_s, float* __restrict__ W, int C) {
    float* Lp = sm;
    float* Up = sm + C * C;
    float* T = sm + 2 * C * C;
    float* Ps = Lp;                                              // P re-uses L' once T = L' U' exists (every operand of the
    for (int e = threadIdx.x; e < C * C; e += blockDim.x) {     // inner loops sits in LDS: a global load per k was ~1 us each)
        const int r = e / C, c = e - r * C;
        Lp[e] = L[e] * Lmask[e] + (r == c ? 1.f : 0.f);
        Up[e] = U[e] * Umask[e] + (r == c ? sign_s[r] * expf(log_s[r]) : 0.f);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < C * C; e += blockDim.x) {
        const int r = e / C, c = e - r * C;
        float acc = 0.f;
        for (int k = 0; k < C; ++k) acc = fmaf(Lp[r * C + k], Up[k * C + c], acc);
        T[e] = acc;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < C * C; e += blockDim.x) Ps[e] = Pm[e];
    __syncthreads();
    for (int e = threadIdx.x; e < C * C; e += blockDim.x) {
        const int r = e / C, c = e - r * C;
        float acc = 0.f;
        for (int k = 0; k < C; ++k) acc = fmaf(Ps[r * C + k], T[k * C + c], acc);
        W[e] = acc;
    }
}
__global__ void __launch_bounds__(NF_BLOCK) k_invconv_weight_fwd(const float* __restrict__ Pm, const float* __restrict__ L,
                                                                 const float* __restrict__ U, const float* __restrict__ Lmask,
                                                                 const float* __restrict__ Umask,
                                                                 const float* __restrict__ sign_s,
                                                                 const float* __restrict__ log_s, float* __restrict__ W, int C) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    nf_plu_weight_fwd_body(sm, Pm, L, U, Lmask, Umask, sign_s, log_s, W, C);
}

__device__ __forceinline__ void nf_plu_weight_bwd_body(float* sm, float* scratch, float* sum_gld_p, const float* __restrict__ gW,
                                                       const float* __restrict__ Pm, const float* __restrict__ L,
                                                       const float* __restrict__ U, const float* __restrict__ Lmask,
                                                       const float* __restrict__ Umask, const float* __restrict__ sign_s,
                                                       const float* __restrict__ log_s, const float* __restrict__ gld,
                                                       float* __restrict__ gL, float* __restrict__ gU, float* __restrict__ glog_s,
                                                       int accumulate, int C, int64_t B, float pixels) {
    float& sum_gld = *sum_gld_p;
    float* Lp = sm;
    float* UpT = sm + C * C;                                     // U' TRANSPOSED: the product below walks U' rows with the lane
                                                                 // index as the ROW -- a stride of C floats (C = 48: two banks)
    float* A = sm + 2 * C * C;                                   // P^T g_W
    float* Ps = sm + 3 * C * C;                                  // P and g_W staged: no global load inside the k loops
    float* Gs = sm + 4 * C * C;
    for (int e = threadIdx.x; e < C * C; e += blockDim.x) {
        const int r = e / C, c = e - r * C;
        Lp[e] = L[e] * Lmask[e] + (r == c ? 1.f : 0.f);
        UpT[c * C + r] = U[e] * Umask[e] + (r == c ? sign_s[r] * expf(log_s[r]) : 0.f);
        Ps[e] = Pm[e];
        Gs[e] = gW[e];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < C * C; e += blockDim.x) {
        const int r = e / C, c = e - r * C;
        float acc = 0.f;
        for (int k = 0; k < C; ++k) acc = fmaf(Ps[k * C + r], Gs[k * C + c], acc);
        A[e] = acc;
    }
    float part = 0.f;
    if (gld != nullptr)
        for (int64_t b = threadIdx.x; b < B; b += blockDim.x) part += gld[b];
    const float tot = nf_block_sum(part, scratch);
    if (threadIdx.x == 0) sum_gld = tot;
    __syncthreads();
    for (int e = threadIdx.x; e < C * C; e += blockDim.x) {
        const int r = e / C, c = e - r * C;
        float gl = 0.f, gu = 0.f;
        for (int k = 0; k < C; ++k) {
            gl = fmaf(A[r * C + k], UpT[k * C + c], gl);         // (A U'^T)[r][c]
            gu = fmaf(Lp[k * C + r], A[k * C + c], gu);          // (L'^T A)[r][c]
        }
        gL[e] = (accumulate ? gL[e] : 0.f) + gl * Lmask[e];
        gU[e] = (accumulate ? gU[e] : 0.f) + gu * Umask[e];
        if (r == c) glog_s[r] = (accumulate ? glog_s[r] : 0.f) + gu * sign_s[r] * expf(log_s[r]) + pixels * sum_gld;
    }
}
__global__ void __launch_bounds__(NF_BLOCK) k_invconv_weight_bwd(const float* __restrict__ gW, const float* __restrict__ Pm,
                                                                 const float* __restrict__ L, const float* __restrict__ U,
                                                                 const float* __restrict__ Lmask,
                                                                 const float* __restrict__ Umask,
                                                                 const float* __restrict__ sign_s,
                                                                 const float* __restrict__ log_s, const float* __restrict__ gld,
                                                                 float* __restrict__ gL, float* __restrict__ gU,
                                                                 float* __restrict__ glog_s, int accumulate, int C, int64_t B, float pixels) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float scratch[NF_BLOCK / NF_WAVE];
    __shared__ float sum_gld;
    nf_plu_weight_bwd_body(sm, scratch, &sum_gld, gW, Pm, L, U, Lmask, Umask, sign_s, log_s, gld, gL, gU, glog_s, accumulate, C, B,
                           pixels);
}

// ---- every invertible 1x1 convolution of a model in a few launches: workgroup = layer (an image Glow has 129 of them, and a
//      single-workgroup launch of three C x C x C products is 14 + 24 us of pure latency each) -------------------------------------
// (1 024 threads per layer since round 5: the products are C^3 / threads dependent LDS -> FMA steps per thread -- 432 at C = 48 with 256
//  threads, 19 / 23.5 us per launch, 14 launches per CIFAR Glow step -- and the other 230 compute units have nothing to do meanwhile)
#define NF_PLU_MULTI_THREADS 1024
struct NfPluArgs { nf_plu_desc d[NF_PLU_MAX_LAYERS]; };
__global__ void __launch_bounds__(NF_PLU_MULTI_THREADS) k_invconv_weight_fwd_multi(NfPluArgs args) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const nf_plu_desc& d = args.d[blockIdx.x];
    nf_plu_weight_fwd_body(sm, d.P, d.L, d.U, d.L_mask, d.U_mask, d.sign_s, d.log_s, d.W, d.C);
}
__global__ void __launch_bounds__(NF_PLU_MULTI_THREADS) k_invconv_weight_bwd_multi(NfPluArgs args) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float scratch[NF_PLU_MULTI_THREADS / NF_WAVE];
    __shared__ float sum_gld;
    const nf_plu_desc& d = args.d[blockIdx.x];
    nf_plu_weight_bwd_body(sm, scratch, &sum_gld, d.g_W, d.P, d.L, d.U, d.L_mask, d.U_mask, d.sign_s, d.log_s, d.g_ld, d.g_L, d.g_U,
                           d.g_log_s, d.accumulate, d.C, d.B, d.pixels);
}

// ---------------------------------------------------------------------------------------------------------------
template <bool TR>
static void nf_launch_apply(int C, dim3 grid, hipStream_t st, const float* z, const float* M, float* y, float* ld,
                            const float* log_s, float ld_sign, int64_t B, int P) {
#define NF_CASE(CT)                                                                                                   \
    case CT:                                                                                                          \
        hipLaunchKernelGGL((k_invconv_apply<CT, TR>), grid, dim3(NF_BLOCK), 0, st, z, M, y, ld, log_s, ld_sign, B, P); \
        break;
    switch (C) {
        NF_CASE(1) NF_CASE(2) NF_CASE(3) NF_CASE(4) NF_CASE(6) NF_CASE(8) NF_CASE(12) NF_CASE(16) NF_CASE(24)
        NF_CASE(32) NF_CASE(48)
        default:
            hipLaunchKernelGGL(k_invconv_apply_any<TR>, grid, dim3(NF_BLOCK), 0, st, z, M, y, ld, log_s, ld_sign, B, C, P);
    }
#undef NF_CASE
}

// fp32-MFMA forms for 9 <= C <= 64 (invconv_mfma.hip); return 1 when they took the launch
__attribute__((visibility("hidden"))) int nf_invconv_apply_mfma_try(const float* z, const float* M, int transpose, float* y, float* ld,
                                         const float* log_s, float ld_sign, int64_t B, int C, int P, void* stream);
__attribute__((visibility("hidden"))) int nf_invconv_wgrad_mfma_try(const float* g_y, const float* z, float* g_M, int64_t B, int C, int P,
                                         void* stream);

extern "C" int nf_invconv_apply(const float* z, const float* M, int transpose, float* y, float* ld, const float* log_s,
                                float ld_sign, int64_t B, int C, int P, nf_stream_t stream) {
    if (C <= 0 || P <= 0 || C > 1024) return NF_E_BADARG;
    if (ld != nullptr && log_s == nullptr) return NF_E_BADARG;
    if (B == 0) return 0;
    if (nf_invconv_apply_mfma_try(z, M, transpose, y, ld, log_s, ld_sign, B, C, P, stream) == 1) {
        NF_CHECK_LAUNCH();
        return 0;
    }
    unsigned g = nf_grid_for(B * P);
    const unsigned g_ld = nf_grid_for(B);
    if (ld != nullptr && g < g_ld) g = g_ld;
    if (transpose) nf_launch_apply<true>(C, dim3(g), (hipStream_t)stream, z, M, y, ld, log_s, ld_sign, B, P);
    else nf_launch_apply<false>(C, dim3(g), (hipStream_t)stream, z, M, y, ld, log_s, ld_sign, B, P);
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_invconv_wgrad(const float* g_y, const float* z, float* g_M, int64_t B, int C, int P,
                                nf_stream_t stream) {
    if (C <= 0 || P <= 0) return NF_E_BADARG;
    if (B == 0) return 0;
    if (nf_invconv_wgrad_mfma_try(g_y, z, g_M, B, C, P, stream) == 1) {
        NF_CHECK_LAUNCH();
        return 0;
    }
    if (C <= 4) {
        unsigned g = nf_grid_for(B * P, NF_IW_BIG * 2);
        if (g > 256) g = 256;
        hipStream_t st = (hipStream_t)stream;
        switch (C) {
            case 1: hipLaunchKernelGGL(k_invconv_wgrad_small<1>, dim3(g), dim3(NF_IW_BIG), 0, st, g_y, z, g_M, B, P); break;
            case 2: hipLaunchKernelGGL(k_invconv_wgrad_small<2>, dim3(g), dim3(NF_IW_BIG), 0, st, g_y, z, g_M, B, P); break;
            case 3: hipLaunchKernelGGL(k_invconv_wgrad_small<3>, dim3(g), dim3(NF_IW_BIG), 0, st, g_y, z, g_M, B, P); break;
            default: hipLaunchKernelGGL(k_invconv_wgrad_small<4>, dim3(g), dim3(NF_IW_BIG), 0, st, g_y, z, g_M, B, P); break;
        }
        NF_CHECK_LAUNCH();
        return 0;
    }
    if (C * C > 12 * NF_BLOCK) return NF_E_UNSUPPORTED;
    const int64_t npix = B * P;
    const int64_t tiles = (npix + NF_TP - 1) / NF_TP;
    int64_t blocks = tiles < 512 ? tiles : 512;
    const int64_t tpb = (tiles + blocks - 1) / blocks;
    blocks = (tiles + tpb - 1) / tpb;
    const size_t lds = (size_t)2 * C * (NF_TP + 1) * sizeof(float);
    hipLaunchKernelGGL(k_invconv_wgrad, dim3((unsigned)blocks), dim3(NF_BLOCK), lds, (hipStream_t)stream, g_y, z, g_M, B,
                       C, P, tpb);
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_invconv_weight_fwd(const float* P, const float* L, const float* U, const float* L_mask,
                                     const float* U_mask, const float* sign_s, const float* log_s, float* W, int C,
                                     nf_stream_t stream) {
    if (C <= 0 || C > NF_PLU_MAXC) return C <= 0 ? NF_E_BADARG : NF_E_UNSUPPORTED;
    hipLaunchKernelGGL(k_invconv_weight_fwd, dim3(1), dim3(NF_BLOCK), (size_t)3 * C * C * sizeof(float), (hipStream_t)stream,
                       P, L, U, L_mask, U_mask, sign_s, log_s, W, C);
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_invconv_weight_bwd(const float* g_W, const float* P, const float* L, const float* U,
                                     const float* L_mask, const float* U_mask, const float* sign_s, const float* log_s,
                                     const float* g_ld, float* g_L, float* g_U, float* g_log_s, int accumulate, int C, int64_t B,
                                     int pixels, nf_stream_t stream) {
    if (C <= 0 || C > NF_PLU_MAXC) return C <= 0 ? NF_E_BADARG : NF_E_UNSUPPORTED;
    const size_t lds_b = (size_t)5 * C * C * sizeof(float);      // 80 KB at C = 64: above the 64 KB default, opt in once
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_invconv_weight_bwd, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           5 * NF_PLU_MAXC * NF_PLU_MAXC * (int)sizeof(float));
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_invconv_weight_bwd, dim3(1), dim3(NF_BLOCK), lds_b, (hipStream_t)stream,
                       g_W, P, L, U, L_mask, U_mask, sign_s, log_s, g_ld, g_L, g_U, g_log_s, accumulate, C, B, (float)pixels);
    NF_CHECK_LAUNCH();
    return 0;
}

static int nf_plu_multi_check(const nf_plu_desc* descs, int n, int& cmax) {
    if (descs == nullptr || n < 1 || n > NF_PLU_MAX_LAYERS) return NF_E_BADARG;
    cmax = 0;
    for (int i = 0; i < n; ++i) {
        if (descs[i].C <= 0) return NF_E_BADARG;
        if (descs[i].C > NF_PLU_MAXC) return NF_E_UNSUPPORTED;
        if (descs[i].C > cmax) cmax = descs[i].C;
    }
    return 0;
}

extern "C" int nf_invconv_weight_fwd_multi(const nf_plu_desc* descs, int n_layers, nf_stream_t stream) {
    int cmax;
    const int rc = nf_plu_multi_check(descs, n_layers, cmax);
    if (rc) return rc;
    NfPluArgs args;
    for (int i = 0; i < n_layers; ++i) args.d[i] = descs[i];
    hipLaunchKernelGGL(k_invconv_weight_fwd_multi, dim3((unsigned)n_layers), dim3(NF_PLU_MULTI_THREADS), (size_t)3 * cmax * cmax * sizeof(float),
                       (hipStream_t)stream, args);
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_invconv_weight_bwd_multi(const nf_plu_desc* descs, int n_layers, nf_stream_t stream) {
    int cmax;
    const int rc = nf_plu_multi_check(descs, n_layers, cmax);
    if (rc) return rc;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_invconv_weight_bwd_multi, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           5 * NF_PLU_MAXC * NF_PLU_MAXC * (int)sizeof(float));
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    NfPluArgs args;
    for (int i = 0; i < n_layers; ++i) args.d[i] = descs[i];
    hipLaunchKernelGGL(k_invconv_weight_bwd_multi, dim3((unsigned)n_layers), dim3(NF_PLU_MULTI_THREADS), (size_t)5 * cmax * cmax * sizeof(float),
                       (hipStream_t)stream, args);
    NF_CHECK_LAUNCH();
    return 0;
}

