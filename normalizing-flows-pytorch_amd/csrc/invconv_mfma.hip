// Invertible 1x1 convolution for 12 <= C <= 64 on the fp32 matrix cores (v_mfma_f32_16x16x4_f32).
//   apply : y[:, pix] = M z[:, pix]   -- A = M (held in registers for the whole kernel), B = z with the 16 pixels of a
//           block across lanes (coalesced along the NCHW pixel axis), D rows = output channels.
//   wgrad : g_M = sum_pix g_y[:, pix] z[:, pix]^T -- both operands staged through LDS (coalesced global reads), every
//           wave accumulates all (C/16)^2 output tiles over its quarter of the pixel tile.
// Same arithmetic as the scalar kernels (exact fp32, k-ordered FMA chains); what changes is the issue cost: the scalar
// kernel spends its time on 2304 scalar-operand FMAs + s_loads per pixel at C = 48 (1.1 TB/s), the MFMA form leaves the
// kernel HBM-bound (8 B/element).
#include "nf_common.h"
#include "nf_det.h"

NF_DET_STATE(nf_icm)
NF_DET_HOST_API(nf_icm)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------
template <int RT, int KQ, bool TRANSPOSE>
__global__ void __launch_bounds__(NF_BLOCK) k_invconv_apply_mfma(const float* __restrict__ z, const float* __restrict__ M,
                                                                 float* __restrict__ y, float* __restrict__ ld,
                                                                 const float* __restrict__ log_s, float ld_sign, int64_t B,
                                                                 int C, int P) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    // A fragments: A[i = li][k = lk] of row tile rt, k-step q  ->  M[16 rt + li][4 q + lk]   (or M^T)
    float a[RT][KQ];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const int r = 16 * rt + li, c = 4 * q + lk;
            a[rt][q] = (r < C && c < C) ? (TRANSPOSE ? M[c * C + r] : M[r * C + c]) : 0.f;
        }
    const int64_t nblk = B * (P / 16);                       // 16-pixel blocks (P % 16 == 0: never straddle samples)
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int bpp = P / 16;
    for (int64_t blk = wave; blk < nblk; blk += nwaves) {
        const int64_t b = blk / bpp;
        const int p0 = (int)(blk - b * bpp) * 16;
        const float* zb = z + b * C * P + p0 + li;
        float* yb = y + b * C * P + p0 + li;
        f32x4 acc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float bv[KQ];
#pragma unroll
        for (int q = 0; q < KQ; ++q) {                       // B[k = lk][j = li] = z[c = 4q + lk][pixel p0 + li]
            const int c = 4 * q + lk;
            bv[q] = c < C ? zb[(int64_t)c * P] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < KQ; ++q)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][q], bv[q], acc[rt], 0, 0, 0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {                    // D: col = li (pixel), row = 4 lk + j
                const int r = 16 * rt + 4 * lk + j;
                if (r < C) yb[(int64_t)r * P] = acc[rt][j];
            }
    }
    if (ld != nullptr) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += log_s[c];
        const float d = ld_sign * (float)P * s;
        const int64_t gstride = (int64_t)gridDim.x * blockDim.x;
        for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gstride) ld[b] += d;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Large-batch form: a wave owns 64 consecutive pixels of a sample.  Lane (lk, li) reads ONE float4 per k-step (channel 4q + lk,
// pixels 4 li .. 4 li + 3: 256 contiguous bytes per channel row and instruction instead of 64) and treats its four components as four
// 16-pixel MFMA blocks {4 li + pb}; the D fragments of the four blocks hold, per lane and output row, exactly the four consecutive
// pixels of a float4 store.  Same products, same k order as the 16-pixel kernel: the bits do not change.
template <int RT, int KQ, bool TRANSPOSE>
__global__ void __launch_bounds__(NF_BLOCK, 2) k_invconv_apply_mfma4(const float* __restrict__ z, const float* __restrict__ M,
                                                                  float* __restrict__ y, float* __restrict__ ld,
                                                                  const float* __restrict__ log_s, float ld_sign, int64_t B,
                                                                  int C, int P) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    float a[RT][KQ];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const int r = 16 * rt + li, c = 4 * q + lk;
            a[rt][q] = (r < C && c < C) ? (TRANSPOSE ? M[c * C + r] : M[r * C + c]) : 0.f;
        }
    const int bpp = P / 64;
    const int64_t nblk = B * bpp;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    // software pipeline: the loads of block i + 1 are in flight while the 12 RT KQ matrix instructions of block i issue
    f32x4 bv[KQ], nx[KQ];
    auto fetch = [&](int64_t blk, f32x4* dst) {
        const bool ok = blk < nblk;
        const int64_t b = ok ? blk / bpp : 0;
        const float* zb = z + b * C * P + (int)(ok ? blk - b * bpp : 0) * 64 + 4 * li;
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const int c = 4 * q + lk;
            dst[q] = (ok && c < C) ? *reinterpret_cast<const f32x4*>(zb + (int64_t)c * P) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    fetch(wave, nx);
    for (int64_t blk = wave; blk < nblk; blk += nwaves) {
        const int64_t b = blk / bpp;
        const int p0 = (int)(blk - b * bpp) * 64 + 4 * li;
        float* yb = y + b * C * P + p0;
#pragma unroll
        for (int q = 0; q < KQ; ++q) bv[q] = nx[q];
        fetch(blk + nwaves, nx);
        f32x4 acc[4][RT];
#pragma unroll
        for (int pb = 0; pb < 4; ++pb)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[pb][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < KQ; ++q)
#pragma unroll
            for (int pb = 0; pb < 4; ++pb)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    acc[pb][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][q], bv[q][pb], acc[pb][rt], 0, 0, 0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 16 * rt + 4 * lk + j;
                if (r < C)
                    *reinterpret_cast<f32x4*>(yb + (int64_t)r * P) = (f32x4){acc[0][rt][j], acc[1][rt][j], acc[2][rt][j], acc[3][rt][j]};
            }
    }
    if (ld != nullptr) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += log_s[c];
        const float d = ld_sign * (float)P * s;
        const int64_t gstride = (int64_t)gridDim.x * blockDim.x;
        for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gstride) ld[b] += d;
    }
}

// ---------------------------------------------------------------------------------------------------------------
#define NF_WTP 128  // pixels per staged tile
template <int RT>
__global__ void __launch_bounds__(NF_BLOCK) k_invconv_wgrad_mfma(const float* __restrict__ gy, const float* __restrict__ z,
                                                                 float* __restrict__ gM, int64_t B, int C, int P,
                                                                 int64_t tiles_per_block) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int RS = NF_WTP + 1;
    const int CP = RT * 16;
    float* gT = lds;                       // [CP][RS]
    float* zT = lds + (size_t)CP * RS;     // [CP][RS]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, li = lane & 15, lk = lane >> 4;
    const int64_t npix = B * P;
    f32x4 acc[RT][RT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < RT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // Staging: a thread owns ONE pixel column of the tile (256 threads = 128 pixels x 2 channel phases) and the channels c = phase,
    // phase + 2, ...; the loads of tile i + 1 are issued into registers before the MFMAs of tile i (the kernel is a stream of
    // 8 bytes per element: what limits it is how much of that stream is in flight).
    constexpr int NCH = RT * 16 / (NF_BLOCK / NF_WTP);     // channel rows per thread
    const int q = threadIdx.x & (NF_WTP - 1), ph = threadIdx.x >> 7;
    float ra[NCH], rz[NCH];
    const int64_t tile0 = (int64_t)blockIdx.x * tiles_per_block;
    auto fetch = [&](int64_t tile) {
        const int64_t t = tile * NF_WTP + q;
        const bool ok = tile < tile0 + tiles_per_block && t < npix;
        const int64_t b = ok ? t / P : 0;
        const int64_t base = b * C * P + (ok ? t - b * P : 0);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = ph + 2 * k;
            const bool in = ok && c < C;
            ra[k] = in ? gy[base + (int64_t)c * P] : 0.f;
            rz[k] = in ? z[base + (int64_t)c * P] : 0.f;
        }
    };
    fetch(tile0);
    for (int64_t tile = tile0; tile < tile0 + tiles_per_block; ++tile) {
        if (tile * NF_WTP >= npix) break;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            gT[(ph + 2 * k) * RS + q] = ra[k];
            zT[(ph + 2 * k) * RS + q] = rz[k];
        }
        __syncthreads();
        fetch(tile + 1);
        // this wave's quarter of the tile: pixels [32 wid, 32 wid + 32), 8 k-steps of 4 pixels
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int pix = 32 * wid + 4 * ks + lk;
            float av[RT], bv[RT];
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                av[i] = gT[(16 * i + li) * RS + pix];            // A[i = r][k = pix] = g_y[r][pix]
                bv[i] = zT[(16 * i + li) * RS + pix];            // B[k = pix][j = c] = z[c][pix]
            }
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < RT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    }
    // cross-wave reduction through LDS, then one atomic per entry per block
    __syncthreads();
    float* red = lds;                      // [4][CP][CP]  (CP*CP*4 <= 2*CP*RS for CP <= 64)
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < RT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)              // D: row = 4 lk + e (r), col = li (c)
                red[(wid * CP + 16 * i + 4 * lk + e) * CP + 16 * j + li] = acc[i][j][e];
    __syncthreads();
    NF_DET_ENTER_ALL(nf_icm);              // (one thread per entry and workgroup)
    for (int e = threadIdx.x; e < C * C; e += blockDim.x) {
        const int r = e / C, c = e - r * C;
        const float t = red[(0 * CP + r) * CP + c] + red[(1 * CP + r) * CP + c] + red[(2 * CP + r) * CP + c] +
                        red[(3 * CP + r) * CP + c];
        atomicAdd(gM + e, t);
    }
    NF_DET_LEAVE_ALL(nf_icm);
}

// ---------------------------------------------------------------------------------------------------------------
template <bool TR>
static bool nf_launch_apply_mfma(int C, dim3 grid, hipStream_t st, const float* z, const float* M, float* y, float* ld,
                                 const float* log_s, float ld_sign, int64_t B, int P, bool wide) {
    const int rt = (C + 15) / 16, kq = (C + 3) / 4;
#define NF_CASE(RT, KQ)                                                                                                \
    if (rt == RT && kq == KQ) {                                                                                        \
        if (wide)                                                                                                      \
            hipLaunchKernelGGL((k_invconv_apply_mfma4<RT, KQ, TR>), grid, dim3(NF_BLOCK), 0, st, z, M, y, ld, log_s, ld_sign, B, C, P); \
        else                                                                                                           \
            hipLaunchKernelGGL((k_invconv_apply_mfma<RT, KQ, TR>), grid, dim3(NF_BLOCK), 0, st, z, M, y, ld, log_s, ld_sign, B, C, P); \
        return true;                                                                                                   \
    }
    NF_CASE(1, 3) NF_CASE(1, 4) NF_CASE(2, 5) NF_CASE(2, 6) NF_CASE(2, 7) NF_CASE(2, 8) NF_CASE(3, 9) NF_CASE(3, 10)
    NF_CASE(3, 11) NF_CASE(3, 12) NF_CASE(4, 13) NF_CASE(4, 14) NF_CASE(4, 15) NF_CASE(4, 16)
#undef NF_CASE
    return false;
}

// returns 1 when the MFMA path took the launch, 0 when the caller must use the scalar kernels, < 0 never
__attribute__((visibility("hidden"))) int nf_invconv_apply_mfma_try(const float* z, const float* M, int transpose, float* y, float* ld,
                                         const float* log_s, float ld_sign, int64_t B, int C, int P, void* stream) {
    if (C < 9 || C > 64 || (P % 16) != 0 || B == 0) return 0;
    // 64-pixel blocks once there are enough of them to give every SIMD of the chip two waves
    const bool wide = (P % 64) == 0 && B * (P / 64) >= 2048 && ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    const int64_t nblk = wide ? B * (P / 64) : B * (P / 16);
    int64_t g = (nblk + 3) / 4;                              // 4 waves per block, >= 1 pixel block per wave
    if (g > 2048) g = 2048;
    if (wide && g > 256) g = 256;                            // one workgroup per CU, every wave pipelines over its blocks (probe:
                                                             // tools/probes/invconv_apply_probe.hip -- 256: 32 us, 512: 40, 2048: 48 at B = 8192)
    const int64_t g_ld = (B + NF_BLOCK - 1) / NF_BLOCK;
    if (ld != nullptr && g < g_ld) g = g_ld > 4096 ? 4096 : g_ld;
    const bool ok = transpose ? nf_launch_apply_mfma<true>(C, dim3((unsigned)g), (hipStream_t)stream, z, M, y, ld, log_s, ld_sign, B, P, wide)
                              : nf_launch_apply_mfma<false>(C, dim3((unsigned)g), (hipStream_t)stream, z, M, y, ld, log_s, ld_sign, B, P, wide);
    return ok ? 1 : 0;
}

__attribute__((visibility("hidden"))) int nf_invconv_wgrad_mfma_try(const float* g_y, const float* z, float* g_M, int64_t B, int C, int P,
                                         void* stream) {
    if (C < 9 || C > 64 || B == 0) return 0;
    const int rt = (C + 15) / 16;
    const int64_t npix = B * P;
    const int64_t tiles = (npix + NF_WTP - 1) / NF_WTP;
    const int cap = 512;
    int64_t blocks = tiles < cap ? tiles : cap;             // (the kernel ends in C * C same-address atomics per block)
    const int64_t tpb = (tiles + blocks - 1) / blocks;
    blocks = (tiles + tpb - 1) / tpb;
    const size_t lds = (size_t)2 * rt * 16 * (NF_WTP + 1) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    switch (rt) {
        case 1: hipLaunchKernelGGL(k_invconv_wgrad_mfma<1>, dim3((unsigned)blocks), dim3(NF_BLOCK), lds, st, g_y, z, g_M, B, C, P, tpb); break;
        case 2: hipLaunchKernelGGL(k_invconv_wgrad_mfma<2>, dim3((unsigned)blocks), dim3(NF_BLOCK), lds, st, g_y, z, g_M, B, C, P, tpb); break;
        case 3: hipLaunchKernelGGL(k_invconv_wgrad_mfma<3>, dim3((unsigned)blocks), dim3(NF_BLOCK), lds, st, g_y, z, g_M, B, C, P, tpb); break;
        case 4: hipLaunchKernelGGL(k_invconv_wgrad_mfma<4>, dim3((unsigned)blocks), dim3(NF_BLOCK), lds, st, g_y, z, g_M, B, C, P, tpb); break;
        default: return 0;
    }
    return 1;
}
