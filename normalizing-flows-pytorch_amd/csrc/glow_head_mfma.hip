// The head of a Glow flow step on 9 <= C <= 64 channels of image data -- ActNorm, invertible 1x1 convolution and the gather of the
// coupling's conditioner input (flows/modules.py:246-249, :470-482; flows/coupling.py:33) -- as ONE launch per direction on the fp32
// matrix cores.  (glow_head.hip is the C <= 4 form with the PLU product assembled in-kernel; here W arrives assembled, see
// nf_invconv_weight_fwd_multi.)  The three layers are HBM-bound elementwise / small-matrix work: fused, x is read once and the
// ActNorm output never exists in memory.
//   forward :  a = (x - bias) / exp(log_scale)  on the way into the B operand;  h = W a  (v_mfma_f32_16x16x4_f32, W in registers);
//              h -> global, the untouched half of the split map -> z1c;  ld += P (sum log_s - sum log_scale)
//   backward:  per 128-pixel tile, g_h and a staged through LDS:  g_W += g_h a^T  (all (C/16)^2 tiles per wave over its quarter of
//              the pixels),  g_a = W^T g_h,  g_x = g_a / exp(log_scale),  g_log_scale -= sum g_a a (+ P sum g_ld),
//              g_bias -= sum g_a / exp(log_scale)
#include "nf_common.h"
#include "nf_det.h"

NF_DET_STATE(nf_ghm)
NF_DET_HOST_API(nf_ghm)

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef NF_GH_PROF     // phase stamps of workgroup 0 (tools/probes/head_prof.py builds this file with -DNF_GH_PROF=1; 100 MHz wall clock)
__device__ long long nf_gh_prof[16];
extern "C" int nf_gh_prof_read(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(nf_gh_prof), sizeof(long long) * 16); }
#define NF_GH_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) nf_gh_prof[i] = wall_clock64(); } while (0)
#else
#define NF_GH_STAMP(i)
#endif

// which half (0 = transformed, 1 = the conditioner's input) full-tensor element (c, y, x) belongs to and its offset inside the half
__device__ __forceinline__ void nf_gh_full_to_half(const NfSplit& s, int c, int p, int y, int x, int& which, int& e) {
    if (s.mode == NF_SPLIT_CHANNEL) {
        const int hc = s.C >> 1, sel = c >= hc ? 1 : 0;
        which = sel ^ s.odd;
        e = (c - sel * hc) * (s.H * s.W) + p;
    } else {                                                // NF_SPLIT_CHECKER, squeeze.py:36-41 (k / C without the division)
        const int k = 4 * c + 2 * (y & 1) + (x & 1);
        const int q = (k >= s.C ? 1 : 0) + (k >= 2 * s.C ? 1 : 0) + (k >= 3 * s.C ? 1 : 0);
        const int sel = (q == 1 || q == 2) ? 1 : 0;
        const int m = sel ? k - s.C : (q == 0 ? k : k - 2 * s.C);
        which = sel ^ s.odd;
        e = (m * s.h + (y >> 1)) * s.w + (x >> 1);
    }
}

template <int RT, int KQ>
__global__ void __launch_bounds__(NF_BLOCK) k_glow_head_w_fwd(const float* __restrict__ x, const float* __restrict__ als,
                                                              const float* __restrict__ abias, const float* __restrict__ M,
                                                              const float* __restrict__ log_s, float* __restrict__ h,
                                                              float* __restrict__ z1c, float* __restrict__ ld, NfSplit s, int64_t B,
                                                              int C, int P) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    NF_GH_STAMP(0);
    // W through LDS: read from memory row by row (coalesced), fragments from LDS -- a lane's fragment elements M[16 rt + li][4 q + lk]
    // sit C floats apart across lanes: gathered straight from memory that was 64 cache lines per load instruction, 36 instructions
    // per wave, ~10 of the kernel's 14 us at C = 48
    __shared__ float Ws[64 * 65];
    const int64_t nblk = B * (P / 16);                       // 16-pixel blocks (P % 16 == 0: never straddle samples)
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int bpp = P / 16;
    // everything a wave needs from memory is requested up front -- the first block's pixels, the ActNorm vectors, W -- so that the
    // launch is ONE memory round trip deep (at the BASELINE sizes a wave has one block: loads issued where they are used made it four)
    float xv[KQ], ab[KQ], ad[KQ];
    {
        const int64_t blk = wave < nblk ? wave : 0, b = blk / bpp;
        const float* xb = x + b * C * P + (int)(blk - b * bpp) * 16 + li;
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const int c = 4 * q + lk;
            xv[q] = c < C ? xb[(int64_t)c * P] : 0.f;
            ab[q] = c < C ? abias[c] : 0.f;
            ad[q] = c < C ? als[c] : 0.f;
        }
    }
    // the log-det behind them: its wave sum WAITS for its two loads, so anything requested after it starts a second round trip (in
    // front of the x / W requests it was 1.7 of the launch's 8.2 us at C = 48, B = 64); the read-modify-write of ld travels under the
    // matrix work
    {
        constexpr int NWR = (64 * 64 + NF_BLOCK - 1) / NF_BLOCK;
        float sl = lane < C ? log_s[lane] - als[lane] : 0.f;     // C <= 64: one value per lane, wave sum
        float wreg[NWR];                                         // W requested too before anything is waited for
#pragma unroll
        for (int u = 0; u < NWR; ++u) {
            const int e = threadIdx.x + u * NF_BLOCK;
            wreg[u] = (u * NF_BLOCK < C * C && e < C * C) ? M[e] : 0.f;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sl += __shfl_xor(sl, off, NF_WAVE);
        const float dl = (float)P * sl;
        const int64_t gstride = (int64_t)gridDim.x * blockDim.x;
        for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gstride) ld[b] += dl;
        NF_GH_STAMP(1);
#pragma unroll
        for (int u = 0; u < NWR; ++u) {
            const int e = threadIdx.x + u * NF_BLOCK;
            if (e < C * C) Ws[(e / C) * 65 + (e - (e / C) * C)] = wreg[u];
        }
    }
    __syncthreads();
    NF_GH_STAMP(2);
    float a[RT][KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
        const int c = 4 * q + lk;
        ad[q] = expf(ad[q]);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int r = 16 * rt + li;
            a[rt][q] = (r < C && c < C) ? Ws[r * 65 + c] : 0.f;
        }
    }
    NF_GH_STAMP(3);
    for (int64_t blk = wave; blk < nblk; blk += nwaves) {
        const int64_t b = blk / bpp;
        const int p = (int)(blk - b * bpp) * 16 + li;
        const float* xb = x + b * C * P + p;
        float* hb = h + b * C * P + p;
        f32x4 acc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float bv[KQ];
#pragma unroll
        for (int q = 0; q < KQ; ++q) {                       // B[k = lk][j = li] = ActNorm(x)[c = 4q + lk][pixel]
            const int c = 4 * q + lk;
            const float xq = blk == wave ? xv[q] : (c < C ? xb[(int64_t)c * P] : 0.f);
            bv[q] = c < C ? (xq - ab[q]) / ad[q] : 0.f;          // (the reference's own rounding: a true division)
        }
#pragma unroll
        for (int q = 0; q < KQ; ++q)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][q], bv[q], acc[rt], 0, 0, 0);
        NF_GH_STAMP(4);
        const int yy = p / s.W, xx = p - yy * s.W;
        float* zb = z1c + b * s.n_half;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {                    // D: col = li (pixel), row = 4 lk + j
                const int r = 16 * rt + 4 * lk + j;
                if (r < C) {
                    hb[(int64_t)r * P] = acc[rt][j];
                    int which, e;
                    nf_gh_full_to_half(s, r, p, yy, xx, which, e);
                    if (which == 1) zb[e] = acc[rt][j];
                }
            }
    }
    NF_GH_STAMP(5);
}

// The same forward on 64-PIXEL blocks (large batches): a wave's lane li owns the four pixels 4 li .. 4 li + 3 of the block -- 16-byte loads
// and stores, 256 contiguous bytes per channel row and wave instruction instead of 64 (an 8 x 8 plane is ONE row; with 16-pixel blocks
// every access is a 64-byte segment and the kernel reaches 25 % of the HBM rate at C = 48).  The four MFMA column tiles are the
// pixel sets {4 li + t}: columns are independent, so any assignment of pixels to columns is a valid GEMM.
template <int RT, int KQ>
__global__ void __launch_bounds__(NF_BLOCK) k_glow_head_w_fwd4(const float* __restrict__ x, const float* __restrict__ als,
                                                               const float* __restrict__ abias, const float* __restrict__ M,
                                                               const float* __restrict__ log_s, float* __restrict__ h,
                                                               float* __restrict__ z1c, float* __restrict__ ld, NfSplit s, int64_t B,
                                                               int C, int P) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    {
        float sl = lane < C ? log_s[lane] - als[lane] : 0.f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sl += __shfl_xor(sl, off, NF_WAVE);
        const float dl = (float)P * sl;
        const int64_t gstride = (int64_t)gridDim.x * blockDim.x;
        for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gstride) ld[b] += dl;
    }
    // W, the ActNorm bias and 1 / exp(log_scale) stay in LDS: with the A fragments (RT * KQ registers) and the vectors in registers
    // next to 48 accumulators and 48 pixel registers a wave needs 256 VGPRs at C = 48 -- one wave per SIMD, 38 % of the HBM rate;
    // from LDS (five reads per k-step) it is four waves per SIMD, and the waves hide each other's memory latency
    __shared__ float Ws[64 * 65 + 128];
    float* sab = Ws + 64 * 65;
    float* sad = sab + 64;
    const int bpp = P / 64;
    const int64_t nblk = B * bpp;                            // 64-pixel blocks (P % 64 == 0: never straddle samples)
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int e = threadIdx.x; e < 64 * 65; e += blockDim.x) {
        const int r = e / 65, c = e - r * 65;
        Ws[e] = (r < C && c < C) ? M[r * C + c] : 0.f;
    }
    if (threadIdx.x < 64) {
        sab[threadIdx.x] = threadIdx.x < C ? abias[threadIdx.x] : 0.f;
        sad[threadIdx.x] = threadIdx.x < C ? expf(-als[threadIdx.x]) : 0.f;     // (x - b) / e^ls as (x - b) * e^-ls: within 1 ulp of the division
    }
    __syncthreads();
    for (int64_t blk = wave; blk < nblk; blk += nwaves) {
        const int64_t b = blk / bpp;
        const int p0 = (int)(blk - b * bpp) * 64 + 4 * li;   // this lane's first pixel
        const float* xb = x + b * C * P + p0;
        float* hb = h + b * C * P + p0;
        float4 xv[KQ];
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const int c = 4 * q + lk;
            xv[q] = c < C ? *(const float4*)(xb + (int64_t)c * P) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        f32x4 acc[RT][4];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const int c = 4 * q + lk;                        // (c >= C: bias 0, scale 0 -> B operand 0)
            const float ab = sab[c], ad = sad[c];
            const float b0 = (xv[q].x - ab) * ad, b1 = (xv[q].y - ab) * ad, b2 = (xv[q].z - ab) * ad, b3 = (xv[q].w - ab) * ad;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const float a = Ws[(16 * rt + li) * 65 + c];
                acc[rt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, acc[rt][0], 0, 0, 0);
                acc[rt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, acc[rt][1], 0, 0, 0);
                acc[rt][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b2, acc[rt][2], 0, 0, 0);
                acc[rt][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b3, acc[rt][3], 0, 0, 0);
            }
        }
        float* zb = z1c + b * s.n_half;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {                    // D of tile t: col = li (pixel 4 li + t), row = 4 lk + j
                const int r = 16 * rt + 4 * lk + j;
                if (r < C) {
                    const float4 v = make_float4(acc[rt][0][j], acc[rt][1][j], acc[rt][2][j], acc[rt][3][j]);
                    *(float4*)(hb + (int64_t)r * P) = v;
                    if (s.mode == NF_SPLIT_CHANNEL) {        // the half is a channel range: the same four pixels, one 16-byte store
                        const int hc = s.C >> 1, sel = r >= hc ? 1 : 0;
                        if ((sel ^ s.odd) == 1) *(float4*)(zb + (int64_t)(r - sel * hc) * P + p0) = v;
                    } else {
                        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int p = p0 + t, yy = p / s.W, xx = p - yy * s.W;
                            int which, e;
                            nf_gh_full_to_half(s, r, p, yy, xx, which, e);
                            if (which == 1) zb[e] = vv[t];
                        }
                    }
                }
            }
    }
}

// TP = pixels per staged tile: 128, or 64 where 128 would leave fewer than 64 workgroups (the 8 x 8 level at B = 64: 32 -> 64 workgroups,
// each wave 16 pixels instead of 32 -- the launch is a latency chain, a wave's part of it halves)
// PART: 0 = everything; 1 = the DATA gradient g_x only (what the backward pass waits for); 2 = the PARAMETER gradients only (g_W, g_log_scale,
// g_bias: contractions over the batch that nothing but the optimizer waits for -- run for many steps per launch where the pass ends,
// k_glow_head_w_params_multi).  Same arithmetic per element in the same order, so 1 + 2 reproduce 0 bit for bit.
template <int RT, int KQ, int TP, int PART>
__device__ __forceinline__ void nf_gh_w_bwd_body(float* lds, const float* __restrict__ gh, const float* __restrict__ gld,
                                                 const float* __restrict__ x, const float* __restrict__ als,
                                                 const float* __restrict__ abias, const float* __restrict__ M,
                                                 float* __restrict__ gx, float* __restrict__ g_ls, float* __restrict__ g_b,
                                                 float* __restrict__ gM, int64_t B, int C, int P, int64_t tiles_per_block) {
    constexpr bool DATA = PART != 2, PARAMS = PART != 1;
    NF_GH_STAMP(8);
    constexpr int PW = TP / 4;            // pixels per wave
    constexpr bool THROUGH_TILE = RT >= 3;   // g_x leaves through the LDS tile (see below)
    const int RS = TP + 1;
    const int CP = RT * 16;
    float* gT = lds;                       // [CP][RS]  g_h
    float* aT = lds + (size_t)CP * RS;     // [CP][RS]  ActNorm(x)
    float* cst = lds + (size_t)2 * CP * 129;   // [2][CP]   bias, exp(log_scale): behind the 128-pixel layout for either tile (the
                                           // reductions at the end alias [0, 4 CP CP) of the buffer and still read it)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, li = lane & 15, lk = lane >> 4;
    const int64_t npix = B * P;
    // sum_b g_ld (every channel's log_scale gradient carries P times it): each block takes a slice of the batch, requested here and
    // consumed at the very end (as a loop of dependent loads in block 0 it was 16 of 91 us at B = 8192)
    float sg = 0.f;
    if (PARAMS) {
        float part[4] = {0.f, 0.f, 0.f, 0.f};
        const int64_t stride = (int64_t)gridDim.x * blockDim.x;
        for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += 4 * stride) {
#pragma unroll
            for (int u = 0; u < 4; ++u) part[u] += b + u * stride < B ? gld[b + u * stride] : 0.f;
        }
        sg = (part[0] + part[1]) + (part[2] + part[3]);
    }
    // A fragments of W^T: A[i = li][k = lk] of row tile rt, k-step q -> W[4 q + lk][16 rt + li]
    float wt[RT][KQ];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const int r = 16 * rt + li, c = 4 * q + lk;
            wt[rt][q] = (r < C && c < C) ? M[c * C + r] : 0.f;
        }
    f32x4 acc[RT][RT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < RT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float s1[RT][4], s2[RT][4];            // this lane's sums of g_a a and g_a over its pixels, rows 16 rt + 4 lk + j
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s1[i][j] = s2[i][j] = 0.f;
    // Staging: a thread owns ONE pixel column of the tile (256 threads = 128 pixels x 2 channel phases); the loads of tile i + 1 are
    // issued into registers before the MFMAs of tile i.
    constexpr int NPH = NF_BLOCK / TP;     // channel phases
    constexpr int NCH = RT * 16 / NPH;     // channel rows per thread
    const int sq = threadIdx.x & (TP - 1), ph = threadIdx.x / TP;
    float rg[NCH], rx[NCH];
    const int64_t tile0 = (int64_t)blockIdx.x * tiles_per_block;
    int64_t fbase = 0, sbase = 0;          // element offset of this thread's pixel column: of the tile just fetched / of the tile at work
    auto fetch = [&](int64_t tile) {
        const int64_t t = tile * TP + sq;
        const bool ok = tile < tile0 + tiles_per_block && t < npix;
        const int64_t b = ok ? t / P : 0;
        const int64_t base = b * C * P + (ok ? t - b * P : 0);
        fbase = base;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = ph + NPH * k;
            const bool in = ok && c < C;
            rg[k] = in ? gh[base + (int64_t)c * P] : 0.f;
            rx[k] = (PARAMS && in) ? x[base + (int64_t)c * P] : 0.f;
        }
    };
    fetch(tile0);
    // (the constants LAST: their LDS stores wait for their loads -- in front of the tile request that was a round trip of its own)
    for (int c = threadIdx.x; c < CP; c += blockDim.x) {
        cst[c] = c < C ? abias[c] : 0.f;
        cst[CP + c] = c < C ? 1.f / expf(als[c]) : 1.f;     // 1 / exp(log_scale): the tile loop multiplies
    }
    NF_GH_STAMP(9);
    for (int64_t tile = tile0; tile < tile0 + tiles_per_block; ++tile) {
        const int64_t t0 = tile * TP;
        if (t0 >= npix) break;
        const int np = (int)min((int64_t)TP, npix - t0);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = ph + NPH * k;
            gT[c * RS + sq] = rg[k];
            if (PARAMS) aT[c * RS + sq] = (sq < np && c < C) ? (rx[k] - cst[c]) * cst[CP + c] : 0.f;
        }
        __syncthreads();
        NF_GH_STAMP(10);
        sbase = fbase;
        fetch(tile + 1);
        // ---- g_W: this wave's quarter of the tile, pixels [PW wid, PW wid + PW), k-steps of 4 pixels ----
#pragma unroll
        for (int ks = 0; ks < (PARAMS ? PW / 4 : 0); ++ks) {
            const int pix = PW * wid + 4 * ks + lk;
            float av[RT], bv[RT];
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                av[i] = gT[(16 * i + li) * RS + pix];            // A[i = r][k = pix] = g_h[r][pix]
                bv[i] = aT[(16 * i + li) * RS + pix];            // B[k = pix][j = c] = a[c][pix]
            }
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < RT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        NF_GH_STAMP(11);
        // ---- g_a = W^T g_h for the same 32 pixels (two blocks of 16), g_x, and the ActNorm sums ----
#pragma unroll
        for (int half = 0; half < PW / 16; ++half) {
            const int pix = PW * wid + 16 * half + li;
            f32x4 ga[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) ga[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                const float bq = gT[(4 * q + lk) * RS + pix];      // B[k = lk][j = li] = g_h[c = 4q + lk][pixel]   (rows >= C are zero)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) ga[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[rt][q], bq, ga[rt], 0, 0, 0);
            }
            // g_x leaves through the tile: the lane that finishes (row r, pixel) is the only reader of aT[r][pixel] (a wave touches its
            // own pixel quarter only), so the value takes that cell and the block writes the tile out below with the mapping of the
            // loads -- whole 256-byte runs per wave instruction (from the accumulator layout it was 64-byte pieces of four rows:
            // 1.7 TB/s on (48, 8, 8) at large batches)
            // (few channels: the extra barrier costs more than the pieces -- straight from the accumulator layout there)
            if (pix < np) {
                const int64_t t = t0 + pix, b = THROUGH_TILE ? 0 : t / P;
                float* gxb = gx + b * C * P + (t - b * P);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = 16 * rt + 4 * lk + j;
                        if (r < C) {
                            const float g = ga[rt][j];
                            if (PARAMS) {
                                s1[rt][j] = fmaf(g, aT[r * RS + pix], s1[rt][j]);
                                s2[rt][j] += g;
                            }
                            if (DATA) {
                                if (THROUGH_TILE) aT[r * RS + pix] = g * cst[CP + r];
                                else gxb[(int64_t)r * P] = g * cst[CP + r];
                            }
                        }
                    }
            }
        }
        if (DATA && THROUGH_TILE) {
            __syncthreads();
            if (sq < np) {
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    const int c = ph + NPH * k;
                    if (c < C) gx[sbase + (int64_t)c * P] = aT[c * RS + sq];
                }
            }
        }
    }
    NF_GH_STAMP(12);
    if (!PARAMS) return;
    // ---- cross-wave reductions through LDS, then one atomic per entry per block ----
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
    NF_DET_ROW_CHAIN();                    // (blockIdx.y = the head of a multi-launch, 0 otherwise: one chain per head)
    NF_DET_ENTER_ALL_K(nf_ghm);            // (one thread per entry / channel and workgroup; the turn is held over both groups of sums)
    for (int e = threadIdx.x; e < C * C; e += blockDim.x) {
        const int r = e / C, c = e - r * C;
        const float t = red[(0 * CP + r) * CP + c] + red[(1 * CP + r) * CP + c] + red[(2 * CP + r) * CP + c] +
                        red[(3 * CP + r) * CP + c];
        atomicAdd(gM + e, t);
    }
    __syncthreads();
    NF_GH_STAMP(13);
    // per-channel sums: over the 16 pixel lanes by shuffles, over the four waves through LDS
    float* rs = lds;                       // [2][4][CP]
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float u = s1[i][j], v = s2[i][j];
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) {
                u += __shfl_xor(u, off, NF_WAVE);
                v += __shfl_xor(v, off, NF_WAVE);
            }
            if (li == 0) {
                rs[(0 * 4 + wid) * CP + 16 * i + 4 * lk + j] = u;
                rs[(1 * 4 + wid) * CP + 16 * i + 4 * lk + j] = v;
            }
        }
    {                                      // this block's slice of sum_b g_ld
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sg += __shfl_xor(sg, off, NF_WAVE);
        if (lane == 0) rs[2 * 4 * CP + wid] = sg;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float R2 = (rs[(0 * 4 + 0) * CP + c] + rs[(0 * 4 + 1) * CP + c]) + (rs[(0 * 4 + 2) * CP + c] + rs[(0 * 4 + 3) * CP + c]);
        const float R1 = (rs[(1 * 4 + 0) * CP + c] + rs[(1 * 4 + 1) * CP + c]) + (rs[(1 * 4 + 2) * CP + c] + rs[(1 * 4 + 3) * CP + c]);
        const float SG = (rs[8 * CP] + rs[8 * CP + 1]) + (rs[8 * CP + 2] + rs[8 * CP + 3]);
        atomicAdd(g_ls + c, -R2 - (float)P * SG);        // modules.py:246-249 differentiated
        atomicAdd(g_b + c, -R1 * cst[CP + c]);
    }
    NF_DET_LEAVE_ALL_K(nf_ghm);
    NF_GH_STAMP(14);
}

template <int RT, int KQ, int TP, int PART>
__global__ void __launch_bounds__(NF_BLOCK, (TP == 64 && RT <= 3) ? 2 : 1) k_glow_head_w_bwd(const float* __restrict__ gh, const float* __restrict__ gld,
                                                              const float* __restrict__ x, const float* __restrict__ als,
                                                              const float* __restrict__ abias, const float* __restrict__ M,
                                                              float* __restrict__ gx, float* __restrict__ g_ls, float* __restrict__ g_b,
                                                              float* __restrict__ gM, int64_t B, int C, int P,
                                                              int64_t tiles_per_block) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    nf_gh_w_bwd_body<RT, KQ, TP, PART>(lds, gh, gld, x, als, abias, M, gx, g_ls, g_b, gM, B, C, P, tiles_per_block);
}

// the parameter gradients of up to NF_GLOW_HEAD_MULTI_MAX heads of one shape in one launch: blockIdx.y = head
struct NfGhMulti { nf_glow_head_params_desc d[NF_GLOW_HEAD_MULTI_MAX]; };
template <int RT, int KQ, int TP>
__global__ void __launch_bounds__(NF_BLOCK, (TP == 64 && RT <= 3) ? 2 : 1) k_glow_head_w_params_multi(NfGhMulti m, int64_t B, int C, int P,
                                                                                                        int64_t tiles_per_block) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const nf_glow_head_params_desc& d = m.d[blockIdx.y];
    nf_gh_w_bwd_body<RT, KQ, TP, 2>(lds, d.g_h, d.g_ld, d.x, d.act_log_scale, d.act_bias, d.W, nullptr, d.g_log_scale, d.g_bias, d.g_W, B, C, P,
                                    tiles_per_block);
}

static inline bool nf_gh_shape_ok(int64_t B, int C, int H, int W) {
    return B > 0 && C >= 9 && C <= 64 && H > 0 && W > 0 && ((H * W) % 16) == 0 && B * (int64_t)C * H * W < ((int64_t)1 << 40);
}

extern "C" int nf_glow_head_w_usable(int64_t B, int C, int H, int W, int mode) {
    if (mode != NF_SPLIT_CHANNEL && mode != NF_SPLIT_CHECKER) return 0;
    NfSplit s;
    return nf_gh_shape_ok(B, C, H, W) && nf_make_split(s, mode, 0, C, H, W) ? 1 : 0;
}

extern "C" int nf_glow_head_w_fwd(const float* x, const float* act_log_scale, const float* act_bias, const float* Wm,
                                  const float* log_s, float* h, float* z1c, float* ld, int mode, int odd, int64_t B, int C, int H,
                                  int W, nf_stream_t stream) {
    NfSplit s;
    if (!nf_glow_head_w_usable(B, C, H, W, mode) || !nf_make_split(s, mode, odd, C, H, W)) return NF_E_BADARG;
    if (x == nullptr || act_log_scale == nullptr || act_bias == nullptr || Wm == nullptr || log_s == nullptr || h == nullptr ||
        z1c == nullptr || ld == nullptr)
        return NF_E_BADARG;
    const int P = H * W;
    const int64_t nblk = B * (P / 16);
    int64_t g = (nblk + 3) / 4;                              // 4 waves per block, >= 1 block of 16 pixels per wave
    if (g > 2048) g = 2048;
    const int64_t g_ld = (B + NF_BLOCK - 1) / NF_BLOCK;
    if (g < g_ld) g = g_ld > 4096 ? 4096 : g_ld;
    const int rt = (C + 15) / 16, kq = (C + 3) / 4;
    hipStream_t st = (hipStream_t)stream;
#define NF_CASE(RT, KQ)                                                                                                         \
    if (rt == RT && kq == KQ) {                                                                                                 \
        hipLaunchKernelGGL((k_glow_head_w_fwd<RT, KQ>), dim3((unsigned)g), dim3(NF_BLOCK), 0, st, x, act_log_scale, act_bias, Wm, \
                           log_s, h, z1c, ld, s, B, C, P);                                                                      \
        NF_CHECK_LAUNCH();                                                                                                      \
        return 0;                                                                                                               \
    }
    // large batches (>= 4 blocks of 64 pixels per compute unit): 64-pixel blocks, 16-byte accesses
    const int64_t nblk64 = (P % 64 == 0) ? B * (P / 64) : 0;
    // (float4 accesses: a contiguous tensor that is a view at an odd storage offset takes the 16-pixel kernel below)
    const bool al16 = (((uintptr_t)x | (uintptr_t)h | (uintptr_t)z1c) & 15) == 0;
    if (nblk64 >= 1024 && al16) {
        int64_t g4 = (nblk64 + 3) / 4;
        if (g4 > 512) g4 = 512;                            // two workgroups per compute unit (256: 61 %, 512: 64 %, 1024: 60 % of 8 TB/s at (48,8,8)): a wave walks several blocks per W staging
        if (g4 < g_ld) g4 = g_ld > 4096 ? 4096 : g_ld;
#define NF_CASE4(RT, KQ)                                                                                                        \
        if (rt == RT && kq == KQ) {                                                                                             \
            hipLaunchKernelGGL((k_glow_head_w_fwd4<RT, KQ>), dim3((unsigned)g4), dim3(NF_BLOCK), 0, st, x, act_log_scale, act_bias, Wm, \
                               log_s, h, z1c, ld, s, B, C, P);                                                                  \
            NF_CHECK_LAUNCH();                                                                                                  \
            return 0;                                                                                                           \
        }
        NF_CASE4(1, 3) NF_CASE4(1, 4) NF_CASE4(2, 5) NF_CASE4(2, 6) NF_CASE4(2, 7) NF_CASE4(2, 8) NF_CASE4(3, 9) NF_CASE4(3, 10)
        NF_CASE4(3, 11) NF_CASE4(3, 12) NF_CASE4(4, 13) NF_CASE4(4, 14) NF_CASE4(4, 15) NF_CASE4(4, 16)
#undef NF_CASE4
    }
    NF_CASE(1, 3) NF_CASE(1, 4) NF_CASE(2, 5) NF_CASE(2, 6) NF_CASE(2, 7) NF_CASE(2, 8) NF_CASE(3, 9) NF_CASE(3, 10)
    NF_CASE(3, 11) NF_CASE(3, 12) NF_CASE(4, 13) NF_CASE(4, 14) NF_CASE(4, 15) NF_CASE(4, 16)
#undef NF_CASE
    return NF_E_BADARG;
}

// Tile and grid by the kernel's register footprint (-Rpass-analysis=kernel-resource-usage): the 128-pixel tile of 33 .. 64 channels
// holds 388 registers -- one workgroup per compute unit, so 512 workgroups ran in two rounds (137 us on (48, 8, 8) x 8192; with the
// 64-pixel tile, two workgroups per unit, 99) -- and the grid is ONE round of what fits (12 channels: three per unit, 91 -> 83 us).
// 33 .. 64 channels: the 128-pixel tile while its tiles are ONE round of one workgroup per compute unit (B = 512 at (48, 8, 8): 256 tiles,
// 27.5 us against 36.7 with 512 64-pixel tiles -- profiles/r04_c4_b512_step_kernels.txt before / after the tile rule above), the 64-pixel
// tile beyond (two workgroups per unit instead of a second round) and below 64 tiles (latency: a wave's share of the chain halves)
struct NfGhPlan { int rt, kq, TP; int64_t blocks, tpb; size_t lds; };
static NfGhPlan nf_gh_bwd_plan(int64_t B, int C, int P, int heads) {
    NfGhPlan p;
    p.rt = (C + 15) / 16; p.kq = (C + 3) / 4;
    const int64_t npix = B * P;
    const int64_t t128 = (npix + 127) / 128;
    p.TP = (t128 < 64 || (p.rt >= 3 && t128 > 256)) ? 64 : 128;
    int cap = p.rt == 1 ? 768 : (p.rt <= 3 ? 512 : 256);
    if (heads > 1) cap = cap / heads > 8 ? cap / heads : 8;      // many heads per launch: the round is shared
    const int64_t tiles = (npix + p.TP - 1) / p.TP;
    p.blocks = tiles < cap ? tiles : cap;                    // ends in C * C + 2 C same-address atomics per block
    p.tpb = (tiles + p.blocks - 1) / p.blocks;
    p.blocks = (tiles + p.tpb - 1) / p.tpb;
    p.lds = ((size_t)2 * p.rt * 16 * (128 + 1) + 2 * p.rt * 16) * sizeof(float);   // (sized for either tile; the reductions alias it)
    return p;
}

template <int PART>
static int nf_gh_w_bwd_launch(const float* g_h, const float* g_ld, const float* x, const float* act_log_scale, const float* act_bias,
                              const float* Wm, float* g_x, float* g_log_scale, float* g_bias, float* g_W, int64_t B, int C, int H, int W,
                              nf_stream_t stream) {
    const int P = H * W;
    const NfGhPlan p = nf_gh_bwd_plan(B, C, P, 1);
    hipStream_t st = (hipStream_t)stream;
#define NF_CASE(RT, KQ)                                                                                                                 \
    if (p.rt == RT && p.kq == KQ) {                                                                                                     \
        if (p.TP == 64)                                                                                                                 \
            hipLaunchKernelGGL((k_glow_head_w_bwd<RT, KQ, 64, PART>), dim3((unsigned)p.blocks), dim3(NF_BLOCK), p.lds, st, g_h, g_ld, x,  \
                               act_log_scale, act_bias, Wm, g_x, g_log_scale, g_bias, g_W, B, C, P, p.tpb);                             \
        else                                                                                                                            \
            hipLaunchKernelGGL((k_glow_head_w_bwd<RT, KQ, 128, PART>), dim3((unsigned)p.blocks), dim3(NF_BLOCK), p.lds, st, g_h, g_ld, x, \
                               act_log_scale, act_bias, Wm, g_x, g_log_scale, g_bias, g_W, B, C, P, p.tpb);                             \
        NF_CHECK_LAUNCH();                                                                                                              \
        return 0;                                                                                                                       \
    }
    NF_CASE(1, 3) NF_CASE(1, 4) NF_CASE(2, 5) NF_CASE(2, 6) NF_CASE(2, 7) NF_CASE(2, 8) NF_CASE(3, 9) NF_CASE(3, 10)
    NF_CASE(3, 11) NF_CASE(3, 12) NF_CASE(4, 13) NF_CASE(4, 14) NF_CASE(4, 15) NF_CASE(4, 16)
#undef NF_CASE
    return NF_E_BADARG;
}

extern "C" int nf_glow_head_w_bwd(const float* g_h, const float* g_ld, const float* x, const float* act_log_scale,
                                  const float* act_bias, const float* Wm, float* g_x, float* g_log_scale, float* g_bias, float* g_W,
                                  int64_t B, int C, int H, int W, nf_stream_t stream) {
    if (!nf_gh_shape_ok(B, C, H, W)) return NF_E_BADARG;
    if (g_h == nullptr || g_ld == nullptr || x == nullptr || act_log_scale == nullptr || act_bias == nullptr || Wm == nullptr ||
        g_x == nullptr || g_log_scale == nullptr || g_bias == nullptr || g_W == nullptr)
        return NF_E_BADARG;
    return nf_gh_w_bwd_launch<0>(g_h, g_ld, x, act_log_scale, act_bias, Wm, g_x, g_log_scale, g_bias, g_W, B, C, H, W, stream);
}

// the data gradient alone: g_x = diag(exp(-log_scale)) W^T g_h -- what the backward pass waits for (the parameter gradients of the same
// head follow through nf_glow_head_w_bwd_params_multi, many heads per launch)
extern "C" int nf_glow_head_w_bwd_data(const float* g_h, const float* act_log_scale, const float* Wm, float* g_x, int64_t B, int C, int H,
                                       int W, nf_stream_t stream) {
    if (!nf_gh_shape_ok(B, C, H, W)) return NF_E_BADARG;
    if (g_h == nullptr || act_log_scale == nullptr || Wm == nullptr || g_x == nullptr) return NF_E_BADARG;
    return nf_gh_w_bwd_launch<1>(g_h, nullptr, nullptr, act_log_scale, act_log_scale, Wm, g_x, nullptr, nullptr, nullptr, B, C, H, W, stream);
}

// g_W, g_log_scale, g_bias (accumulated: += like every sink) of n heads of ONE shape (B, C, H, W) in one launch
extern "C" int nf_glow_head_w_bwd_params_multi(const nf_glow_head_params_desc* descs, int n, int64_t B, int C, int H, int W,
                                               nf_stream_t stream) {
    if (!nf_gh_shape_ok(B, C, H, W) || descs == nullptr || n < 1 || n > NF_GLOW_HEAD_MULTI_MAX) return NF_E_BADARG;
    NfGhMulti m;
    for (int i = 0; i < n; ++i) {
        const nf_glow_head_params_desc& d = descs[i];
        if (d.g_h == nullptr || d.g_ld == nullptr || d.x == nullptr || d.act_log_scale == nullptr || d.act_bias == nullptr || d.W == nullptr ||
            d.g_log_scale == nullptr || d.g_bias == nullptr || d.g_W == nullptr)
            return NF_E_BADARG;
        m.d[i] = d;
    }
    const int P = H * W;
    const NfGhPlan p = nf_gh_bwd_plan(B, C, P, n);
    hipStream_t st = (hipStream_t)stream;
#define NF_CASE(RT, KQ)                                                                                                                   \
    if (p.rt == RT && p.kq == KQ) {                                                                                                       \
        if (p.TP == 64)                                                                                                                   \
            hipLaunchKernelGGL((k_glow_head_w_params_multi<RT, KQ, 64>), dim3((unsigned)p.blocks, (unsigned)n), dim3(NF_BLOCK), p.lds, st, m, B, \
                               C, P, p.tpb);                                                                                              \
        else                                                                                                                              \
            hipLaunchKernelGGL((k_glow_head_w_params_multi<RT, KQ, 128>), dim3((unsigned)p.blocks, (unsigned)n), dim3(NF_BLOCK), p.lds, st, m, B, \
                               C, P, p.tpb);                                                                                              \
        NF_CHECK_LAUNCH();                                                                                                                \
        return 0;                                                                                                                         \
    }
    NF_CASE(1, 3) NF_CASE(1, 4) NF_CASE(2, 5) NF_CASE(2, 6) NF_CASE(2, 7) NF_CASE(2, 8) NF_CASE(3, 9) NF_CASE(3, 10)
    NF_CASE(3, 11) NF_CASE(3, 12) NF_CASE(4, 13) NF_CASE(4, 14) NF_CASE(4, 15) NF_CASE(4, 16)
#undef NF_CASE
    return NF_E_BADARG;
}
