// Fused head of a Glow flow step for small channel counts (2-D data and C <= 4 images):
//   ActNorm -> invertible 1x1 (PLU weight assembled in-kernel) -> gather of the conditioning half,
//   flows/modules.py:246-250 + :470-482 + flows/coupling.py:33 / flows/squeeze.py  in ONE launch, and its autograd
//   (scatter-add of the conditioner's input gradient + W^T apply + ActNorm backward + the C x C weight-gradient
//   reduction) in ONE launch followed by the PLU backward.  At these sizes every launch is pure latency (~4 us for a
//   131 KB problem), so the four launches forward / seven backward this replaces are the cost.
#include "nf_common.h"
#include "nf_det.h"
#include "nf_small_plu.h"

NF_DET_STATE(nf_gh)
NF_DET_HOST_API(nf_gh)

#define NF_HEAD_MAXC 4

template <int CT>
__global__ void __launch_bounds__(NF_BLOCK) k_glow_head_fwd(const float* __restrict__ z, const float* __restrict__ ls,
                                                            const float* __restrict__ bs, const float* __restrict__ Pm,
                                                            const float* __restrict__ L, const float* __restrict__ U,
                                                            const float* __restrict__ Lm, const float* __restrict__ Um,
                                                            const float* __restrict__ sign_s,
                                                            const float* __restrict__ log_s, float* __restrict__ h,
                                                            float* __restrict__ z1c, float* __restrict__ Wout,
                                                            float* __restrict__ ld, NfSplit s, int64_t B, int P) {
    float Wm[CT][CT], es[CT], bb[CT];
    nf_small_plu<CT>(Pm, L, U, Lm, Um, sign_s, log_s, Wm);
    float dld = 0.f;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        es[c] = expf(ls[c]);
        bb[c] = bs[c];
        dld += log_s[c] - ls[c];                                   // modules.py:249, :480
    }
    dld *= (float)P;
    const int64_t gstride = (int64_t)gridDim.x * blockDim.x;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gtid == 0 && Wout != nullptr) {
#pragma unroll
        for (int r = 0; r < CT; ++r)
#pragma unroll
            for (int c = 0; c < CT; ++c) Wout[r * CT + c] = Wm[r][c];   // saved for the backward pass
    }
    const int64_t npix = B * P;
    for (int64_t t = gtid; t < npix; t += gstride) {
        const int64_t b = t / P;
        const int p = (int)(t - b * P);
        const int64_t base = b * CT * P + p;
        float zn[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) zn[c] = (z[base + (int64_t)c * P] - bb[c]) / es[c];     // modules.py:246
#pragma unroll
        for (int r = 0; r < CT; ++r) {
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < CT; ++c) a = fmaf(Wm[r][c], zn[c], a);                       // modules.py:477
            h[base + (int64_t)r * P] = a;
            int which, e;
            nf_full_to_half(s, r, p, which, e);
            if (which == 1) z1c[b * s.n_half + e] = a;                                       // conditioner input
        }
    }
    for (int64_t b = gtid; b < B; b += gstride) ld[b] += dld;
}

// backward.  G = g_h + scatter(g_z1c);  g_zn = W^T G;  g_z = g_zn / exp(ls);
// g_bias[c] += -sum g_zn[c]/exp(ls[c]);  g_ls[c] += -sum g_zn[c] zn[c] - P sum_b g_ld;  g_W[r][c] += sum G[r] zn[c]
#define NF_GH_BIG 1024     // threads of the backward kernel: it ends in same-address atomics (256 blocks of sixteen waves, see coupling.hip)
template <int CT>
__device__ __forceinline__ void nf_gh_bwd_load(const float* __restrict__ gh, const float* __restrict__ gz1c, const float* __restrict__ z,
                                               const NfSplit& s, int64_t t, int P, int64_t& base, float (&zr)[CT], float (&G)[CT]) {
    const int64_t b = t / P;
    const int p = (int)(t - b * P);
    base = b * CT * P + p;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        zr[c] = z[base + (int64_t)c * P];
        float g = gh[base + (int64_t)c * P];
        int which, e;
        nf_full_to_half(s, c, p, which, e);
        if (which == 1 && gz1c != nullptr) g += gz1c[b * s.n_half + e];
        G[c] = g;
    }
}

// PART: 0 = everything; 1 = the data gradient g_z alone; 2 = the parameter sums alone (many heads per launch: k_glow_head_params_multi).
// Same per-pixel arithmetic, same block partition: 1 + 2 reproduce 0.
template <int CT, int PART>
__device__ __forceinline__ void nf_gh_bwd_body(const float* __restrict__ gh, const float* __restrict__ gz1c,
                                               const float* __restrict__ gld, const float* __restrict__ z,
                                               const float* __restrict__ ls, const float* __restrict__ bs,
                                               const float* __restrict__ Wsaved, float* __restrict__ gz,
                                               float* __restrict__ g_ls, float* __restrict__ g_bias,
                                               float* __restrict__ gW, float* __restrict__ sum_gld, const NfSplit& s, int64_t B, int P) {
    constexpr bool DATA = PART != 2, PARAMS = PART != 1;
    float Wm[CT][CT], es[CT], bb[CT];
#pragma unroll
    for (int r = 0; r < CT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) Wm[r][c] = Wsaved[r * CT + c];
#pragma unroll
    for (int c = 0; c < CT; ++c) { es[c] = 1.f / expf(ls[c]); bb[c] = bs[c]; }      // (the loop multiplies)
    float aW[CT][CT], aB[CT], aL[CT];
#pragma unroll
    for (int r = 0; r < CT; ++r) {
        aB[r] = 0.f; aL[r] = 0.f;
#pragma unroll
        for (int c = 0; c < CT; ++c) aW[r][c] = 0.f;
    }
    const int64_t gstride = (int64_t)gridDim.x * blockDim.x;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t npix = B * P;
    float sg = 0.f;                                  // this block's share of sum_b g_ld: requested with the first pixels (behind the
    if (PARAMS)
        for (int64_t b = gtid; b < B; b += gstride) sg += gld[b]; // loop it was a round trip of its own at the end of the launch)
    for (int64_t t = gtid; t < npix; t += 2 * gstride) {     // two pixels per trip: their loads are in flight together
        const int64_t t2 = t + gstride;
        const bool has2 = t2 < npix;
        int64_t base[2];
        float zr[2][CT], G[2][CT];
        nf_gh_bwd_load<CT>(gh, gz1c, z, s, t, P, base[0], zr[0], G[0]);
        nf_gh_bwd_load<CT>(gh, gz1c, z, s, has2 ? t2 : t, P, base[1], zr[1], G[1]);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !has2) break;
            float zn[CT];
#pragma unroll
            for (int c = 0; c < CT; ++c) zn[c] = (zr[u][c] - bb[c]) * es[c];
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                float a = 0.f;
#pragma unroll
                for (int r = 0; r < CT; ++r) {
                    a = fmaf(Wm[r][c], G[u][r], a);
                    if (PARAMS) aW[r][c] = fmaf(G[u][r], zn[c], aW[r][c]);
                }
                const float gzc = a * es[c];
                if (DATA) gz[base[u] + (int64_t)c * P] = gzc;
                if (PARAMS) {
                    aB[c] -= gzc;
                    aL[c] = fmaf(-a, zn[c], aL[c]);
                }
            }
        }
    }
    if (!PARAMS) return;
    // the block's 1 + CT (2 + CT) sums in ONE pass: wave sums by shuffles, the sixteen wave partials of every value side by side in
    // LDS, one barrier, thread i finishes value i (one nf_block_sum per value was 2 barriers each: 32 in a row at CT = 3, ~8 of the
    // launch's 19 us); partials are added in wave order, as nf_block_sum does
    constexpr int NV = 1 + CT * (2 + CT);
    constexpr int NWV = NF_GH_BIG / NF_WAVE;
    __shared__ float part[NWV][NV + 1];
    const int lane = threadIdx.x & (NF_WAVE - 1), wid = threadIdx.x >> 6;
    {
        float v = nf_wave_sum(sg);
        if (lane == 0) part[wid][0] = v;
#pragma unroll
        for (int r = 0; r < CT; ++r) {
            const int i0 = 1 + r * (2 + CT);
            v = nf_wave_sum(aB[r]);
            if (lane == 0) part[wid][i0] = v;
            v = nf_wave_sum(aL[r]);
            if (lane == 0) part[wid][i0 + 1] = v;
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                v = nf_wave_sum(aW[r][c]);
                if (lane == 0) part[wid][i0 + 2 + c] = v;
            }
        }
    }
    __syncthreads();
    NF_DET_ROW_CHAIN();                    // (blockIdx.y = the head of a multi-launch, 0 otherwise: one chain per head)
    NF_DET_ENTER_ALL_K(nf_gh);             // (thread i of the workgroup owns value i)
    if (threadIdx.x < NV) {
        const int i = threadIdx.x;
        const int nw = (blockDim.x + NF_WAVE - 1) >> 6;
        float t = 0.f, SG = 0.f;
        for (int w = 0; w < nw; ++w) { t += part[w][i]; SG += part[w][0]; }
        if (i == 0) {
            if (sum_gld != nullptr) atomicAdd(sum_gld, SG);
        } else {
            const int r = (i - 1) / (2 + CT), k = (i - 1) - r * (2 + CT);
            if (k == 0) atomicAdd(g_bias + r, t);
            else if (k == 1) atomicAdd(g_ls + r, t - (float)P * SG);
            else atomicAdd(gW + r * CT + (k - 2), t);
        }
    }
    NF_DET_LEAVE_ALL_K(nf_gh);
}

template <int CT, int PART>
__global__ void __launch_bounds__(NF_GH_BIG) k_glow_head_bwd(const float* __restrict__ gh, const float* __restrict__ gz1c,
                                                             const float* __restrict__ gld, const float* __restrict__ z,
                                                             const float* __restrict__ ls, const float* __restrict__ bs,
                                                             const float* __restrict__ Wsaved, float* __restrict__ gz,
                                                             float* __restrict__ g_ls, float* __restrict__ g_bias,
                                                             float* __restrict__ gW, float* __restrict__ sum_gld, NfSplit s, int64_t B, int P) {
    nf_gh_bwd_body<CT, PART>(gh, gz1c, gld, z, ls, bs, Wsaved, gz, g_ls, g_bias, gW, sum_gld, s, B, P);
}

struct NfGhsMulti { nf_glow_head_small_params_desc d[NF_GLOW_HEAD_MULTI_MAX]; };
template <int CT>
__global__ void __launch_bounds__(NF_GH_BIG) k_glow_head_params_multi(NfGhsMulti m, NfSplit s, int64_t B, int P) {
    const nf_glow_head_small_params_desc& d = m.d[blockIdx.y];
    NfSplit sd = s;
    sd.odd = d.odd;
    nf_gh_bwd_body<CT, 2>(d.g_h, d.g_z1c, d.g_ld, d.z, d.log_scale, d.bias, d.W_saved, nullptr, d.g_log_scale, d.g_bias, d.g_W, d.sum_g_ld, sd, B, P);
}

extern "C" int nf_glow_head_fwd(const float* z, const float* log_scale, const float* bias, const float* P,
                                const float* L, const float* U, const float* L_mask, const float* U_mask,
                                const float* sign_s, const float* log_s, float* h, float* z1c, float* W_out, float* ld,
                                int mode, int odd, int64_t B, int C, int H, int W, nf_stream_t stream) {
    NfSplit s;
    if (!nf_make_split(s, mode, odd, C, H, W) || mode == NF_SPLIT_NONE) return NF_E_BADARG;
    if (C > NF_HEAD_MAXC) return NF_E_UNSUPPORTED;
    if (B == 0) return 0;
    const int Px = H * W;
    unsigned g = nf_grid_for(B * Px);
    const unsigned g_ld = nf_grid_for(B);
    if (g < g_ld) g = g_ld;
    hipStream_t st = (hipStream_t)stream;
#define NF_CASE(CT) case CT: hipLaunchKernelGGL(k_glow_head_fwd<CT>, dim3(g), dim3(NF_BLOCK), 0, st, z, log_scale, bias, P, L, U, L_mask, U_mask, sign_s, log_s, h, z1c, W_out, ld, s, B, Px); break;
    switch (C) { NF_CASE(1) NF_CASE(2) NF_CASE(3) NF_CASE(4) default: return NF_E_UNSUPPORTED; }
#undef NF_CASE
    NF_CHECK_LAUNCH();
    return 0;
}

template <int PART>
static int nf_gh_bwd_launch(const float* g_h, const float* g_z1c, const float* g_ld, const float* z, const float* log_scale, const float* bias,
                            const float* W_saved, float* g_z, float* g_log_scale, float* g_bias, float* g_W, float* sum_g_ld, int mode, int odd,
                            int64_t B, int C, int H, int W, nf_stream_t stream) {
    NfSplit s;
    if (!nf_make_split(s, mode, odd, C, H, W) || mode == NF_SPLIT_NONE) return NF_E_BADARG;
    if (C > NF_HEAD_MAXC) return NF_E_UNSUPPORTED;
    if (B == 0) return 0;
    const int Px = H * W;
    const int thr = 512;                                     // (B = 64, 32 x 32: 10.4 us at 1024 threads, 8.8 at 512, 9.1 at 256)
    unsigned g = nf_grid_for(B * Px, thr * 2);
    if (g > 256) g = 256;
    hipStream_t st = (hipStream_t)stream;
#define NF_CASE(CT) case CT: hipLaunchKernelGGL((k_glow_head_bwd<CT, PART>), dim3(g), dim3(thr), 0, st, g_h, g_z1c, g_ld, z, log_scale, bias, W_saved, g_z, g_log_scale, g_bias, g_W, sum_g_ld, s, B, Px); break;
    switch (C) { NF_CASE(1) NF_CASE(2) NF_CASE(3) NF_CASE(4) default: return NF_E_UNSUPPORTED; }
#undef NF_CASE
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_glow_head_bwd(const float* g_h, const float* g_z1c, const float* g_ld, const float* z,
                                const float* log_scale, const float* bias, const float* W_saved, float* g_z,
                                float* g_log_scale, float* g_bias, float* g_W, float* sum_g_ld, int mode, int odd,
                                int64_t B, int C, int H, int W, nf_stream_t stream) {
    return nf_gh_bwd_launch<0>(g_h, g_z1c, g_ld, z, log_scale, bias, W_saved, g_z, g_log_scale, g_bias, g_W, sum_g_ld, mode, odd, B, C, H, W, stream);
}

// the data gradient alone (the backward pass waits for nothing else) ...
extern "C" int nf_glow_head_bwd_data(const float* g_h, const float* g_z1c, const float* log_scale, const float* W_saved, float* g_z, int mode,
                                     int odd, int64_t B, int C, int H, int W, nf_stream_t stream) {
    if (g_h == nullptr || log_scale == nullptr || W_saved == nullptr || g_z == nullptr) return NF_E_BADARG;
    // (z and bias are not read for the data gradient: the compiler drops their loads; any valid address serves)
    return nf_gh_bwd_launch<1>(g_h, g_z1c, nullptr, g_h, log_scale, log_scale, W_saved, g_z, nullptr, nullptr, nullptr, nullptr, mode, odd, B, C, H,
                               W, stream);
}

// ... and g_log_scale, g_bias, g_W, sum_g_ld (+=) of n heads of one shape and split mode in one launch (their `odd` per head)
extern "C" int nf_glow_head_bwd_params_multi(const nf_glow_head_small_params_desc* descs, int n, int mode, int64_t B, int C, int H, int W,
                                             nf_stream_t stream) {
    NfSplit s;
    if (!nf_make_split(s, mode, 0, C, H, W) || mode == NF_SPLIT_NONE) return NF_E_BADARG;
    if (C > NF_HEAD_MAXC) return NF_E_UNSUPPORTED;
    if (descs == nullptr || n < 1 || n > NF_GLOW_HEAD_MULTI_MAX) return NF_E_BADARG;
    if (B == 0) return 0;
    NfGhsMulti m;
    for (int i = 0; i < n; ++i) {
        const nf_glow_head_small_params_desc& d = descs[i];
        if (d.g_h == nullptr || d.g_ld == nullptr || d.z == nullptr || d.log_scale == nullptr || d.bias == nullptr || d.W_saved == nullptr ||
            d.g_log_scale == nullptr || d.g_bias == nullptr || d.g_W == nullptr)
            return NF_E_BADARG;
        m.d[i] = d;
    }
    const int Px = H * W;
    const int thr = 512;
    unsigned g = nf_grid_for(B * Px, thr * 2);
    const unsigned cap = 256u / (unsigned)n > 8u ? 256u / (unsigned)n : 8u;
    if (g > cap) g = cap;
    hipStream_t st = (hipStream_t)stream;
#define NF_CASE(CT) case CT: hipLaunchKernelGGL(k_glow_head_params_multi<CT>, dim3(g, (unsigned)n), dim3(thr), 0, st, m, s, B, Px); break;
    switch (C) { NF_CASE(1) NF_CASE(2) NF_CASE(3) NF_CASE(4) default: return NF_E_UNSUPPORTED; }
#undef NF_CASE
    NF_CHECK_LAUNCH();
    return 0;
}
