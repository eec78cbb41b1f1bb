// Fused training-mode flow BatchNorm (flows/modules.py:283-307) for the head of a RealNVP / MAF flow step:
//   stats : ONE pass of shifted sums  sum_c += sum (x - k_c),  sqsum_c += sum (x - k_c)^2  with the centre k = the running
//           mean (a good estimate after the first steps, and any centre is exact in exact arithmetic);
//   apply : mean / biased variance (+eps inside, modules.py:286-287) from the sums in the prologue, batch / running
//           buffers updated by block 0, y = (x - mean)/sqrt(var) * exp(log_gamma) + beta, ld += P * sum(log_gamma -
//           0.5 log var), and -- optionally -- the conditioning half of the following coupling gathered in the same
//           launch (coupling.py:33).
// 2 launches instead of 6 (sum, sqdev, finalize, apply, zero-fill, gather); backward (affine=False: statistics are
// constants for autograd) = scale + scatter-add of the conditioner-input gradient in 1 launch instead of 3.
#include "nf_common.h"
#include "nf_det.h"

NF_DET_STATE(nf_fbh)
NF_DET_HOST_API(nf_fbh)

__global__ void __launch_bounds__(NF_BLOCK) k_flowbn_stats(const float* __restrict__ x, const float* __restrict__ center,
                                                           float* __restrict__ ws, int64_t B, int C, int P,
                                                           int64_t items_per_block) {
    __shared__ float scratch[NF_BLOCK / NF_WAVE];
    const int c = blockIdx.x;
    const float k = center[c];
    const int64_t n = B * P;
    const int64_t q0 = (int64_t)blockIdx.y * items_per_block;
    const int64_t q1 = min(q0 + items_per_block, n);
    float s1 = 0.f, s2 = 0.f;
    for (int64_t q = q0 + threadIdx.x; q < q1; q += blockDim.x) {
        const int64_t b = q / P;
        const float v = x[(b * C + c) * P + (q - b * P)] - k;
        s1 += v;
        s2 = fmaf(v, v, s2);
    }
    const float t1 = nf_block_sum(s1, scratch);
    const float t2 = nf_block_sum(s2, scratch);
    if (threadIdx.x == 0) {
        NF_DET_ENTER_COL(nf_fbh);     // (grid (channel, block): the blocks of a channel take turns, the channels side by side)
        atomicAdd(ws + c, t1);
        atomicAdd(ws + C + c, t2);
        NF_DET_LEAVE_COL(nf_fbh);
        if (blockIdx.y == 0) ws[2 * C + c] = k;       // the centre that was used (running_mean changes in the next launch)
    }
}

struct NfBnCoef { float mean, sd, eg, beta; };

__global__ void __launch_bounds__(NF_BLOCK) k_flowbn_head_fwd(const float* __restrict__ x, const float* __restrict__ ws,
                                                              const float* __restrict__ log_gamma,
                                                              const float* __restrict__ beta, float* __restrict__ bmean,
                                                              float* __restrict__ bvar, float* __restrict__ rmean,
                                                              float* __restrict__ rvar, float eps, float mom,
                                                              float* __restrict__ y, float* __restrict__ z1c,
                                                              float* __restrict__ ld, NfSplit s, int64_t B, int P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    NfBnCoef* coef = reinterpret_cast<NfBnCoef*>(lds);
    const int C = s.C;
    const float n = (float)B * (float)P;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float m1 = ws[c] / n;
        const float mean = ws[2 * C + c] + m1;
        const float var = fmaxf(ws[C + c] / n - m1 * m1, 0.f) + eps;            // biased, eps inside (modules.py:287)
        if (blockIdx.x == 0) {
            bmean[c] = mean;
            bvar[c] = var;
            rmean[c] = rmean[c] * (1.f - mom) + mean * mom;                      // modules.py:291-294
            rvar[c] = rvar[c] * (1.f - mom) + var * mom;
        }
        NfBnCoef k;
        k.mean = mean; k.sd = sqrtf(var); k.eg = expf(log_gamma[c]); k.beta = beta[c];
        coef[c] = k;
        lds[4 * C + c] = log_gamma[c] - 0.5f * logf(var);
    }
    __syncthreads();
    float dld = 0.f;
    for (int c = 0; c < C; ++c) dld += lds[4 * C + c];
    dld *= (float)P;                                                             // modules.py:303-305
    const int64_t total = B * C * P;
    const int64_t gstride = (int64_t)gridDim.x * blockDim.x;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool gather = z1c != nullptr;
    for (int64_t t = gtid; t < total; t += gstride) {
        const int64_t bc = t / P;
        const int p = (int)(t - bc * P);
        const int64_t b = bc / C;
        const int c = (int)(bc - b * C);
        const NfBnCoef k = coef[c];
        const float v = ((x[t] - k.mean) / k.sd) * k.eg + k.beta;               // modules.py:300-301
        y[t] = v;
        if (gather) {
            int which, e;
            nf_full_to_half(s, c, p, which, e);
            if (which == 1) z1c[b * s.n_half + e] = v;
        }
    }
    for (int64_t b = gtid; b < B; b += gstride) ld[b] += dld;
}

// g_x = (g_h + scatter(g_z1c)) * exp(log_gamma) / sqrt(var)         (appendix B4, affine=False)
__global__ void __launch_bounds__(NF_BLOCK) k_flowbn_head_bwd(const float* __restrict__ gh, const float* __restrict__ gz1c,
                                                              const float* __restrict__ var,
                                                              const float* __restrict__ log_gamma, float* __restrict__ gx,
                                                              NfSplit s, int64_t B, int P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int C = s.C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) { lds[c] = sqrtf(var[c]); lds[C + c] = expf(log_gamma[c]); }
    __syncthreads();
    const int64_t total = B * C * P;
    const bool scatter = gz1c != nullptr;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t bc = t / P;
        const int p = (int)(t - bc * P);
        const int64_t b = bc / C;
        const int c = (int)(bc - b * C);
        float g = gh[t];
        if (scatter) {
            int which, e;
            nf_full_to_half(s, c, p, which, e);
            if (which == 1) g += gz1c[b * s.n_half + e];
        }
        gx[t] = g / lds[c] * lds[C + c];
    }
}

extern "C" int nf_flowbn_stats(const float* x, const float* center, float* ws, int64_t B, int C, int P,
                               nf_stream_t stream) {
    if (C <= 0 || P <= 0) return NF_E_BADARG;
    if (B == 0) return 0;
    const int64_t n = B * P;
    int64_t chunks = (n + 4 * NF_BLOCK - 1) / (4 * NF_BLOCK);
    const int64_t cap = (1024 + C - 1) / C;
    if (chunks > cap) chunks = cap;
    if (chunks < 1) chunks = 1;
    const int64_t ipb = (n + chunks - 1) / chunks;
    chunks = (n + ipb - 1) / ipb;
    hipLaunchKernelGGL(k_flowbn_stats, dim3((unsigned)C, (unsigned)chunks), dim3(NF_BLOCK), 0, (hipStream_t)stream, x,
                       center, ws, B, C, P, ipb);
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_flowbn_head_fwd(const float* x, const float* ws, const float* log_gamma, const float* beta,
                                  float* batch_mean, float* batch_var, float* running_mean, float* running_var, float eps,
                                  float momentum, float* y, float* z1c, float* ld, int mode, int odd, int64_t B, int C,
                                  int H, int W, nf_stream_t stream) {
    NfSplit s;
    if (!nf_make_split(s, z1c != nullptr ? mode : NF_SPLIT_NONE, odd, C, H, W) || C > 2048) return NF_E_BADARG;
    if (B == 0) return 0;
    const int Px = H * W;
    unsigned g = nf_grid_for(B * C * Px);
    const unsigned g_ld = nf_grid_for(B);
    if (g < g_ld) g = g_ld;
    hipLaunchKernelGGL(k_flowbn_head_fwd, dim3(g), dim3(NF_BLOCK), (size_t)5 * C * sizeof(float), (hipStream_t)stream, x, ws,
                       log_gamma, beta, batch_mean, batch_var, running_mean, running_var, eps, momentum, y, z1c, ld, s, B, Px);
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_flowbn_head_bwd(const float* g_h, const float* g_z1c, const float* var, const float* log_gamma,
                                  float* g_x, int mode, int odd, int64_t B, int C, int H, int W, nf_stream_t stream) {
    NfSplit s;
    if (!nf_make_split(s, g_z1c != nullptr ? mode : NF_SPLIT_NONE, odd, C, H, W) || C > 2048) return NF_E_BADARG;
    if (B == 0) return 0;
    const int Px = H * W;
    hipLaunchKernelGGL(k_flowbn_head_bwd, dim3(nf_grid_for(B * C * Px)), dim3(NF_BLOCK), (size_t)2 * C * sizeof(float),
                       (hipStream_t)stream, g_h, g_z1c, var, log_gamma, g_x, s, B, Px);
    NF_CHECK_LAUNCH();
    return 0;
}
