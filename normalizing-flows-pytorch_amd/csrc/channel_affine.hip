// Per-channel affine bijectors: ActNorm (flows/modules.py:225-256) and the flow BatchNorm (modules.py:259-322),
// plus the per-channel batch statistics both need (ActNorm data-dependent init :238-244, flow-BN :284-294).
// All HBM-bound: 8 B/element forward/inverse, 4 B/element for a statistics pass.
//
// Every variant is y = ((x - A_c) / D_c) * M_c + S_c with the four per-channel coefficients staged in LDS once per
// block, chosen so that the operation ORDER is the reference's (division where it divides, multiply where it
// multiplies) -- keeps us within an ulp of the CPU path instead of relying on the 1e-5 budget.
#include "nf_common.h"

struct NfCoef { float A, D, M, S; };

__device__ __forceinline__ NfCoef nf_coef(int op, int inverse, const float* p0, const float* p1, const float* p2,
                                          const float* p3, int c) {
    NfCoef k;
    if (op == NF_ACTNORM) {
        const float e = expf(p0[c]), b = p1[c];
        if (!inverse) { k.A = b; k.D = e; k.M = 1.f; k.S = 0.f; }       // (z - bias) / exp(log_scale)   modules.py:246
        else          { k.A = 0.f; k.D = 1.f; k.M = e; k.S = b; }       // y * exp(log_scale) + bias     modules.py:253
    } else {
        const float mean = p0[c], sd = sqrtf(p1[c]), eg = expf(p2[c]), beta = p3[c];
        if (!inverse) { k.A = mean; k.D = sd; k.M = eg; k.S = beta; }    // modules.py:300-301
        else          { k.A = beta; k.D = eg; k.M = sd; k.S = mean; }    // modules.py:315-316
    }
    return k;
}

// the layer's scalar log-det per pixel (same for every sample)
__device__ __forceinline__ float nf_chan_logdet(int op, int inverse, const float* p0, const float* p1, const float* p2,
                                                int C) {
    float s = 0.f;
    if (op == NF_ACTNORM) {
        for (int c = 0; c < C; ++c) s += p0[c];
        return inverse ? s : -s;                                          // modules.py:249, :255
    }
    for (int c = 0; c < C; ++c) s += p2[c] - 0.5f * logf(p1[c]);          // modules.py:304, :319
    return inverse ? -s : s;
}

__global__ void __launch_bounds__(NF_BLOCK) k_chan_affine_fwd(int op, int inverse, const float* __restrict__ x,
                                                              const float* __restrict__ p0, const float* __restrict__ p1,
                                                              const float* __restrict__ p2, const float* __restrict__ p3,
                                                              float* __restrict__ y, float* __restrict__ ld, int64_t B,
                                                              int C, int P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];          // 4*C coefficients + 1 log-det
    NfCoef* coef = reinterpret_cast<NfCoef*>(lds);
    for (int c = threadIdx.x; c < C; c += blockDim.x) coef[c] = nf_coef(op, inverse, p0, p1, p2, p3, c);
    if (threadIdx.x == 0) lds[4 * C] = nf_chan_logdet(op, inverse, p0, p1, p2, C) * (float)P;
    __syncthreads();
    const int64_t total = B * C * P;
    const int64_t gstride = (int64_t)gridDim.x * blockDim.x;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t t = gtid; t < total; t += gstride) {
        const int c = (int)((t / P) % C);
        const NfCoef k = coef[c];
        y[t] = ((x[t] - k.A) / k.D) * k.M + k.S;
    }
    if (ld != nullptr) {
        const float d = lds[4 * C];
        for (int64_t b = gtid; b < B; b += gstride) ld[b] += d;
    }
}

// g_x = g_y / D_c * M_c  (flow-BN with affine=False: the statistics are constants for autograd, appendix B4)
__global__ void __launch_bounds__(NF_BLOCK) k_chan_scale_bwd(int op, const float* __restrict__ gy,
                                                             const float* __restrict__ p0, const float* __restrict__ p1,
                                                             const float* __restrict__ p2, const float* __restrict__ p3,
                                                             float* __restrict__ gx, int64_t B, int C, int P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    NfCoef* coef = reinterpret_cast<NfCoef*>(lds);
    for (int c = threadIdx.x; c < C; c += blockDim.x) coef[c] = nf_coef(op, 0, p0, p1, p2, p3, c);
    __syncthreads();
    const int64_t total = B * C * P;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const NfCoef k = coef[(int)((t / P) % C)];
        gx[t] = gy[t] / k.D * k.M;
    }
}

// channel-major traversal: block (c, chunk) walks items q = b*P + p of channel c
__device__ __forceinline__ int64_t nf_chan_addr(int64_t q, int c, int C, int P) {
    const int64_t b = q / P;
    return (b * C + c) * P + (q - b * P);
}

// autograd with parameter gradients (appendix B2; B4 with affine=True)
__global__ void __launch_bounds__(NF_BLOCK) k_chan_affine_bwd(int op, const float* __restrict__ gy,
                                                              const float* __restrict__ gld, const float* __restrict__ x,
                                                              const float* __restrict__ p0, const float* __restrict__ p1,
                                                              const float* __restrict__ p2, const float* __restrict__ p3,
                                                              float* __restrict__ gx, float* __restrict__ g_pa,
                                                              float* __restrict__ g_pb, int64_t B, int C, int P,
                                                              int64_t items_per_block) {
    __shared__ float scratch[NF_BLOCK / NF_WAVE];
    const int c = blockIdx.x;
    const NfCoef k = nf_coef(op, 0, p0, p1, p2, p3, c);
    const int64_t n = B * P;
    const int64_t q0 = (int64_t)blockIdx.y * items_per_block;
    const int64_t q1 = min(q0 + items_per_block, n);
    float r1 = 0.f, r2 = 0.f;
    for (int64_t q = q0 + threadIdx.x; q < q1; q += blockDim.x) {
        const int64_t a = nf_chan_addr(q, c, C, P);
        const float g = gy[a];
        const float yv = ((x[a] - k.A) / k.D) * k.M;   // y - S
        gx[a] = g / k.D * k.M;
        r1 += g;
        r2 += g * yv;
    }
    float sg = 0.f;                                  // this block's share of sum_b g_ld (every channel needs the total)
    for (int64_t b = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; b < B; b += (int64_t)gridDim.y * blockDim.x)
        sg += gld[b];
    const float R1 = nf_block_sum(r1, scratch);
    const float R2 = nf_block_sum(r2, scratch);
    const float SG = nf_block_sum(sg, scratch);
    if (threadIdx.x == 0) {
        if (op == NF_ACTNORM) {   // g_log_scale = -sum g*y - P*sum g_ld ; g_bias = -sum g / exp(log_scale)
            atomicAdd(g_pa + c, -R2 - (float)P * SG);
            atomicAdd(g_pb + c, -R1 / k.D);
        } else {                  // g_log_gamma = sum g*(y-beta) + P*sum g_ld ; g_beta = sum g
            atomicAdd(g_pa + c, R2 + (float)P * SG);
            atomicAdd(g_pb + c, R1);
        }
    }
}

template <bool SQDEV>
__global__ void __launch_bounds__(NF_BLOCK) k_chan_stat(const float* __restrict__ x, const float* __restrict__ sum,
                                                        float* __restrict__ out, int64_t B, int C, int P,
                                                        int64_t items_per_block) {
    __shared__ float scratch[NF_BLOCK / NF_WAVE];
    const int c = blockIdx.x;
    const int64_t n = B * P;
    const float mean = SQDEV ? sum[c] / (float)n : 0.f;
    const int64_t q0 = (int64_t)blockIdx.y * items_per_block;
    const int64_t q1 = min(q0 + items_per_block, n);
    float acc = 0.f;
    for (int64_t q = q0 + threadIdx.x; q < q1; q += blockDim.x) {
        const float v = x[nf_chan_addr(q, c, C, P)] - mean;
        acc += SQDEV ? v * v : v;
    }
    const float tot = nf_block_sum(acc, scratch);
    if (threadIdx.x == 0) atomicAdd(out + c, tot);
}

__global__ void k_flowbn_finalize(const float* __restrict__ sum, const float* __restrict__ sqdev,
                                  float* __restrict__ bmean, float* __restrict__ bvar, float* __restrict__ rmean,
                                  float* __restrict__ rvar, float eps, float momentum, float n, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float m = sum[c] / n, v = sqdev[c] / n + eps;     // biased variance, eps stored inside  modules.py:286-287
    bmean[c] = m;
    bvar[c] = v;
    rmean[c] = rmean[c] * (1.f - momentum) + m * momentum;  // modules.py:291-294
    rvar[c] = rvar[c] * (1.f - momentum) + v * momentum;
}

__global__ void k_actnorm_init_finalize(const float* __restrict__ sum, const float* __restrict__ sqdev,
                                        float* __restrict__ log_scale, float* __restrict__ bias, float eps, float n,
                                        int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    log_scale[c] = logf(sqrtf(sqdev[c] / (n - 1.f)) + eps);  // unbiased std   modules.py:240
    bias[c] = sum[c] / n;                                    // modules.py:241
}

// ---------------------------------------------------------------------------------------------------------------
static inline void nf_chan_grid(int64_t n, int C, dim3& grid, int64_t& ipb) {
    int64_t chunks = (n + 4 * NF_BLOCK - 1) / (4 * NF_BLOCK);      // >= 4 items per thread
    const int64_t cap = (2048 + C - 1) / C;                        // ~2048 blocks in flight overall
    if (chunks > cap) chunks = cap;
    if (chunks < 1) chunks = 1;
    ipb = (n + chunks - 1) / chunks;
    chunks = (n + ipb - 1) / ipb;
    grid = dim3((unsigned)C, (unsigned)chunks);
}

extern "C" int nf_chan_affine_fwd(int op, const float* x, const float* p0, const float* p1, const float* p2,
                                  const float* p3, float* y, float* ld, int inverse, int64_t B, int C, int P,
                                  nf_stream_t stream) {
    if ((op != NF_ACTNORM && op != NF_FLOWBN) || C <= 0 || P <= 0 || C > 8192) return NF_E_BADARG;
    if (B == 0) return 0;
    const int64_t total = B * C * P;
    unsigned g = nf_grid_for(total);
    const unsigned g_ld = nf_grid_for(B);
    if (ld != nullptr && g < g_ld) g = g_ld;
    hipLaunchKernelGGL(k_chan_affine_fwd, dim3(g), dim3(NF_BLOCK), (4 * C + 1) * sizeof(float), (hipStream_t)stream, op,
                       inverse, x, p0, p1, p2, p3, y, ld, B, C, P);
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_chan_affine_bwd(int op, const float* g_y, const float* g_ld, const float* x, const float* p0,
                                  const float* p1, const float* p2, const float* p3, float* g_x, float* g_pa,
                                  float* g_pb, int64_t B, int C, int P, nf_stream_t stream) {
    if ((op != NF_ACTNORM && op != NF_FLOWBN) || C <= 0 || P <= 0 || C > 8192) return NF_E_BADARG;
    if (B == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (g_pa == nullptr || g_pb == nullptr) {
        hipLaunchKernelGGL(k_chan_scale_bwd, dim3(nf_grid_for(B * C * P)), dim3(NF_BLOCK), 4 * C * sizeof(float), st, op,
                           g_y, p0, p1, p2, p3, g_x, B, C, P);
    } else {
        dim3 grid;
        int64_t ipb;
        nf_chan_grid(B * P, C, grid, ipb);
        hipLaunchKernelGGL(k_chan_affine_bwd, grid, dim3(NF_BLOCK), 0, st, op, g_y, g_ld, x, p0, p1, p2, p3, g_x, g_pa,
                           g_pb, B, C, P, ipb);
    }
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_chan_sum(const float* x, float* sum, int64_t B, int C, int P, nf_stream_t stream) {
    if (C <= 0 || P <= 0) return NF_E_BADARG;
    if (B == 0) return 0;
    dim3 grid;
    int64_t ipb;
    nf_chan_grid(B * P, C, grid, ipb);
    hipLaunchKernelGGL(k_chan_stat<false>, grid, dim3(NF_BLOCK), 0, (hipStream_t)stream, x, (const float*)nullptr, sum, B,
                       C, P, ipb);
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_chan_sqdev(const float* x, const float* sum, float* sqdev, int64_t B, int C, int P,
                             nf_stream_t stream) {
    if (C <= 0 || P <= 0) return NF_E_BADARG;
    if (B == 0) return 0;
    dim3 grid;
    int64_t ipb;
    nf_chan_grid(B * P, C, grid, ipb);
    hipLaunchKernelGGL(k_chan_stat<true>, grid, dim3(NF_BLOCK), 0, (hipStream_t)stream, x, sum, sqdev, B, C, P, ipb);
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_flowbn_finalize(const float* sum, const float* sqdev, float* batch_mean, float* batch_var,
                                  float* running_mean, float* running_var, float eps, float momentum, int64_t n, int C,
                                  nf_stream_t stream) {
    if (C <= 0 || n <= 0) return NF_E_BADARG;
    hipLaunchKernelGGL(k_flowbn_finalize, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, sum, sqdev, batch_mean,
                       batch_var, running_mean, running_var, eps, momentum, (float)n, C);
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_actnorm_init_finalize(const float* sum, const float* sqdev, float* log_scale, float* bias, float eps,
                                        int64_t n, int C, nf_stream_t stream) {
    if (C <= 0 || n <= 1) return NF_E_BADARG;
    hipLaunchKernelGGL(k_actnorm_init_finalize, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, sum, sqdev,
                       log_scale, bias, eps, (float)n, C);
    NF_CHECK_LAUNCH();
    return 0;
}
