// The whole MLP conditioner of AffineCoupling (flows/modules.py:393-413: WN-Linear -> 2 x [BN, ReLU, WN-Linear, BN, ReLU,
// WN-Linear, + skip] -> BN, ReLU, WN-Linear) as ONE persistent launch per direction, for batches that fit the chip at one
// 16-row tile per wave (N <= NF_MLP_MAX_ROWS).
//
// Why: at the reference's batch sizes a linear + BatchNorm launch is pure latency (7.6 us forward / 10.2 us backward for a
// 4096 x 32 x 32 layer, 8 MFLOP), and training-mode BatchNorm forces a device-wide reduction between any two linears.
// A software grid barrier over <= 64 co-resident workgroups costs 1.1 - 4 us including the statistics exchange
// (tools/probes/grid_barrier_probe.hip: 16 workgroups 1.7 us), a kernel boundary + prologue costs 6 - 8 us.  So: one
// workgroup of 16 waves per 256 rows, every wave keeps its 16-row tile of every activation in registers in the
// row-per-lane layout of nf_mfma16.h (linears chain register to register on v_mfma_f32_16x16x4_f32), the weights of all
// six linears sit in LDS, and the only global traffic between two linears is 64 atomics per workgroup + the barrier.
//
// Numerics are those of linear_bn.hip (the multi-launch path used for larger batches): statistics centred at the
// producing linear's bias, biased variance for normalisation, unbiased for the running estimate, weight-norm as a scale of
// the activation column (weight_norm.py:40).
#include "nf_common.h"
#include "nf_mfma16.h"

#define NF_MC_WAVES 16
#define NF_MC_THREADS (NF_MC_WAVES * NF_WAVE)
#define NF_MC_NL NF_MLP_LINEARS
#define NF_MC_NB NF_MLP_BNS

struct NfMlpP {
    const float* v[NF_MC_NL]; const float* g[NF_MC_NL]; const float* b[NF_MC_NL];
    const float* gamma[NF_MC_NB]; const float* beta[NF_MC_NB];
    float* rmean[NF_MC_NB]; float* rvar[NF_MC_NB]; int64_t* nbt[NF_MC_NB];
};

static inline void nf_mlp_unpack(const void* const* t, NfMlpP& p) {
    for (int l = 0; l < NF_MC_NL; ++l) {
        p.v[l] = (const float*)t[3 * l]; p.g[l] = (const float*)t[3 * l + 1]; p.b[l] = (const float*)t[3 * l + 2];
    }
    for (int j = 0; j < NF_MC_NB; ++j) {
        const void* const* q = t + 3 * NF_MC_NL + 5 * j;
        p.gamma[j] = (const float*)q[0]; p.beta[j] = (const float*)q[1];
        p.rmean[j] = (float*)q[2]; p.rvar[j] = (float*)q[3]; p.nbt[j] = (int64_t*)q[4];
    }
}

// LDS (floats)
#define NF_MC_W 0                                         // [6][32 * 36] weight_v, zero padded
#define NF_MC_WS (NF_MC_W + NF_MC_NL * 32 * NF_FP_ST)     // [6][32] weight-norm column scales g_k / (||v[:, k]|| + eps)
#define NF_MC_B (NF_MC_WS + NF_MC_NL * 32)                // [6][32] biases
#define NF_MC_GA (NF_MC_B + NF_MC_NL * 32)                // [5][32] gamma
#define NF_MC_BE (NF_MC_GA + NF_MC_NB * 32)               // [5][32] beta
#define NF_MC_BNC (NF_MC_BE + NF_MC_NB * 32)              // [5][4][32] per BatchNorm: scale, shift, mean, invstd
#define NF_MC_RED (NF_MC_BNC + NF_MC_NB * 4 * 32)         // [16][64] cross-wave reduction
#define NF_MC_TILES (NF_MC_RED + NF_MC_WAVES * 64)        // per-wave 16 x 36 tiles

// arrive + spin on a monotonically increasing counter (zero at launch); every workgroup of the grid is resident by
// construction (grid <= NF_MLP_MAX_BLOCKS, one workgroup per CU fits), the spin is bounded so a mistake cannot hang the box
__device__ __forceinline__ void nf_grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(counter, 1u);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) break;
        }
        __threadfence();
    }
    __syncthreads();
}

__device__ __forceinline__ void nf_mc_stage(const NfMlpP& p, float* sm, int I0, int O_out, float wn_eps) {
    const int tid = threadIdx.x, oo = tid >> 5, k = tid & 31;
    float w[NF_MC_NL];
#pragma unroll
    for (int l = 0; l < NF_MC_NL; ++l) {                  // six independent loads in flight, one latency
        const int I = l == 0 ? I0 : 32, O = l == NF_MC_NL - 1 ? O_out : 32;
        w[l] = (oo < O && k < I) ? p.v[l][oo * I + k] : 0.f;
    }
    float gk = 0.f, bk = 0.f, ga = 0.f, be = 0.f;
    if (tid < NF_MC_NL * 32) {
        const int l = tid >> 5;
        const int I = l == 0 ? I0 : 32, O = l == NF_MC_NL - 1 ? O_out : 32;
        gk = k < I ? p.g[l][k] : 0.f;
        bk = k < O ? p.b[l][k] : 0.f;
    }
    if (tid < NF_MC_NB * 32) { ga = p.gamma[tid >> 5][k]; be = p.beta[tid >> 5][k]; }
#pragma unroll
    for (int l = 0; l < NF_MC_NL; ++l) sm[NF_MC_W + l * 32 * NF_FP_ST + oo * NF_FP_ST + k] = w[l];
    if (tid < NF_MC_NL * 32) sm[NF_MC_B + tid] = bk;
    if (tid < NF_MC_NB * 32) { sm[NF_MC_GA + tid] = ga; sm[NF_MC_BE + tid] = be; }
    __syncthreads();
    if (tid < NF_MC_NL * 32) {                            // weight_norm.py:40: norm over the output index, per input column
        const float* W = sm + NF_MC_W + (tid >> 5) * 32 * NF_FP_ST;
        float ss = 0.f;
#pragma unroll 8
        for (int o = 0; o < 32; ++o) ss = fmaf(W[o * NF_FP_ST + k], W[o * NF_FP_ST + k], ss);
        const int I = (tid >> 5) == 0 ? I0 : 32;
        sm[NF_MC_WS + tid] = k < I ? gk / (sqrtf(ss) + wn_eps) : 0.f;
    }
    __syncthreads();
}

// out^T = W_l act^T for a 32-wide (zero padded) layer
__device__ __forceinline__ void nf_mc_linear(const float* sm, int l, const float (&av)[8], float (&dv)[8], int c16, int g) {
    f32x4 acc[2] = {nf_fp_zero4(), nf_fp_zero4()};
    nf_fp_gemm<2>(sm + NF_MC_W + l * 32 * NF_FP_ST, NF_FP_ST, 0, av, acc, c16, g);
#pragma unroll
    for (int j = 0; j < 8; ++j) dv[j] = acc[j >> 2][j & 3];
}
// BatchNorm j -> ReLU -> weight-norm scale of linear l (what linear l multiplies with)
__device__ __forceinline__ void nf_mc_activate(const float* sm, int j, int l, const float (&a)[8], float (&av)[8], int g) {
    float sc[8], sh[8], ws[8];
    nf_fp_ldvec(sm + NF_MC_BNC + (4 * j + 0) * 32, g, sc);
    nf_fp_ldvec(sm + NF_MC_BNC + (4 * j + 1) * 32, g, sh);
    nf_fp_ldvec(sm + NF_MC_WS + l * 32, g, ws);
#pragma unroll
    for (int k = 0; k < 8; ++k) av[k] = fmaxf(fmaf(a[k], sc[k], sh[k]), 0.f) * ws[k];
}

// batch statistics of BatchNorm j over the whole grid (dv = pre-bias output of the producing linear, zero in invalid rows),
// then its constants -> sm[NF_MC_BNC + 4 j ..]; workgroup 0 does the running-statistics bookkeeping
__device__ __forceinline__ void nf_mc_batchnorm_stats(float* sm, const NfMlpP& p, int j, int lprod, const float (&dv)[8], bool rv,
                                                      float* stats, unsigned* counter, float* save, int64_t N, float eps,
                                                      float mom, int c16, int g, int wid) {
    float* tile = sm + NF_MC_TILES + wid * 16 * NF_FP_ST;
    float m[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) m[k] = rv ? dv[k] : 0.f;
    nf_fp_store_rows(m, tile, c16, g);
    nf_fp_wsync();
    float c[2][4];
    nf_fp_load_cols<2>(tile, c, c16, g);
    nf_fp_wsync();
    float s1[2], s2[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        s1[cb] = (c[cb][0] + c[cb][1]) + (c[cb][2] + c[cb][3]);
        s2[cb] = fmaf(c[cb][0], c[cb][0], fmaf(c[cb][1], c[cb][1], fmaf(c[cb][2], c[cb][2], c[cb][3] * c[cb][3])));
        s1[cb] = nf_fp_rowsum(s1[cb]);
        s2[cb] = nf_fp_rowsum(s2[cb]);
    }
    float* red = sm + NF_MC_RED;
    if (g == 0) {
        red[wid * 64 + c16] = s1[0]; red[wid * 64 + 16 + c16] = s1[1];
        red[wid * 64 + 32 + c16] = s2[0]; red[wid * 64 + 48 + c16] = s2[1];
    }
    __syncthreads();
    float* st = stats + (size_t)j * NF_STAT_REPL * 64;
    if (threadIdx.x < 64) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NF_MC_WAVES; ++w) t += red[w * 64 + threadIdx.x];
        atomicAdd(st + (blockIdx.x % NF_STAT_REPL) * 64 + threadIdx.x, t);
    }
    nf_grid_barrier(counter, (unsigned)(j + 1) * gridDim.x);
    if (threadIdx.x < 32) {
        const int k = threadIdx.x;
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int r = 0; r < NF_STAT_REPL; ++r) {
            t1 += __hip_atomic_load(st + r * 64 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            t2 += __hip_atomic_load(st + r * 64 + 32 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const float invN = 1.f / (float)N;
        const float m1 = t1 * invN;
        const float mean = sm[NF_MC_B + lprod * 32 + k] + m1;
        const float var = fmaxf(t2 * invN - m1 * m1, 0.f);                         // biased, as BatchNorm normalises
        const float invstd = 1.f / sqrtf(var + eps);
        const float sc = sm[NF_MC_GA + j * 32 + k] * invstd;
        sm[NF_MC_BNC + (4 * j + 0) * 32 + k] = sc;
        sm[NF_MC_BNC + (4 * j + 1) * 32 + k] = sm[NF_MC_BE + j * 32 + k] - mean * sc;
        sm[NF_MC_BNC + (4 * j + 2) * 32 + k] = mean;
        sm[NF_MC_BNC + (4 * j + 3) * 32 + k] = invstd;
        if (blockIdx.x == 0) {
            const float unb = N > 1 ? var * ((float)N / (float)(N - 1)) : var;
            p.rmean[j][k] = (1.f - mom) * p.rmean[j][k] + mom * mean;
            p.rvar[j][k] = (1.f - mom) * p.rvar[j][k] + mom * unb;
            save[(2 * j + 0) * 32 + k] = mean;
            save[(2 * j + 1) * 32 + k] = invstd;
            if (k == 0 && p.nbt[j] != nullptr) p.nbt[j][0] += 1;
        }
    }
    __syncthreads();
}
// evaluation mode / the backward's recomputation: constants from given mean / invstd
__device__ __forceinline__ void nf_mc_batchnorm_consts(float* sm, int j, float mean, float invstd) {
    const int k = threadIdx.x & 31;
    const float sc = sm[NF_MC_GA + j * 32 + k] * invstd;
    sm[NF_MC_BNC + (4 * j + 0) * 32 + k] = sc;
    sm[NF_MC_BNC + (4 * j + 1) * 32 + k] = sm[NF_MC_BE + j * 32 + k] - mean * sc;
    sm[NF_MC_BNC + (4 * j + 2) * 32 + k] = mean;
    sm[NF_MC_BNC + (4 * j + 3) * 32 + k] = invstd;
}

__device__ __forceinline__ void nf_mc_load_x(const float* x, int64_t row, bool rv, int I0, float (&xa)[8], int g) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 16 * (j >> 2) + 4 * g + (j & 3);
        const float v = x[(rv ? row : 0) * I0 + (k < I0 ? k : 0)];
        xa[j] = (rv && k < I0) ? v : 0.f;
    }
}

__global__ void __launch_bounds__(NF_MC_THREADS) k_mlp_chain_fwd(const float* __restrict__ x, NfMlpP p, float* __restrict__ out,
                                                                 float* save, float* stats, int64_t N, int I0, int O_out,
                                                                 int training, float eps, float mom, float wn_eps) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    nf_mc_stage(p, sm, I0, O_out, wn_eps);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, c16 = lane & 15, g = lane >> 4;
    const int64_t row = ((int64_t)blockIdx.x * NF_MC_WAVES + wid) * 16 + c16;
    const bool rv = row < N;
    unsigned* counter = (unsigned*)(stats + NF_MC_NB * NF_STAT_REPL * 64);
    if (!training) {
        if (threadIdx.x < NF_MC_NB * 32) {
            const int j = threadIdx.x >> 5, k = threadIdx.x & 31;
            nf_mc_batchnorm_consts(sm, j, p.rmean[j][k], 1.f / sqrtf(p.rvar[j][k] + eps));
        }
        __syncthreads();
    }
    float a_in[8], av[8], dv[8], bias[8], stream[8];
    nf_mc_load_x(x, row, rv, I0, a_in, g);
    {   // linear 0: no BatchNorm in front, only the weight-norm scale
        float ws[8];
        nf_fp_ldvec(sm + NF_MC_WS, g, ws);
#pragma unroll
        for (int k = 0; k < 8; ++k) av[k] = a_in[k] * ws[k];
    }
    nf_mc_linear(sm, 0, av, dv, c16, g);
#pragma unroll 1
    for (int l = 0; l < NF_MC_NL - 1; ++l) {              // dv = pre-bias output of linear l = input of BatchNorm l
        if (l > 0 && (l & 1) == 0) {                      // second linear of a residual block: add the block input
#pragma unroll
            for (int k = 0; k < 8; ++k) dv[k] += stream[k];
        }
        if (training) nf_mc_batchnorm_stats(sm, p, l, l, dv, rv, stats, counter, save, N, eps, mom, c16, g, wid);
        nf_fp_ldvec(sm + NF_MC_B + l * 32, g, bias);
#pragma unroll
        for (int k = 0; k < 8; ++k) a_in[k] = dv[k] + bias[k];
        if ((l & 1) == 0) {                               // acts[0], acts[2], acts[4] are the residual stream
#pragma unroll
            for (int k = 0; k < 8; ++k) stream[k] = a_in[k];
        }
        nf_mc_activate(sm, l, l + 1, a_in, av, g);
        nf_mc_linear(sm, l + 1, av, dv, c16, g);
    }
    nf_fp_ldvec(sm + NF_MC_B + (NF_MC_NL - 1) * 32, g, bias);
    if (rv) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * (j >> 2) + 4 * g + (j & 3);
            if (k < O_out) out[row * O_out + k] = dv[j] + bias[j];
        }
    }
}

static inline size_t nf_mc_lds_bytes(int tiles_per_wave) {
    return (size_t)(NF_MC_TILES + NF_MC_WAVES * tiles_per_wave * 16 * NF_FP_ST) * sizeof(float);
}

extern "C" int nf_mlp_chain_fwd(const float* x, const void* const* params, float* out, float* save_stats, float* ws_zero,
                                int64_t N, int I0, int O_out, int training, float bn_eps, float bn_momentum, float wn_eps,
                                nf_stream_t stream) {
    if (params == nullptr || I0 < 1 || I0 > 32 || O_out < 1 || O_out > 32 || N > NF_MLP_MAX_ROWS) return NF_E_BADARG;
    if (N <= 0) return N == 0 ? 0 : NF_E_BADARG;
    NfMlpP p;
    nf_mlp_unpack(params, p);
    const unsigned grid = (unsigned)((N + NF_MLP_ROWS_PER_BLOCK - 1) / NF_MLP_ROWS_PER_BLOCK);
    const size_t lds = nf_mc_lds_bytes(1);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_mlp_chain_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_mlp_chain_fwd, dim3(grid), dim3(NF_MC_THREADS), lds, (hipStream_t)stream, x, p, out, save_stats, ws_zero,
                       N, I0, O_out, training, bn_eps, bn_momentum, wn_eps);
    NF_CHECK_LAUNCH();
    return 0;
}
