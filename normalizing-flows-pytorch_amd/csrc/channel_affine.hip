// Per-channel affine bijectors: ActNorm (flows/modules.py:225-256) and the flow BatchNorm (modules.py:259-322),
// plus the per-channel batch statistics both need (ActNorm data-dependent init :238-244, flow-BN :284-294).
// All HBM-bound: 8 B/element forward/inverse, 4 B/element for a statistics pass.
//
// Every variant is y = ((x - A_c) / D_c) * M_c + S_c with the four per-channel coefficients staged in LDS once per
// block, chosen so that the operation ORDER is the reference's (division where it divides, multiply where it
// multiplies) -- keeps us within an ulp of the CPU path instead of relying on the 1e-5 budget.
#include "nf_common.h"
#include "nf_det.h"

NF_DET_STATE(nf_ca)
NF_DET_HOST_API(nf_ca)

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
        NF_DET_ENTER_COL(nf_ca);      // (one thread per workgroup adds; the blocks of a channel meet in block order when the mode is on)
        if (op == NF_ACTNORM) {   // g_log_scale = -sum g*y - P*sum g_ld ; g_bias = -sum g / exp(log_scale)
            atomicAdd(g_pa + c, -R2 - (float)P * SG);
            atomicAdd(g_pb + c, -R1 / k.D);
        } else {                  // g_log_gamma = sum g*(y-beta) + P*sum g_ld ; g_beta = sum g
            atomicAdd(g_pa + c, R2 + (float)P * SG);
            atomicAdd(g_pb + c, R1);
        }
        NF_DET_LEAVE_COL(nf_ca);
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
    if (threadIdx.x == 0) { NF_DET_ENTER_COL(nf_ca); atomicAdd(out + c, tot); NF_DET_LEAVE_COL(nf_ca); }
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
// Vectorised variants (16-byte accesses, 32-bit index arithmetic -- the scalar kernels above spend their time on one 64-bit
// division per ELEMENT and on 4-byte loads: 31 .. 57 % of the copy rate; they remain the fallback for odd shapes).
//   IMG : P % 4 == 0  -> a float4 lies inside one (sample, channel) row of P pixels: one coefficient per vector
//   ROW : P == 1, C in {1, 2, 4} (2-D data) -> a float4 holds 4 / C samples, component j belongs to channel j % C
// total4 = B * C * P / 4 must fit 31 bits (the launchers check).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float nf_coef_apply(const NfCoef& k, float x) { return ((x - k.A) / k.D) * k.M + k.S; }

template <bool ROW>
__global__ void __launch_bounds__(NF_BLOCK) k_chan_affine_fwd_v4(int op, int inverse, const float4* __restrict__ x,
                                                                 const float* __restrict__ p0, const float* __restrict__ p1,
                                                                 const float* __restrict__ p2, const float* __restrict__ p3,
                                                                 float4* __restrict__ y, float* __restrict__ ld, int64_t B, int C,
                                                                 int P, unsigned total4) {
    extern __shared__ __attribute__((aligned(16))) float lds[];          // 4*C coefficients + 1 log-det
    NfCoef* coef = reinterpret_cast<NfCoef*>(lds);
    for (int c = threadIdx.x; c < C; c += blockDim.x) coef[c] = nf_coef(op, inverse, p0, p1, p2, p3, c);
    if (threadIdx.x == 0) lds[4 * C] = nf_chan_logdet(op, inverse, p0, p1, p2, C) * (float)P;
    __syncthreads();
    const unsigned gstride = gridDim.x * blockDim.x, gtid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned P4 = ROW ? 1u : (unsigned)P >> 2;
    NfCoef kr[4];
    if (ROW) {
#pragma unroll
        for (int j = 0; j < 4; ++j) kr[j] = coef[j % C];
    }
    for (unsigned t = gtid; t < total4; t += 2 * gstride) {              // two independent vectors in flight per trip
        const unsigned t2 = t + gstride;
        const bool has2 = t2 < total4;
        const float4 a = x[t], b = has2 ? x[t2] : a;
        float4 ra, rb;
        if (ROW) {
            ra = make_float4(nf_coef_apply(kr[0], a.x), nf_coef_apply(kr[1], a.y), nf_coef_apply(kr[2], a.z), nf_coef_apply(kr[3], a.w));
            rb = make_float4(nf_coef_apply(kr[0], b.x), nf_coef_apply(kr[1], b.y), nf_coef_apply(kr[2], b.z), nf_coef_apply(kr[3], b.w));
        } else {
            const NfCoef ka = coef[(t / P4) % (unsigned)C], kb2 = coef[((has2 ? t2 : t) / P4) % (unsigned)C];
            ra = make_float4(nf_coef_apply(ka, a.x), nf_coef_apply(ka, a.y), nf_coef_apply(ka, a.z), nf_coef_apply(ka, a.w));
            rb = make_float4(nf_coef_apply(kb2, b.x), nf_coef_apply(kb2, b.y), nf_coef_apply(kb2, b.z), nf_coef_apply(kb2, b.w));
        }
        y[t] = ra;
        if (has2) y[t2] = rb;
    }
    if (ld != nullptr) {
        const float d = lds[4 * C];
        for (int64_t b = gtid; b < B; b += gstride) ld[b] += d;
    }
}

template <bool ROW>
__global__ void __launch_bounds__(NF_BLOCK) k_chan_scale_bwd_v4(int op, const float4* __restrict__ gy, const float* __restrict__ p0,
                                                                const float* __restrict__ p1, const float* __restrict__ p2,
                                                                const float* __restrict__ p3, float4* __restrict__ gx, int C, int P,
                                                                unsigned total4) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    NfCoef* coef = reinterpret_cast<NfCoef*>(lds);
    for (int c = threadIdx.x; c < C; c += blockDim.x) coef[c] = nf_coef(op, 0, p0, p1, p2, p3, c);
    __syncthreads();
    const unsigned P4 = ROW ? 1u : (unsigned)P >> 2;
    for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < total4; t += gridDim.x * blockDim.x) {
        const float4 g = gy[t];
        float4 r;
        if (ROW) {
            const NfCoef k0 = coef[0], k1 = coef[1 % C], k2 = coef[2 % C], k3 = coef[3 % C];
            r = make_float4(g.x / k0.D * k0.M, g.y / k1.D * k1.M, g.z / k2.D * k2.M, g.w / k3.D * k3.M);
        } else {
            const NfCoef k = coef[(t / P4) % (unsigned)C];
            r = make_float4(g.x / k.D * k.M, g.y / k.D * k.M, g.z / k.D * k.M, g.w / k.D * k.M);
        }
        gx[t] = r;
    }
}

// IMG backward with parameter gradients: block (c, chunk) walks the float4 items q4 = b * P4 + p4 of channel c
__global__ void __launch_bounds__(NF_BLOCK) k_chan_affine_bwd_img(int op, const float4* __restrict__ gy, const float* __restrict__ gld,
                                                                  const float4* __restrict__ x, const float* __restrict__ p0,
                                                                  const float* __restrict__ p1, const float* __restrict__ p2,
                                                                  const float* __restrict__ p3, float4* __restrict__ gx,
                                                                  float* __restrict__ g_pa, float* __restrict__ g_pb, int64_t B, int C,
                                                                  int P, unsigned items_per_block) {
    __shared__ float scratch[NF_BLOCK / NF_WAVE];
    const int c = blockIdx.x;
    const NfCoef k = nf_coef(op, 0, p0, p1, p2, p3, c);
    const unsigned P4 = (unsigned)P >> 2, n4 = (unsigned)B * P4;
    const unsigned q0 = blockIdx.y * items_per_block;
    const unsigned q1 = min(q0 + items_per_block, n4);
    float r1 = 0.f, r2 = 0.f;
    for (unsigned q = q0 + threadIdx.x; q < q1; q += blockDim.x) {
        const unsigned b = q / P4;
        const unsigned a = (b * (unsigned)C + (unsigned)c) * P4 + (q - b * P4);
        const float4 g = gy[a], xv = x[a];
        gx[a] = make_float4(g.x / k.D * k.M, g.y / k.D * k.M, g.z / k.D * k.M, g.w / k.D * k.M);
        r1 += (g.x + g.y) + (g.z + g.w);
        r2 += g.x * (((xv.x - k.A) / k.D) * k.M) + g.y * (((xv.y - k.A) / k.D) * k.M) + g.z * (((xv.z - k.A) / k.D) * k.M) +
              g.w * (((xv.w - k.A) / k.D) * k.M);
    }
    float sg = 0.f;                                  // this block's share of sum_b g_ld (every channel needs the total)
    for (int64_t b = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; b < B; b += (int64_t)gridDim.y * blockDim.x) sg += gld[b];
    const float R1 = nf_block_sum(r1, scratch);
    const float R2 = nf_block_sum(r2, scratch);
    const float SG = nf_block_sum(sg, scratch);
    if (threadIdx.x == 0) {
        NF_DET_ENTER_COL(nf_ca);
        if (op == NF_ACTNORM) {
            atomicAdd(g_pa + c, -R2 - (float)P * SG);
            atomicAdd(g_pb + c, -R1 / k.D);
        } else {
            atomicAdd(g_pa + c, R2 + (float)P * SG);
            atomicAdd(g_pb + c, R1);
        }
        NF_DET_LEAVE_COL(nf_ca);
    }
}

// ROW backward with parameter gradients (P == 1, C in {1, 2, 4}): every block walks whole vectors and keeps C pairs of sums
#define NF_BIG 1024       // see coupling.hip: kernels that end in same-address atomics run 256 blocks of 16 waves
__global__ void __launch_bounds__(NF_BIG) k_chan_affine_bwd_row(int op, const float4* __restrict__ gy, const float* __restrict__ gld,
                                                                  const float4* __restrict__ x, const float* __restrict__ p0,
                                                                  const float* __restrict__ p1, const float* __restrict__ p2,
                                                                  const float* __restrict__ p3, float4* __restrict__ gx,
                                                                  float* __restrict__ g_pa, float* __restrict__ g_pb, int64_t B, int C,
                                                                  unsigned total4) {
    __shared__ float scratch[NF_BIG / NF_WAVE];
    NfCoef kr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) kr[j] = nf_coef(op, 0, p0, p1, p2, p3, j % C);
    float r1[4] = {0.f, 0.f, 0.f, 0.f}, r2[4] = {0.f, 0.f, 0.f, 0.f};
    const unsigned gstride = gridDim.x * blockDim.x, gtid = blockIdx.x * blockDim.x + threadIdx.x;
    for (unsigned t = gtid; t < total4; t += gstride) {
        const float4 g = gy[t], xv = x[t];
        const float gv[4] = {g.x, g.y, g.z, g.w}, xs[4] = {xv.x, xv.y, xv.z, xv.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[j] = gv[j] / kr[j].D * kr[j].M;
            r1[j] += gv[j];
            r2[j] += gv[j] * (((xs[j] - kr[j].A) / kr[j].D) * kr[j].M);
        }
        gx[t] = make_float4(o[0], o[1], o[2], o[3]);
    }
    float sg = 0.f;
    for (int64_t b = gtid; b < B; b += gstride) sg += gld[b];
    const float SG = nf_block_sum(sg, scratch);
    const bool det = nf_det_on(nf_ca_det);           // thread 0 adds for the workgroup: it holds the turn across the C channels
    if (det && threadIdx.x == 0) nf_det_wait(nf_ca_det);
    for (int c = 0; c < C; ++c) {                    // component j belongs to channel j % C
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j % C == c) { a1 += r1[j]; a2 += r2[j]; }
        const float R1 = nf_block_sum(a1, scratch);
        const float R2 = nf_block_sum(a2, scratch);
        if (threadIdx.x == 0) {
            const float Dc = nf_coef(op, 0, p0, p1, p2, p3, c).D;
            if (op == NF_ACTNORM) {
                atomicAdd(g_pa + c, -R2 - SG);       // P == 1
                atomicAdd(g_pb + c, -R1 / Dc);
            } else {
                atomicAdd(g_pa + c, R2 + SG);
                atomicAdd(g_pb + c, R1);
            }
        }
    }
    if (det && threadIdx.x == 0) nf_det_pass(nf_ca_det);
}

// statistics, vectorised.  IMG: block (c, chunk) over the channel's float4 items; ROW: all channels per block.
template <bool SQDEV>
__global__ void __launch_bounds__(NF_BLOCK) k_chan_stat_img(const float4* __restrict__ x, const float* __restrict__ sum,
                                                            float* __restrict__ out, int64_t B, int C, int P, unsigned items_per_block) {
    __shared__ float scratch[NF_BLOCK / NF_WAVE];
    const int c = blockIdx.x;
    const unsigned P4 = (unsigned)P >> 2, n4 = (unsigned)B * P4;
    const float mean = SQDEV ? sum[c] / ((float)B * (float)P) : 0.f;
    const unsigned q0 = blockIdx.y * items_per_block;
    const unsigned q1 = min(q0 + items_per_block, n4);
    float acc0 = 0.f, acc1 = 0.f;
    for (unsigned q = q0 + threadIdx.x; q < q1; q += 2 * blockDim.x) {
        const unsigned qb = q + blockDim.x;
        const unsigned b = q / P4, b2 = (qb < q1 ? qb : q) / P4;
        const float4 v = x[(b * (unsigned)C + (unsigned)c) * P4 + (q - b * P4)];
        const float4 w = qb < q1 ? x[(b2 * (unsigned)C + (unsigned)c) * P4 + (qb - b2 * P4)] : make_float4(mean, mean, mean, mean);
        const float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
        const float c0 = w.x - mean, c1 = w.y - mean, c2 = w.z - mean, c3 = w.w - mean;
        acc0 += SQDEV ? (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3) : (a0 + a1) + (a2 + a3);
        acc1 += SQDEV ? (c0 * c0 + c1 * c1) + (c2 * c2 + c3 * c3) : (c0 + c1) + (c2 + c3);
    }
    const float tot = nf_block_sum(acc0 + acc1, scratch);
    if (threadIdx.x == 0) { NF_DET_ENTER_COL(nf_ca); atomicAdd(out + c, tot); NF_DET_LEAVE_COL(nf_ca); }
}
template <bool SQDEV>
__global__ void __launch_bounds__(NF_BIG) k_chan_stat_row(const float4* __restrict__ x, const float* __restrict__ sum,
                                                            float* __restrict__ out, int64_t B, int C, unsigned total4) {
    __shared__ float scratch[NF_BIG / NF_WAVE];
    float mean[4], acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) mean[j] = SQDEV ? sum[j % C] / (float)B : 0.f;
    const unsigned gstride = gridDim.x * blockDim.x;
    for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < total4; t += gstride) {
        const float4 v = x[t];
        const float d0 = v.x - mean[0], d1 = v.y - mean[1], d2 = v.z - mean[2], d3 = v.w - mean[3];
        acc[0] += SQDEV ? d0 * d0 : d0; acc[1] += SQDEV ? d1 * d1 : d1; acc[2] += SQDEV ? d2 * d2 : d2; acc[3] += SQDEV ? d3 * d3 : d3;
    }
    const bool det = nf_det_on(nf_ca_det);
    if (det && threadIdx.x == 0) nf_det_wait(nf_ca_det);
    for (int c = 0; c < C; ++c) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j % C == c) a += acc[j];
        const float tot = nf_block_sum(a, scratch);
        if (threadIdx.x == 0) atomicAdd(out + c, tot);
    }
    if (det && threadIdx.x == 0) nf_det_pass(nf_ca_det);
}

static inline bool nf_al16(const void* a, const void* b = nullptr, const void* c = nullptr) {
    return (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) == 0;
}
static inline bool nf_chan_vec_img(int64_t B, int C, int P) { return P % 4 == 0 && B * C * (int64_t)P / 4 < (int64_t)1 << 31; }
static inline bool nf_chan_vec_row(int64_t B, int C, int P) {
    return P == 1 && (C == 1 || C == 2 || C == 4) && (B * C) % 4 == 0 && B * C / 4 < (int64_t)1 << 31;
}
// grid of a channel-major vector kernel: blocks (C, chunks), ~1024 blocks overall so that the closing same-address atomics stay few
static inline void nf_chan_grid4(int64_t n4, int C, dim3& grid, unsigned& ipb) {
    int64_t chunks = (n4 + 2 * NF_BLOCK - 1) / (2 * NF_BLOCK);
    const int64_t cap = (1024 + C - 1) / C;
    if (chunks > cap) chunks = cap;
    if (chunks < 1) chunks = 1;
    int64_t per = (n4 + chunks - 1) / chunks;
    per = (per + 2 * NF_BLOCK - 1) / (2 * NF_BLOCK) * (2 * NF_BLOCK);      // whole trips of the two-vector loop
    chunks = (n4 + per - 1) / per;
    ipb = (unsigned)per;
    grid = dim3((unsigned)C, (unsigned)chunks);
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
    const bool al = nf_al16(x, y);
    const bool img = al && nf_chan_vec_img(B, C, P), row = al && !img && nf_chan_vec_row(B, C, P);
    if (img || row) {
        const unsigned total4 = (unsigned)(total / 4);
        unsigned gv = nf_grid_for((total4 + 1) / 2);
        const unsigned gl = nf_grid_for(B);
        if (ld != nullptr && gv < gl) gv = gl;
        if (img) hipLaunchKernelGGL(k_chan_affine_fwd_v4<false>, dim3(gv), dim3(NF_BLOCK), (4 * C + 1) * sizeof(float), (hipStream_t)stream, op,
                                    inverse, (const float4*)x, p0, p1, p2, p3, (float4*)y, ld, B, C, P, total4);
        else hipLaunchKernelGGL(k_chan_affine_fwd_v4<true>, dim3(gv), dim3(NF_BLOCK), (4 * C + 1) * sizeof(float), (hipStream_t)stream, op,
                                inverse, (const float4*)x, p0, p1, p2, p3, (float4*)y, ld, B, C, P, total4);
        NF_CHECK_LAUNCH();
        return 0;
    }
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
    const bool al = nf_al16(g_y, x, g_x);
    const bool img = al && nf_chan_vec_img(B, C, P), row = al && !img && nf_chan_vec_row(B, C, P);
    if (img || row) {
        const unsigned total4 = (unsigned)(B * C * P / 4);
        if (g_pa == nullptr || g_pb == nullptr) {
            if (img) hipLaunchKernelGGL(k_chan_scale_bwd_v4<false>, dim3(nf_grid_for(total4)), dim3(NF_BLOCK), 4 * C * sizeof(float), st, op,
                                        (const float4*)g_y, p0, p1, p2, p3, (float4*)g_x, C, P, total4);
            else hipLaunchKernelGGL(k_chan_scale_bwd_v4<true>, dim3(nf_grid_for(total4)), dim3(NF_BLOCK), 4 * C * sizeof(float), st, op,
                                    (const float4*)g_y, p0, p1, p2, p3, (float4*)g_x, C, P, total4);
        } else if (img) {
            dim3 grid;
            unsigned ipb;
            nf_chan_grid4(B * (P / 4), C, grid, ipb);
            hipLaunchKernelGGL(k_chan_affine_bwd_img, grid, dim3(NF_BLOCK), 0, st, op, (const float4*)g_y, g_ld, (const float4*)x, p0, p1,
                               p2, p3, (float4*)g_x, g_pa, g_pb, B, C, P, ipb);
        } else {
            unsigned gv = nf_grid_for(total4, NF_BIG);
            if (gv > 256) gv = 256;
            hipLaunchKernelGGL(k_chan_affine_bwd_row, dim3(gv), dim3(NF_BIG), 0, st, op, (const float4*)g_y, g_ld, (const float4*)x, p0,
                               p1, p2, p3, (float4*)g_x, g_pa, g_pb, B, C, total4);
        }
        NF_CHECK_LAUNCH();
        return 0;
    }
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
    if (nf_al16(x) && nf_chan_vec_img(B, C, P)) {
        dim3 g4;
        unsigned ipb4;
        nf_chan_grid4(B * (P / 4), C, g4, ipb4);
        hipLaunchKernelGGL(k_chan_stat_img<false>, g4, dim3(NF_BLOCK), 0, (hipStream_t)stream, (const float4*)x, (const float*)nullptr,
                           sum, B, C, P, ipb4);
        NF_CHECK_LAUNCH();
        return 0;
    }
    if (nf_al16(x) && nf_chan_vec_row(B, C, P)) {
        const unsigned total4 = (unsigned)(B * C / 4);
        unsigned gv = nf_grid_for(total4, NF_BIG);
        if (gv > 256) gv = 256;
        hipLaunchKernelGGL(k_chan_stat_row<false>, dim3(gv), dim3(NF_BIG), 0, (hipStream_t)stream, (const float4*)x,
                           (const float*)nullptr, sum, B, C, total4);
        NF_CHECK_LAUNCH();
        return 0;
    }
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
    if (nf_al16(x) && nf_chan_vec_img(B, C, P)) {
        dim3 g4;
        unsigned ipb4;
        nf_chan_grid4(B * (P / 4), C, g4, ipb4);
        hipLaunchKernelGGL(k_chan_stat_img<true>, g4, dim3(NF_BLOCK), 0, (hipStream_t)stream, (const float4*)x, sum, sqdev, B, C, P, ipb4);
        NF_CHECK_LAUNCH();
        return 0;
    }
    if (nf_al16(x) && nf_chan_vec_row(B, C, P)) {
        const unsigned total4 = (unsigned)(B * C / 4);
        unsigned gv = nf_grid_for(total4, NF_BIG);
        if (gv > 256) gv = 256;
        hipLaunchKernelGGL(k_chan_stat_row<true>, dim3(gv), dim3(NF_BIG), 0, (hipStream_t)stream, (const float4*)x, sum, sqdev, B, C,
                           total4);
        NF_CHECK_LAUNCH();
        return 0;
    }
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
