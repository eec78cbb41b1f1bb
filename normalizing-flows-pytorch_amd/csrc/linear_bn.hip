// Fused (masked | weight-normed) linear + BatchNorm1d + ReLU chain on fp32 MFMA -- the conditioner building block.
//   MADE            : flows/maf.py:49-64     F.linear(z, W*M, b) -> BatchNorm1d -> relu      (mask baked into the tile loader)
//   MLP conditioner : flows/modules.py:342-413, flows/weight_norm.py:35-41   BN -> ReLU -> WeightNorm(Linear) (+ residual)
// One launch = one linear layer for up to NF_MAX_NETS independent nets.  "Normalise on load, statistics on store":
// the BatchNorm that precedes a linear is folded into its operand load, the batch statistics the NEXT BatchNorm needs
// are accumulated in the epilogue, so a BN-MLP costs one launch per linear instead of ~10 framework kernels.
//
// GEMM shape: rows (samples) x 32 outputs x <=32 inputs, fp32 exact -> v_mfma_f32_32x32x2_f32, one 32x32 tile per wave
// per 16 issues (157 TF peak class, 1e-5 parity rules out bf16).  K is walked in the permuted order
// k = half*KH + kk (half = lane>>5) so that a lane's A fragment is CONTIGUOUS in memory (one 64-byte run per row-half).
// At the sizes of the reference models these launches are latency-bound (a 4096 x 32 x 32 layer is 8 MFLOP); the win
// is the launch count and the removed HBM round trips of the intermediate tensors.
#include "nf_common.h"
#include "nf_det.h"

NF_DET_STATE(nf_lb)
NF_DET_HOST_API(nf_lb)

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct NfLinArgs { nf_linear_desc d[NF_MAX_NETS]; };
struct NfLinBwdArgs { nf_linear_bwd_desc d[NF_MAX_NETS]; };
#define NF_MAX_WG_LAYERS 16
struct NfWGradArgs { nf_weight_grad_desc d[NF_MAX_WG_LAYERS]; };

#define NF_LB_WAVES 4
#define NF_TS 33  // padded row stride (words) of the 32x32 LDS tiles: column walks are conflict-free

__device__ __forceinline__ float nf_half32_sum(float v) {  // sum over the 32 lanes of this wave half
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, NF_WAVE);
    return v;
}

// row index of accumulator register r in the 32x32 C/D layout (col = lane & 31)
__device__ __forceinline__ int nf_cd_row(int r, int hs) { return (r & 3) + 8 * (r >> 2) + 4 * hs; }

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
// Prologue (once per block, ONE round of global loads, everything else through LDS):
//   W tile (mask applied) -> LDS;  thread k < I: column norm of W (weight-norm) and the folded BatchNorm constants of
//   input feature k;  the weight-norm column scale s_k = g_k / (||v_k|| + eps) is applied on the A side
//   (out = sum_k (act_k s_k) v[o][k]), which keeps the B fragment a plain copy of the stored weight.
template <int KH>
__global__ void __launch_bounds__(NF_LB_WAVES * NF_WAVE) k_linear_bn_fwd(NfLinArgs args, int64_t N, int I, int O,
                                                                         int training, float eps, float mom,
                                                                         float wn_eps, int64_t tiles) {
    __shared__ float Wl[32 * NF_TS];
    __shared__ float kc[3][32];                       // per input feature: BN scale, BN shift, weight-norm scale
    __shared__ float red[2][NF_LB_WAVES][32];
    const nf_linear_desc& d = args.d[blockIdx.y];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, o = lane & 31, hs = lane >> 5;
    const bool has_bn = d.bn_gamma != nullptr;
    const float invN = 1.f / (float)N;

    for (int e = threadIdx.x; e < 32 * 32; e += blockDim.x) {
        const int oo = e >> 5, k = e & 31;
        float w = 0.f;
        if (oo < O && k < I) {
            w = d.weight[oo * I + k];
            if (d.mask != nullptr) w *= d.mask[oo * I + k];                        // maf.py:54
        }
        Wl[oo * NF_TS + k] = w;
    }
    float sc_k = has_bn ? 0.f : 1.f, sh_k = 0.f;
    if (threadIdx.x < 32 && has_bn && (int)threadIdx.x < I) {
        const int k = threadIdx.x;
        float mean, invstd;
        if (training) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int r = 0; r < NF_STAT_REPL; ++r) { t1 += d.bn_sum[32 * r + k]; t2 += d.bn_sqsum[32 * r + k]; }
            const float m1 = t1 * invN;
            mean = d.bn_center[k] + m1;
            const float var = fmaxf(t2 * invN - m1 * m1, 0.f);                     // biased, as BatchNorm normalises
            invstd = 1.f / sqrtf(var + eps);
            if (blockIdx.x == 0) {                                                 // bookkeeping, once per feature
                const float rm = d.bn_running_mean[k], rv = d.bn_running_var[k];
                const float unb = N > 1 ? var * ((float)N / (float)(N - 1)) : var;
                d.bn_running_mean[k] = (1.f - mom) * rm + mom * mean;
                d.bn_running_var[k] = (1.f - mom) * rv + mom * unb;
            }
        } else {
            mean = d.bn_running_mean[k];
            invstd = 1.f / sqrtf(d.bn_running_var[k] + eps);
        }
        if (blockIdx.x == 0 && d.bn_save_mean != nullptr) {                        // constants of the backward pass
            d.bn_save_mean[k] = mean;
            d.bn_save_invstd[k] = invstd;
        }
        sc_k = d.bn_gamma[k] * invstd;
        sh_k = d.bn_beta[k] - mean * sc_k;
    }
    if (training && has_bn && d.bn_num_batches != nullptr && blockIdx.x == 0 && threadIdx.x == 0)
        d.bn_num_batches[0] += 1;
    const float wg_k = (d.weight_g != nullptr && threadIdx.x < 32 && (int)threadIdx.x < I) ? d.weight_g[threadIdx.x] : 0.f;
    __syncthreads();
    if (threadIdx.x < 32) {
        const int k = threadIdx.x;
        float ws = 1.f;
        if (d.weight_g != nullptr) {                                               // weight_norm.py:40
            float ss = 0.f;
#pragma unroll 8
            for (int oo = 0; oo < 32; ++oo) ss = fmaf(Wl[oo * NF_TS + k], Wl[oo * NF_TS + k], ss);
            ws = (k < I) ? wg_k / (sqrtf(ss) + wn_eps) : 0.f;
        }
        kc[0][k] = sc_k; kc[1][k] = sh_k; kc[2][k] = ws;
    }
    __syncthreads();
    float b[KH], sc[KH], sh[KH], wsc[KH];
#pragma unroll
    for (int kk = 0; kk < KH; ++kk) {
        const int k = hs * KH + kk;                   // KH <= 16 -> k < 32
        b[kk] = Wl[o * NF_TS + k];
        sc[kk] = kc[0][k]; sh[kk] = kc[1][k]; wsc[kk] = kc[2][k];
    }

    const float bias_o = (o < O) ? d.bias[o] : 0.f;
    const bool want_stats = d.stat_sum != nullptr;
    const bool has_res = d.residual != nullptr;
    float s1 = 0.f, s2 = 0.f;
    for (int64_t tile = (int64_t)blockIdx.x * NF_LB_WAVES + wid; tile < tiles; tile += (int64_t)gridDim.x * NF_LB_WAVES) {
        const int64_t row0 = tile * 32;
        const int64_t row = row0 + o;                     // A fragment row of this lane (o doubles as row-in-tile)
        const bool rv = row < N;
        float a[KH];
        if (KH == 16 && I == 32) {
            const float4* p = reinterpret_cast<const float4*>(d.in + (rv ? row : 0) * 32 + hs * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = p[q];
                a[4 * q + 0] = rv ? v.x : 0.f; a[4 * q + 1] = rv ? v.y : 0.f;
                a[4 * q + 2] = rv ? v.z : 0.f; a[4 * q + 3] = rv ? v.w : 0.f;
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < KH; ++kk) {
                const int k = hs * KH + kk;
                a[kk] = (rv && k < I) ? d.in[row * I + k] : 0.f;
            }
        }
        float rres[16];                                   // residual tile prefetched: 16 loads in flight under the MFMAs
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t gr = row0 + nf_cd_row(r, hs);
            rres[r] = (has_res && o < O && gr < N) ? d.residual[gr * O + o] : 0.f;
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < KH; ++kk) {
            const float av = (has_bn ? fmaxf(fmaf(a[kk], sc[kk], sh[kk]), 0.f) : a[kk]) * wsc[kk];   // BN -> ReLU -> WN scale
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[kk], acc, 0, 0, 0);
        }
        if (o < O) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t gr = row0 + nf_cd_row(r, hs);
                if (gr < N) {
                    const float dv = acc[r] + rres[r];
                    d.out[gr * O + o] = dv + bias_o;
                    s1 += dv;                                 // statistics centred at the bias (shifted sums)
                    s2 = fmaf(dv, dv, s2);
                }
            }
        }
    }
    if (want_stats) {                                         // block-uniform branch
        s1 += __shfl_xor(s1, 32, NF_WAVE);
        s2 += __shfl_xor(s2, 32, NF_WAVE);
        if (hs == 0) { red[0][wid][o] = s1; red[1][wid][o] = s2; }
        __syncthreads();
        if (wid == 0) {
            NF_DET_ENTER_WAVE(nf_lb);                          // (wave 0 adds for the workgroup, one lane per feature)
            if (hs == 0 && o < O) {
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int w = 0; w < NF_LB_WAVES; ++w) { t1 += red[0][w][o]; t2 += red[1][w][o]; }
                const int rep = 32 * (blockIdx.x % NF_STAT_REPL);
                atomicAdd(d.stat_sum + rep + o, t1);
                atomicAdd(d.stat_sqsum + rep + o, t2);
            }
            NF_DET_LEAVE_WAVE(nf_lb);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward (training mode)
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(NF_LB_WAVES * NF_WAVE) k_linear_bn_bwd(NfLinBwdArgs args, int64_t N, int I, int O,
                                                                         float wn_eps, int64_t tiles, int iters) {
    // per wave: G tile and raw-input tile (32 x 32, padded); afterwards reused for the cross-wave reduction of g_weff
    __shared__ float lds[NF_LB_WAVES * 2 * 32 * NF_TS];
    __shared__ float cbn[5][32];     // consumer BatchNorm constants per output feature
    __shared__ float red[3][NF_LB_WAVES][32];
    const nf_linear_bwd_desc& d = args.d[blockIdx.y];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, c32 = lane & 31, hs = lane >> 5;
    float* Gt = lds + wid * 2 * 32 * NF_TS;
    float* It = Gt + 32 * NF_TS;
    const bool has_bn = d.bn_gamma != nullptr;
    const bool has_src = d.gn_src != nullptr;
    const float invN = 1.f / (float)N;

    if (threadIdx.x < 32) {
        const int oo = threadIdx.x;
        float c1 = 0.f, mean = 0.f, invstd = 0.f, mg = 0.f, mgx = 0.f;
        if (has_src && oo < O) {
            invstd = d.cbn_save_invstd[oo];
            mean = d.cbn_save_mean[oo];
            c1 = d.cbn_gamma[oo] * invstd;
            if (d.cbn_sum_g != nullptr) {
#pragma unroll
                for (int r = 0; r < NF_STAT_REPL; ++r) { mg += d.cbn_sum_g[32 * r + oo]; mgx += d.cbn_sum_gx[32 * r + oo]; }
                mg *= invN;
                mgx *= invN;
            }
        }
        cbn[0][oo] = c1; cbn[1][oo] = mean; cbn[2][oo] = invstd; cbn[3][oo] = mg; cbn[4][oo] = mgx;
    }
    // B fragment of the data-gradient GEMM: Weff[o = hs*16+kk][i = c32]
    float bW[16];
    {
        float ss = 0.f;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const int oo = hs * 16 + kk;
            const bool ok = (oo < O) && (c32 < I);
            float w = ok ? d.weight[oo * I + c32] : 0.f;
            if (ok && d.mask != nullptr) w *= d.mask[oo * I + c32];
            bW[kk] = w;
            ss = fmaf(w, w, ss);
        }
        if (d.weight_g != nullptr) {
            ss += __shfl_xor(ss, 32, NF_WAVE);                    // full column norm over o
            const float s_i = (c32 < I) ? d.weight_g[c32] / (sqrtf(ss) + wn_eps) : 0.f;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) bW[kk] *= s_i;
        }
    }
    // input-side BatchNorm constants of column i = c32
    float mean_i = 0.f, invstd_i = 0.f, sc_i = 1.f, sh_i = 0.f;
    if (has_bn && c32 < I) {
        mean_i = d.bn_save_mean[c32];
        invstd_i = d.bn_save_invstd[c32];
        sc_i = d.bn_gamma[c32] * invstd_i;
        sh_i = d.bn_beta[c32] - mean_i * sc_i;
    }
    __syncthreads();

    f32x16 accW;
#pragma unroll
    for (int r = 0; r < 16; ++r) accW[r] = 0.f;
    float gb = 0.f, sg = 0.f, sgx = 0.f;

    for (int it = 0; it < iters; ++it) {
        const int64_t tile = ((int64_t)it * gridDim.x + blockIdx.x) * NF_LB_WAVES + wid;
        const bool active = tile < tiles;                         // wave-uniform
        const int64_t row0 = tile * 32;
        float a1[16];
        if (active) {
            // ---- assemble G[row][o] for o = hs*16 + kk (row = c32), stash it in LDS, keep it as the A fragment ----
            const int64_t row = row0 + c32;
            const bool rv = row < N;
            if (O == 32) {                                         // 64-byte runs per lane: four 16-byte loads per tensor
                const int64_t base = (rv ? row : 0) * 32 + hs * 16;    // invalid rows read row 0 and are zeroed below
                float g[16];
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) g[kk] = 0.f;
                if (d.g_direct != nullptr) {
                    const float4* p4 = reinterpret_cast<const float4*>(d.g_direct + base);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = p4[q];
                        g[4 * q] += v.x; g[4 * q + 1] += v.y; g[4 * q + 2] += v.z; g[4 * q + 3] += v.w;
                    }
                }
                if (d.g_skip != nullptr) {
                    const float4* p4 = reinterpret_cast<const float4*>(d.g_skip + base);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = p4[q];
                        g[4 * q] += v.x; g[4 * q + 1] += v.y; g[4 * q + 2] += v.z; g[4 * q + 3] += v.w;
                    }
                }
                if (has_src) {
                    const float4* s4 = reinterpret_cast<const float4*>(d.gn_src + base);
                    const float4* o4 = reinterpret_cast<const float4*>(d.out + base);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 sv = s4[q], ov = o4[q];
                        const float se[4] = {sv.x, sv.y, sv.z, sv.w}, oe[4] = {ov.x, ov.y, ov.z, ov.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int oo = hs * 16 + 4 * q + j;
                            const float xh = (oe[j] - cbn[1][oo]) * cbn[2][oo];
                            g[4 * q + j] += cbn[0][oo] * (se[j] - cbn[3][oo] - xh * cbn[4][oo]);   // BatchNorm backward on load
                        }
                    }
                }
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    a1[kk] = rv ? g[kk] : 0.f;
                    Gt[c32 * NF_TS + hs * 16 + kk] = a1[kk];
                }
                if (rv && d.g_store != nullptr) {
                    float4* p4 = reinterpret_cast<float4*>(d.g_store + base);
#pragma unroll
                    for (int q = 0; q < 4; ++q) p4[q] = make_float4(a1[4 * q], a1[4 * q + 1], a1[4 * q + 2], a1[4 * q + 3]);
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    const int oo = hs * 16 + kk;
                    float g = 0.f;
                    if (rv && oo < O) {
                        const int64_t idx = row * O + oo;
                        if (d.g_direct != nullptr) g += d.g_direct[idx];
                        if (d.g_skip != nullptr) g += d.g_skip[idx];
                        if (has_src) {
                            const float xh = (d.out[idx] - cbn[1][oo]) * cbn[2][oo];
                            g += cbn[0][oo] * (d.gn_src[idx] - cbn[3][oo] - xh * cbn[4][oo]);
                        }
                        if (d.g_store != nullptr) d.g_store[idx] = g;
                    }
                    a1[kk] = g;
                    Gt[c32 * NF_TS + oo] = g;
                }
            }
            // ---- raw input tile, coalesced ----
            if (I == 32) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int idx4 = lane + 64 * q;                // float4 index inside the 32 x 32 tile
                    const int rr = idx4 >> 3, cc = (idx4 & 7) * 4;
                    const bool ok = row0 + rr < N;
                    const float4 v = reinterpret_cast<const float4*>(d.in + (ok ? row0 + rr : 0) * 32)[idx4 & 7];
                    float* dst = It + rr * NF_TS + cc;
                    dst[0] = ok ? v.x : 0.f; dst[1] = ok ? v.y : 0.f; dst[2] = ok ? v.z : 0.f; dst[3] = ok ? v.w : 0.f;
                }
            } else {
                for (int idx = lane; idx < 32 * I; idx += NF_WAVE) {
                    const int rr = idx / I, cc = idx - rr * I;
                    It[rr * NF_TS + cc] = (row0 + rr < N) ? d.in[(row0 + rr) * I + cc] : 0.f;
                }
            }
        }
        __syncthreads();
        if (active) {
            // ---- data gradient: (G Weff)[row][i] -> ReLU mask -> pre-BatchNorm gradient + its two batch sums ----
            f32x16 acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[kk], bW[kk], acc1, 0, 0, 0);
            if (d.gn_out != nullptr && c32 < I) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = nf_cd_row(r, hs);
                    const int64_t gr = row0 + rr;
                    if (gr < N) {
                        float gn = acc1[r];
                        if (has_bn) {
                            const float xin = It[rr * NF_TS + c32];
                            gn = (fmaf(xin, sc_i, sh_i) > 0.f) ? gn : 0.f;
                            sg += gn;
                            sgx = fmaf(gn, (xin - mean_i) * invstd_i, sgx);
                        }
                        d.gn_out[gr * I + c32] = gn;
                    }
                }
            }
            // ---- weight gradient: g_weff[o][i] += sum_n G[n][o] act[n][i]  (A = G^T, B = act, K = the tile's 32 rows) ----
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int n = hs * 16 + kk;
                const float ga = Gt[n * NF_TS + c32];             // G[n][o = c32]
                const float xin = It[n * NF_TS + c32];            // in[n][i = c32]
                const float act = has_bn ? fmaxf(fmaf(xin, sc_i, sh_i), 0.f) : xin;
                gb += ga;
                accW = __builtin_amdgcn_mfma_f32_32x32x2f32(ga, act, accW, 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // ---- block reductions, then one atomic per entry per block ----
    gb += __shfl_xor(gb, 32, NF_WAVE);
    sg += __shfl_xor(sg, 32, NF_WAVE);
    sgx += __shfl_xor(sgx, 32, NF_WAVE);
    if (hs == 0) { red[0][wid][c32] = gb; red[1][wid][c32] = sg; red[2][wid][c32] = sgx; }
    float* Wred = lds;                                            // [wave][32][32]
#pragma unroll
    for (int r = 0; r < 16; ++r) Wred[(wid * 32 + nf_cd_row(r, hs)) * 32 + c32] = accW[r];
    __syncthreads();
    for (int e = threadIdx.x; e < 32 * 32; e += blockDim.x) {
        const int oo = e >> 5, ii = e & 31;
        if (oo < O && ii < I) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NF_LB_WAVES; ++w) t += Wred[(w * 32 + oo) * 32 + ii];
            d.g_weff[((int64_t)blockIdx.x * O + oo) * I + ii] = t;       // this workgroup's slab (no atomics, no zero-fill)
        }
    }
    if (wid == 0) {
        NF_DET_ENTER_WAVE(nf_lb);                              // (wave 0 adds for the workgroup, one lane per feature)
        if (hs == 0) {
            float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < NF_LB_WAVES; ++w) { t0 += red[0][w][c32]; t1 += red[1][w][c32]; t2 += red[2][w][c32]; }
            const int rep = 32 * (blockIdx.x % NF_STAT_REPL);
            if (c32 < O && d.g_bias != nullptr) atomicAdd(d.g_bias + rep + c32, t0);
            if (has_bn && c32 < I && d.sum_g != nullptr) {
                atomicAdd(d.sum_g + rep + c32, t1);
                atomicAdd(d.sum_gx + rep + c32, t2);
            }
        }
        NF_DET_LEAVE_WAVE(nf_lb);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// g_weff slabs -> gradients of the stored parameters (mask / weight-norm) + plain vector gradients, one block per job
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(NF_BLOCK) k_weight_grad_finalize(NfWGradArgs args, float wn_eps) {
    __shared__ float gW[32 * 32];
    __shared__ float nrm2[32], dot[32];
    const nf_weight_grad_desc& d = args.d[blockIdx.x];
    const int I = d.I, O = d.O;
    const bool acc = d.accumulate != 0;
    if (d.g_weff != nullptr) {
        {   // deterministic slab reduction; <= 4 entries per thread, 4 slabs per trip -> 16 loads in flight
            const int OI = O * I;
            float t[4] = {0.f, 0.f, 0.f, 0.f};
            int sl = 0;
            for (; sl + 4 <= d.n_slabs; sl += 4) {
                float v[4][4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int e = threadIdx.x + q * NF_BLOCK;
                        v[u][q] = e < OI ? d.g_weff[(int64_t)(sl + u) * OI + e] : 0.f;
                    }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) t[q] += v[u][q];
            }
            for (; sl < d.n_slabs; ++sl)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = threadIdx.x + q * NF_BLOCK;
                    if (e < OI) t[q] += d.g_weff[(int64_t)sl * OI + e];
                }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = threadIdx.x + q * NF_BLOCK;
                if (e < OI) gW[e] = t[q];
            }
        }
        __syncthreads();
        if (d.weight_g != nullptr) {
            if (threadIdx.x < 32) {
                const int i = threadIdx.x;
                float a = 0.f, b = 0.f;
                if (i < I)
                    for (int o = 0; o < O; ++o) {
                        const float v = d.weight[o * I + i];
                        a = fmaf(v, v, a);
                        b = fmaf(gW[o * I + i], v, b);
                    }
                nrm2[i] = a;
                dot[i] = b;
            }
            __syncthreads();
            for (int e = threadIdx.x; e < O * I; e += blockDim.x) {
                const int i = e % I;
                const float nrm = sqrtf(nrm2[i]), den = nrm + wn_eps, g = d.weight_g[i];
                float gv = gW[e] * (g / den);
                if (nrm > 0.f) gv -= d.weight[e] * (dot[i] * g / (den * den * nrm));
                d.g_weight[e] = (acc ? d.g_weight[e] : 0.f) + gv;
            }
            if (d.g_weight_g != nullptr && (int)threadIdx.x < I) {
                const int i = threadIdx.x;
                d.g_weight_g[i] = (acc ? d.g_weight_g[i] : 0.f) + dot[i] / (sqrtf(nrm2[i]) + wn_eps);
            }
        } else {
            for (int e = threadIdx.x; e < O * I; e += blockDim.x)
                d.g_weight[e] = (acc ? d.g_weight[e] : 0.f) + (d.mask != nullptr ? gW[e] * d.mask[e] : gW[e]);
        }
    }
    const int repl = d.vec_repl > 1 ? d.vec_repl : 1;
    if (d.vec_dst0 != nullptr)
        for (int e = threadIdx.x; e < d.vec_n0; e += blockDim.x) {
            float t = 0.f;
            for (int r = 0; r < repl; ++r) t += d.vec_src0[32 * r + e];
            d.vec_dst0[e] = (acc ? d.vec_dst0[e] : 0.f) + t;
        }
    if (d.vec_dst1 != nullptr)
        for (int e = threadIdx.x; e < d.vec_n1; e += blockDim.x) {
            float t = 0.f;
            for (int r = 0; r < repl; ++r) t += d.vec_src1[32 * r + e];
            d.vec_dst1[e] = (acc ? d.vec_dst1[e] : 0.f) + t;
        }
}

// ---------------------------------------------------------------------------------------------------------------
static inline unsigned nf_lb_grid(int64_t tiles, int64_t cap = 1024) {
    int64_t g = (tiles + NF_LB_WAVES - 1) / NF_LB_WAVES;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

#define NF_LB_BWD_MAX_SLABS 256
extern "C" int nf_linear_bwd_slabs(int64_t N) {
    if (N <= 0) return 0;
    return (int)nf_lb_grid((N + 31) / 32, NF_LB_BWD_MAX_SLABS);
}

extern "C" int nf_linear_bn_fwd(const nf_linear_desc* descs, int n_nets, int64_t N, int I, int O, int training,
                                float bn_eps, float bn_momentum, float wn_eps, nf_stream_t stream) {
    if (descs == nullptr || n_nets < 1 || n_nets > NF_MAX_NETS || I < 1 || O < 1 || I > 32 || O > 32) return NF_E_BADARG;
    if (N <= 0) return N == 0 ? 0 : NF_E_BADARG;
    NfLinArgs args;
    for (int i = 0; i < n_nets; ++i) args.d[i] = descs[i];
    const int64_t tiles = (N + 31) / 32;
    const dim3 grid(nf_lb_grid(tiles), (unsigned)n_nets), block(NF_LB_WAVES * NF_WAVE);
    hipStream_t st = (hipStream_t)stream;
    const int kh = (I + 1) / 2;
#define NF_LAUNCH(KH) hipLaunchKernelGGL(k_linear_bn_fwd<KH>, grid, block, 0, st, args, N, I, O, training, bn_eps, bn_momentum, wn_eps, tiles)
    if (kh <= 1) NF_LAUNCH(1);
    else if (kh <= 2) NF_LAUNCH(2);
    else if (kh <= 4) NF_LAUNCH(4);
    else if (kh <= 8) NF_LAUNCH(8);
    else NF_LAUNCH(16);
#undef NF_LAUNCH
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_linear_bn_bwd(const nf_linear_bwd_desc* descs, int n_nets, int64_t N, int I, int O, float wn_eps,
                                nf_stream_t stream) {
    if (descs == nullptr || n_nets < 1 || n_nets > NF_MAX_NETS || I < 1 || O < 1 || I > 32 || O > 32) return NF_E_BADARG;
    if (N <= 0) return N == 0 ? 0 : NF_E_BADARG;
    NfLinBwdArgs args;
    for (int i = 0; i < n_nets; ++i) args.d[i] = descs[i];
    const int64_t tiles = (N + 31) / 32;
    const unsigned gx = nf_lb_grid(tiles, NF_LB_BWD_MAX_SLABS);
    const int iters = (int)((tiles + (int64_t)gx * NF_LB_WAVES - 1) / ((int64_t)gx * NF_LB_WAVES));
    hipLaunchKernelGGL(k_linear_bn_bwd, dim3(gx, (unsigned)n_nets), dim3(NF_LB_WAVES * NF_WAVE), 0, (hipStream_t)stream,
                       args, N, I, O, wn_eps, tiles, iters);
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_weight_grad_finalize(const nf_weight_grad_desc* descs, int n_layers, float wn_eps,
                                       nf_stream_t stream) {
    if (descs == nullptr || n_layers < 1 || n_layers > NF_MAX_WG_LAYERS) return NF_E_BADARG;
    NfWGradArgs args;
    for (int i = 0; i < n_layers; ++i) {
        if (descs[i].g_weff != nullptr && (descs[i].I < 1 || descs[i].O < 1 || descs[i].I > 32 || descs[i].O > 32))
            return NF_E_BADARG;
        args.d[i] = descs[i];
    }
    hipLaunchKernelGGL(k_weight_grad_finalize, dim3((unsigned)n_layers), dim3(NF_BLOCK), 0, (hipStream_t)stream, args,
                       wn_eps);
    NF_CHECK_LAUNCH();
    return 0;
}
