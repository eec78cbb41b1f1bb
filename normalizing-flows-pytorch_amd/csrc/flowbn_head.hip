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

// ---- statistics AND apply in ONE persistent launch (round 6) ---------------------------------------------------------------------------
// The two launches above are ~7 + ~5 us for a tensor of a few hundred KB (196 608 elements at the CIFAR levels): launch latency twice,
// the tensor read twice.  Here a thread keeps four consecutive elements of one channel plane in registers (a workgroup = 1 024
// elements = whole planes, or a whole number of workgroups per plane), every (sample, channel) plane's shifted sums go through ONE
// {generation : value} slot pair (zeroed by the caller), every workgroup gathers the B C pairs and adds each channel's B rows in
// sample order -- the same bits in every workgroup and in every run, no float atomics: nothing for the ordered mode to do -- and
// normalises what it holds.  Grid <= the compute units (co-resident by construction: 256 threads, 25 KB of LDS); a workgroup that
// never arrives ends the wait after the spin limit, counted in nf_persistent_timeouts like every bounded wait of the library.
// Workgroup 0 updates running_mean -- the centre every workgroup has read by the time its planes are published.
NF_PERSIST_STATE(nf_fbh)
NF_PERSIST_HOST_API(nf_fbh)
#define NF_FBF_EPT 4                        // elements per thread
#define NF_FBF_WGE (NF_BLOCK * NF_FBF_EPT)  // elements per workgroup
#define NF_FBF_MAX_VALUES 6144              // 2 B C values gathered in LDS
__global__ void __launch_bounds__(NF_BLOCK) k_flowbn_head_fused(const float* __restrict__ x, const float* __restrict__ log_gamma,
                                                                const float* __restrict__ beta, float* __restrict__ bmean,
                                                                float* __restrict__ bvar, float* __restrict__ rmean,
                                                                float* __restrict__ rvar, float eps, float mom, float* __restrict__ y,
                                                                float* __restrict__ z1c, float* __restrict__ ld,
                                                                unsigned long long* __restrict__ slots, NfSplit s, int B, int P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int C = s.C, G = gridDim.x, tid = threadIdx.x, lane = tid & (NF_WAVE - 1);
    float* val = lds;                                   // [B][C][2] gathered plane sums; before that: [waves][2] of this workgroup
    NfBnCoef* coef = reinterpret_cast<NfBnCoef*>(lds + NF_FBF_MAX_VALUES);
    float* ldt = lds + NF_FBF_MAX_VALUES + 4 * C;       // [C] log-det terms
    float* cen = ldt + C;                               // [C] the centres (running_mean as it was when the launch began)
    if (tid < C) cen[tid] = rmean[tid];
    const int total = B * C * P;                        // (< 2^31: checked by the host)
    const int e0 = (blockIdx.x * NF_BLOCK + tid) * NF_FBF_EPT;
    const bool valid = e0 < total;
    const int lgP = 31 - __clz(P);
    const int plane = valid ? e0 >> lgP : 0;            // = b C + c
    const int bb = plane / C, c = plane - bb * C;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) v = *reinterpret_cast<const float4*>(x + e0);
    const float k = rmean[c];
    float s1 = 0.f, s2 = 0.f;
    if (valid) {
        const float d0 = v.x - k, d1 = v.y - k, d2 = v.z - k, d3 = v.w - k;
        s1 = (d0 + d1) + (d2 + d3);
        s2 = fmaf(d0, d0, fmaf(d1, d1, fmaf(d2, d2, d3 * d3)));
    }
    // the threads of one plane are L = P / 4 consecutive ones (a power of two): a butterfly over min(L, 64) lanes; a plane of 512 or
    // 1 024 elements spans two or four waves of this workgroup (or, beyond 1 024, several workgroups: their partial sums share the slot
    // pair -- not taken, see nf_fbf_grid)
    const int L = P / NF_FBF_EPT, Lw = L < NF_WAVE ? L : NF_WAVE;
    for (int off = 1; off < Lw; off <<= 1) { s1 += __shfl_xor(s1, off, NF_WAVE); s2 += __shfl_xor(s2, off, NF_WAVE); }
    if (L > NF_WAVE) {                                  // (block-uniform) wave sums -> LDS, the first thread of the plane adds them in wave order
        if (lane == 0) { val[2 * (tid >> 6)] = s1; val[2 * (tid >> 6) + 1] = s2; }
        __syncthreads();
        if ((tid & (L - 1)) == 0) {
            s1 = 0.f; s2 = 0.f;
            for (int w = 0; w < L / NF_WAVE; ++w) { s1 += val[2 * ((tid >> 6) + w)]; s2 += val[2 * ((tid >> 6) + w) + 1]; }
        }
    }
    if (valid && (tid & (L - 1)) == 0) {
        unsigned long long* dst = slots + (size_t)plane * 2;
        __hip_atomic_store(dst, (1ull << 32) | (unsigned long long)__float_as_uint(s1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + 1, (1ull << 32) | (unsigned long long)__float_as_uint(s2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();                                    // (val was read above, is rewritten below; cen is written)
    const int nval = 2 * B * C;
    for (int e0v = tid; e0v < nval; e0v += 4 * NF_BLOCK) {      // four polls in flight per trip
        unsigned long long w[4];
        unsigned spins = 0;
        bool ok;
        do {
            ok = true;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = e0v + q * NF_BLOCK;
                w[q] = __hip_atomic_load(slots + (e < nval ? e : e0v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) ok = ok && (unsigned)(w[q] >> 32) == 1u;
            if (ok) break;
            if (++spins > nf_fbh_spin_limit) { NF_PERSIST_GIVE_UP(nf_fbh); break; }
            __builtin_amdgcn_s_sleep(1);
        } while (true);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = e0v + q * NF_BLOCK;
            if (e < nval) val[e] = __uint_as_float((unsigned)w[q]);
        }
    }
    __syncthreads();
    const float n = (float)B * (float)P;
    if (tid < C) {
        float S1 = 0.f, S2 = 0.f;
        for (int b = 0; b < B; ++b) { S1 += val[(b * C + tid) * 2]; S2 += val[(b * C + tid) * 2 + 1]; }
        const float kc = cen[tid];
        const float m1 = S1 / n;
        const float mean = kc + m1;
        const float var = fmaxf(S2 / n - m1 * m1, 0.f) + eps;                    // biased, eps inside (modules.py:287)
        NfBnCoef kf;
        kf.mean = mean; kf.sd = sqrtf(var); kf.eg = expf(log_gamma[tid]); kf.beta = beta[tid];
        coef[tid] = kf;
        ldt[tid] = log_gamma[tid] - 0.5f * logf(var);
        if (blockIdx.x == 0) {
            bmean[tid] = mean;
            bvar[tid] = var;
            rmean[tid] = kc * (1.f - mom) + mean * mom;                          // modules.py:291-294
            rvar[tid] = rvar[tid] * (1.f - mom) + var * mom;
        }
    }
    __syncthreads();
    float dld = 0.f;
    for (int cc = 0; cc < C; ++cc) dld += ldt[cc];
    dld *= (float)P;                                                             // modules.py:303-305
    if (valid) {
        const NfBnCoef kf = coef[c];
        float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = ((o[j] - kf.mean) / kf.sd) * kf.eg + kf.beta;      // modules.py:300-301
        *reinterpret_cast<float4*>(y + e0) = make_float4(o[0], o[1], o[2], o[3]);
        if (z1c != nullptr) {
            const int p0 = e0 & (P - 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int which, e;
                nf_full_to_half(s, c, p0 + j, which, e);
                if (which == 1) z1c[(int64_t)bb * s.n_half + e] = o[j];
            }
        }
    }
    for (int b = blockIdx.x * NF_BLOCK + tid; b < B; b += G * NF_BLOCK) ld[b] += dld;
}

static int nf_fbf_grid(int64_t B, int C, int P) {       // 0: the fused launch does not take this shape
    if (B < 1 || C < 1 || C > 64 || P < 16 || P > NF_FBF_WGE || (P & (P - 1)) != 0) return 0;
    const int64_t total = B * C * P;
    if (total >= ((int64_t)1 << 30) || 2 * B * C > NF_FBF_MAX_VALUES) return 0;
    const int64_t G = (total + NF_FBF_WGE - 1) / NF_FBF_WGE;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (G > cus) return 0;
    return (int)G;
}
// floats of zeroed slot memory the fused launch needs (0: not usable for this shape -> nf_flowbn_stats + nf_flowbn_head_fwd)
extern "C" int nf_flowbn_head_fused_ws_floats(int64_t B, int C, int H, int W) {
    return nf_fbf_grid(B, C, H * W) > 0 ? (int)(4 * B * C) : 0;
}
extern "C" int nf_flowbn_head_fused(const float* x, const float* log_gamma, const float* beta, float* batch_mean, float* batch_var,
                                    float* running_mean, float* running_var, float eps, float momentum, float* y, float* z1c, float* ld,
                                    float* ws_zero, int mode, int odd, int64_t B, int C, int H, int W, nf_stream_t stream) {
    NfSplit s;
    if (!nf_make_split(s, z1c != nullptr ? mode : NF_SPLIT_NONE, odd, C, H, W) || ws_zero == nullptr) return NF_E_BADARG;
    const int G = nf_fbf_grid(B, C, H * W);
    if (G <= 0) return NF_E_BADARG;
    const size_t lds = sizeof(float) * (NF_FBF_MAX_VALUES + 6 * (size_t)C);
    hipLaunchKernelGGL(k_flowbn_head_fused, dim3((unsigned)G), dim3(NF_BLOCK), lds, (hipStream_t)stream, x, log_gamma, beta, batch_mean,
                       batch_var, running_mean, running_var, eps, momentum, y, z1c, ld, reinterpret_cast<unsigned long long*>(ws_zero), s,
                       (int)B, H * W);
    NF_CHECK_LAUNCH();
    return 0;
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
