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
#include "nf_det.h"

NF_DET_STATE(nf_mcd)
NF_DET_HOST_API(nf_mcd)
#include "nf_mfma16.h"
#include "nf_flow_rec.h"

#ifndef NF_MC_PROF
#define NF_MC_PROF 0      // 1 (tools/probes/mlp_chain_prof.py builds that variant): time stamps of workgroup 0 at phase boundaries
#endif
__device__ long long nf_mc_prof_buf[128];
NF_PERSIST_STATE(nf_mc)                                   // spin loops that gave up (a grid that was not co-resident): must stay 0
#define NF_MC_T(i)                                                                          \
    do {                                                                                    \
        if (NF_MC_PROF && blockIdx.x == 0 && threadIdx.x == 0) nf_mc_prof_buf[i] = wall_clock64(); \
    } while (0)
#if NF_MC_PROF
extern "C" int nf_mlp_chain_prof_read(long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(nf_mc_prof_buf), sizeof(long long) * 128);
}
#endif


// gradient store of the fold: atomic += (barrier-free fold, `afold` in scope), += or = otherwise
#define NF_MC_AFOLD_MAX_BLOCKS 2
#define NF_MC_ACC(ptr, val)                                                            \
    do {                                                                               \
        auto* p_ = (ptr);                                                              \
        const float v_ = (val);                                                        \
        if (afold) atomicAdd((float*)p_, v_);                                          \
        else *p_ = (accumulate ? *p_ : 0.f) + v_;                                      \
    } while (0)
static inline void nf_mlp_unpack(const void* const* t, NfMlpP& p) {
    for (int l = 0; l < NF_MC_NL; ++l) {
        NF_GSET(p.v[l], t[3 * l]); NF_GSET(p.g[l], t[3 * l + 1]); NF_GSET(p.b[l], t[3 * l + 2]);
    }
    for (int j = 0; j < NF_MC_NB; ++j) {
        const void* const* q = t + 3 * NF_MC_NL + 5 * j;
        NF_GSET(p.gamma[j], q[0]); NF_GSET(p.beta[j], q[1]);
        NF_GSET(p.rmean[j], q[2]); NF_GSET(p.rvar[j], q[3]); NF_GSET(p.nbt[j], q[4]);
    }
}

static inline void nf_glow_unpack(const void* const* t, NfGlowV& h) {
    NF_GSET(h.ls, t[0]); NF_GSET(h.bs, t[1]); NF_GSET(h.P, t[2]); NF_GSET(h.L, t[3]);
    NF_GSET(h.U, t[4]); NF_GSET(h.Lm, t[5]); NF_GSET(h.Um, t[6]); NF_GSET(h.sign_s, t[7]);
    NF_GSET(h.log_s, t[8]); NF_GSET(h.a, t[9]); NF_GSET(h.c, t[10]);
}

// LDS (floats)
#define NF_MC_W 0                                         // [6][32 * 36] weight_v, zero padded
#define NF_MC_WS (NF_MC_W + NF_MC_NL * 32 * NF_FP_ST)     // [6][32] weight-norm column scales g_k / (||v[:, k]|| + eps)
#define NF_MC_G (NF_MC_WS + NF_MC_NL * 32)                // [6][32] weight-norm gains g_k (the backward's fold reads them)
#define NF_MC_B (NF_MC_G + NF_MC_NL * 32)                 // [6][32] biases
#define NF_MC_GA (NF_MC_B + NF_MC_NL * 32)                // [5][32] gamma
#define NF_MC_BE (NF_MC_GA + NF_MC_NB * 32)               // [5][32] beta
#define NF_MC_BNC (NF_MC_BE + NF_MC_NB * 32)              // [5][4][32] per BatchNorm: scale, shift, mean, invstd
#define NF_MC_VAR (NF_MC_BNC + NF_MC_NB * 4 * 32)         // [5][32] biased batch variance (training bookkeeping)
#define NF_MC_RED (NF_MC_VAR + NF_MC_NB * 32)             // [16][64] cross-wave reduction
#define NF_MC_GB (NF_MC_RED + NF_MC_WAVES * 64)           // [5][64] backward: grid totals sum_g | sum_gx per BatchNorm (= g_beta | g_gamma)
#define NF_MC_TOT (NF_MC_GB + NF_MC_NB * 64)              // [2][64] grid totals of the exchange, double buffered by round parity
#define NF_MC_HEAD (NF_MC_TOT + 2 * 64)                   // fused Glow step: W (4 x 4), exp(log_scale) [4], bias [4], dld, a, c | L' | U' | P
#define NF_MC_TILES (NF_MC_HEAD + 32 + 48 + 16)           // + [80..83] statistics centre, [84..87] additive offset of the head                  // per-wave 16 x 36 tiles: scratch | (backward) G | activation
// the [blocks][64] gather buffer of the exchange aliases the scratch tiles when it fits in them (they are idle while it is
// live), else it follows the last tile
#define NF_MC_GATHER_IN_SCRATCH (NF_MLP_MAX_BLOCKS * 64 <= NF_MC_WAVES * 16 * NF_FP_ST)
#define NF_MC_GATHER(tiles_per_wave) (NF_MC_GATHER_IN_SCRATCH ? NF_MC_TILES : NF_MC_TILES + NF_MC_WAVES * (tiles_per_wave) * 16 * NF_FP_ST)

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
            if (++spins > nf_mc_spin_limit) { NF_PERSIST_GIVE_UP(nf_mc); break; }
        }
        __threadfence();
    }
    __syncthreads();
}

// Grid-wide sum of 64 per-workgroup values in ONE memory round trip: workgroup b publishes {value, generation} as single
// 64-bit stores into its own slots of round `round`, then every workgroup polls all slots until they carry the
// generation and adds them up in workgroup order (deterministic, no atomics, no fences: a slot is one naturally aligned
// 8-byte word).  The slots are zero at launch, generations are >= 1.  Measured against atomics + counter barrier: 1.5 vs 4 us.
// Split in two so that independent work (the backward's weight-gradient products) runs while the stores travel.
//   publish: red[w * 64 + i] per-wave partials (w < NF_MC_WAVES) -> this workgroup's slots
//   collect: sm[NF_MC_TOT + (round & 1) * 64 + i] (i < 64) = grid totals, valid after the call; double buffered because
//            there is no barrier between a wave reading them and a faster wave starting the next round
// one slot of the partner workgroup (two-workgroup grids): bounded poll, returns the value
__device__ __forceinline__ float nf_mc_poll1(const unsigned long long* p, unsigned gen) {
    unsigned long long v;
    unsigned spins = 0;
    do {
        v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(v >> 32) == gen) break;
        if (++spins > nf_mc_spin_limit) { NF_PERSIST_GIVE_UP(nf_mc); break; }
        __builtin_amdgcn_s_sleep(1);
    } while (true);
    return __uint_as_float((unsigned)v);
}
__device__ __forceinline__ void nf_mc_publish(float* sm, unsigned long long* slots, int round, unsigned gen) {
    float* red = sm + NF_MC_RED;
    __syncthreads();                                     // red (and any LDS tile written before) complete
    if (threadIdx.x < 64) {
        float mine = 0.f;
#pragma unroll
        for (int w = 0; w < NF_MC_WAVES; ++w) mine += red[w * 64 + threadIdx.x];
        if (gridDim.x <= 2) red[threadIdx.x] = mine;     // row 0 of red doubles as the result / keeps this workgroup's part (its partial is consumed)
        if (gridDim.x > 1) {
            const unsigned long long pk = ((unsigned long long)gen << 32) | (unsigned long long)__float_as_uint(mine);
            __hip_atomic_store(slots + ((size_t)round * NF_MLP_MAX_BLOCKS + blockIdx.x) * 64 + threadIdx.x, pk, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
__device__ __forceinline__ const float* nf_mc_collect(float* sm, int gather, unsigned long long* slots, int round, unsigned gen) {
    float* xs = sm + gather;                             // see NF_MC_GATHER
    float* tot = sm + NF_MC_TOT + (round & 1) * 64;
    const int G = gridDim.x;
    if (G == 1) {                                        // red[i] was written by the thread that copies it
        if (threadIdx.x < 64) tot[threadIdx.x] = sm[NF_MC_RED + threadIdx.x];
        __syncthreads();
        return tot;
    }
    const unsigned long long* rs = slots + (size_t)round * NF_MLP_MAX_BLOCKS * 64;
    if (G == 2) {
        // two workgroups (C1: 256 rows): the 64 publishing threads poll the partner's slot themselves and add their own part -- no
        // gather through LDS, one barrier (the general path below: 2.1 us per exchange at two workgroups, tools/probes/mlp_chain_prof.py)
        if (threadIdx.x < 64) {
            const float other = nf_mc_poll1(rs + (size_t)(1 - blockIdx.x) * 64 + threadIdx.x, gen);
            const float mine = sm[NF_MC_RED + threadIdx.x];
            tot[threadIdx.x] = blockIdx.x == 0 ? mine + other : other + mine;       // workgroup order, as below
        }
        __syncthreads();
        return tot;
    }
    // four slots per thread and trip, all four loads in flight before any is looked at: a poll round costs ONE memory latency
    // (polling them one after the other cost four: 2.2 us per exchange at 32 workgroups instead of ~1.3)
    for (int e0 = threadIdx.x; e0 < G * 64; e0 += 4 * NF_MC_THREADS) {
        unsigned long long v[4];
        unsigned spins = 0;
        bool ok;
        do {
            ok = true;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = e0 + k * NF_MC_THREADS;
                v[k] = __hip_atomic_load(rs + (e < G * 64 ? e : e0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) ok = ok && (unsigned)(v[k] >> 32) == gen;
            if (ok) break;
            if (++spins > nf_mc_spin_limit) { NF_PERSIST_GIVE_UP(nf_mc); break; }   // bounded: a mistake cannot hang the box
            __builtin_amdgcn_s_sleep(1);
        } while (true);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = e0 + k * NF_MC_THREADS;
            if (e < G * 64) xs[e] = __uint_as_float((unsigned)v[k]);
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
        for (int b = 0; b < G; ++b) t += xs[b * 64 + threadIdx.x];
        tot[threadIdx.x] = t;
    }
    __syncthreads();
    return tot;
}

// ---- batch STATISTICS through the exchange: (sum, M2) pairs combined by the parallel-variance rule -----------------------------------
// A one-pass E[x^2] - E[x]^2 (even centred at the producing linear's bias) loses (mean / std)^2 * 6e-8 of the variance, and the
// conditioner of a 2-D flow is full of features with |mean| >> std (its first linear has ONE input, all 32 outputs are affine in
// the same scalar): measured on C1 (RealNVP, B = 256) that cost 6e-4 per flow step in the backward's data gradient, 50x the fp32
// CPU path's own distance from float64 (tools/probes/parity_depth.py).  So every level keeps M2 = sum (x - mean_level)^2 about
// ITS OWN mean and levels are merged with  M2 = sum_i M2_i + sum_i n_i (mean_i - mean)^2  (Chan et al.): wave tile (16 rows) ->
// workgroup -> grid.  Same slots, same single memory round trip as the plain sums.
//   tile:    red[w * 64 + i] = tile sum, red[w * 64 + 32 + i] = tile M2 (i < 32 features), rows beyond N contribute nothing
//   publish: slots[i] = workgroup sum, slots[32 + i] = workgroup M2
//   collect: tot[i] = grid sum, tot[32 + i] = grid M2  (var = M2 / N, biased, as BatchNorm normalises)
__device__ __forceinline__ void nf_mc_tile_stats(const float (&c)[4], int g, int nt, float& s1, float& m2) {
    s1 = nf_fp_rowsum((c[0] + c[1]) + (c[2] + c[3]));
    const float mt = s1 / (float)max(nt, 1);
    float q = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const float d = (4 * s + g < nt) ? c[s] - mt : 0.f;                       // element [row 4 s + g] of the tile (nf_fp_load_cols)
        q = fmaf(d, d, q);
    }
    m2 = nf_fp_rowsum(q);
}
__device__ __forceinline__ int nf_mc_rows_of_block(int64_t N, int b) {
    const int64_t left = N - (int64_t)b * NF_MLP_ROWS_PER_BLOCK;
    return (int)(left < NF_MLP_ROWS_PER_BLOCK ? (left > 0 ? left : 0) : NF_MLP_ROWS_PER_BLOCK);
}
// two blocks of m rows each (sums S, S', squared deviations M2, M2' about their own means) merge into S + S', M2 + M2' + (S' - S)^2 / (2 m)
__device__ __forceinline__ void nf_mc_merge(float& S, float& M2, float So, float Mo, float half_inv_m) {
    const float dl = So - S;
    M2 = (M2 + Mo) + dl * dl * half_inv_m;
    S += So;
}
__device__ __forceinline__ void nf_mc_publish_stats(float* sm, unsigned long long* slots, int round, unsigned gen, int64_t N) {
    float* red = sm + NF_MC_RED;
    __syncthreads();
    if (threadIdx.x < 32) {
        const int i = threadIdx.x;
        const int nb = nf_mc_rows_of_block(N, blockIdx.x);
        float S, M2;
        if (nb == NF_MLP_ROWS_PER_BLOCK) {               // every tile full: pairwise tree over the waves, no division
            float sv[NF_MC_WAVES], mv[NF_MC_WAVES];
#pragma unroll
            for (int w = 0; w < NF_MC_WAVES; ++w) { sv[w] = red[w * 64 + i]; mv[w] = red[w * 64 + 32 + i]; }
#pragma unroll
            for (int st = 1; st < NF_MC_WAVES; st *= 2)
#pragma unroll
                for (int w = 0; w < NF_MC_WAVES; w += 2 * st) nf_mc_merge(sv[w], mv[w], sv[w + st], mv[w + st], 0.5f / (float)(16 * st));
            S = sv[0]; M2 = mv[0];
        } else {                                         // the last workgroup of a batch that does not fill it
            S = 0.f; M2 = 0.f;
            for (int w = 0; w < NF_MC_WAVES; ++w) S += red[w * 64 + i];
            const float mb = S / (float)max(nb, 1);
            for (int w = 0; w < NF_MC_WAVES; ++w) {
                const int nw = min(max(nb - 16 * w, 0), 16);
                const float d = red[w * 64 + i] / (float)max(nw, 1) - mb;
                M2 += nw > 0 ? fmaf((float)nw * d, d, red[w * 64 + 32 + i]) : 0.f;
            }
        }
        if (gridDim.x <= 2) { red[i] = S; red[32 + i] = M2; }     // one wave: every read above precedes these stores
        if (gridDim.x > 1) {
            unsigned long long* dst = slots + ((size_t)round * NF_MLP_MAX_BLOCKS + blockIdx.x) * 64 + i;
            __hip_atomic_store(dst, ((unsigned long long)gen << 32) | (unsigned long long)__float_as_uint(S), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst + 32, ((unsigned long long)gen << 32) | (unsigned long long)__float_as_uint(M2), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// collect: the merge over the workgroups is spread over the whole workgroup (thread = (channel i, part p): workgroups p, p + NP, ...);
// the parts meet in `red` (consumed by publish).  No division per workgroup: every block but the last holds NF_MLP_ROWS_PER_BLOCK rows.
__device__ __forceinline__ const float* nf_mc_collect_stats(float* sm, int gather, unsigned long long* slots, int round, unsigned gen,
                                                            int64_t N) {
    float* xs = sm + gather;
    float* part = sm + NF_MC_RED;                        // [NP][32]
    float* tot = sm + NF_MC_TOT + (round & 1) * 64;
    const int G = gridDim.x;
    if (G == 1) {
        if (threadIdx.x < 64) tot[threadIdx.x] = sm[NF_MC_RED + threadIdx.x];
        __syncthreads();
        return tot;
    }
    const unsigned long long* rs = slots + (size_t)round * NF_MLP_MAX_BLOCKS * 64;
    if (G == 2) {
        // two workgroups: the publishing lanes poll the partner's (sum, M2) themselves and merge -- same formula, same order as the
        // general path (block 0, then block 1), one barrier instead of five
        if (threadIdx.x < 32) {
            const int i = threadIdx.x;
            const unsigned long long* po = rs + (size_t)(1 - blockIdx.x) * 64 + i;
            const float So = nf_mc_poll1(po, gen), Mo = nf_mc_poll1(po + 32, gen);
            const float Sm = sm[NF_MC_RED + i], Mm = sm[NF_MC_RED + 32 + i];
            const float S0 = blockIdx.x == 0 ? Sm : So, M0 = blockIdx.x == 0 ? Mm : Mo;
            const float S1 = blockIdx.x == 0 ? So : Sm, M1 = blockIdx.x == 0 ? Mo : Mm;
            const int n0 = nf_mc_rows_of_block(N, 0), n1 = nf_mc_rows_of_block(N, 1);
            const float S = S0 + S1, mean = S / (float)N;
            const float inv_full = 1.f / (float)NF_MLP_ROWS_PER_BLOCK;
            const float d0 = S0 * (n0 == NF_MLP_ROWS_PER_BLOCK ? inv_full : 1.f / (float)max(n0, 1)) - mean;
            const float d1 = S1 * (n1 == NF_MLP_ROWS_PER_BLOCK ? inv_full : 1.f / (float)max(n1, 1)) - mean;
            tot[i] = S;
            tot[32 + i] = fmaf((float)n0 * d0, d0, M0) + fmaf((float)n1 * d1, d1, M1);
        }
        __syncthreads();
        return tot;
    }
    for (int e0 = threadIdx.x; e0 < G * 64; e0 += 4 * NF_MC_THREADS) {
        unsigned long long v[4];
        unsigned spins = 0;
        bool ok;
        do {
            ok = true;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = e0 + k * NF_MC_THREADS;
                v[k] = __hip_atomic_load(rs + (e < G * 64 ? e : e0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) ok = ok && (unsigned)(v[k] >> 32) == gen;
            if (ok) break;
            if (++spins > nf_mc_spin_limit) { NF_PERSIST_GIVE_UP(nf_mc); break; }
            __builtin_amdgcn_s_sleep(1);
        } while (true);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = e0 + k * NF_MC_THREADS;
            if (e < G * 64) xs[e] = __uint_as_float((unsigned)v[k]);
        }
    }
    __syncthreads();
    constexpr int NP = NF_MC_THREADS / 32;
    const int i = threadIdx.x & 31, p = threadIdx.x >> 5;
    float ps = 0.f;
    for (int b = p; b < G; b += NP) ps += xs[b * 64 + i];
    part[p * 32 + i] = ps;
    __syncthreads();
    float S = 0.f;
#pragma unroll
    for (int q = 0; q < NP; ++q) S += part[q * 32 + i];
    const float mean = S / (float)N;
    const float inv_full = 1.f / (float)NF_MLP_ROWS_PER_BLOCK;
    float pm = 0.f;
    for (int b = p; b < G; b += NP) {
        const int nb = nf_mc_rows_of_block(N, b);
        const float d = xs[b * 64 + i] * (nb == NF_MLP_ROWS_PER_BLOCK ? inv_full : 1.f / (float)max(nb, 1)) - mean;
        pm += fmaf((float)nb * d, d, xs[b * 64 + 32 + i]);
    }
    __syncthreads();                                     // every thread has read the sums' parts
    part[p * 32 + i] = pm;
    __syncthreads();
    if (threadIdx.x < 32) {
        float M2 = 0.f;
#pragma unroll
        for (int q = 0; q < NP; ++q) M2 += part[q * 32 + i];
        tot[i] = S;
        tot[32 + i] = M2;
    }
    __syncthreads();
    return tot;
}

struct NfGlowRaw;
__device__ __forceinline__ void nf_glow_head_phase_a(float* sm, const NfGlowV& h, const NfGlowRaw& raw);
__device__ __forceinline__ void nf_glow_head_phase_b(float* sm);

// what a thread brings from global memory for the staging of one MLP: requested by nf_mc_stage_load, placed by nf_mc_stage_store --
// together they are nf_mc_stage; apart, the whole-flow backward requests step s - 1's while step s computes (k_glow_flow_bwd)
struct NfMcStageRegs {
    float w[NF_MC_NL][32 * 32 / NF_MC_THREADS];
    float gk, bk, ga, be;
};
__device__ __forceinline__ void nf_mc_stage_load(const NfMlpP& p, int I0, int O_out, NfMcStageRegs& r) {
    const int tid = threadIdx.x, k = tid & 31;
    constexpr int RPT = 32 * 32 / NF_MC_THREADS;           // rows of a 32 x 32 matrix per thread (1 or 2)
#pragma unroll
    for (int l = 0; l < NF_MC_NL; ++l) {                  // 6 * RPT independent loads in flight, one latency
        const int I = l == 0 ? I0 : 32, O = l == NF_MC_NL - 1 ? O_out : 32;
#pragma unroll
        for (int h = 0; h < RPT; ++h) {
            const int oo = (tid >> 5) + h * (NF_MC_THREADS / 32);
            r.w[l][h] = (oo < O && k < I) ? p.v[l][oo * I + k] : 0.f;
        }
    }
    r.gk = r.bk = r.ga = r.be = 0.f;
    if (tid < NF_MC_NL * 32) {
        const int l = tid >> 5;
        const int I = l == 0 ? I0 : 32, O = l == NF_MC_NL - 1 ? O_out : 32;
        r.gk = k < I ? p.g[l][k] : 0.f;
        r.bk = k < O ? p.b[l][k] : 0.f;
    }
    if (tid < NF_MC_NB * 32) { r.ga = p.gamma[tid >> 5][k]; r.be = p.beta[tid >> 5][k]; }
}
__device__ __forceinline__ void nf_mc_stage_store(const NfMcStageRegs& r, float* sm, int I0, int O_out, float wn_eps,
                                                  const NfGlowV* head = nullptr, const NfGlowRaw* raw = nullptr) {
    const int tid = threadIdx.x, k = tid & 31;
    constexpr int RPT = 32 * 32 / NF_MC_THREADS;
    const float gk = r.gk, bk = r.bk, ga = r.ga, be = r.be;
#pragma unroll
    for (int l = 0; l < NF_MC_NL; ++l)
#pragma unroll
        for (int h = 0; h < RPT; ++h)
            sm[NF_MC_W + l * 32 * NF_FP_ST + ((tid >> 5) + h * (NF_MC_THREADS / 32)) * NF_FP_ST + k] = r.w[l][h];
    if (tid < NF_MC_NL * 32) sm[NF_MC_B + tid] = bk;
    if (tid < NF_MC_NB * 32) { sm[NF_MC_GA + tid] = ga; sm[NF_MC_BE + tid] = be; }
    if (head != nullptr) nf_glow_head_phase_a(sm, *head, *raw);
    __syncthreads();
    if (head != nullptr) nf_glow_head_phase_b(sm);
    if (tid < NF_MC_NL * 32) {                            // weight_norm.py:40: norm over the output index, per input column
        const float* W = sm + NF_MC_W + (tid >> 5) * 32 * NF_FP_ST;
        float ss = 0.f;
#pragma unroll 8
        for (int o = 0; o < 32; ++o) ss = fmaf(W[o * NF_FP_ST + k], W[o * NF_FP_ST + k], ss);
        const int I = (tid >> 5) == 0 ? I0 : 32;
        sm[NF_MC_WS + tid] = k < I ? gk / (sqrtf(ss) + wn_eps) : 0.f;
        sm[NF_MC_G + tid] = gk;
    }
    __syncthreads();
}
__device__ __forceinline__ void nf_mc_stage(const NfMlpP& p, float* sm, int I0, int O_out, float wn_eps, const NfGlowV* head = nullptr,
                                            const NfGlowRaw* raw = nullptr) {
    NfMcStageRegs r;
    nf_mc_stage_load(p, I0, O_out, r);
    nf_mc_stage_store(r, sm, I0, O_out, wn_eps, head, raw);
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

// batch statistics of BatchNorm j over the whole grid (dv = pre-bias output of the producing linear l = j), then its
// constants -> sm[NF_MC_BNC + 4 j ..] (one half-wave derives them: per-lane redundancy costs more issue slots on the
// 16-wave workgroup than the barrier it would save)
__device__ __forceinline__ void nf_mc_batchnorm_train(float* sm, int j, const float (&dv)[8], bool rv, unsigned long long* slots,
                                                      int64_t N, float eps, int c16, int g, int wid) {
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
    const int nt = min(max(nf_mc_rows_of_block(N, blockIdx.x) - 16 * wid, 0), 16);  // valid rows of this wave's tile
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) nf_mc_tile_stats(c[cb], g, nt, s1[cb], s2[cb]);
    float* red = sm + NF_MC_RED;
    if (g == 0) {
        red[wid * 64 + c16] = s1[0]; red[wid * 64 + 16 + c16] = s1[1];
        red[wid * 64 + 32 + c16] = s2[0]; red[wid * 64 + 48 + c16] = s2[1];
    }
    if (j == 1) NF_MC_T(40);
    nf_mc_publish_stats(sm, slots, j, (unsigned)(j + 1), N);
    const float* tot = nf_mc_collect_stats(sm, NF_MC_GATHER(1), slots, j, (unsigned)(j + 1), N);
    if (j == 1) NF_MC_T(41);
    if (threadIdx.x < 32) {
        const int k = threadIdx.x;
        const float invN = 1.f / (float)N;
        const float m1 = tot[k] * invN;
        const float mean = sm[NF_MC_B + j * 32 + k] + m1;                          // the linear's output is pre-bias
        const float var = tot[32 + k] * invN;                                      // biased, as BatchNorm normalises; M2 >= 0
        const float invstd = 1.f / sqrtf(var + eps);
        const float sc = sm[NF_MC_GA + j * 32 + k] * invstd;
        sm[NF_MC_BNC + (4 * j + 0) * 32 + k] = sc;
        sm[NF_MC_BNC + (4 * j + 1) * 32 + k] = sm[NF_MC_BE + j * 32 + k] - mean * sc;
        sm[NF_MC_BNC + (4 * j + 2) * 32 + k] = mean;
        sm[NF_MC_BNC + (4 * j + 3) * 32 + k] = invstd;
        sm[NF_MC_VAR + j * 32 + k] = var;                 // running-statistics bookkeeping happens once, after the last layer
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

// Head constants of the fused Glow step -> sm[NF_MC_HEAD ..], in two phases around the staging's first barrier so that every
// global load of a thread is issued up front (one memory latency):
//   phase A (threads 0..15: one entry each of L' = L o Lm + I, U' = U o Um + diag(sign_s exp(log_s)), P; threads 16..20: the
//            ActNorm scale / bias, the per-sample log-det of the head (modules.py:249, :480), the coupling scalars)
//   phase B (threads 0..15): W = P L' U' from LDS (modules.py:470-476)
struct NfGlowRaw { float v[8]; };                        // what a thread of phase A needs from global memory
__device__ __forceinline__ void nf_glow_head_load(const NfGlowV& h, NfGlowRaw& raw) {   // issued with the kernel's first loads
    const int t = threadIdx.x, D = h.D;
#pragma unroll
    for (int k = 0; k < 8; ++k) raw.v[k] = 0.f;
    if (t < 16) {
        const int r = t >> 2, c = t & 3;
        const bool ok = r < D && c < D;
        const int e = ok ? r * D + c : 0;
        raw.v[0] = h.L[e]; raw.v[1] = h.Lm[e]; raw.v[2] = h.U[e]; raw.v[3] = h.Um[e]; raw.v[4] = h.P[e];
        raw.v[5] = h.sign_s[r < D ? r : 0]; raw.v[6] = h.log_s[r < D ? r : 0];
    } else if (t < 20) {
        const int c = t - 16;
        raw.v[0] = h.ls[c < D ? c : 0]; raw.v[1] = h.bs[c < D ? c : 0];
    } else if (t == 20) {
#pragma unroll
        for (int c = 0; c < 4; ++c) raw.v[c] = c < D ? h.log_s[c] - h.ls[c] : 0.f;
        raw.v[4] = h.a[0]; raw.v[5] = h.c[0];
    }
}
__device__ __forceinline__ void nf_glow_head_phase_a(float* sm, const NfGlowV& h, const NfGlowRaw& raw) {
    const int t = threadIdx.x, D = h.D;
    if (t < 16) {
        const int r = t >> 2, c = t & 3;
        const bool ok = r < D && c < D;
        sm[NF_MC_HEAD + 32 + t] = ok ? raw.v[0] * raw.v[1] + (r == c ? 1.f : 0.f) : 0.f;
        sm[NF_MC_HEAD + 48 + t] = ok ? raw.v[2] * raw.v[3] + (r == c ? raw.v[5] * expf(raw.v[6]) : 0.f) : 0.f;
        sm[NF_MC_HEAD + 64 + t] = ok ? raw.v[4] : 0.f;
    } else if (t < 20) {
        const int c = t - 16;
        sm[NF_MC_HEAD + 16 + c] = c < D ? expf(raw.v[0]) : 1.f;
        sm[NF_MC_HEAD + 20 + c] = c < D ? raw.v[1] : 0.f;
    } else if (t >= 24 && t < 28) {
        sm[NF_MC_HEAD + 84 + (t - 24)] = 0.f;
    } else if (t == 20) {
        sm[NF_MC_HEAD + 24] = (raw.v[0] + raw.v[1]) + (raw.v[2] + raw.v[3]);           // modules.py:249, :480
        sm[NF_MC_HEAD + 25] = raw.v[4];
        sm[NF_MC_HEAD + 26] = raw.v[5];
    }
}
__device__ __forceinline__ void nf_glow_head_phase_b(float* sm) {
    const int t = threadIdx.x;
    if (t < 16) {
        const int r = t >> 2, c = t & 3;
        float w = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float tk = 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m) tk = fmaf(sm[NF_MC_HEAD + 32 + 4 * k + m], sm[NF_MC_HEAD + 48 + 4 * m + c], tk);
            w = fmaf(sm[NF_MC_HEAD + 64 + 4 * r + k], tk, w);
        }
        sm[NF_MC_HEAD + t] = w;
    }
}
// flow-BatchNorm head (RealNVP step, modules.py:283-307, affine=False) expressed in the same per-row form: bias := batch
// mean, exp(log_scale) := sqrt(var), W := diag(exp(log_gamma)), offset := beta, per-sample log-det sum(log_gamma - log(var)/2)
__device__ __forceinline__ void nf_fbn_head_consts(float* sm, const NfGlowV& h, int c, float mean, float var) {
    const bool ok = c < h.D;
    const float lg = ok ? h.ls[c] : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) sm[NF_MC_HEAD + 4 * c + k] = (ok && k == c) ? expf(lg) : 0.f;
    sm[NF_MC_HEAD + 16 + c] = ok ? sqrtf(var) : 1.f;
    sm[NF_MC_HEAD + 20 + c] = ok ? mean : 0.f;
    sm[NF_MC_HEAD + 84 + c] = ok ? h.bs[c] : 0.f;
    sm[NF_MC_HEAD + 28 + c] = ok ? lg - 0.5f * logf(var) : 0.f;
    if (c == 0) { sm[NF_MC_HEAD + 25] = h.a[0]; sm[NF_MC_HEAD + 26] = h.c[0]; }
}

// zn = (z - bias) / exp(log_scale);  hh = W zn        (every lane of the row's four does this: D <= 4)
__device__ __forceinline__ void nf_glow_head_row(const float* sm, const float (&zr)[4], float (&zn)[4], float (&hh)[4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) zn[c] = (zr[c] - sm[NF_MC_HEAD + 20 + c]) / sm[NF_MC_HEAD + 16 + c];      // modules.py:246
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) acc = fmaf(sm[NF_MC_HEAD + 4 * r + c], zn[c], acc);                     // modules.py:477
        hh[r] = acc + sm[NF_MC_HEAD + 84 + r];             // (beta of the flow-BatchNorm head; 0 for the Glow head)
    }
}
// INVERSE step (modules.py:250-256, :484-497): W^-1 = U'^-1 L'^-1 Pp by two triangular solves on one thread (D <= 4), Pp = the
// row-swap matrix of the LAPACK pivots (what torch.lu_solve applies to its right-hand side; the caller passes it in the P
// slot of the head) -> sm[NF_MC_HEAD + 0 .. 15], replacing W; then z = (W^-1 h) exp(log_scale) + bias per row.
__device__ __forceinline__ void nf_glow_head_inverse_weight(float* sm, int D) {
    const float* Lp = sm + NF_MC_HEAD + 32;
    const float* Up = sm + NF_MC_HEAD + 48;
    const float* Pp = sm + NF_MC_HEAD + 64;
    float X[4][4];
    for (int c = 0; c < 4; ++c) {
        for (int r = 0; r < 4; ++r) {                     // forward substitution, unit lower triangular
            float v = (r < D && c < D) ? Pp[4 * r + c] : 0.f;
            for (int k = 0; k < r; ++k) v -= Lp[4 * r + k] * X[k][c];
            X[r][c] = v;
        }
        for (int r = 3; r >= 0; --r) {                    // back substitution
            float v = X[r][c];
            for (int k = r + 1; k < 4; ++k) v -= Up[4 * r + k] * X[k][c];
            X[r][c] = r < D ? v / Up[4 * r + r] : 0.f;
        }
    }
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) sm[NF_MC_HEAD + 4 * r + c] = X[r][c];
}
__device__ __forceinline__ void nf_glow_head_inverse_row(const float* sm, const float (&hh)[4], float (&zr)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) acc = fmaf(sm[NF_MC_HEAD + 4 * r + c], hh[c] - sm[NF_MC_HEAD + 84 + c], acc);
        zr[r] = fmaf(acc, sm[NF_MC_HEAD + 16 + r], sm[NF_MC_HEAD + 20 + r]);
    }
}
// the conditioner input in R: feature e < D / 2 is element e of the conditioning half (squeeze.py:68-69: interleaved)
__device__ __forceinline__ void nf_glow_cond_input(const float (&hh)[4], int D, int odd, float (&xa)[8], int g) {
#pragma unroll
    for (int j = 0; j < 8; ++j) xa[j] = 0.f;
    if (g == 0) {
        xa[0] = odd ? hh[0] : hh[1];
        if (D == 4) xa[1] = odd ? hh[2] : hh[3];
    }
}
__device__ __forceinline__ void nf_glow_load_row(const float* z, int64_t row, bool rv, int D, float (&zr)[4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float v = z[(rv ? row : 0) * D + (c < D ? c : 0)];
        zr[c] = (rv && c < D) ? v : 0.f;
    }
}

__device__ __forceinline__ void nf_mc_load_x(const float* x, int64_t row, bool rv, int I0, float (&xa)[8], int g) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 16 * (j >> 2) + 4 * g + (j & 3);
        const float v = x[(rv ? row : 0) * I0 + (k < I0 ? k : 0)];
        xa[j] = (rv && k < I0) ? v : 0.f;
    }
}

// what a whole-flow kernel hands from one step to the next in registers (every lane of a row holds the row's values): the
// step's output row and log-det (forward) or the gradient of its input row (backward).  Going through global memory instead
// would put a store drain and a load round trip (~4 us) between two steps of the same workgroup.
struct NfMcCarry { float v[4]; float ld; int have; };

// The body of one launch is a device function so that the whole-flow kernels (k_glow_flow_*) can run it once per flow step
// inside ONE launch; hz / hy / hld are the step's input, output and log-det rows (h.z / h.y / h.ld of a single-step launch).
// INV (HEAD == 1 only): the INVERSE of the Glow step -- hz is the step's OUTPUT row y, the conditioner runs on its unchanged
// half, then coupling^-1, (1x1)^-1, ActNorm^-1; hy receives the step's input, hld -= the step's log-det (coupling.py:115-122).
template <int HEAD, bool INV = false>   // 0: the conditioner alone; 1: whole Glow step (ActNorm + 1x1 head); 2: whole RealNVP step (flow-BatchNorm head)
__device__ __forceinline__ void nf_mc_fwd_body(float* sm, const float* __restrict__ x, const NfMlpP& p, float* __restrict__ out,
                                               float* save, float* stats, int64_t N, int I0, int O_out, int training, float eps,
                                               float mom, float wn_eps, const NfGlowV& h, const float* hz, float* hy, float* hld,
                                               NfMcCarry* carry = nullptr) {
    constexpr bool GLOW = HEAD != 0, FBN = HEAD == 2;     // GLOW: a fused flow step (either head)
    static_assert(!INV || HEAD != 0, "the inverse body is a flow step's");
    NF_MC_T(0);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, c16 = lane & 15, g = lane >> 4;
    const int64_t row = ((int64_t)blockIdx.x * NF_MC_WAVES + wid) * 16 + c16;
    const bool rv = row < N;
    float xa[8], zr[4], hh[4], rm_old = 0.f, rv_old = 0.f, ld_in = 0.f;   // issued before the staging: one memory latency
    float frm = 0.f, frv = 0.f;
    NfGlowRaw head_raw;
    if (FBN && threadIdx.x < 4) {
        const NF_G float* ms = h.fbn_mode == 2 ? h.bmean : h.rmean;         // constants of the evaluation / inverse modes
        const NF_G float* vs = h.fbn_mode == 2 ? h.bvar : h.rvar;
        frm = (int)threadIdx.x < h.D ? ms[threadIdx.x] : 0.f;
        frv = (int)threadIdx.x < h.D ? vs[threadIdx.x] : 1.f;
        sm[NF_MC_HEAD + 80 + threadIdx.x] = frm;          // the centre of the shifted sums (flowbn_head.hip)
    }
    if (GLOW) {
        if (!FBN) nf_glow_head_load(h, head_raw);
        if (carry != nullptr && carry->have) {
#pragma unroll
            for (int c = 0; c < 4; ++c) zr[c] = carry->v[c];
            ld_in = carry->ld;
        } else {
            nf_glow_load_row(hz, row, rv, h.D, zr);
            if (rv && g == 0) ld_in = hld[row];
        }
    }
    else nf_mc_load_x(x, row, rv, I0, xa, g);
    if (training && blockIdx.x == 0 && threadIdx.x < NF_MC_NB * 32) {
        rm_old = p.rmean[threadIdx.x >> 5][threadIdx.x & 31];
        rv_old = p.rvar[threadIdx.x >> 5][threadIdx.x & 31];
    }
    nf_mc_stage(p, sm, I0, O_out, wn_eps, (GLOW && !FBN) ? &h : nullptr, (GLOW && !FBN) ? &head_raw : nullptr);
    unsigned long long* slots = (unsigned long long*)stats;
    if (FBN && h.fbn_mode != 0) {   // evaluation mode / inverse (modules.py:296-298, :309-322): the statistics are buffers, no exchange
        if (threadIdx.x < 4) nf_fbn_head_consts(sm, h, threadIdx.x, frm, frv);
        __syncthreads();
        if (threadIdx.x == 0) sm[NF_MC_HEAD + 24] = (sm[NF_MC_HEAD + 28] + sm[NF_MC_HEAD + 29]) + (sm[NF_MC_HEAD + 30] + sm[NF_MC_HEAD + 31]);
        __syncthreads();
    } else if (FBN) {   // batch statistics of z itself: one more exchange (round NF_MC_NB), then the head constants
        float* tile = sm + NF_MC_TILES + wid * 16 * NF_FP_ST;
        float v8[8], c4[2][4];
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = 0.f;
        if (g == 0 && rv) {
#pragma unroll
            for (int c = 0; c < 4; ++c) v8[c] = c < h.D ? zr[c] - sm[NF_MC_HEAD + 80 + c] : 0.f;
        }
        nf_fp_store_rows(v8, tile, c16, g);
        nf_fp_wsync();
        nf_fp_load_cols<2>(tile, c4, c16, g);
        nf_fp_wsync();
        float s1, s2;
        nf_mc_tile_stats(c4[0], g, min(max(nf_mc_rows_of_block(N, blockIdx.x) - 16 * wid, 0), 16), s1, s2);
        float* red = sm + NF_MC_RED + wid * 64;
        red[lane] = 0.f;
        nf_fp_wsync();
        if (g == 0 && c16 < 4) { red[c16] = s1; red[32 + c16] = s2; }
        nf_mc_publish_stats(sm, slots, NF_MC_NB, (unsigned)(NF_MC_NB + 1), N);
        const float* tot = nf_mc_collect_stats(sm, NF_MC_GATHER(1), slots, NF_MC_NB, (unsigned)(NF_MC_NB + 1), N);
        if (threadIdx.x < 4) {
            const int c = threadIdx.x;
            const float n = (float)N, m1 = tot[c] / n;
            const float mean = sm[NF_MC_HEAD + 80 + c] + m1;
            const float var = tot[32 + c] / n + h.fbn_eps;                           // biased, eps inside (modules.py:287)
            nf_fbn_head_consts(sm, h, c, mean, var);
            if (blockIdx.x == 0 && c < h.D) {
                h.bmean[c] = mean; h.bvar[c] = var;
                h.rmean[c] = frm * (1.f - h.fbn_mom) + mean * h.fbn_mom;             // modules.py:291-294
                h.rvar[c] = frv * (1.f - h.fbn_mom) + var * h.fbn_mom;
                save[2 * NF_MC_NB * 32 + c] = mean; save[2 * NF_MC_NB * 32 + 4 + c] = var;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) sm[NF_MC_HEAD + 24] = (sm[NF_MC_HEAD + 28] + sm[NF_MC_HEAD + 29]) + (sm[NF_MC_HEAD + 30] + sm[NF_MC_HEAD + 31]);
        __syncthreads();
    }
    if (INV) {
        if (FBN) {                                        // W = diag(exp(log_gamma)): invert in place
            if ((int)threadIdx.x < h.D) sm[NF_MC_HEAD + 5 * threadIdx.x] = 1.f / sm[NF_MC_HEAD + 5 * threadIdx.x];
        } else if (threadIdx.x == 0) nf_glow_head_inverse_weight(sm, h.D);
#pragma unroll
        for (int c = 0; c < 4; ++c) hh[c] = zr[c];         // y: its conditioning half is the head output's
        nf_glow_cond_input(hh, h.D, h.odd, xa, g);
        __syncthreads();
    } else if (GLOW) {
        float zn[4];
        nf_glow_head_row(sm, zr, zn, hh);
        nf_glow_cond_input(hh, h.D, h.odd, xa, g);
    }
    NF_MC_T(1);
    if (!training) {
        if (threadIdx.x < NF_MC_NB * 32) {
            const int j = threadIdx.x >> 5, k = threadIdx.x & 31;
            nf_mc_batchnorm_consts(sm, j, p.rmean[j][k], 1.f / sqrtf(p.rvar[j][k] + eps));
        }
        __syncthreads();
    }
    float a_in[8], av[8], dv[8], bias[8], stream[8];
    {   // linear 0: no BatchNorm in front, only the weight-norm scale
        float ws[8];
        nf_fp_ldvec(sm + NF_MC_WS, g, ws);
#pragma unroll
        for (int k = 0; k < 8; ++k) av[k] = xa[k] * ws[k];
    }
    nf_mc_linear(sm, 0, av, dv, c16, g);
    NF_MC_T(2);
#pragma unroll 1
    for (int l = 0; l < NF_MC_NL - 1; ++l) {              // dv = pre-bias output of linear l = input of BatchNorm l
        if (l > 0 && (l & 1) == 0) {                      // second linear of a residual block: add the block input
#pragma unroll
            for (int k = 0; k < 8; ++k) dv[k] += stream[k];
        }
        if (training) nf_mc_batchnorm_train(sm, l, dv, rv, slots, N, eps, c16, g, wid);
        nf_fp_ldvec(sm + NF_MC_B + l * 32, g, bias);
#pragma unroll
        for (int k = 0; k < 8; ++k) a_in[k] = dv[k] + bias[k];
        nf_mc_activate(sm, l, l + 1, a_in, av, g);
        if ((l & 1) == 0) {                               // acts[0], acts[2], acts[4] are the residual stream
#pragma unroll
            for (int k = 0; k < 8; ++k) stream[k] = a_in[k];
        }
        nf_mc_linear(sm, l + 1, av, dv, c16, g);
        NF_MC_T(3 + l);
    }
    if (training && blockIdx.x == 0) {                    // BatchNorm bookkeeping (nn.BatchNorm1d semantics), off the chain
        if (threadIdx.x < NF_MC_NB * 32) {
            const int j = threadIdx.x >> 5, k = threadIdx.x & 31;
            const float mean = sm[NF_MC_BNC + (4 * j + 2) * 32 + k], var = sm[NF_MC_VAR + j * 32 + k];
            const float unb = N > 1 ? var * ((float)N / (float)(N - 1)) : var;
            p.rmean[j][k] = (1.f - mom) * rm_old + mom * mean;
            p.rvar[j][k] = (1.f - mom) * rv_old + mom * unb;
            save[(2 * j + 0) * 32 + k] = mean;
            save[(2 * j + 1) * 32 + k] = sm[NF_MC_BNC + (4 * j + 3) * 32 + k];
            if (k == 0 && p.nbt[j] != nullptr) p.nbt[j][0] += 1;
        }
    }
    nf_fp_ldvec(sm + NF_MC_B + (NF_MC_NL - 1) * 32, g, bias);
    if (GLOW) {                                           // the affine coupling itself, coupling.py:104-113 (lane g = 0 has t | s_raw)
        float yv[4] = {0.f, 0.f, 0.f, 0.f}, ld_out = 0.f;
        if (rv && g == 0) {
            const int D = h.D, nh = D >> 1, sel0 = h.odd, sel1 = 1 ^ h.odd;
            const float ca = sm[NF_MC_HEAD + 25], cc = sm[NF_MC_HEAD + 26];
            float o4[4], dld = sm[NF_MC_HEAD + 24];
#pragma unroll
            for (int j = 0; j < 4; ++j) o4[j] = dv[j] + bias[j];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if (INV && e < nh) {                      // coupling^-1 (coupling.py:115-122): h0 = (y0 - t) exp(-s)
                    const float sv = tanhf(e == 0 ? o4[nh] : o4[nh + 1]) * ca + cc;
                    const float t = e == 0 ? o4[0] : o4[1];
                    const float en = expf(-sv);
                    if (sel0) hh[2 * e + 1] = (hh[2 * e + 1] - t) * en;
                    else hh[2 * e] = (hh[2 * e] - t) * en;
                    dld += sv;
                }
            }
            if (INV) {                                    // (1x1)^-1, ActNorm^-1; the log-det of the whole step is subtracted
                nf_glow_head_inverse_row(sm, hh, yv);
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < D) hy[row * D + c] = yv[c];
                dld = -dld;
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if (!INV && e < nh) {
                    const float sv = tanhf(e == 0 ? o4[nh] : o4[nh + 1]) * ca + cc;
                    const float t = e == 0 ? o4[0] : o4[1];
                    const float h0 = sel0 ? hh[2 * e + 1] : hh[2 * e], h1 = sel0 ? hh[2 * e] : hh[2 * e + 1];
                    const float y0 = h0 * expf(sv) + t;
                    hy[row * D + 2 * e + sel0] = y0;
                    hy[row * D + 2 * e + sel1] = h1;
                    if (sel0) { yv[2 * e + 1] = y0; yv[2 * e] = h1; } else { yv[2 * e] = y0; yv[2 * e + 1] = h1; }
                    dld += sv;
                }
            }
            ld_out = ld_in + dld;
            hld[row] = ld_out;
        }
        if (carry != nullptr) {                           // the next step of a whole-flow launch takes its row from here
#pragma unroll
            for (int c = 0; c < 4; ++c) carry->v[c] = __shfl(yv[c], c16, NF_WAVE);
            carry->ld = __shfl(ld_out, c16, NF_WAVE);
            carry->have = 1;
        }
    } else if (rv) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * (j >> 2) + 4 * g + (j & 3);
            if (k < O_out) out[row * O_out + k] = dv[j] + bias[j];
        }
    }
    NF_MC_T(8);
}

template <int HEAD, bool INV = false>
__global__ void __launch_bounds__(NF_MC_THREADS) k_mlp_chain_fwd(const float* __restrict__ x, NfMlpP p, float* __restrict__ out,
                                                                 float* save, float* stats, int64_t N, int I0, int O_out,
                                                                 int training, float eps, float mom, float wn_eps, NfGlowV h) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    nf_mc_fwd_body<HEAD, INV>(sm, x, p, out, save, stats, N, I0, O_out, training, eps, mom, wn_eps, h, h.z, h.y, h.ld);
}

static inline size_t nf_mc_lds_bytes(int tiles_per_wave) {
    const size_t tiles = (size_t)NF_MC_WAVES * tiles_per_wave * 16 * NF_FP_ST;
    return (NF_MC_TILES + tiles + (NF_MC_GATHER_IN_SCRATCH ? 0 : (size_t)NF_MLP_MAX_BLOCKS * 64)) * sizeof(float);
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
        hipError_t e = hipFuncSetAttribute((const void*)k_mlp_chain_fwd<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_mlp_chain_fwd<0>, dim3(grid), dim3(NF_MC_THREADS), lds, (hipStream_t)stream, x, p, out, save_stats,
                       ws_zero, N, I0, O_out, training, bn_eps, bn_momentum, wn_eps, NfGlowV{});
    NF_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// backward: forward recomputed from x and the saved batch statistics (no barrier needed for that), then the chain of
//   G_l -> [weight / bias gradient of linear l] -> G_l Weff_l -> ReLU mask -> two batch sums (grid exchange) -> BatchNorm
//   backward -> G_{l-1}
// with every activation and gradient of the wave's 16 rows in registers.  Weight gradients are formed per workgroup:
// all waves park their G and activation tiles in LDS, then wave w owns output block (w & 1, (w >> 1) & 1) over the rows
// of waves 4 (w >> 2) .. +3 -- 16 MFMAs each, partial results to this workgroup's slab.  After a final grid barrier
// workgroup l folds the slabs of linear l and applies the weight-norm backward (weight_norm.py:35-41).
// ---------------------------------------------------------------------------------------------------------------

static_assert(NF_MC_WAVES % 4 == 0 && NF_MLP_MAX_BLOCKS * NF_MLP_ROWS_PER_BLOCK == NF_MLP_MAX_ROWS, "geometry in include/nfhip.h");
static_assert((NF_MC_NB + 1) * NF_MLP_MAX_BLOCKS * 64 * 2 + 64 == NF_MLP_WS_FLOATS, "exchange workspace size in include/nfhip.h");
static_assert(NF_MC_SLAB * NF_MLP_MAX_BLOCKS == NF_MLP_BWD_SLAB_FLOATS, "slab workspace size in include/nfhip.h");
static_assert(NF_MC_SLAB == NF_MLP_BWD_SLAB_WG_FLOATS, "per-workgroup slab size in include/nfhip.h");

// this wave's share of g_Weff[L] and g_bias[L]: output block (wid & 1, (wid >> 1) & 1) over the rows of waves 4 (wid >> 2) .. + 3
template <int L>
__device__ __forceinline__ void nf_mc_wgrad_job(const float* sm, float* slab, int lane, int wid) {
    const int c16 = lane & 15, g = lane >> 4;
    const int ob = wid & 1, ib = (wid >> 1) & 1, kq = wid >> 2;
    f32x4 d = nf_fp_zero4();
    float bs = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float* Gt = sm + NF_MC_TILES + (NF_MC_WAVES + 4 * kq + q) * 16 * NF_FP_ST + 16 * ob + c16;
        const float* At = sm + NF_MC_TILES + (2 * NF_MC_WAVES + 4 * kq + q) * 16 * NF_FP_ST + 16 * ib + c16;
        float ga[4], av[4];
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) { ga[s2] = Gt[(4 * s2 + g) * NF_FP_ST]; av[s2] = At[(4 * s2 + g) * NF_FP_ST]; }
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            bs += ga[s2];
            d = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s2], av[s2], d, 0, 0, 0);
        }
    }
    float* sl = slab + L * NF_MC_SLAB_L + kq * NF_MC_SLAB_Q;       // [i][o]: the fold walks columns (fixed i) contiguously
    *(float4*)(sl + (16 * ib + c16) * 32 + 16 * ob + 4 * g) = make_float4(d[0], d[1], d[2], d[3]);
    bs = nf_fp_rowsum(bs);
    if (ib == 0 && g == 0) sl[1024 + 16 * ob + c16] = bs;
}

// fused Glow step: the 29 sums over the rows that the ActNorm / PLU / coupling-scalar gradients need, as entries of the
// product U^T V of two parked 16 x 32 tiles (U: G_h | g_zn | g_s, g_s tanh | g_ld;  V: zn | 1) -> red[wid][slot]:
//   slots 0..15 g_W[r][c] = sum G_h[r] zn[c];  16..19 sum g_zn[c] zn[c];  20..23 sum g_zn[c];  24..25 sum g_s[e];
//   26..27 sum g_s[e] tanh(s_raw[e]);  28 sum g_ld.       Waves 0, 4, .. (output block (0, 0)) own 64 rows each.
__device__ __forceinline__ void nf_glow_head_product(float* sm, int lane, int wid) {
    const int c16 = lane & 15, g = lane >> 4;
    float* red = sm + NF_MC_RED + wid * 64;
    red[lane] = 0.f;
    if ((wid & 3) != 0) return;
    nf_fp_wsync();
    const int kq = wid >> 2;
    f32x4 d = nf_fp_zero4();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float* Gt = sm + NF_MC_TILES + (NF_MC_WAVES + 4 * kq + q) * 16 * NF_FP_ST + c16;
        const float* At = sm + NF_MC_TILES + (2 * NF_MC_WAVES + 4 * kq + q) * 16 * NF_FP_ST + c16;
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) d = __builtin_amdgcn_mfma_f32_16x16x4f32(Gt[(4 * s2 + g) * NF_FP_ST], At[(4 * s2 + g) * NF_FP_ST], d, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {                                // d[r] = M[o = 4 g + r][i = c16]
        const int o = 4 * g + r, i = c16;
        int slot = -1;
        if (o < 4 && i < 4) slot = 4 * o + i;
        else if (o < 8 && i == o - 4) slot = 16 + (o - 4);
        else if (o < 8 && o >= 4 && i == 4) slot = 20 + (o - 4);
        else if (o >= 8 && o < 12 && i == 4) slot = 24 + (o - 8);
        else if (o == 12 && i == 4) slot = 28;
        if (slot >= 0) red[slot] = d[r];
    }
}

template <int L>
__device__ __forceinline__ void nf_mc_bwd_layer(float* sm, const float (&xa)[8], const float (&a)[NF_MC_NB][8], float (&G)[8],
                                                float (&Gs)[8], float* slab, unsigned long long* slots, float* g_x, int64_t row,
                                                bool rv, int64_t N, int I0, int training, int lane, int wid) {   // L == 0: G <- g_x
    const int c16 = lane & 15, g = lane >> 4;
    float* TS = sm + NF_MC_TILES + wid * 16 * NF_FP_ST;
    float* TG = sm + NF_MC_TILES + (NF_MC_WAVES + wid) * 16 * NF_FP_ST;
    float* TA = sm + NF_MC_TILES + (2 * NF_MC_WAVES + wid) * 16 * NF_FP_ST;
    {   // what linear L multiplied with (before the weight-norm column scale): x, or ReLU(BatchNorm_{L-1}(a_{L-1}))
        float act[8];
        if (L == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) act[k] = xa[k];
        } else {
            float sc[8], sh[8];
            nf_fp_ldvec(sm + NF_MC_BNC + (4 * (L - 1) + 0) * 32, g, sc);
            nf_fp_ldvec(sm + NF_MC_BNC + (4 * (L - 1) + 1) * 32, g, sh);
#pragma unroll
            for (int k = 0; k < 8; ++k) act[k] = fmaxf(fmaf(a[L > 0 ? L - 1 : 0][k], sc[k], sh[k]), 0.f);
        }
        nf_fp_store_rows(G, TG, c16, g);
        nf_fp_store_rows(act, TA, c16, g);
    }
    if (L == 4) NF_MC_T(80);
    float t[8];
    {   // G Weff_L
        f32x4 acc[2] = {nf_fp_zero4(), nf_fp_zero4()};
        nf_fp_gemm_d<2>(sm + NF_MC_W + L * 32 * NF_FP_ST, NF_FP_ST, 0, G, acc, c16, g);
        float ws[8];
        nf_fp_ldvec(sm + NF_MC_WS + L * 32, g, ws);
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = acc[k >> 2][k & 3] * ws[k];
    }
    if (L == 0) {                                         // gradient of the conditioner input; last weight-gradient products
        if (g_x != nullptr && rv) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 16 * (j >> 2) + 4 * g + (j & 3);
                if (k < I0) g_x[row * I0 + k] = t[j];
            }
        }
        __syncthreads();
        nf_mc_wgrad_job<L>(sm, slab, lane, wid);
#pragma unroll
        for (int k = 0; k < 8; ++k) G[k] = t[k];
        return;
    }
    constexpr int J = L > 0 ? L - 1 : 0;                  // the BatchNorm between a_J and linear L
    float sc[8], sh[8], mean[8], invstd[8], gn[8], xh[8], gnx[8];
    nf_fp_ldvec(sm + NF_MC_BNC + (4 * J + 0) * 32, g, sc);
    nf_fp_ldvec(sm + NF_MC_BNC + (4 * J + 1) * 32, g, sh);
    nf_fp_ldvec(sm + NF_MC_BNC + (4 * J + 2) * 32, g, mean);
    nf_fp_ldvec(sm + NF_MC_BNC + (4 * J + 3) * 32, g, invstd);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        gn[k] = fmaf(a[J][k], sc[k], sh[k]) > 0.f ? t[k] : 0.f;      // ReLU mask; rows beyond N carry G = 0 -> t = 0
        xh[k] = (a[J][k] - mean[k]) * invstd[k];
        gnx[k] = gn[k] * xh[k];
    }
    {   // column sums of gn and gn * xhat over this wave's rows -> red[wid]
        float c[2][4], s1[2], s2[2];
        nf_fp_store_rows(gn, TS, c16, g);
        nf_fp_wsync();
        nf_fp_load_cols<2>(TS, c, c16, g);
        nf_fp_wsync();
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) s1[cb] = nf_fp_rowsum((c[cb][0] + c[cb][1]) + (c[cb][2] + c[cb][3]));
        nf_fp_store_rows(gnx, TS, c16, g);
        nf_fp_wsync();
        nf_fp_load_cols<2>(TS, c, c16, g);
        nf_fp_wsync();
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) s2[cb] = nf_fp_rowsum((c[cb][0] + c[cb][1]) + (c[cb][2] + c[cb][3]));
        float* red = sm + NF_MC_RED;
        if (g == 0) {
            red[wid * 64 + c16] = s1[0]; red[wid * 64 + 16 + c16] = s1[1];
            red[wid * 64 + 32 + c16] = s2[0]; red[wid * 64 + 48 + c16] = s2[1];
        }
    }
    if (L == 4) NF_MC_T(81);
    nf_mc_publish(sm, slots, NF_MC_NB - 1 - J, (unsigned)(NF_MC_NB - J));    // its barrier also covers the G / act tiles
    if (L == 4) NF_MC_T(82);
    nf_mc_wgrad_job<L>(sm, slab, lane, wid);                                 // runs while the partial sums travel
    if (L == 4) NF_MC_T(83);
    const float* tot = nf_mc_collect(sm, NF_MC_GATHER(3), slots, NF_MC_NB - 1 - J, (unsigned)(NF_MC_NB - J));
    if (L == 4) NF_MC_T(84);
    if (threadIdx.x < 64) sm[NF_MC_GB + J * 64 + threadIdx.x] = tot[threadIdx.x];
    float mg[8], mgx[8];
    nf_fp_ldvec(tot, g, mg);
    nf_fp_ldvec(tot + 32, g, mgx);
    const float invN = training ? 1.f / (float)N : 0.f;   // evaluation mode: the statistics are constants
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float v = sc[k] * (gn[k] - mg[k] * invN - xh[k] * (mgx[k] * invN));      // sc = gamma * invstd
        if (J == 0 || J == 2) v += Gs[k];                 // a_J also feeds the residual connection two linears on
        v = rv ? v : 0.f;
        G[k] = v;
        if (J == 2 || J == 4) Gs[k] = v;
    }
}

// the column-owner fold (more than two slabs) and the head gradients, shared by the backward body (fold workgroup = the
// workgroup itself, after the grid barrier) and by k_glow_fold_all (the deferred fold of every step of a flow in one launch):
// unit = one column (l, i) of a weight matrix or one bias vector, owned by a half wave (lane = output index o): sums the
// nslabs partials (coalesced), reduces <g_Weff, v> and ||v||^2 over o by shuffles, applies the weight-norm backward.  Needs
// the staged weights / gains / head constants of nf_mc_stage in sm.
__device__ __forceinline__ void nf_mc_fold_units(const float* sm, const float* __restrict__ slabs, int nslabs, int fb, int nfb,
                                                 const NfMlpG& gr, int accumulate, int I0, int O_out, float wn_eps) {
    const int o = threadIdx.x & 31;
    const int G_ = nslabs;
    constexpr int HW = NF_MC_THREADS / 32;                           // half waves per workgroup
    for (int u = fb * HW + (threadIdx.x >> 5); u < NF_MC_NL * 33; u += nfb * HW) {
        const int l = u / 33, i = u - l * 33;                    // i == 32: the bias
        const int I = l == 0 ? I0 : 32, O = l == NF_MC_NL - 1 ? O_out : 32;
        if (i < 32 && i >= I) continue;                          // half-wave uniform
        float tsum = 0.f;
        {
            const float* base = slabs + (size_t)l * NF_MC_SLAB_L + i * 32 + o;
            for (int b0 = 0; b0 < G_; b0 += 8) {                 // 8 NKQ independent loads in flight: one latency per trip
                float v[8][NF_MC_NKQ];
#pragma unroll
                for (int q8 = 0; q8 < 8; ++q8) {
                    const int b = b0 + q8 < G_ ? b0 + q8 : G_ - 1;
#pragma unroll
                    for (int q = 0; q < NF_MC_NKQ; ++q) v[q8][q] = base[(size_t)b * NF_MC_SLAB + q * NF_MC_SLAB_Q];
                }
#pragma unroll
                for (int q8 = 0; q8 < 8; ++q8)
#pragma unroll
                    for (int q = 0; q < NF_MC_NKQ; ++q)
                        if (b0 + q8 < G_) tsum += v[q8][q];
            }
        }
        if (i == 32) {
            if (o < O) gr.b[l][o] = (accumulate ? gr.b[l][o] : 0.f) + tsum;
            continue;
        }
        const float v = sm[NF_MC_W + l * 32 * NF_FP_ST + o * NF_FP_ST + i];
        float n2 = v * v, dt = tsum * v;
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) {
            n2 += __shfl_xor(n2, off, NF_WAVE);
            dt += __shfl_xor(dt, off, NF_WAVE);
        }
        const float nrm = sqrtf(n2), den = nrm + wn_eps, gi = sm[NF_MC_G + l * 32 + i];
        if (o < O) {
            float gv = tsum * (gi / den);
            if (nrm > 0.f) gv -= v * (dt * gi / (den * den * nrm));
            float* dst = gr.v[l] + o * I + i;
            *dst = (accumulate ? *dst : 0.f) + gv;
        }
        if (o == 0) gr.g[l][i] = (accumulate ? gr.g[l][i] : 0.f) + dt / den;
    }
}

// ActNorm / PLU / coupling-scalar gradients of a fused Glow (HEAD 1) or RealNVP (HEAD 2) step from the 64 grid totals of the
// head product (hs) and the staged head constants (nf_glow_head_phase_a / _b); one thread
template <int HEAD>
__device__ __forceinline__ void nf_mc_head_grads(const float* sm, const NfGlowV& h, const float* head_tot, int accumulate, bool afold) {
    constexpr bool FBN = HEAD == 2;
    {
        const float* hs = head_tot;
        const int D = h.D;
        const float sum_gld = hs[28];
        NF_MC_ACC(h.g_a, hs[26] + hs[27]);                                           // d/d s_log_scale: sum g_s tanh(s_raw)
        NF_MC_ACC(h.g_c, hs[24] + hs[25]);                                           // d/d s_bias
        if (!FBN) {
        const float* Lp = sm + NF_MC_HEAD + 32;                  // [4][4] each, staged at kernel start
        const float* Up = sm + NF_MC_HEAD + 48;
        const float* Pm = sm + NF_MC_HEAD + 64;
        float A[4][4];
        for (int r = 0; r < D; ++r) {
            const float es = sm[NF_MC_HEAD + 16 + r];
            NF_MC_ACC(h.g_bs + r, -hs[20 + r] / es);                                 // modules.py:246
            NF_MC_ACC(h.g_ls + r, -hs[16 + r] - sum_gld);                            // pixels = 1
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {                        // A = P^T g_W
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) acc = fmaf(Pm[k * 4 + r], hs[4 * k + c], acc);
                A[r][c] = acc;
            }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float gl = 0.f, gu = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    gl = fmaf(A[r][k], Up[c * 4 + k], gl);       // (A U'^T)[r][c]
                    gu = fmaf(Lp[k * 4 + r], A[k][c], gu);       // (L'^T A)[r][c]
                }
                if (r < D && c < D) {
                    const int e = r * D + c;
                    NF_MC_ACC(h.g_L + e, gl * h.Lm[e]);
                    NF_MC_ACC(h.g_U + e, gu * h.Um[e]);
                    if (r == c) NF_MC_ACC(h.g_log_s + r, gu * Up[r * 4 + r] + sum_gld);
                }
            }
        }
    }
}

// Whole-flow backward: everything a step reads from global memory before its first barrier -- weights, BatchNorm vectors and saved
// statistics, the head's parameters, its input row -- is requested for step s - 1 right after step s has staged its own (NfMcNext
// names where from), travels under step s's matrix work, and is placed at the top of step s - 1 without a wait: the start-of-step
// round trip (2.7 of a step's 25 us at two workgroups, tools/probes/mlp_chain_prof.py) leaves the latency chain.
struct NfMcPre {
    NfMcStageRegs st;
    NfGlowRaw raw;
    float bn_mean, bn_istd_raw, fm, fv, zr[4], gld;
    int have, have_gld;
};
struct NfMcNext { const NfGlowFlowStep* st; const float* save; const float* hz; };

template <int HEAD>
__device__ __forceinline__ void nf_mc_bwd_body(float* sm, const float* __restrict__ x, const NfMlpP& p, const float* __restrict__ save,
                                               const float* __restrict__ g_out, float* __restrict__ g_x, const NfMlpG& gr,
                                               int accumulate, float* ws, float* __restrict__ slabs, int64_t N, int I0, int O_out,
                                               int training, float eps, float wn_eps, const NfGlowV& h, const float* hz,
                                               const float* hgy, const float* hgld, float* hgz, NfMcCarry* carry = nullptr,
                                               float* head_rec = nullptr, NfMcPre* pre = nullptr, const NfMcNext* nx = nullptr) {
    constexpr bool GLOW = HEAD != 0, FBN = HEAD == 2;
    NF_MC_T(64);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, c16 = lane & 15, g = lane >> 4;
    const int64_t row = ((int64_t)blockIdx.x * NF_MC_WAVES + wid) * 16 + c16;
    const bool rv = row < N;
    float xa[8], a[NF_MC_NB][8], G[8], Gs[8];
    float zr[4], gy[4], gld = 0.f;                        // fused Glow step: this row of z and of the incoming gradients
    NfGlowRaw head_raw;
    float fm = 0.f, fv = 1.f;
    const bool pf = pre != nullptr && pre->have;          // (uniform over the grid: a step counter)
    if (FBN && pf) {
        fm = pre->fm; fv = pre->fv;
    } else if (FBN && threadIdx.x < 4) {
        fm = (int)threadIdx.x < h.D ? save[2 * NF_MC_NB * 32 + threadIdx.x] : 0.f;
        fv = (int)threadIdx.x < h.D ? save[2 * NF_MC_NB * 32 + 4 + threadIdx.x] : 1.f;
    }
    if (GLOW) {
        if (!FBN) {
            if (pf) head_raw = pre->raw;
            else nf_glow_head_load(h, head_raw);
        }
        if (pf) {
#pragma unroll
            for (int c = 0; c < 4; ++c) zr[c] = pre->zr[c];
        } else {
            nf_glow_load_row(hz, row, rv, h.D, zr);
        }
        if (carry != nullptr && carry->have) {
#pragma unroll
            for (int c = 0; c < 4; ++c) gy[c] = carry->v[c];
        } else {
            nf_glow_load_row(hgy, row, rv, h.D, gy);
        }
        if (pre != nullptr && pre->have_gld) {
            gld = pre->gld;                               // (the same vector for every step of the run)
        } else {
            if (hgld != nullptr && rv) gld = hgld[row];
            if (pre != nullptr) { pre->gld = gld; pre->have_gld = 1; }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) Gs[j] = 0.f;
    } else {
        nf_mc_load_x(x, row, rv, I0, xa, g);              // both loads are in flight while the weights are staged
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * (j >> 2) + 4 * g + (j & 3);
            const float v = g_out[(rv ? row : 0) * O_out + (k < O_out ? k : 0)];
            G[j] = (rv && k < O_out) ? v : 0.f;
            Gs[j] = 0.f;
        }
    }
    float bn_mean = 0.f, bn_invstd = 0.f;
    NfMcStageRegs sr;
    if (pf) {
        bn_mean = pre->bn_mean;
        bn_invstd = training ? pre->bn_istd_raw : 1.f / sqrtf(pre->bn_istd_raw + eps);
        sr = pre->st;
    } else {
        if (threadIdx.x < NF_MC_NB * 32) {
            const int j = threadIdx.x >> 5, k = threadIdx.x & 31;
            bn_mean = training ? save[(2 * j + 0) * 32 + k] : p.rmean[j][k];
            bn_invstd = training ? save[(2 * j + 1) * 32 + k] : 1.f / sqrtf(p.rvar[j][k] + eps);
        }
        nf_mc_stage_load(p, I0, O_out, sr);
    }
    nf_mc_stage_store(sr, sm, I0, O_out, wn_eps, (GLOW && !FBN) ? &h : nullptr, (GLOW && !FBN) ? &head_raw : nullptr);
    NF_MC_T(65);
    if (threadIdx.x < NF_MC_NB * 32) nf_mc_batchnorm_consts(sm, threadIdx.x >> 5, bn_mean, bn_invstd);
    if (FBN && threadIdx.x < 4) nf_fbn_head_consts(sm, h, threadIdx.x, fm, fv);
    __syncthreads();
    if (pre != nullptr) {                                 // the next step's requests (NfMcPre), behind this step's own staging
        pre->have = 0;
        if (nx != nullptr && nx->st != nullptr) {
            const NfGlowFlowStep& ns = *nx->st;
            nf_mc_stage_load(ns.p, I0, O_out, pre->st);
            pre->bn_mean = 0.f; pre->bn_istd_raw = 1.f;
            if (threadIdx.x < NF_MC_NB * 32) {
                const int j = threadIdx.x >> 5, k = threadIdx.x & 31;
                pre->bn_mean = training ? nx->save[(2 * j + 0) * 32 + k] : ns.p.rmean[j][k];
                pre->bn_istd_raw = training ? nx->save[(2 * j + 1) * 32 + k] : ns.p.rvar[j][k];
            }
            pre->fm = 0.f; pre->fv = 1.f;
            if (FBN && threadIdx.x < 4) {
                pre->fm = (int)threadIdx.x < ns.h.D ? nx->save[2 * NF_MC_NB * 32 + threadIdx.x] : 0.f;
                pre->fv = (int)threadIdx.x < ns.h.D ? nx->save[2 * NF_MC_NB * 32 + 4 + threadIdx.x] : 1.f;
            }
            if (GLOW) {
                nf_glow_load_row(nx->hz, row, rv, ns.h.D, pre->zr);
                if (!FBN) nf_glow_head_load(ns.h, pre->raw);
            }
            pre->have = 1;
        }
    }
    float zn[4], hh[4];
    if (GLOW) {
        nf_glow_head_row(sm, zr, zn, hh);
        nf_glow_cond_input(hh, h.D, h.odd, xa, g);
    }
    unsigned long long* slots = (unsigned long long*)ws;
    unsigned* counter = (unsigned*)(ws + (NF_MC_NB + 1) * NF_MLP_MAX_BLOCKS * 64 * 2);
    float* slab = slabs + (size_t)blockIdx.x * NF_MC_SLAB;

    // ---- forward, activations kept ------------------------------------------------------------------------------
    {
        float av[8], dv[8], bias[8], ws0[8];
        nf_fp_ldvec(sm + NF_MC_WS, g, ws0);
#pragma unroll
        for (int k = 0; k < 8; ++k) av[k] = xa[k] * ws0[k];
        nf_mc_linear(sm, 0, av, dv, c16, g);
        nf_fp_ldvec(sm + NF_MC_B, g, bias);
#pragma unroll
        for (int k = 0; k < 8; ++k) a[0][k] = dv[k] + bias[k];
#pragma unroll
        for (int l = 1; l < NF_MC_NB; ++l) {
            nf_mc_activate(sm, l - 1, l, a[l - 1], av, g);
            nf_mc_linear(sm, l, av, dv, c16, g);
            nf_fp_ldvec(sm + NF_MC_B + l * 32, g, bias);
#pragma unroll
            for (int k = 0; k < 8; ++k) a[l][k] = dv[k] + bias[k] + ((l & 1) == 0 ? a[l - 2][k] : 0.f);
        }
    }
    float gsv[2] = {0.f, 0.f}, gsvth[2] = {0.f, 0.f}, Gh[4] = {0.f, 0.f, 0.f, 0.f};
    if (GLOW) {   // conditioner output (t | s_raw) once more, then the affine coupling's backward (coupling.py:104-113)
        float av[8], dv[8], bias[8], o4[4];
        nf_mc_activate(sm, NF_MC_NB - 1, NF_MC_NL - 1, a[NF_MC_NB - 1], av, g);
        nf_mc_linear(sm, NF_MC_NL - 1, av, dv, c16, g);
        nf_fp_ldvec(sm + NF_MC_B + (NF_MC_NL - 1) * 32, g, bias);
#pragma unroll
        for (int j = 0; j < 4; ++j) o4[j] = __shfl(dv[j] + bias[j], c16, NF_WAVE);      // features 0..3 live in the g = 0 lane
        const int D = h.D, nh = D >> 1, sel0 = h.odd;
        const float ca = sm[NF_MC_HEAD + 25], cc = sm[NF_MC_HEAD + 26];
#pragma unroll
        for (int j = 0; j < 8; ++j) G[j] = 0.f;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (e < nh) {
                const float th = tanhf(e == 0 ? o4[nh] : o4[nh + 1]);
                const float ev = expf(th * ca + cc);
                const float h0 = sel0 ? hh[2 * e + 1] : hh[2 * e];
                const float gy0 = sel0 ? gy[2 * e + 1] : gy[2 * e], gy1 = sel0 ? gy[2 * e] : gy[2 * e + 1];
                gsv[e] = gy0 * h0 * ev + gld;                                  // ld += s: the log-det gradient enters here
                gsvth[e] = gsv[e] * th;
                const float gsraw = gsv[e] * ca * (1.f - th * th);
                if (sel0) { Gh[2 * e + 1] = gy0 * ev; Gh[2 * e] = gy1; } else { Gh[2 * e] = gy0 * ev; Gh[2 * e + 1] = gy1; }
                if (g == 0 && rv) {                                            // R: feature e = g_t, feature nh + e = g_s_raw
                    if (e == 0) { G[0] = gy0; if (nh == 1) G[1] = gsraw; else G[2] = gsraw; }
                    else { G[1] = gy0; G[3] = gsraw; }
                }
            }
        }
    }
    NF_MC_T(66);
    // ---- backward -----------------------------------------------------------------------------------------------
    nf_mc_bwd_layer<5>(sm, xa, a, G, Gs, slab, slots, g_x, row, rv, N, I0, training, lane, wid);
    NF_MC_T(67);
    nf_mc_bwd_layer<4>(sm, xa, a, G, Gs, slab, slots, g_x, row, rv, N, I0, training, lane, wid);
    NF_MC_T(68);
    nf_mc_bwd_layer<3>(sm, xa, a, G, Gs, slab, slots, g_x, row, rv, N, I0, training, lane, wid);
    NF_MC_T(69);
    nf_mc_bwd_layer<2>(sm, xa, a, G, Gs, slab, slots, g_x, row, rv, N, I0, training, lane, wid);
    NF_MC_T(70);
    nf_mc_bwd_layer<1>(sm, xa, a, G, Gs, slab, slots, g_x, row, rv, N, I0, training, lane, wid);
    NF_MC_T(71);
    nf_mc_bwd_layer<0>(sm, xa, a, G, Gs, slab, slots, g_x, row, rv, N, I0, training, lane, wid);
    if (GLOW) {   // ActNorm + 1x1 backward per row; their parameter gradients are one more 32 x 32 product over the rows
        const int D = h.D, sel0 = h.odd;
        const float gz0 = __shfl(G[0], c16, NF_WAVE), gz1 = __shfl(G[1], c16, NF_WAVE);   // g_x of the conditioner (lane g = 0)
        if (sel0) { Gh[0] += gz0; if (D == 4) Gh[2] += gz1; } else { Gh[1] += gz0; if (D == 4) Gh[3] += gz1; }
        float gzn[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float acc = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = fmaf(sm[NF_MC_HEAD + 4 * r + c], Gh[r], acc);
            gzn[c] = rv ? acc : 0.f;
        }
        float gzv[4] = {0.f, 0.f, 0.f, 0.f};
        if (g == 0 && rv) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < D) {
                    gzv[c] = gzn[c] / sm[NF_MC_HEAD + 16 + c];
                    hgz[row * D + c] = gzv[c];
                }
        }
        if (carry != nullptr) {                           // = the incoming gradient of the previous step of a whole-flow launch
#pragma unroll
            for (int c = 0; c < 4; ++c) carry->v[c] = __shfl(gzv[c], c16, NF_WAVE);
            carry->have = 1;
        }
        float u8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, v8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (rv) {
            if (g == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { u8[c] = Gh[c]; v8[c] = c < D ? zn[c] : 0.f; }
            } else if (g == 1) {
#pragma unroll
                for (int c = 0; c < 4; ++c) u8[c] = gzn[c];
                v8[0] = 1.f;
            } else if (g == 2) {
                u8[0] = gsv[0]; u8[1] = gsv[1]; u8[2] = gsvth[0]; u8[3] = gsvth[1];
            } else {
                u8[0] = gld;
            }
        }
        __syncthreads();                                  // every wave is done with the tiles of linear 0
        nf_fp_store_rows(u8, sm + NF_MC_TILES + (NF_MC_WAVES + wid) * 16 * NF_FP_ST, c16, g);
        nf_fp_store_rows(v8, sm + NF_MC_TILES + (2 * NF_MC_WAVES + wid) * 16 * NF_FP_ST, c16, g);
        __syncthreads();
        nf_glow_head_product(sm, lane, wid);
    }
    NF_MC_T(72);

    // ---- slabs -> parameter gradients: workgroup l (mod grid) owns linear l, workgroup 0 the BatchNorm affines ------------
    // One or two workgroups (and += semantics): NO grid barrier -- the weight-norm backward and the head's PLU algebra are
    // linear in the summed partials, so every workgroup folds ITS OWN slab / head sums and adds the result into the gradient
    // buffers with float atomics (C1, two workgroups: 1.84 -> 1.71 ms per step).  With more workgroups every one of them
    // repeats the whole fold instead of a 1/G share and the shared addresses contend: measured 1.72 -> 1.88 ms at 8 and
    // 1.90 -> 2.13 ms at 32 workgroups, so the barrier + owner fold below stays there.  The atomic order varies: gradients
    // repeat to ~1e-7.
    // Deferred fold (head_rec != nullptr, a run of steps launched one by one: nf_glow_flow_steps_bwd): the grid barrier and the
    // fold are the only part of a step nothing downstream waits for, yet they sit on the critical path of its launch (6.4 us
    // of ~27 at 32 workgroups).  The workgroup leaves its slab and its 64 head sums in memory and ends; ONE launch after the
    // last step folds every step of the run at once (k_glow_fold_all).
    const bool defer = GLOW && head_rec != nullptr;
    // (deterministic mode: no barrier-free atomic fold -- with two workgroups adding into a NON-zero gradient the order of the two adds
    // shows in the last bit; the grid fold behind the barrier has one writer per address)
    const bool afold = !defer && accumulate != 0 && gridDim.x <= NF_MC_AFOLD_MAX_BLOCKS && !nf_det_on(nf_mcd_det);
    const float* head_tot = nullptr;
    if (defer) {
        __syncthreads();                                  // red rows of the head product
        if (threadIdx.x < 64) {
            float mine = 0.f;
#pragma unroll
            for (int w = 0; w < NF_MC_WAVES; ++w) mine += sm[NF_MC_RED + w * 64 + threadIdx.x];
            head_rec[(size_t)blockIdx.x * 64 + threadIdx.x] = mine;
        }
    } else if (afold) {
        __syncthreads();                                  // red rows of the head product / the tiles are complete
        if (GLOW && threadIdx.x < 64) {
            float mine = 0.f;
#pragma unroll
            for (int w = 0; w < NF_MC_WAVES; ++w) mine += sm[NF_MC_RED + w * 64 + threadIdx.x];
            sm[NF_MC_TOT + threadIdx.x] = mine;           // this workgroup's part of the head sums
        }
        head_tot = sm + NF_MC_TOT;
        __threadfence_block();                            // this workgroup's slab, written by all its waves
        __syncthreads();
    } else if (GLOW) {   // one more exchange carries the head sums AND, fenced on both sides, is the grid barrier in front of the fold
        __syncthreads();
        if (threadIdx.x == 0) __threadfence();            // release: the slabs of every wave (cumulative through the barrier)
        nf_mc_publish(sm, slots, NF_MC_NB, (unsigned)(NF_MC_NB + 1));
        head_tot = nf_mc_collect(sm, NF_MC_GATHER(3), slots, NF_MC_NB, (unsigned)(NF_MC_NB + 1));
        if (threadIdx.x == 0) __threadfence();            // acquire
        __syncthreads();
    } else {
        nf_grid_barrier(counter, gridDim.x);
    }
    NF_MC_T(73);
    // unit = one column (l, i) of a weight matrix or one bias vector, owned by a half wave (lane = output index o): sums the
    // partials (coalesced), reduces <g_Weff, v> and ||v||^2 over o by shuffles, applies the weight-norm backward.  No LDS.
    if (!defer) {
        const int G_ = afold ? 1 : (int)gridDim.x, blk = afold ? 0 : (int)blockIdx.x;
        const float* sl_base = afold ? slabs + (size_t)blockIdx.x * NF_MC_SLAB : slabs;
        constexpr int HW = NF_MC_THREADS / 32;                       // half waves per workgroup
        if (G_ <= 2) {   // one or two workgroups: too few half waves for the column scheme -- whole layers through LDS instead,
                         // the workgroup's layers (l = blockIdx, blockIdx + G, ..) side by side: three barriers in all
            float* gW = sm + NF_MC_TILES;                            // [layer slot][1024 (i-major) + 32 bias]
            float* nd = gW + NF_MC_NL * NF_MC_SLAB_Q;                // [layer slot][3][32] per-column weight-norm backward factors
            const int nown = (NF_MC_NL - blk + G_ - 1) / G_;
            for (int e = threadIdx.x; e < nown * NF_MC_SLAB_Q; e += NF_MC_THREADS) {
                const int sl = e / NF_MC_SLAB_Q, ee = e - sl * NF_MC_SLAB_Q, l = blk + sl * G_;
                float t4 = 0.f;
                for (int b = 0; b < G_; ++b)
#pragma unroll
                    for (int q = 0; q < NF_MC_NKQ; ++q) t4 += sl_base[(size_t)b * NF_MC_SLAB + l * NF_MC_SLAB_L + q * NF_MC_SLAB_Q + ee];
                gW[e] = t4;
            }
            __syncthreads();
            if ((int)threadIdx.x < nown * 32) {
                const int sl = threadIdx.x >> 5, i = threadIdx.x & 31, l = blk + sl * G_;
                const float* W = sm + NF_MC_W + l * 32 * NF_FP_ST;
                float n2 = 0.f, dt = 0.f;
#pragma unroll 8
                for (int oo = 0; oo < 32; ++oo) {
                    const float v = W[oo * NF_FP_ST + i];
                    n2 = fmaf(v, v, n2);
                    dt = fmaf(gW[sl * NF_MC_SLAB_Q + i * 32 + oo], v, dt);
                }
                const float nrm = sqrtf(n2), den = nrm + wn_eps, gi = sm[NF_MC_G + l * 32 + i];
                nd[sl * 96 + i] = gi / den;                                          // weight_norm.py:35-41, per column
                nd[sl * 96 + 32 + i] = nrm > 0.f ? dt * gi / (den * den * nrm) : 0.f;
                nd[sl * 96 + 64 + i] = dt / den;
            }
            __syncthreads();
            for (int e = threadIdx.x; afold && e < nown * NF_MC_SLAB_Q; e += NF_MC_THREADS) {
                // atomic fold: consecutive threads -> consecutive addresses of the gradient (an atomic instruction that touches
                // 64 cache lines instead of 4 costs sixteen times as much: the i-major order below it took 22 us per launch)
                const int sl = e / NF_MC_SLAB_Q, ee = e - sl * NF_MC_SLAB_Q, l = blk + sl * G_;
                const int I = l == 0 ? I0 : 32, O = l == NF_MC_NL - 1 ? O_out : 32;
                if (ee >= 1024) {
                    const int oo = ee - 1024;
                    if (oo < O) atomicAdd((float*)(gr.b[l] + oo), gW[e]);
                    continue;
                }
                if (ee >= O * I) continue;
                const int oo = ee / I, i = ee - oo * I;
                const float gv = gW[sl * NF_MC_SLAB_Q + i * 32 + oo] * nd[sl * 96 + i] -
                                 sm[NF_MC_W + l * 32 * NF_FP_ST + oo * NF_FP_ST + i] * nd[sl * 96 + 32 + i];
                atomicAdd((float*)(gr.v[l] + ee), gv);
                if (ee < I) atomicAdd((float*)(gr.g[l] + ee), nd[sl * 96 + 64 + ee]);
            }
            for (int e = threadIdx.x; !afold && e < nown * NF_MC_SLAB_Q; e += NF_MC_THREADS) {
                const int sl = e / NF_MC_SLAB_Q, ee = e - sl * NF_MC_SLAB_Q, l = blk + sl * G_;
                const int I = l == 0 ? I0 : 32, O = l == NF_MC_NL - 1 ? O_out : 32;
                if (ee >= 1024) {                                    // bias
                    const int oo = ee - 1024;
                    if (oo < O) NF_MC_ACC(gr.b[l] + oo, gW[e]);
                    continue;
                }
                const int i = ee >> 5, oo = ee & 31;
                if (i >= I) continue;
                if (oo < O) {
                    const float gv = gW[e] * nd[sl * 96 + i] - sm[NF_MC_W + l * 32 * NF_FP_ST + oo * NF_FP_ST + i] * nd[sl * 96 + 32 + i];
                    NF_MC_ACC(gr.v[l] + oo * I + i, gv);
                }
                if (oo == 0) NF_MC_ACC(gr.g[l] + i, nd[sl * 96 + 64 + i]);
            }
        }
        if (G_ > 2) nf_mc_fold_units(sm, slabs, G_, (int)blockIdx.x, G_, gr, accumulate, I0, O_out, wn_eps);
    }
    if (!defer && GLOW && (afold || blockIdx.x == gridDim.x - 1) && threadIdx.x == NF_MC_THREADS - 1)
        nf_mc_head_grads<HEAD>(sm, h, head_tot, accumulate, afold);
    if (blockIdx.x == 0 && threadIdx.x < NF_MC_NB * 32) {
        const int j = threadIdx.x >> 5, k = threadIdx.x & 31;
        gr.beta[j][k] = (accumulate ? gr.beta[j][k] : 0.f) + sm[NF_MC_GB + j * 64 + k];
        gr.gamma[j][k] = (accumulate ? gr.gamma[j][k] : 0.f) + sm[NF_MC_GB + j * 64 + 32 + k];
    }
    NF_MC_T(74);
}

template <int HEAD>
__global__ void __launch_bounds__(NF_MC_THREADS) k_mlp_chain_bwd(const float* __restrict__ x, NfMlpP p, const float* __restrict__ save,
                                                                 const float* __restrict__ g_out, float* __restrict__ g_x, NfMlpG gr,
                                                                 int accumulate, float* ws, float* __restrict__ slabs, int64_t N,
                                                                 int I0, int O_out, int training, float eps, float wn_eps, NfGlowV h,
                                                                 float* head_rec) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    nf_mc_bwd_body<HEAD>(sm, x, p, save, g_out, g_x, gr, accumulate, ws, slabs, N, I0, O_out, training, eps, wn_eps, h, h.z, h.g_y,
                         h.g_ld, h.g_z, nullptr, head_rec);
}

extern "C" int nf_mlp_chain_bwd(const float* x, const void* const* params, const float* save_stats, const float* g_out, float* g_x,
                                void* const* grads, int accumulate, float* ws_zero, float* slabs, int64_t N, int I0, int O_out,
                                int training, float bn_eps, float wn_eps, nf_stream_t stream) {
    if (params == nullptr || grads == nullptr || ws_zero == nullptr || slabs == nullptr || I0 < 1 || I0 > 32 || O_out < 1 ||
        O_out > 32 || N > NF_MLP_MAX_ROWS)
        return NF_E_BADARG;
    if (N <= 0) return N == 0 ? 0 : NF_E_BADARG;
    NfMlpP p;
    nf_mlp_unpack(params, p);
    NfMlpG g;
    for (int l = 0; l < NF_MC_NL; ++l) { NF_GSET(g.v[l], grads[3 * l]); NF_GSET(g.g[l], grads[3 * l + 1]); NF_GSET(g.b[l], grads[3 * l + 2]); }
    for (int j = 0; j < NF_MC_NB; ++j) { NF_GSET(g.gamma[j], grads[3 * NF_MC_NL + 2 * j]); NF_GSET(g.beta[j], grads[3 * NF_MC_NL + 2 * j + 1]); }
    const unsigned grid = (unsigned)((N + NF_MLP_ROWS_PER_BLOCK - 1) / NF_MLP_ROWS_PER_BLOCK);
    const size_t lds = nf_mc_lds_bytes(3);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_mlp_chain_bwd<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_mlp_chain_bwd<0>, dim3(grid), dim3(NF_MC_THREADS), lds, (hipStream_t)stream, x, p, save_stats, g_out,
                       g_x, g, accumulate, ws_zero, slabs, N, I0, O_out, training, bn_eps, wn_eps, NfGlowV{}, (float*)nullptr);
    NF_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// the fused vector Glow step: ActNorm -> invertible 1x1 -> affine coupling (MLP conditioner) in one launch per direction
// ---------------------------------------------------------------------------------------------------------------
static int nf_glow_args_ok(int64_t N, int D) { return (D == 2 || D == 4) && N <= NF_MLP_MAX_ROWS; }

extern "C" int nf_glow_step_vec_fwd(const float* z, float* y, float* ld, const void* const* head, const void* const* mlp_params,
                                    float* save_stats, float* ws_zero, int64_t N, int D, int odd, int training, float bn_eps,
                                    float bn_momentum, float wn_eps, nf_stream_t stream) {
    if (z == nullptr || y == nullptr || ld == nullptr || head == nullptr || mlp_params == nullptr || !nf_glow_args_ok(N, D))
        return NF_E_BADARG;
    if (N <= 0) return N == 0 ? 0 : NF_E_BADARG;
    NfMlpP p;
    nf_mlp_unpack(mlp_params, p);
    NfGlowV h{};
    nf_glow_unpack(head, h);
    h.z = z; h.y = y; h.ld = ld; h.D = D; h.odd = odd ? 1 : 0;
    const unsigned grid = (unsigned)((N + NF_MLP_ROWS_PER_BLOCK - 1) / NF_MLP_ROWS_PER_BLOCK);
    const size_t lds = nf_mc_lds_bytes(1);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_mlp_chain_fwd<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_mlp_chain_fwd<1>, dim3(grid), dim3(NF_MC_THREADS), lds, (hipStream_t)stream, (const float*)nullptr, p,
                       (float*)nullptr, save_stats, ws_zero, N, D / 2, D, training, bn_eps, bn_momentum, wn_eps, h);
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_glow_step_vec_inv(const float* y, float* z, float* ld, const void* const* head, const void* const* mlp_params,
                                    float* save_stats, float* ws_zero, int64_t N, int D, int odd, int training, float bn_eps,
                                    float bn_momentum, float wn_eps, nf_stream_t stream) {
    if (y == nullptr || z == nullptr || ld == nullptr || head == nullptr || mlp_params == nullptr || save_stats == nullptr ||
        ws_zero == nullptr || !nf_glow_args_ok(N, D))
        return NF_E_BADARG;
    if (N <= 0) return N == 0 ? 0 : NF_E_BADARG;
    NfMlpP p;
    nf_mlp_unpack(mlp_params, p);
    NfGlowV h{};
    nf_glow_unpack(head, h);
    h.z = y; h.y = z; h.ld = ld; h.D = D; h.odd = odd ? 1 : 0;
    const unsigned grid = (unsigned)((N + NF_MLP_ROWS_PER_BLOCK - 1) / NF_MLP_ROWS_PER_BLOCK);
    const size_t lds = nf_mc_lds_bytes(1);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_mlp_chain_fwd<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL((k_mlp_chain_fwd<1, true>), dim3(grid), dim3(NF_MC_THREADS), lds, (hipStream_t)stream, (const float*)nullptr, p,
                       (float*)nullptr, save_stats, ws_zero, N, D / 2, D, training, bn_eps, bn_momentum, wn_eps, h);
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_glow_step_vec_bwd(const float* z, const float* g_y, const float* g_ld, float* g_z, const void* const* head,
                                    const void* const* mlp_params, const float* save_stats, void* const* head_grads,
                                    void* const* mlp_grads, int accumulate, float* ws_zero, float* slabs, int64_t N, int D, int odd,
                                    int training, float bn_eps, float wn_eps, nf_stream_t stream) {
    if (z == nullptr || g_y == nullptr || g_z == nullptr || head == nullptr || mlp_params == nullptr || head_grads == nullptr ||
        mlp_grads == nullptr || ws_zero == nullptr || slabs == nullptr || !nf_glow_args_ok(N, D))
        return NF_E_BADARG;
    if (N <= 0) return N == 0 ? 0 : NF_E_BADARG;
    NfMlpP p;
    nf_mlp_unpack(mlp_params, p);
    NfMlpG g;
    for (int l = 0; l < NF_MC_NL; ++l) { NF_GSET(g.v[l], mlp_grads[3 * l]); NF_GSET(g.g[l], mlp_grads[3 * l + 1]); NF_GSET(g.b[l], mlp_grads[3 * l + 2]); }
    for (int j = 0; j < NF_MC_NB; ++j) { NF_GSET(g.gamma[j], mlp_grads[3 * NF_MC_NL + 2 * j]); NF_GSET(g.beta[j], mlp_grads[3 * NF_MC_NL + 2 * j + 1]); }
    NfGlowV h{};
    nf_glow_unpack(head, h);
    h.z = z; h.g_y = g_y; h.g_ld = g_ld; h.g_z = g_z; h.D = D; h.odd = odd ? 1 : 0;
    NF_GSET(h.g_ls, head_grads[0]); NF_GSET(h.g_bs, head_grads[1]); NF_GSET(h.g_L, head_grads[2]); NF_GSET(h.g_U, head_grads[3]);
    NF_GSET(h.g_log_s, head_grads[4]); NF_GSET(h.g_a, head_grads[5]); NF_GSET(h.g_c, head_grads[6]);
    const unsigned grid = (unsigned)((N + NF_MLP_ROWS_PER_BLOCK - 1) / NF_MLP_ROWS_PER_BLOCK);
    const size_t lds = nf_mc_lds_bytes(3);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_mlp_chain_bwd<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_mlp_chain_bwd<1>, dim3(grid), dim3(NF_MC_THREADS), lds, (hipStream_t)stream, (const float*)nullptr, p,
                       save_stats, (const float*)nullptr, (float*)nullptr, g, accumulate, ws_zero, slabs, N, D / 2, D, training, bn_eps,
                       wn_eps, h, (float*)nullptr);
    NF_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// A whole flow of S fused vector Glow steps in ONE launch per direction.  Rows never leave their workgroup between steps
// (a flow step is row-local apart from the batch statistics, which the step bodies already exchange grid-wide), so the
// launch boundary between steps buys nothing: it costs ~2.7 us of dispatch / drain per step and direction (64 of them per
// C2 train step).  The step bodies are the single-step kernels' bodies, run back to back on per-step exchange workspaces;
// the backward's weight-gradient slabs alternate between two regions (a workgroup may start writing step s-1 while a
// slower one still folds step s; it cannot get further ahead than that, the first exchange of a step needs everybody).
// ---------------------------------------------------------------------------------------------------------------
static void nf_fbn_unpack(const void* const* t, NfGlowV& h);
static inline int nf_fbn_mode_of(float momentum) { return momentum == NF_FBN_RUNNING ? 1 : momentum == NF_FBN_BATCH_BUFFERS ? 2 : 0; }

extern "C" int nf_glow_flow_step_bytes(void) { return (int)sizeof(NfGlowFlowStep); }

extern "C" int nf_glow_flow_pack(void* dst_host, const void* const* head, const void* const* mlp_params, void* const* head_grads,
                                 void* const* mlp_grads, int D, int odd) {
    if (dst_host == nullptr || head == nullptr || mlp_params == nullptr || (D != 2 && D != 4)) return NF_E_BADARG;
    NfGlowFlowStep st{};
    nf_mlp_unpack(mlp_params, st.p);
    nf_glow_unpack(head, st.h);
    st.h.D = D; st.h.odd = odd ? 1 : 0;
    if (mlp_grads != nullptr) {
        for (int l = 0; l < NF_MC_NL; ++l) {
            NF_GSET(st.g.v[l], mlp_grads[3 * l]); NF_GSET(st.g.g[l], mlp_grads[3 * l + 1]); NF_GSET(st.g.b[l], mlp_grads[3 * l + 2]);
        }
        for (int j = 0; j < NF_MC_NB; ++j) {
            NF_GSET(st.g.gamma[j], mlp_grads[3 * NF_MC_NL + 2 * j]); NF_GSET(st.g.beta[j], mlp_grads[3 * NF_MC_NL + 2 * j + 1]);
        }
    }
    if (head_grads != nullptr) {
        NF_GSET(st.h.g_ls, head_grads[0]); NF_GSET(st.h.g_bs, head_grads[1]); NF_GSET(st.h.g_L, head_grads[2]);
        NF_GSET(st.h.g_U, head_grads[3]); NF_GSET(st.h.g_log_s, head_grads[4]); NF_GSET(st.h.g_a, head_grads[5]);
        NF_GSET(st.h.g_c, head_grads[6]);
    }
    *reinterpret_cast<NfGlowFlowStep*>(dst_host) = st;
    return 0;
}

// The step records live in global memory; reading ~100 pointers from there at the top of every step would put a memory
// round trip in front of each step (the single-step kernels get them in SGPRs from the kernarg segment).  So the record of
// step s+1 is fetched by the first threads while step s runs and parked in LDS (double buffered) for its turn.
#define NF_GF_REC_WORDS ((int)((sizeof(NfGlowFlowStep) + 7) / 8))
static_assert(NF_GF_REC_WORDS <= NF_MC_THREADS, "one 8-byte word of a step record per thread");

template <int HEAD>   // 1: Glow steps, 2: RealNVP steps (flow-BatchNorm head)
__global__ void __launch_bounds__(NF_MC_THREADS) k_glow_flow_fwd(const NfGlowFlowStep* __restrict__ steps, int S, const float* z0,
                                                                 float* ys, float* ld, float* saves, int save_stride, float* ws,
                                                                 int64_t N, int D, int training, float eps, float mom,
                                                                 float wn_eps, int rec_off) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    // the records sit BEHIND the step body's LDS (a static array in front of it would shift every tile of the body)
    unsigned long long (*rec)[NF_GF_REC_WORDS] = reinterpret_cast<unsigned long long (*)[NF_GF_REC_WORDS]>(sm + rec_off);
    const int64_t ND = N * D;
    const bool rt = (int)threadIdx.x < NF_GF_REC_WORDS;
    if (rt) rec[0][threadIdx.x] = reinterpret_cast<const unsigned long long*>(steps)[threadIdx.x];
    __syncthreads();
    NfMcCarry carry;
    carry.have = 0;
#pragma unroll 1
    for (int s = 0; s < S; ++s) {
        NF_MC_T(100);
        unsigned long long nxt = 0;
        if (rt && s + 1 < S) nxt = reinterpret_cast<const unsigned long long*>(steps + s + 1)[threadIdx.x];
        const NfGlowFlowStep& st = *reinterpret_cast<const NfGlowFlowStep*>(rec[s & 1]);
        nf_mc_fwd_body<HEAD>(sm, nullptr, st.p, nullptr, saves + (int64_t)s * save_stride, ws + (int64_t)s * NF_MLP_WS_FLOATS, N,
                          D / 2, D, training, eps, mom, wn_eps, st.h, s == 0 ? z0 : ys + (int64_t)(s - 1) * ND, ys + (int64_t)s * ND, ld,
                          &carry);
        NF_MC_T(101);
        if (rt) rec[(s + 1) & 1][threadIdx.x] = nxt;
        __syncthreads();                                // LDS is restaged; the rows travel in registers (carry)
        NF_MC_T(102);
    }
}

// the INVERSE of the whole run, last step first: rows travel in registers, zs (2, N, D) ping-pongs the steps' results (slice
// s & 1: the flow's input ends up in slice 0)
template <int HEAD>
__global__ void __launch_bounds__(NF_MC_THREADS) k_glow_flow_inv(const NfGlowFlowStep* __restrict__ steps, int S, const float* y,
                                                                 float* zs, float* ld, float* saves, int save_stride, float* ws,
                                                                 int64_t N, int D, int training, float eps, float mom,
                                                                 float wn_eps, int rec_off) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    unsigned long long (*rec)[NF_GF_REC_WORDS] = reinterpret_cast<unsigned long long (*)[NF_GF_REC_WORDS]>(sm + rec_off);
    const int64_t ND = N * D;
    const bool rt = (int)threadIdx.x < NF_GF_REC_WORDS;
    if (rt) rec[(S - 1) & 1][threadIdx.x] = reinterpret_cast<const unsigned long long*>(steps + S - 1)[threadIdx.x];
    __syncthreads();
    NfMcCarry carry;
    carry.have = 0;
#pragma unroll 1
    for (int s = S - 1; s >= 0; --s) {
        unsigned long long nxt = 0;
        if (rt && s > 0) nxt = reinterpret_cast<const unsigned long long*>(steps + s - 1)[threadIdx.x];
        const NfGlowFlowStep& st = *reinterpret_cast<const NfGlowFlowStep*>(rec[s & 1]);
        nf_mc_fwd_body<HEAD, true>(sm, nullptr, st.p, nullptr, saves + (int64_t)s * save_stride, ws + (int64_t)s * NF_MLP_WS_FLOATS, N,
                                D / 2, D, training, eps, mom, wn_eps, st.h, y, zs + (int64_t)(s & 1) * ND, ld, &carry);
        if (rt) rec[(s + 1) & 1][threadIdx.x] = nxt;    // parity of s - 1
        __syncthreads();
    }
}

template <int HEAD>
__global__ void __launch_bounds__(NF_MC_THREADS) k_glow_flow_bwd(const NfGlowFlowStep* __restrict__ steps, int S, const float* z0,
                                                                 const float* ys, const float* g_y, const float* g_ld, float* gzs,
                                                                 const float* saves, int save_stride, int accumulate, float* ws,
                                                                 float* slabs, int64_t N, int D, int training, float eps,
                                                                 float wn_eps, int rec_off, float* head_rec) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    unsigned long long (*rec)[NF_GF_REC_WORDS] = reinterpret_cast<unsigned long long (*)[NF_GF_REC_WORDS]>(sm + rec_off);
    const int64_t ND = N * D;
    const bool rt = (int)threadIdx.x < NF_GF_REC_WORDS;
    // head_rec != nullptr: deferred fold -- `slabs` then holds a region per step and workgroup (S x grid x NF_MC_SLAB), every step
    // leaves its slab and head sums behind and ONE k_glow_fold_all launch after this kernel turns them into parameter gradients (the
    // in-kernel fold is 8.4 of a step's 34 us at two workgroups: tools/probes/mlp_chain_prof.py)
    // three records in LDS: step s (at work), step s - 1 (whose loads step s requests, NfMcPre) and the slot step s - 2 arrives in
    if (rt) {
        rec[(S - 1) % 3][threadIdx.x] = reinterpret_cast<const unsigned long long*>(steps + S - 1)[threadIdx.x];
        if (S >= 2) rec[(S - 2) % 3][threadIdx.x] = reinterpret_cast<const unsigned long long*>(steps + S - 2)[threadIdx.x];
    }
    __syncthreads();
    NfMcCarry carry;
    carry.have = 0;
    NfMcPre pre;
    pre.have = 0;
    pre.have_gld = 0;
#pragma unroll 1
    for (int s = S - 1; s >= 0; --s) {
        NF_MC_T(104);
        unsigned long long nxt = 0;
        if (rt && s > 1) nxt = reinterpret_cast<const unsigned long long*>(steps + s - 2)[threadIdx.x];
        const NfGlowFlowStep& st = *reinterpret_cast<const NfGlowFlowStep*>(rec[s % 3]);
        NfMcNext nx;
        nx.st = s > 0 ? reinterpret_cast<const NfGlowFlowStep*>(rec[(s + 2) % 3]) : nullptr;      // (s - 1) mod 3
        nx.save = saves + (int64_t)(s > 0 ? s - 1 : 0) * save_stride;
        nx.hz = s > 1 ? ys + (int64_t)(s - 2) * ND : z0;
        nf_mc_bwd_body<HEAD>(sm, nullptr, st.p, saves + (int64_t)s * save_stride, nullptr, nullptr, st.g, accumulate,
                          ws + (int64_t)s * NF_MLP_WS_FLOATS,
                          head_rec != nullptr ? slabs + (size_t)s * gridDim.x * NF_MC_SLAB : slabs + (int64_t)(s & 1) * NF_MLP_BWD_SLAB_FLOATS,
                          N, D / 2, D, training, eps, wn_eps, st.h, s == 0 ? z0 : ys + (int64_t)(s - 1) * ND,
                          s == S - 1 ? g_y : gzs + (int64_t)(s + 1) * ND, g_ld, gzs + (int64_t)s * ND, &carry,
                          head_rec != nullptr ? head_rec + (size_t)s * gridDim.x * 64 : nullptr, &pre, &nx);
        NF_MC_T(105);
        if (rt && s > 1) rec[(s + 1) % 3][threadIdx.x] = nxt;    // step s - 2 takes the slot of step s + 1
        __syncthreads();
        NF_MC_T(106);
    }
}

template <int HEAD>
static int nf_flow_launch_fwd(const void* steps_dev, int S, const float* z0, float* ys, float* ld, float* saves, int save_stride,
                              float* ws_zero, int64_t N, int D, int training, float bn_eps, float bn_momentum, float wn_eps,
                              nf_stream_t stream) {
    if (steps_dev == nullptr || S < 1 || S > NF_GLOW_FLOW_MAX_STEPS || z0 == nullptr || ys == nullptr || ld == nullptr ||
        saves == nullptr || ws_zero == nullptr || !nf_glow_args_ok(N, D))
        return NF_E_BADARG;
    if (N <= 0) return N == 0 ? 0 : NF_E_BADARG;
    const unsigned grid = (unsigned)((N + NF_MLP_ROWS_PER_BLOCK - 1) / NF_MLP_ROWS_PER_BLOCK);
    const size_t body_lds = nf_mc_lds_bytes(1), lds = body_lds + 2 * NF_GF_REC_WORDS * 8;
    static bool attr_set = false;                       // one per HEAD
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_glow_flow_fwd<HEAD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_glow_flow_fwd<HEAD>, dim3(grid), dim3(NF_MC_THREADS), lds, (hipStream_t)stream,
                       (const NfGlowFlowStep*)steps_dev, S, z0, ys, ld, saves, save_stride, ws_zero, N, D, training, bn_eps,
                       bn_momentum, wn_eps, (int)(body_lds / sizeof(float)));
    NF_CHECK_LAUNCH();
    return 0;
}

template <int HEAD>
static int nf_flow_launch_bwd(const void* steps_dev, int S, const float* z0, const float* ys, const float* g_y, const float* g_ld,
                              float* gzs, const float* saves, int save_stride, int accumulate, float* ws_zero, float* slabs2,
                              int64_t N, int D, int training, float bn_eps, float wn_eps, nf_stream_t stream, float* head_rec = nullptr) {
    if (steps_dev == nullptr || S < 1 || S > NF_GLOW_FLOW_MAX_STEPS || z0 == nullptr || ys == nullptr || g_y == nullptr ||
        gzs == nullptr || saves == nullptr || ws_zero == nullptr || slabs2 == nullptr || !nf_glow_args_ok(N, D))
        return NF_E_BADARG;
    if (N <= 0) return N == 0 ? 0 : NF_E_BADARG;
    const unsigned grid = (unsigned)((N + NF_MLP_ROWS_PER_BLOCK - 1) / NF_MLP_ROWS_PER_BLOCK);
    const size_t body_lds = nf_mc_lds_bytes(3), lds = body_lds + 3 * NF_GF_REC_WORDS * 8;     // (three records: see k_glow_flow_bwd)
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_glow_flow_bwd<HEAD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_glow_flow_bwd<HEAD>, dim3(grid), dim3(NF_MC_THREADS), lds, (hipStream_t)stream,
                       (const NfGlowFlowStep*)steps_dev, S, z0, ys, g_y, g_ld, gzs, saves, save_stride, accumulate, ws_zero, slabs2, N, D,
                       training, bn_eps, wn_eps, (int)(body_lds / sizeof(float)), head_rec);
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_glow_flow_vec_fwd(const void* steps_dev, int S, const float* z0, float* ys, float* ld, float* saves, float* ws_zero,
                                    int64_t N, int D, int training, float bn_eps, float bn_momentum, float wn_eps,
                                    nf_stream_t stream) {
    return nf_flow_launch_fwd<1>(steps_dev, S, z0, ys, ld, saves, NF_GLOW_FLOW_SAVE_FLOATS, ws_zero, N, D, training, bn_eps, bn_momentum,
                                 wn_eps, stream);
}
template <int HEAD>
static int nf_flow_launch_inv(const void* steps_dev, int S, const float* y, float* zs2, float* ld, float* saves, int save_stride,
                              float* ws_zero, int64_t N, int D, int training, float bn_eps, float bn_momentum, float wn_eps,
                              nf_stream_t stream) {
    if (steps_dev == nullptr || S < 1 || S > NF_GLOW_FLOW_MAX_STEPS || y == nullptr || zs2 == nullptr || ld == nullptr ||
        saves == nullptr || ws_zero == nullptr || !nf_glow_args_ok(N, D))
        return NF_E_BADARG;
    if (N <= 0) return N == 0 ? 0 : NF_E_BADARG;
    const unsigned grid = (unsigned)((N + NF_MLP_ROWS_PER_BLOCK - 1) / NF_MLP_ROWS_PER_BLOCK);
    const size_t body_lds = nf_mc_lds_bytes(1), lds = body_lds + 2 * NF_GF_REC_WORDS * 8;
    static bool attr_set = false;                       // one per HEAD
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_glow_flow_inv<HEAD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_glow_flow_inv<HEAD>, dim3(grid), dim3(NF_MC_THREADS), lds, (hipStream_t)stream,
                       (const NfGlowFlowStep*)steps_dev, S, y, zs2, ld, saves, save_stride, ws_zero, N, D, training, bn_eps, bn_momentum,
                       wn_eps, (int)(body_lds / sizeof(float)));
    NF_CHECK_LAUNCH();
    return 0;
}
extern "C" int nf_glow_flow_vec_inv(const void* steps_dev, int S, const float* y, float* zs2, float* ld, float* saves, float* ws_zero,
                                    int64_t N, int D, int training, float bn_eps, float bn_momentum, float wn_eps,
                                    nf_stream_t stream) {
    return nf_flow_launch_inv<1>(steps_dev, S, y, zs2, ld, saves, NF_GLOW_FLOW_SAVE_FLOATS, ws_zero, N, D, training, bn_eps, bn_momentum,
                                 wn_eps, stream);
}
extern "C" int nf_glow_flow_vec_bwd(const void* steps_dev, int S, const float* z0, const float* ys, const float* g_y,
                                    const float* g_ld, float* gzs, const float* saves, int accumulate, float* ws_zero, float* slabs2,
                                    int64_t N, int D, int training, float bn_eps, float wn_eps, nf_stream_t stream) {
    return nf_flow_launch_bwd<1>(steps_dev, S, z0, ys, g_y, g_ld, gzs, saves, NF_GLOW_FLOW_SAVE_FLOATS, accumulate, ws_zero, slabs2, N, D,
                                 training, bn_eps, wn_eps, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// The same run of Glow steps as S launches per direction + ONE fold launch, for batches where the whole-flow kernel's
// in-kernel exchanges lose against the launch gaps (32 workgroups and up): the single-step kernels with the records of
// nf_glow_flow_pack as kernel arguments (host copy of the table), the backward in deferred-fold mode -- every step leaves
// its weight-gradient slabs (own region per step) and its workgroups' head sums behind, k_glow_fold_all turns all of
// them into parameter gradients at once: grid (NF_GF_FOLD_BLOCKS, S), one column / bias vector per half wave.
// ---------------------------------------------------------------------------------------------------------------
#define NF_GF_FOLD_BLOCKS ((NF_MC_NL * 33 + NF_MC_THREADS / 32 - 1) / (NF_MC_THREADS / 32))

template <int HEAD>
__global__ void __launch_bounds__(NF_MC_THREADS) k_glow_fold_all(const NfGlowFlowStep* __restrict__ steps, const float* __restrict__ slabs,
                                                                 const float* __restrict__ head_rec, int G, int accumulate, int D,
                                                                 float wn_eps) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int s = blockIdx.y;
    const NfGlowFlowStep& st = steps[s];                 // block-uniform: scalar loads
    const bool last = blockIdx.x == gridDim.x - 1;       // the head gradients' workgroup
    NfGlowRaw raw;
    if (HEAD == 1) nf_glow_head_load(st.h, raw);
    float hsum = 0.f;
    if (last && threadIdx.x < 64) {
        const float* r = head_rec + (size_t)s * G * 64 + threadIdx.x;
        for (int b = 0; b < G; ++b) hsum += r[(size_t)b * 64];
    }
    nf_mc_stage(st.p, sm, D / 2, D, wn_eps, HEAD == 1 ? &st.h : nullptr, HEAD == 1 ? &raw : nullptr);
    if (last && threadIdx.x < 64) sm[NF_MC_TOT + threadIdx.x] = hsum;
    __syncthreads();
    nf_mc_fold_units(sm, slabs + (size_t)s * G * NF_MC_SLAB, G, (int)blockIdx.x, (int)gridDim.x, st.g, accumulate, D / 2, D, wn_eps);
    const bool afold = false;
    (void)afold;
    if (last && threadIdx.x == NF_MC_THREADS - 1) nf_mc_head_grads<HEAD>(sm, st.h, sm + NF_MC_TOT, accumulate, false);
}

template <int HEAD>
static int nf_flow_steps_fwd(const void* steps_host, int S, const float* z0, float* ys, float* ld, float* saves, int save_stride,
                             float* ws_zero, int64_t N, int D, int training, float bn_eps, float bn_momentum, float wn_eps,
                             nf_stream_t stream) {
    if (steps_host == nullptr || S < 1 || S > NF_GLOW_FLOW_MAX_STEPS || z0 == nullptr || ys == nullptr || ld == nullptr ||
        saves == nullptr || ws_zero == nullptr || !nf_glow_args_ok(N, D))
        return NF_E_BADARG;
    if (N <= 0) return N == 0 ? 0 : NF_E_BADARG;
    const unsigned grid = (unsigned)((N + NF_MLP_ROWS_PER_BLOCK - 1) / NF_MLP_ROWS_PER_BLOCK);
    const size_t lds = nf_mc_lds_bytes(1);
    hipError_t e = hipFuncSetAttribute((const void*)k_mlp_chain_fwd<HEAD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    const NfGlowFlowStep* st = reinterpret_cast<const NfGlowFlowStep*>(steps_host);
    const int64_t ND = N * D;
    for (int s = 0; s < S; ++s) {
        NfGlowV h = st[s].h;
        h.z = s == 0 ? z0 : ys + (int64_t)(s - 1) * ND; h.y = ys + (int64_t)s * ND; h.ld = ld;
        hipLaunchKernelGGL(k_mlp_chain_fwd<HEAD>, dim3(grid), dim3(NF_MC_THREADS), lds, (hipStream_t)stream, (const float*)nullptr, st[s].p,
                           (float*)nullptr, saves + (int64_t)s * save_stride, ws_zero + (int64_t)s * NF_MLP_WS_FLOATS, N, D / 2,
                           D, training, bn_eps, bn_momentum, wn_eps, h);
    }
    NF_CHECK_LAUNCH();
    return 0;
}

template <int HEAD>
static int nf_flow_steps_bwd(const void* steps_host, const void* steps_dev, int S, const float* z0, const float* ys,
                             const float* g_y, const float* g_ld, float* gzs, const float* saves, int save_stride, int accumulate,
                             float* ws_zero, float* slabs_all, float* head_rec, int64_t N, int D, int training, float bn_eps,
                             float wn_eps, nf_stream_t stream) {
    if (steps_host == nullptr || steps_dev == nullptr || S < 1 || S > NF_GLOW_FLOW_MAX_STEPS || z0 == nullptr || ys == nullptr ||
        g_y == nullptr || gzs == nullptr || saves == nullptr || ws_zero == nullptr || slabs_all == nullptr || head_rec == nullptr ||
        !nf_glow_args_ok(N, D))
        return NF_E_BADARG;
    if (N <= 0) return N == 0 ? 0 : NF_E_BADARG;
    const unsigned grid = (unsigned)((N + NF_MLP_ROWS_PER_BLOCK - 1) / NF_MLP_ROWS_PER_BLOCK);
    const size_t lds = nf_mc_lds_bytes(3), lds_fold = nf_mc_lds_bytes(1);
    hipError_t e = hipFuncSetAttribute((const void*)k_mlp_chain_bwd<HEAD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)k_glow_fold_all<HEAD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fold);
    if (e != hipSuccess) return (int)e;
    const NfGlowFlowStep* st = reinterpret_cast<const NfGlowFlowStep*>(steps_host);
    const int64_t ND = N * D;
    for (int s = S - 1; s >= 0; --s) {
        NfGlowV h = st[s].h;
        h.z = s == 0 ? z0 : ys + (int64_t)(s - 1) * ND;
        h.g_y = s == S - 1 ? g_y : gzs + (int64_t)(s + 1) * ND;
        h.g_ld = g_ld;
        h.g_z = gzs + (int64_t)s * ND;
        hipLaunchKernelGGL(k_mlp_chain_bwd<HEAD>, dim3(grid), dim3(NF_MC_THREADS), lds, (hipStream_t)stream, (const float*)nullptr, st[s].p,
                           saves + (int64_t)s * save_stride, (const float*)nullptr, (float*)nullptr, st[s].g, accumulate,
                           ws_zero + (int64_t)s * NF_MLP_WS_FLOATS, slabs_all + (size_t)s * grid * NF_MC_SLAB, N, D / 2, D, training,
                           bn_eps, wn_eps, h, head_rec + (size_t)s * grid * 64);
    }
    hipLaunchKernelGGL(k_glow_fold_all<HEAD>, dim3(NF_GF_FOLD_BLOCKS, S), dim3(NF_MC_THREADS), lds_fold, (hipStream_t)stream,
                       (const NfGlowFlowStep*)steps_dev, slabs_all, head_rec, (int)grid, accumulate, D, wn_eps);
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_glow_flow_steps_fwd(const void* steps_host, int S, const float* z0, float* ys, float* ld, float* saves,
                                      float* ws_zero, int64_t N, int D, int training, float bn_eps, float bn_momentum, float wn_eps,
                                      nf_stream_t stream) {
    return nf_flow_steps_fwd<1>(steps_host, S, z0, ys, ld, saves, NF_GLOW_FLOW_SAVE_FLOATS, ws_zero, N, D, training, bn_eps, bn_momentum,
                                wn_eps, stream);
}
extern "C" int nf_glow_flow_steps_bwd(const void* steps_host, const void* steps_dev, int S, const float* z0, const float* ys,
                                      const float* g_y, const float* g_ld, float* gzs, const float* saves, int accumulate,
                                      float* ws_zero, float* slabs_all, float* head_rec, int64_t N, int D, int training, float bn_eps,
                                      float wn_eps, nf_stream_t stream) {
    return nf_flow_steps_bwd<1>(steps_host, steps_dev, S, z0, ys, g_y, g_ld, gzs, saves, NF_GLOW_FLOW_SAVE_FLOATS, accumulate, ws_zero,
                                slabs_all, head_rec, N, D, training, bn_eps, wn_eps, stream);
}
// the RealNVP steps likewise (training mode; records of nf_realnvp_flow_pack, saves = S x NF_REALNVP_SAVE_FLOATS)
extern "C" int nf_realnvp_flow_steps_fwd(const void* steps_host, int S, const float* z0, float* ys, float* ld, float* saves,
                                         float* ws_zero, int64_t N, int D, float bn_eps, float bn_momentum, float wn_eps,
                                         nf_stream_t stream) {
    return nf_flow_steps_fwd<2>(steps_host, S, z0, ys, ld, saves, NF_REALNVP_SAVE_FLOATS, ws_zero, N, D, 1, bn_eps, bn_momentum, wn_eps,
                                stream);
}
extern "C" int nf_realnvp_flow_steps_bwd(const void* steps_host, const void* steps_dev, int S, const float* z0, const float* ys,
                                         const float* g_y, const float* g_ld, float* gzs, const float* saves, int accumulate,
                                         float* ws_zero, float* slabs_all, float* head_rec, int64_t N, int D, float bn_eps,
                                         float wn_eps, nf_stream_t stream) {
    return nf_flow_steps_bwd<2>(steps_host, steps_dev, S, z0, ys, g_y, g_ld, gzs, saves, NF_REALNVP_SAVE_FLOATS, accumulate, ws_zero,
                                slabs_all, head_rec, N, D, 1, bn_eps, wn_eps, stream);
}

// whole-flow backward with the DEFERRED fold: one launch for the data gradients of all S steps (slabs_all: a region per step and
// workgroup, head_rec: the steps' head sums), one k_glow_fold_all launch behind it
template <int HEAD>
static int nf_flow_launch_bwd_deferred(const void* steps_dev, int S, const float* z0, const float* ys, const float* g_y,
                                       const float* g_ld, float* gzs, const float* saves, int save_stride, int accumulate,
                                       float* ws_zero, float* slabs_all, float* head_rec, int64_t N, int D, int training, float bn_eps,
                                       float wn_eps, nf_stream_t stream) {
    if (head_rec == nullptr) return NF_E_BADARG;
    const int rc = nf_flow_launch_bwd<HEAD>(steps_dev, S, z0, ys, g_y, g_ld, gzs, saves, save_stride, accumulate, ws_zero, slabs_all, N, D,
                                            training, bn_eps, wn_eps, stream, head_rec);
    if (rc != 0 || N <= 0) return rc;
    const unsigned grid = (unsigned)((N + NF_MLP_ROWS_PER_BLOCK - 1) / NF_MLP_ROWS_PER_BLOCK);
    const size_t lds_fold = nf_mc_lds_bytes(1);
    hipError_t e = hipFuncSetAttribute((const void*)k_glow_fold_all<HEAD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fold);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_glow_fold_all<HEAD>, dim3(NF_GF_FOLD_BLOCKS, S), dim3(NF_MC_THREADS), lds_fold, (hipStream_t)stream,
                       (const NfGlowFlowStep*)steps_dev, slabs_all, head_rec, (int)grid, accumulate, D, wn_eps);
    NF_CHECK_LAUNCH();
    return 0;
}
extern "C" int nf_glow_flow_vec_bwd_deferred(const void* steps_dev, int S, const float* z0, const float* ys, const float* g_y,
                                             const float* g_ld, float* gzs, const float* saves, int accumulate, float* ws_zero,
                                             float* slabs_all, float* head_rec, int64_t N, int D, int training, float bn_eps,
                                             float wn_eps, nf_stream_t stream) {
    return nf_flow_launch_bwd_deferred<1>(steps_dev, S, z0, ys, g_y, g_ld, gzs, saves, NF_GLOW_FLOW_SAVE_FLOATS, accumulate, ws_zero,
                                          slabs_all, head_rec, N, D, training, bn_eps, wn_eps, stream);
}
extern "C" int nf_realnvp_flow_vec_bwd_deferred(const void* steps_dev, int S, const float* z0, const float* ys, const float* g_y,
                                                const float* g_ld, float* gzs, const float* saves, int accumulate, float* ws_zero,
                                                float* slabs_all, float* head_rec, int64_t N, int D, float bn_eps, float wn_eps,
                                                nf_stream_t stream) {
    if (nf_solo_plan(N, D, 1) && nf_solo_bwd_steps_ok(S) && steps_dev != nullptr && z0 != nullptr && ys != nullptr &&
        g_y != nullptr && gzs != nullptr && saves != nullptr && ws_zero != nullptr && slabs_all != nullptr && head_rec != nullptr) {
        // the whole batch in one workgroup + one for the weight gradients (flow_solo.hip); their slabs and head sums are folded by the
        // same launch as the grid kernel's
        const int rc = nf_solo_bwd(steps_dev, S, z0, ys, g_y, g_ld, gzs, saves, NF_REALNVP_SAVE_FLOATS, accumulate, ws_zero, slabs_all,
                                   head_rec, N, wn_eps, (hipStream_t)stream);
        if (rc != 0) return rc;
        const size_t lds_fold = nf_mc_lds_bytes(1);
        hipError_t e = hipFuncSetAttribute((const void*)k_glow_fold_all<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fold);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(k_glow_fold_all<2>, dim3(NF_GF_FOLD_BLOCKS, S), dim3(NF_MC_THREADS), lds_fold, (hipStream_t)stream,
                           (const NfGlowFlowStep*)steps_dev, slabs_all, head_rec, NF_FLOW_SOLO_REGIONS, accumulate, D, wn_eps);
        NF_CHECK_LAUNCH();
        return 0;
    }
    return nf_flow_launch_bwd_deferred<2>(steps_dev, S, z0, ys, g_y, g_ld, gzs, saves, NF_REALNVP_SAVE_FLOATS, accumulate, ws_zero,
                                          slabs_all, head_rec, N, D, 1, bn_eps, wn_eps, stream);
}

// the same for a run of RealNVP steps [flow BatchNorm (batch statistics), AffineCoupling]: records packed by nf_realnvp_flow_pack
extern "C" int nf_realnvp_flow_pack(void* dst_host, const void* const* head, const void* const* mlp_params, float* g_s_log_scale,
                                    float* g_s_bias, void* const* mlp_grads, int D, int odd, float flow_bn_eps,
                                    float flow_bn_momentum) {
    if (dst_host == nullptr || head == nullptr || mlp_params == nullptr || (D != 2 && D != 4)) return NF_E_BADARG;
    NfGlowFlowStep st{};
    nf_mlp_unpack(mlp_params, st.p);
    nf_fbn_unpack(head, st.h);
    st.h.D = D; st.h.odd = odd ? 1 : 0; st.h.fbn_eps = flow_bn_eps; st.h.fbn_mom = flow_bn_momentum;
    st.h.fbn_mode = nf_fbn_mode_of(flow_bn_momentum);
    if (mlp_grads != nullptr) {
        for (int l = 0; l < NF_MC_NL; ++l) {
            NF_GSET(st.g.v[l], mlp_grads[3 * l]); NF_GSET(st.g.g[l], mlp_grads[3 * l + 1]); NF_GSET(st.g.b[l], mlp_grads[3 * l + 2]);
        }
        for (int j = 0; j < NF_MC_NB; ++j) {
            NF_GSET(st.g.gamma[j], mlp_grads[3 * NF_MC_NL + 2 * j]); NF_GSET(st.g.beta[j], mlp_grads[3 * NF_MC_NL + 2 * j + 1]);
        }
    }
    NF_GSET(st.h.g_a, g_s_log_scale); NF_GSET(st.h.g_c, g_s_bias);
    *reinterpret_cast<NfGlowFlowStep*>(dst_host) = st;
    return 0;
}
// evaluation mode (records packed with flow_bn_momentum = NF_FBN_RUNNING: running statistics everywhere, no exchange) and the inverse
// (NF_FBN_RUNNING, or NF_FBN_BATCH_BUFFERS for the training-mode inverse of modules.py:309-322)
extern "C" int nf_realnvp_flow_vec_fwd_eval(const void* steps_dev, int S, const float* z0, float* ys, float* ld, float* saves,
                                            float* ws_zero, int64_t N, int D, float bn_eps, float wn_eps, nf_stream_t stream) {
    return nf_flow_launch_fwd<2>(steps_dev, S, z0, ys, ld, saves, NF_REALNVP_SAVE_FLOATS, ws_zero, N, D, 0, bn_eps, 0.f, wn_eps, stream);
}
extern "C" int nf_realnvp_flow_vec_inv(const void* steps_dev, int S, const float* y, float* zs2, float* ld, float* saves,
                                       float* ws_zero, int64_t N, int D, int training, float bn_eps, float bn_momentum, float wn_eps,
                                       nf_stream_t stream) {
    return nf_flow_launch_inv<2>(steps_dev, S, y, zs2, ld, saves, NF_REALNVP_SAVE_FLOATS, ws_zero, N, D, training, bn_eps, bn_momentum,
                                 wn_eps, stream);
}
extern "C" int nf_realnvp_flow_vec_fwd(const void* steps_dev, int S, const float* z0, float* ys, float* ld, float* saves,
                                       float* ws_zero, int64_t N, int D, float bn_eps, float bn_momentum, float wn_eps,
                                       nf_stream_t stream) {
    if (nf_solo_plan(N, D, 0) && steps_dev != nullptr && S >= 1 && S <= NF_GLOW_FLOW_MAX_STEPS && z0 != nullptr && ys != nullptr &&
        ld != nullptr && saves != nullptr)
        return nf_solo_fwd(steps_dev, S, z0, ys, ld, saves, NF_REALNVP_SAVE_FLOATS, N, bn_eps, bn_momentum, wn_eps, (hipStream_t)stream);
    return nf_flow_launch_fwd<2>(steps_dev, S, z0, ys, ld, saves, NF_REALNVP_SAVE_FLOATS, ws_zero, N, D, 1, bn_eps, bn_momentum, wn_eps,
                                 stream);
}
extern "C" int nf_realnvp_flow_vec_bwd(const void* steps_dev, int S, const float* z0, const float* ys, const float* g_y,
                                       const float* g_ld, float* gzs, const float* saves, int accumulate, float* ws_zero,
                                       float* slabs2, int64_t N, int D, float bn_eps, float wn_eps, nf_stream_t stream) {
    return nf_flow_launch_bwd<2>(steps_dev, S, z0, ys, g_y, g_ld, gzs, saves, NF_REALNVP_SAVE_FLOATS, accumulate, ws_zero, slabs2, N, D,
                                 1, bn_eps, wn_eps, stream);
}

// number of bounded spin loops that gave up since the library was loaded (0 unless a persistent grid was not co-resident:
// results of such a launch are garbage).  Synchronises the device.
// ---------------------------------------------------------------------------------------------------------------
// the fused vector RealNVP step: flow BatchNorm (batch statistics) -> affine coupling (MLP conditioner), HEAD == 2
// ---------------------------------------------------------------------------------------------------------------
static void nf_fbn_unpack(const void* const* t, NfGlowV& h) {
    NF_GSET(h.ls, t[0]); NF_GSET(h.bs, t[1]); NF_GSET(h.bmean, t[2]); NF_GSET(h.bvar, t[3]);
    NF_GSET(h.rmean, t[4]); NF_GSET(h.rvar, t[5]); NF_GSET(h.a, t[6]); NF_GSET(h.c, t[7]);
}

extern "C" int nf_realnvp_step_vec_fwd(const float* z, float* y, float* ld, const void* const* head, const void* const* mlp_params,
                                       float* save_stats, float* ws_zero, int64_t N, int D, int odd, float flow_bn_eps,
                                       float flow_bn_momentum, float bn_eps, float bn_momentum, float wn_eps, nf_stream_t stream) {
    if (z == nullptr || y == nullptr || ld == nullptr || head == nullptr || mlp_params == nullptr || !nf_glow_args_ok(N, D))
        return NF_E_BADARG;
    if (N <= 0) return N == 0 ? 0 : NF_E_BADARG;
    NfMlpP p;
    nf_mlp_unpack(mlp_params, p);
    NfGlowV h{};
    nf_fbn_unpack(head, h);
    h.z = z; h.y = y; h.ld = ld; h.D = D; h.odd = odd ? 1 : 0; h.fbn_eps = flow_bn_eps; h.fbn_mom = flow_bn_momentum;
    h.fbn_mode = nf_fbn_mode_of(flow_bn_momentum);
    if (h.fbn_mode == 2) return NF_E_BADARG;           // the batch buffers serve the inverse only
    const int cond_training = h.fbn_mode == 0 ? 1 : 0;
    const unsigned grid = (unsigned)((N + NF_MLP_ROWS_PER_BLOCK - 1) / NF_MLP_ROWS_PER_BLOCK);
    const size_t lds = nf_mc_lds_bytes(1);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_mlp_chain_fwd<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_mlp_chain_fwd<2>, dim3(grid), dim3(NF_MC_THREADS), lds, (hipStream_t)stream, (const float*)nullptr, p,
                       (float*)nullptr, save_stats, ws_zero, N, D / 2, D, cond_training, bn_eps, bn_momentum, wn_eps, h);
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_realnvp_step_vec_inv(const float* y, float* z, float* ld, const void* const* head, const void* const* mlp_params,
                                       float* save_stats, float* ws_zero, int64_t N, int D, int odd, int training, float bn_eps,
                                       float bn_momentum, float wn_eps, nf_stream_t stream) {
    if (y == nullptr || z == nullptr || ld == nullptr || head == nullptr || mlp_params == nullptr || save_stats == nullptr ||
        ws_zero == nullptr || !nf_glow_args_ok(N, D))
        return NF_E_BADARG;
    if (N <= 0) return N == 0 ? 0 : NF_E_BADARG;
    NfMlpP p;
    nf_mlp_unpack(mlp_params, p);
    NfGlowV h{};
    nf_fbn_unpack(head, h);
    h.z = y; h.y = z; h.ld = ld; h.D = D; h.odd = odd ? 1 : 0;
    h.fbn_mode = training ? 2 : 1;                     // modules.py:309-322: batch buffers in training mode, running statistics else
    const unsigned grid = (unsigned)((N + NF_MLP_ROWS_PER_BLOCK - 1) / NF_MLP_ROWS_PER_BLOCK);
    const size_t lds = nf_mc_lds_bytes(1);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_mlp_chain_fwd<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL((k_mlp_chain_fwd<2, true>), dim3(grid), dim3(NF_MC_THREADS), lds, (hipStream_t)stream, (const float*)nullptr, p,
                       (float*)nullptr, save_stats, ws_zero, N, D / 2, D, training ? 1 : 0, bn_eps, bn_momentum, wn_eps, h);
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_realnvp_step_vec_bwd(const float* z, const float* g_y, const float* g_ld, float* g_z, const void* const* head,
                                       const void* const* mlp_params, const float* save_stats, float* g_s_log_scale,
                                       float* g_s_bias, void* const* mlp_grads, int accumulate, float* ws_zero, float* slabs,
                                       int64_t N, int D, int odd, float bn_eps, float wn_eps, nf_stream_t stream) {
    if (z == nullptr || g_y == nullptr || g_z == nullptr || head == nullptr || mlp_params == nullptr || g_s_log_scale == nullptr ||
        g_s_bias == nullptr || mlp_grads == nullptr || ws_zero == nullptr || slabs == nullptr || !nf_glow_args_ok(N, D))
        return NF_E_BADARG;
    if (N <= 0) return N == 0 ? 0 : NF_E_BADARG;
    NfMlpP p;
    nf_mlp_unpack(mlp_params, p);
    NfMlpG g;
    for (int l = 0; l < NF_MC_NL; ++l) { NF_GSET(g.v[l], mlp_grads[3 * l]); NF_GSET(g.g[l], mlp_grads[3 * l + 1]); NF_GSET(g.b[l], mlp_grads[3 * l + 2]); }
    for (int j = 0; j < NF_MC_NB; ++j) { NF_GSET(g.gamma[j], mlp_grads[3 * NF_MC_NL + 2 * j]); NF_GSET(g.beta[j], mlp_grads[3 * NF_MC_NL + 2 * j + 1]); }
    NfGlowV h{};
    nf_fbn_unpack(head, h);
    h.z = z; h.g_y = g_y; h.g_ld = g_ld; h.g_z = g_z; h.D = D; h.odd = odd ? 1 : 0; NF_GSET(h.g_a, g_s_log_scale); NF_GSET(h.g_c, g_s_bias);
    const unsigned grid = (unsigned)((N + NF_MLP_ROWS_PER_BLOCK - 1) / NF_MLP_ROWS_PER_BLOCK);
    const size_t lds = nf_mc_lds_bytes(3);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_mlp_chain_bwd<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_mlp_chain_bwd<2>, dim3(grid), dim3(NF_MC_THREADS), lds, (hipStream_t)stream, (const float*)nullptr, p,
                       save_stats, (const float*)nullptr, (float*)nullptr, g, accumulate, ws_zero, slabs, N, D / 2, D, 1, bn_eps,
                       wn_eps, h, (float*)nullptr);
    NF_CHECK_LAUNCH();
    return 0;
}

NF_PERSIST_HOST_API(nf_mc)
__attribute__((visibility("hidden"))) int nf_md_persist_read(unsigned* v);                                 // made_chain.hip
__attribute__((visibility("hidden"))) int nf_md_persist_set(unsigned limit, unsigned* flag_dev, int reset);
__attribute__((visibility("hidden"))) int nf_cc_persist_read(unsigned* v);                                 // conv_chain.hip
__attribute__((visibility("hidden"))) int nf_cc_persist_set(unsigned limit, unsigned* flag_dev, int reset);
__attribute__((visibility("hidden"))) int nf_so_persist_read(unsigned* v);                                 // flow_solo.hip
__attribute__((visibility("hidden"))) int nf_so_persist_set(unsigned limit, unsigned* flag_dev, int reset);
__attribute__((visibility("hidden"))) int nf_fbh_persist_read(unsigned* v);                                // flowbn_head.hip
__attribute__((visibility("hidden"))) int nf_fbh_persist_set(unsigned limit, unsigned* flag_dev, int reset);

extern "C" int nf_persistent_timeouts(int* count) {
    if (count == nullptr) return NF_E_BADARG;
    unsigned v = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0;
    int e = nf_mc_persist_read(&v);
    if (e == 0) e = nf_md_persist_read(&v2);
    if (e == 0) e = nf_cc_persist_read(&v3);
    if (e == 0) e = nf_so_persist_read(&v4);
    if (e == 0) e = nf_fbh_persist_read(&v5);
    if (e != 0) return e;
    *count = (int)(v + v2 + v3 + v4 + v5);
    return 0;
}

static unsigned* g_persist_host_word = nullptr;          // pinned + mapped; the kernels' sticky error word
static unsigned* g_persist_dev_word = nullptr;

extern "C" int nf_persistent_config(int64_t spin_limit, int reset, void** host_error_word) {
    if (spin_limit < 0 || spin_limit > 0xffffffffLL) return NF_E_BADARG;
    if (g_persist_host_word == nullptr) {
        void* hp = nullptr;
        hipError_t e = hipHostMalloc(&hp, 64, hipHostMallocMapped | hipHostMallocPortable);
        if (e != hipSuccess) return (int)e;
        void* dp = nullptr;
        e = hipHostGetDevicePointer(&dp, hp, 0);
        if (e != hipSuccess) return (int)e;
        g_persist_host_word = (unsigned*)hp;
        g_persist_dev_word = (unsigned*)dp;
        *(volatile unsigned*)g_persist_host_word = 0u;
    }
    if (reset) {
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) return (int)e;
        *(volatile unsigned*)g_persist_host_word = 0u;
    }
    const unsigned lim = (unsigned)spin_limit;
    int e = nf_mc_persist_set(lim, g_persist_dev_word, reset);
    if (e == 0) e = nf_md_persist_set(lim, g_persist_dev_word, reset);
    if (e == 0) e = nf_cc_persist_set(lim, g_persist_dev_word, reset);
    if (e == 0) e = nf_so_persist_set(lim, g_persist_dev_word, reset);
    if (e == 0) e = nf_fbh_persist_set(lim, g_persist_dev_word, reset);
    if (e != 0) return e;
    if (host_error_word != nullptr) *host_error_word = (void*)g_persist_host_word;
    return 0;
}

__attribute__((visibility("hidden"))) int nf_md_persist_capacity(int* blocks);                            // made_chain.hip

// co-resident workgroups of the largest persistent kernel of each family on the current device: occupancy x compute units.
// A grid above that would wait on workgroups that cannot start: the host side then takes the multi-launch path instead.
extern "C" int nf_persistent_capacity(int* mlp_blocks, int* maf_blocks) {
    if (mlp_blocks == nullptr || maf_blocks == nullptr) return NF_E_BADARG;
    int dev = 0, cus = 0, per_cu = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const size_t lds = nf_mc_lds_bytes(3) + 3 * NF_GF_REC_WORDS * 8;
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_glow_flow_bwd<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_glow_flow_bwd<1>, NF_MC_THREADS, lds);
    if (e != hipSuccess) return (int)e;
    *mlp_blocks = per_cu * cus;
    return nf_md_persist_capacity(maf_blocks);
}
