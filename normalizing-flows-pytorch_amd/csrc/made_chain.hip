// One whole MAF flow step on vector data as ONE persistent launch per direction (training mode, N <= NF_MAF_MAX_ROWS):
//   flow BatchNorm (flows/modules.py:283-307, batch statistics, affine=False) -> column permutation (maf.py:100, z @ perm)
//   -> the two MADE conditioners s, t (maf.py:49-64: masked Linear -> BatchNorm1d -> ReLU, three hidden layers of 32)
//   -> affine transform y = z exp(tanh(s) a + c) + t over ALL features, log-det += sum s (maf.py:103-106).
// Same machinery as mlp_chain.hip (see there for the reasoning and the measurements): one 8-wave workgroup per 128 rows, a
// 16-row tile per wave, activations in registers in the row-per-lane layout of nf_mfma16.h, the masked weights of both nets
// in LDS, batch statistics across the grid through per-workgroup {value, generation} slots that every workgroup polls
// (one memory round trip, deterministic order), weight gradients formed per workgroup from parked tiles and folded from
// per-workgroup slabs after a final fenced exchange.  Both nets run in lockstep, so one exchange carries both nets'
// statistics (128 values) and the two independent GEMM chains interleave.
// Replaces 7 launches forward (statistics, head, rocBLAS permutation matmul, 4 linear+BN launches, transform) and 13
// backward per flow step; numerics are those of linear_bn.hip / flowbn_head.hip.
#include "nf_common.h"
#include "nf_det.h"

NF_DET_STATE(nf_mdc)
NF_DET_HOST_API(nf_mdc)
#include "nf_mfma16.h"

#define NF_MD_WAVES (NF_MAF_ROWS_PER_BLOCK / 16)
#define NF_MD_THREADS (NF_MD_WAVES * NF_WAVE)
#define NF_MD_NKQ (NF_MD_WAVES / 4)
#define NF_MD_NL 4
#define NF_MD_NB 3
#define NF_MD_XW 128                                      // exchange width: 2 nets x (32 sums + 32 square sums)
static_assert(NF_MD_THREADS == 512 && NF_MD_WAVES % 4 == 0, "geometry");

NF_PERSIST_STATE(nf_md)

struct NfMadeP {                                          // [net][layer]
    const float* w[2][NF_MD_NL]; const float* m[2][NF_MD_NL]; const float* b[2][NF_MD_NL];
    const float* gamma[2][NF_MD_NB]; const float* beta[2][NF_MD_NB];
    float* rmean[2][NF_MD_NB]; float* rvar[2][NF_MD_NB]; int64_t* nbt[2][NF_MD_NB];
};
struct NfMadeG { float* w[2][NF_MD_NL]; float* b[2][NF_MD_NL]; float* gamma[2][NF_MD_NB]; float* beta[2][NF_MD_NB]; };
struct NfMafV {
    const float* z; float* y; float* ld;                  // forward
    const float* g_y; const float* g_ld; float* g_z;      // backward
    const float *log_gamma, *bn_beta;                     // flow BatchNorm (affine=False: buffers)
    float *bmean, *bvar, *rmean, *rvar;
    const float* perm; const float *a, *c;                // permutation matrix (D, D); transform scale / shift scalars
    float *g_a, *g_c;
    float eps, mom;
    int D;
};

static inline void nf_made_unpack(const void* const* t, NfMadeP& p) {
    for (int n = 0; n < 2; ++n) {
        const void* const* q = t + n * (3 * NF_MD_NL + 5 * NF_MD_NB);
        for (int l = 0; l < NF_MD_NL; ++l) {
            p.w[n][l] = (const float*)q[3 * l]; p.m[n][l] = (const float*)q[3 * l + 1]; p.b[n][l] = (const float*)q[3 * l + 2];
        }
        q += 3 * NF_MD_NL;
        for (int j = 0; j < NF_MD_NB; ++j) {
            p.gamma[n][j] = (const float*)q[5 * j]; p.beta[n][j] = (const float*)q[5 * j + 1];
            p.rmean[n][j] = (float*)q[5 * j + 2]; p.rvar[n][j] = (float*)q[5 * j + 3]; p.nbt[n][j] = (int64_t*)q[5 * j + 4];
        }
    }
}

// LDS (floats)
#define NF_MD_W 0                                         // [2][4][32 * 36] masked weights, zero padded
#define NF_MD_B (NF_MD_W + 2 * NF_MD_NL * 32 * NF_FP_ST)  // [2][4][32] biases
#define NF_MD_GA (NF_MD_B + 2 * NF_MD_NL * 32)            // [2][3][32] gamma
#define NF_MD_BE (NF_MD_GA + 2 * NF_MD_NB * 32)           // [2][3][32] beta
#define NF_MD_BNC (NF_MD_BE + 2 * NF_MD_NB * 32)          // [2][3][4][32] scale, shift, mean, invstd
#define NF_MD_VAR (NF_MD_BNC + 2 * NF_MD_NB * 4 * 32)     // [2][3][32] biased batch variance (bookkeeping)
#define NF_MD_HEAD (NF_MD_VAR + 2 * NF_MD_NB * 32)        // flow BN per feature: mean[4], sd[4], exp(log_gamma)[4], beta[4]; perm[16]; dld, a, c
#define NF_MD_RED (NF_MD_HEAD + 48)                       // [waves][128] per-wave partials
#define NF_MD_PART (NF_MD_RED + NF_MD_WAVES * NF_MD_XW)   // [4][128] partial totals of the poll groups
#define NF_MD_TOT (NF_MD_PART + 4 * NF_MD_XW)             // [2][128] grid totals, double buffered by round parity
#define NF_MD_GB (NF_MD_TOT + 2 * NF_MD_XW)               // [3][128] backward: sum_g | sum_gx of both nets per BatchNorm
#define NF_MD_TILES (NF_MD_GB + NF_MD_NB * NF_MD_XW)      // per-wave 16 x 36 tiles

static inline size_t nf_md_lds_bytes(int tiles_per_wave) {
    return (size_t)(NF_MD_TILES + NF_MD_WAVES * tiles_per_wave * 16 * NF_FP_ST) * sizeof(float);
}

// ---- grid exchange of NF_MD_XW values per workgroup (see mlp_chain.hip: nf_mc_publish / nf_mc_collect) --------------------
__device__ __forceinline__ void nf_md_publish(float* sm, unsigned long long* slots, int round, unsigned gen) {
    float* red = sm + NF_MD_RED;
    __syncthreads();
    if (threadIdx.x < NF_MD_XW) {
        float mine = 0.f;
#pragma unroll
        for (int w = 0; w < NF_MD_WAVES; ++w) mine += red[w * NF_MD_XW + threadIdx.x];
        if (gridDim.x == 1) {
            red[threadIdx.x] = mine;
        } else {
            const unsigned long long pk = ((unsigned long long)gen << 32) | (unsigned long long)__float_as_uint(mine);
            __hip_atomic_store(slots + ((size_t)round * NF_MAF_MAX_BLOCKS + blockIdx.x) * NF_MD_XW + threadIdx.x, pk, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
#define NF_MD_POLL 16
// Two-level exchange for grids of more than NF_MD_2LVL workgroups.  All-to-all polling moves G x G x 1 KB per sweep (16.8 MB at
// 128 workgroups: bandwidth-, not latency-bound, ~2.5x the cost at 32).  Instead the first workgroup of every group of
// NF_MD_GRP polls its group's slots, publishes the group total, and everybody polls the <= 8 group totals: G x (16 + 8) KB
// per sweep, two hops.  Fixed summation order (members in order, then groups in order): still deterministic.
#define NF_MD_GRP 16
#define NF_MD_2LVL 32
#define NF_MD_MAX_GROUPS (NF_MAF_MAX_BLOCKS / NF_MD_GRP)
#define NF_MD_ROUNDS 4
// thread t owns value index i = t % 128 of the workgroups b = t / 128 (mod 4): up to 16 polls in flight per trip, partial sums
// in workgroup order, then the four groups are added in order -- deterministic, and no gather buffer in LDS
__device__ __forceinline__ const float* nf_md_collect(float* sm, unsigned long long* slots, int round, unsigned gen) {
    float* tot = sm + NF_MD_TOT + (round & 1) * NF_MD_XW;
    const int G = gridDim.x;
    if (G == 1) {
        if (threadIdx.x < NF_MD_XW) tot[threadIdx.x] = sm[NF_MD_RED + threadIdx.x];
        __syncthreads();
        return tot;
    }
    const int i = threadIdx.x & (NF_MD_XW - 1), grp = threadIdx.x >> 7;
    const unsigned long long* rs = slots + (size_t)round * NF_MAF_MAX_BLOCKS * NF_MD_XW + i;
    if (G > NF_MD_2LVL) {
        unsigned long long* gs = slots + (size_t)NF_MD_ROUNDS * NF_MAF_MAX_BLOCKS * NF_MD_XW +
                                 (size_t)round * NF_MD_MAX_GROUPS * NF_MD_XW + i;
        const int j = blockIdx.x / NF_MD_GRP, ngroups = (G + NF_MD_GRP - 1) / NF_MD_GRP;
        if (blockIdx.x % NF_MD_GRP == 0) {               // group leader (block-uniform): members j*GRP + grp + 4k
            const int b_lo = j * NF_MD_GRP, b_hi = min(G, b_lo + NF_MD_GRP);
            unsigned long long v[NF_MD_GRP / 4];
            unsigned spins = 0;
            bool ok;
            do {
                ok = true;
#pragma unroll
                for (int k = 0; k < NF_MD_GRP / 4; ++k) {
                    const int b = b_lo + grp + 4 * k;
                    v[k] = __hip_atomic_load(rs + (size_t)(b < b_hi ? b : b_lo) * NF_MD_XW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int k = 0; k < NF_MD_GRP / 4; ++k) ok = ok && (unsigned)(v[k] >> 32) == gen;
                if (ok) break;
                if (++spins > nf_md_spin_limit) { NF_PERSIST_GIVE_UP(nf_md); break; }
                __builtin_amdgcn_s_sleep(1);
            } while (true);
            float a1 = 0.f;
#pragma unroll
            for (int k = 0; k < NF_MD_GRP / 4; ++k)
                if (b_lo + grp + 4 * k < b_hi) a1 += __uint_as_float((unsigned)v[k]);
            sm[NF_MD_PART + threadIdx.x] = a1;
            __syncthreads();
            if (threadIdx.x < NF_MD_XW) {
                const float gt = (sm[NF_MD_PART + threadIdx.x] + sm[NF_MD_PART + NF_MD_XW + threadIdx.x]) +
                                 (sm[NF_MD_PART + 2 * NF_MD_XW + threadIdx.x] + sm[NF_MD_PART + 3 * NF_MD_XW + threadIdx.x]);
                const unsigned long long pk = ((unsigned long long)gen << 32) | (unsigned long long)__float_as_uint(gt);
                __hip_atomic_store(gs + (size_t)j * NF_MD_XW, pk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();                             // PART is reused below
        }
        unsigned long long v[NF_MD_MAX_GROUPS / 4];
        unsigned spins = 0;
        bool ok;
        do {
            ok = true;
#pragma unroll
            for (int k = 0; k < NF_MD_MAX_GROUPS / 4; ++k) {
                const int q = grp + 4 * k;
                v[k] = __hip_atomic_load(gs + (size_t)(q < ngroups ? q : 0) * NF_MD_XW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int k = 0; k < NF_MD_MAX_GROUPS / 4; ++k) ok = ok && (unsigned)(v[k] >> 32) == gen;
            if (ok) break;
            if (++spins > nf_md_spin_limit) { NF_PERSIST_GIVE_UP(nf_md); break; }
            __builtin_amdgcn_s_sleep(1);
        } while (true);
        float a2 = 0.f;
#pragma unroll
        for (int k = 0; k < NF_MD_MAX_GROUPS / 4; ++k)
            if (grp + 4 * k < ngroups) a2 += __uint_as_float((unsigned)v[k]);
        sm[NF_MD_PART + threadIdx.x] = a2;
        __syncthreads();
        if (threadIdx.x < NF_MD_XW)
            tot[threadIdx.x] = (sm[NF_MD_PART + threadIdx.x] + sm[NF_MD_PART + NF_MD_XW + threadIdx.x]) +
                               (sm[NF_MD_PART + 2 * NF_MD_XW + threadIdx.x] + sm[NF_MD_PART + 3 * NF_MD_XW + threadIdx.x]);
        __syncthreads();
        return tot;
    }
    float acc = 0.f;
    for (int b0 = grp; b0 < G; b0 += 4 * NF_MD_POLL) {
        unsigned long long v[NF_MD_POLL];
        unsigned spins = 0;
        bool ok;
        do {
            ok = true;
#pragma unroll
            for (int k = 0; k < NF_MD_POLL; ++k) {
                const int b = b0 + 4 * k;
                v[k] = __hip_atomic_load(rs + (size_t)(b < G ? b : b0) * NF_MD_XW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int k = 0; k < NF_MD_POLL; ++k) ok = ok && (unsigned)(v[k] >> 32) == gen;
            if (ok) break;
            if (++spins > nf_md_spin_limit) { NF_PERSIST_GIVE_UP(nf_md); break; }      // bounded: a mistake cannot hang the box
            __builtin_amdgcn_s_sleep(1);
        } while (true);
#pragma unroll
        for (int k = 0; k < NF_MD_POLL; ++k)
            if (b0 + 4 * k < G) acc += __uint_as_float((unsigned)v[k]);
    }
    sm[NF_MD_PART + threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < NF_MD_XW)
        tot[threadIdx.x] = (sm[NF_MD_PART + threadIdx.x] + sm[NF_MD_PART + NF_MD_XW + threadIdx.x]) +
                           (sm[NF_MD_PART + 2 * NF_MD_XW + threadIdx.x] + sm[NF_MD_PART + 3 * NF_MD_XW + threadIdx.x]);
    __syncthreads();
    return tot;
}

// ---- batch STATISTICS through the exchange (see mlp_chain.hip, nf_mc_publish_stats): no E[x^2] - E[x]^2 anywhere --------------------
// Feature f = n * 32 + k (net n, f < 64) travels as ONE 64-bit slot {sum : M2 | 1}: the workgroup's sum and its sum of squared
// deviations about the workgroup's OWN mean (wave tiles merged by the parallel-variance rule).  The workspace is zero at launch
// and every round has its own slots, so "non-zero" means "published" (the forced low mantissa bit of M2 makes it so: 1 ulp).
// A collector turns each pair into deviations about ONE common centre c = the mean of workgroup 0,
//     T_b = M2_b + n_b (mean_b - c)^2 = sum over the rows of b of (x - c)^2,   which is additive,
// and finishes with  M2 = sum_b T_b - N (mean - c)^2:  c is within ~ std / sqrt(128) of the mean, so that last subtraction is
// between terms ~ 1 / 128 of the result -- harmless, unlike the (mean / std)^2 blow-up of a one-pass variance.
//   tile:    red[w][n * 64 + k] = tile sum, red[w][n * 64 + 32 + k] = tile M2 about the tile mean (rows beyond N excluded)
//   collect: tot[n * 64 + k] = grid sum, tot[n * 64 + 32 + k] = grid M2      (thread t: feature t & 63, workgroups (t >> 6) + 8 j)
__device__ __forceinline__ int nf_md_rows_of_block(int64_t N, int b) {
    const int64_t left = N - (int64_t)b * NF_MAF_ROWS_PER_BLOCK;
    return (int)(left < NF_MAF_ROWS_PER_BLOCK ? (left > 0 ? left : 0) : NF_MAF_ROWS_PER_BLOCK);
}
__device__ __forceinline__ void nf_md_publish_stats(float* sm, unsigned long long* slots, int round, int64_t N) {
    float* red = sm + NF_MD_RED;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int f = (threadIdx.x >> 5) * 64 + (threadIdx.x & 31);
        const int nb = nf_md_rows_of_block(N, blockIdx.x);
        float S, M2;
        if (nb == NF_MAF_ROWS_PER_BLOCK) {               // every tile full: pairwise tree over the waves, no division
            float sv[NF_MD_WAVES], mv[NF_MD_WAVES];
#pragma unroll
            for (int w = 0; w < NF_MD_WAVES; ++w) { sv[w] = red[w * NF_MD_XW + f]; mv[w] = red[w * NF_MD_XW + 32 + f]; }
#pragma unroll
            for (int st = 1; st < NF_MD_WAVES; st *= 2)
#pragma unroll
                for (int w = 0; w < NF_MD_WAVES; w += 2 * st) {
                    const float dl = sv[w + st] - sv[w];
                    mv[w] = (mv[w] + mv[w + st]) + dl * dl * (0.5f / (float)(16 * st));
                    sv[w] += sv[w + st];
                }
            S = sv[0]; M2 = mv[0];
        } else {
            S = 0.f; M2 = 0.f;
            for (int w = 0; w < NF_MD_WAVES; ++w) S += red[w * NF_MD_XW + f];
            const float mb = S / (float)max(nb, 1);
            for (int w = 0; w < NF_MD_WAVES; ++w) {
                const int nw = min(max(nb - 16 * w, 0), 16);
                const float d = red[w * NF_MD_XW + f] / (float)max(nw, 1) - mb;
                M2 += nw > 0 ? fmaf((float)nw * d, d, red[w * NF_MD_XW + 32 + f]) : 0.f;
            }
        }
        if (gridDim.x == 1) {
            red[f] = S; red[32 + f] = M2;                 // threads 0..63 are ONE wave: every read above precedes these stores
        } else {
            const unsigned long long pk = ((unsigned long long)__float_as_uint(S) << 32) | (unsigned long long)(__float_as_uint(M2) | 1u);
            __hip_atomic_store(slots + ((size_t)round * NF_MAF_MAX_BLOCKS + blockIdx.x) * NF_MD_XW + threadIdx.x, pk, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
#define NF_MD_SPOLL (NF_MAF_MAX_BLOCKS / 8)
__device__ __forceinline__ const float* nf_md_collect_stats(float* sm, unsigned long long* slots, int round, int64_t N) {
    float* tot = sm + NF_MD_TOT + (round & 1) * NF_MD_XW;
    const int G = gridDim.x;
    if (G == 1) {
        __syncthreads();                                  // wave 0 wrote row 0 of RED for both waves' readers (publish_stats)
        if (threadIdx.x < NF_MD_XW) tot[threadIdx.x] = sm[NF_MD_RED + threadIdx.x];
        __syncthreads();
        return tot;
    }
    const int i = threadIdx.x & 63, grp = threadIdx.x >> 6;                    // 512 threads: 8 poll groups
    const unsigned long long* rs = slots + (size_t)round * NF_MAF_MAX_BLOCKS * NF_MD_XW + i;
    unsigned long long v[NF_MD_SPOLL], v0;
    unsigned spins = 0;
    bool ok;
    do {                                                  // every poll of the thread in flight at once: one latency per round
        ok = true;
        v0 = __hip_atomic_load(rs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int k = 0; k < NF_MD_SPOLL; ++k) {
            const int b = grp + 8 * k;
            v[k] = __hip_atomic_load(rs + (size_t)(b < G ? b : 0) * NF_MD_XW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ok = v0 != 0ull;
#pragma unroll
        for (int k = 0; k < NF_MD_SPOLL; ++k) ok = ok && v[k] != 0ull;
        if (ok) break;
        if (++spins > nf_md_spin_limit) { NF_PERSIST_GIVE_UP(nf_md); break; }
        __builtin_amdgcn_s_sleep(1);
    } while (true);
    const float c = __uint_as_float((unsigned)(v0 >> 32)) / (float)nf_md_rows_of_block(N, 0);
    float aS = 0.f, aT = 0.f;
#pragma unroll
    for (int k = 0; k < NF_MD_SPOLL; ++k) {
        const int b = grp + 8 * k;
        if (b < G) {
            const int nb = nf_md_rows_of_block(N, b);
            const float Sb = __uint_as_float((unsigned)(v[k] >> 32)), Mb = __uint_as_float((unsigned)v[k]);
            const float d = Sb * (nb == NF_MAF_ROWS_PER_BLOCK ? 1.f / (float)NF_MAF_ROWS_PER_BLOCK : 1.f / (float)max(nb, 1)) - c;
            aS += Sb;
            aT += fmaf((float)nb * d, d, Mb);
        }
    }
    __syncthreads();                                      // wave 0 is done reading RED (publish) before it is reused below
    sm[NF_MD_PART + grp * 64 + i] = aS;
    sm[NF_MD_RED + grp * 64 + i] = aT;
    __syncthreads();
    if (threadIdx.x < 64) {
        float S = 0.f, T = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) { S += sm[NF_MD_PART + q * 64 + i]; T += sm[NF_MD_RED + q * 64 + i]; }
        const float dm = S / (float)N - c;
        const int f = (i >> 5) * 64 + (i & 31);
        tot[f] = S;
        tot[32 + f] = fmaxf(T - (float)N * dm * dm, 0.f);
    }
    __syncthreads();
    return tot;
}
// tile sums and M2 about the tile mean of an R-layout vector (nt = valid rows of the tile; rows beyond hold zeros)
__device__ __forceinline__ void nf_md_colstats(const float (&v)[8], float* tile, float (&s1)[2], float (&m2)[2], int c16, int g, int nt) {
    nf_fp_store_rows(v, tile, c16, g);
    nf_fp_wsync();
    float c[2][4];
    nf_fp_load_cols<2>(tile, c, c16, g);
    nf_fp_wsync();
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        s1[cb] = nf_fp_rowsum((c[cb][0] + c[cb][1]) + (c[cb][2] + c[cb][3]));
        const float mt = s1[cb] / (float)max(nt, 1);
        float q = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float d = (4 * s + g < nt) ? c[cb][s] - mt : 0.f;
            q = fmaf(d, d, q);
        }
        m2[cb] = nf_fp_rowsum(q);
    }
}

// ---- staging: masked weights of both nets, biases, BatchNorm affines ------------------------------------------------------------
__device__ __forceinline__ void nf_md_stage(const NfMadeP& p, float* sm, int D) {
    const int tid = threadIdx.x, k = tid & 31;
    float w[2][NF_MD_NL][2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int l = 0; l < NF_MD_NL; ++l) {              // 16 independent (weight, mask) pairs in flight, one latency
            const int I = l == 0 ? D : 32, O = l == NF_MD_NL - 1 ? D : 32;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int oo = (tid >> 5) + 16 * h2;
                const bool ok = oo < O && k < I;
                const int e = ok ? oo * I + k : 0;
                const float wv = p.w[n][l][e], mv = p.m[n][l][e];
                w[n][l][h2] = ok ? wv * mv : 0.f;                                    // maf.py:54: F.linear(h, W * M, b)
            }
        }
    float bk = 0.f, ga = 0.f, be = 0.f;
    if (tid < 2 * NF_MD_NL * 32) {
        const int n = tid >> 7, l = (tid >> 5) & 3;
        const int O = l == NF_MD_NL - 1 ? D : 32;
        bk = k < O ? p.b[n][l][k] : 0.f;
    }
    if (tid < 2 * NF_MD_NB * 32) {
        const int n = tid / 96, j = (tid / 32) % 3;
        ga = p.gamma[n][j][k]; be = p.beta[n][j][k];
    }
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int l = 0; l < NF_MD_NL; ++l)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
                sm[NF_MD_W + (n * NF_MD_NL + l) * 32 * NF_FP_ST + ((tid >> 5) + 16 * h2) * NF_FP_ST + k] = w[n][l][h2];
    if (tid < 2 * NF_MD_NL * 32) sm[NF_MD_B + tid] = bk;
    if (tid < 2 * NF_MD_NB * 32) { sm[NF_MD_GA + tid] = ga; sm[NF_MD_BE + tid] = be; }
}

__device__ __forceinline__ void nf_md_linear(const float* sm, int n, int l, const float (&av)[8], float (&dv)[8], int c16, int g) {
    f32x4 acc[2] = {nf_fp_zero4(), nf_fp_zero4()};
    nf_fp_gemm<2>(sm + NF_MD_W + (n * NF_MD_NL + l) * 32 * NF_FP_ST, NF_FP_ST, 0, av, acc, c16, g);
#pragma unroll
    for (int j = 0; j < 8; ++j) dv[j] = acc[j >> 2][j & 3];
}
// ReLU(BatchNorm_j(a)) of net n
__device__ __forceinline__ void nf_md_activate(const float* sm, int n, int j, const float (&a)[8], float (&av)[8], int g) {
    float sc[8], sh[8];
    nf_fp_ldvec(sm + NF_MD_BNC + ((n * NF_MD_NB + j) * 4 + 0) * 32, g, sc);
    nf_fp_ldvec(sm + NF_MD_BNC + ((n * NF_MD_NB + j) * 4 + 1) * 32, g, sh);
#pragma unroll
    for (int k = 0; k < 8; ++k) av[k] = fmaxf(fmaf(a[k], sc[k], sh[k]), 0.f);
}
__device__ __forceinline__ void nf_md_bn_consts(float* sm, int n, int j, int k, float mean, float invstd) {
    const int q = n * NF_MD_NB + j;
    const float sc = sm[NF_MD_GA + q * 32 + k] * invstd;
    sm[NF_MD_BNC + (q * 4 + 0) * 32 + k] = sc;
    sm[NF_MD_BNC + (q * 4 + 1) * 32 + k] = sm[NF_MD_BE + q * 32 + k] - mean * sc;
    sm[NF_MD_BNC + (q * 4 + 2) * 32 + k] = mean;
    sm[NF_MD_BNC + (q * 4 + 3) * 32 + k] = invstd;
}
// this lane's share of the column sums of an R-layout vector and (SQ) of its squares, reduced over the four lane groups
template <bool SQ>
__device__ __forceinline__ void nf_md_colsums(const float (&v)[8], float* tile, float (&s1)[2], float (&s2)[2], int c16, int g) {
    nf_fp_store_rows(v, tile, c16, g);
    nf_fp_wsync();
    float c[2][4];
    nf_fp_load_cols<2>(tile, c, c16, g);
    nf_fp_wsync();
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        s1[cb] = nf_fp_rowsum((c[cb][0] + c[cb][1]) + (c[cb][2] + c[cb][3]));
        if (SQ) s2[cb] = nf_fp_rowsum(fmaf(c[cb][0], c[cb][0], fmaf(c[cb][1], c[cb][1], fmaf(c[cb][2], c[cb][2], c[cb][3] * c[cb][3]))));
    }
}

// flow BatchNorm constants of feature c from (mean, var incl. eps) -> HEAD
__device__ __forceinline__ void nf_md_head_consts(float* sm, const NfMafV& h, int c, float mean, float var) {
    sm[NF_MD_HEAD + c] = mean;
    sm[NF_MD_HEAD + 4 + c] = sqrtf(var);
    sm[NF_MD_HEAD + 8 + c] = c < h.D ? expf(h.log_gamma[c]) : 1.f;
    sm[NF_MD_HEAD + 12 + c] = c < h.D ? h.bn_beta[c] : 0.f;
    sm[NF_MD_HEAD + 36 + c] = c < h.D ? h.log_gamma[c] - 0.5f * logf(var) : 0.f;     // per-feature log-det (modules.py:303-305)
}
// h = BN(z), z' = h perm  (per row, every lane of the row's four)
__device__ __forceinline__ void nf_md_head_row(const float* sm, const float (&zr)[4], int D, float (&zp)[4]) {
    float hh[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
        hh[c] = c < D ? ((zr[c] - sm[NF_MD_HEAD + c]) / sm[NF_MD_HEAD + 4 + c]) * sm[NF_MD_HEAD + 8 + c] + sm[NF_MD_HEAD + 12 + c] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) acc = fmaf(hh[c], sm[NF_MD_HEAD + 16 + 4 * c + j], acc);     // maf.py:100: z @ perm
        zp[j] = acc;
    }
}
__device__ __forceinline__ void nf_md_load_row(const float* z, int64_t row, bool rv, int D, float (&zr)[4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float v = z[(rv ? row : 0) * D + (c < D ? c : 0)];
        zr[c] = (rv && c < D) ? v : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
// MODE 0: the training-mode forward step.  MODE 1: the evaluation-mode forward (every BatchNorm on its running statistics: no
// grid exchange, no buffer touched).  MODE 2: the INVERSE step (maf.py:108-119, modules.py:309-322): h.z is the step's output,
// D sequential passes of the MADE pair each fix one feature, then perm^-1 and the flow BatchNorm's inverse (batch buffers when
// `training`, running statistics otherwise); with `training` the MADE BatchNorms use batch statistics and update their running
// statistics once per pass, exactly as D module calls do (pass i exchanges through ws + i * NF_MAF_WS_FLOATS).
template <int MODE>
__global__ void __launch_bounds__(NF_MD_THREADS) k_maf_step_fwd(NfMadeP p, NfMafV h, float* save, float* ws, int64_t N, float eps,
                                                                int training) {
    const bool use_batch = MODE == 0 || (MODE == 2 && training != 0);
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, c16 = lane & 15, g = lane >> 4;
    const int64_t row = ((int64_t)blockIdx.x * NF_MD_WAVES + wid) * 16 + c16;
    const bool rv = row < N;
    const int D = h.D;
    float zr[4], ld_in = 0.f, rm_old = 0.f, rv_old = 0.f, kc = 0.f, frm = 0.f, frv = 0.f;
    nf_md_load_row(h.z, row, rv, D, zr);                  // issued before the staging: one memory latency for everything
    if (rv && g == 0) ld_in = h.ld[row];
    if ((blockIdx.x == 0 || !use_batch) && threadIdx.x < 2 * NF_MD_NB * 32) {
        const int n = threadIdx.x / 96, j = (threadIdx.x / 32) % 3, k = threadIdx.x & 31;
        rm_old = p.rmean[n][j][k]; rv_old = p.rvar[n][j][k];
    }
    frv = 1.f;
    if (threadIdx.x < 4 && (int)threadIdx.x < D) {
        const bool buffers = MODE == 2 && training != 0;  // the inverse of a training-mode flow BatchNorm reads its batch buffers
        frm = buffers ? h.bmean[threadIdx.x] : h.rmean[threadIdx.x];
        frv = buffers ? h.bvar[threadIdx.x] : h.rvar[threadIdx.x];
    }
    if (threadIdx.x < 16) sm[NF_MD_HEAD + 16 + threadIdx.x] = ((threadIdx.x >> 2) < D && (threadIdx.x & 3) < D) ? h.perm[(threadIdx.x >> 2) * D + (threadIdx.x & 3)] : 0.f;
    if (threadIdx.x == 16) { sm[NF_MD_HEAD + 33] = h.a[0]; sm[NF_MD_HEAD + 34] = h.c[0]; }
    if (threadIdx.x >= 32 && threadIdx.x < 36) sm[NF_MD_HEAD + 40 + (threadIdx.x - 32)] = (int)(threadIdx.x - 32) < D ? h.rmean[threadIdx.x - 32] : 0.f;
    nf_md_stage(p, sm, D);
    __syncthreads();
    unsigned long long* slots = (unsigned long long*)ws;
    float* tile = sm + NF_MD_TILES + wid * 16 * NF_FP_ST;
    float* red = sm + NF_MD_RED + wid * NF_MD_XW;

    if (MODE != 0) {                                      // constants: no exchange
        if (threadIdx.x < 4) nf_md_head_consts(sm, h, threadIdx.x, frm, frv);
        if (!use_batch && threadIdx.x < 2 * NF_MD_NB * 32) {
            const int n = threadIdx.x / 96, j = (threadIdx.x / 32) % 3, k = threadIdx.x & 31;
            nf_md_bn_consts(sm, n, j, k, rm_old, 1.f / sqrtf(rv_old + eps));
        }
        __syncthreads();
    }
    // ---- flow BatchNorm statistics: shifted sums around the running mean (flowbn_head.hip) ------------------------------
    if (MODE == 0) {
        float v8[8], s1[2], s2[2];
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = 0.f;
        if (g == 0 && rv) {
#pragma unroll
            for (int c = 0; c < 4; ++c) v8[c] = c < D ? zr[c] - sm[NF_MD_HEAD + 40 + c] : 0.f;
        }
        const int nt = min(max(nf_md_rows_of_block(N, blockIdx.x) - 16 * wid, 0), 16);
        nf_md_colstats(v8, tile, s1, s2, c16, g, nt);
        red[lane] = 0.f; red[64 + lane] = 0.f;
        nf_fp_wsync();
        if (g == 0 && c16 < 4) { red[c16] = s1[0]; red[32 + c16] = s2[0]; }
        nf_md_publish_stats(sm, slots, 0, N);
        const float* tot = nf_md_collect_stats(sm, slots, 0, N);
        if (threadIdx.x < 4) {
            const int c = threadIdx.x;
            const float n = (float)N;
            const float m1 = tot[c] / n;
            kc = sm[NF_MD_HEAD + 40 + c];
            const float mean = kc + m1;
            const float var = tot[32 + c] / n + eps;                                 // biased, eps inside (modules.py:287)
            nf_md_head_consts(sm, h, c, mean, var);
            if (blockIdx.x == 0 && c < D) {
                h.bmean[c] = mean; h.bvar[c] = var;
                h.rmean[c] = frm * (1.f - h.mom) + mean * h.mom;                    // modules.py:291-294
                h.rvar[c] = frv * (1.f - h.mom) + var * h.mom;
                save[2 * NF_MD_NB * 2 * 32 + c] = mean; save[2 * NF_MD_NB * 2 * 32 + 4 + c] = var;
            }
        }
        __syncthreads();
    }
    float zp[4], xa[8], inv_ld = 0.f;
    if (MODE == 2) {
#pragma unroll
        for (int c = 0; c < 4; ++c) zp[c] = zr[c];          // the step's output; feature i is replaced by pass i
    } else {
        nf_md_head_row(sm, zr, D, zp);
    }
    const int passes = MODE == 2 ? D : 1;
    float av[2][8], dv[2][8];
#pragma unroll 1
    for (int pass = 0; pass < passes; ++pass) {
    if (MODE == 2 && pass > 0) {
        slots += NF_MAF_WS_FLOATS / 2;                    // a fresh exchange workspace per pass
        __syncthreads();                                  // the previous pass is done with the BatchNorm constants in LDS
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) xa[j] = 0.f;
    if (g == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) xa[c] = zp[c];
    }
    // ---- the two MADE nets in lockstep ------------------------------------------------------------------------------------
#pragma unroll
    for (int n = 0; n < 2; ++n) nf_md_linear(sm, n, 0, xa, dv[n], c16, g);
#pragma unroll 1
    for (int l = 0; l < NF_MD_NB; ++l) {                  // dv = pre-bias output of linear l = input of BatchNorm l
#pragma unroll
        for (int n = 0; use_batch && n < 2; ++n) {
            float m8[8], s1[2], s2[2];
#pragma unroll
            for (int k = 0; k < 8; ++k) m8[k] = rv ? dv[n][k] : 0.f;
            nf_md_colstats(m8, tile, s1, s2, c16, g, min(max(nf_md_rows_of_block(N, blockIdx.x) - 16 * wid, 0), 16));
            if (g == 0) {
                red[n * 64 + c16] = s1[0]; red[n * 64 + 16 + c16] = s1[1];
                red[n * 64 + 32 + c16] = s2[0]; red[n * 64 + 48 + c16] = s2[1];
            }
        }
        const float* tot = nullptr;
        if (use_batch) {                                  // kernel-uniform
            nf_md_publish_stats(sm, slots, 1 + l, N);
            tot = nf_md_collect_stats(sm, slots, 1 + l, N);
        }
        if (use_batch && threadIdx.x < 64) {
            const int n = threadIdx.x >> 5, k = threadIdx.x & 31;
            const float invN = 1.f / (float)N;
            const float m1 = tot[n * 64 + k] * invN;
            const float mean = sm[NF_MD_B + (n * NF_MD_NL + l) * 32 + k] + m1;      // the linear's output is pre-bias
            const float var = tot[n * 64 + 32 + k] * invN;                          // biased, as BatchNorm normalises; M2 >= 0
            nf_md_bn_consts(sm, n, l, k, mean, 1.f / sqrtf(var + eps));
            sm[NF_MD_VAR + (n * NF_MD_NB + l) * 32 + k] = var;
        }
        if (use_batch) __syncthreads();
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            float bias[8], a_in[8];
            nf_fp_ldvec(sm + NF_MD_B + (n * NF_MD_NL + l) * 32, g, bias);
#pragma unroll
            for (int k = 0; k < 8; ++k) a_in[k] = dv[n][k] + bias[k];
            nf_md_activate(sm, n, l, a_in, av[n], g);
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) nf_md_linear(sm, n, l + 1, av[n], dv[n], c16, g);
    }
    if (use_batch && blockIdx.x == 0 && threadIdx.x < 2 * NF_MD_NB * 32) {       // BatchNorm1d bookkeeping, off the chain
        const int n = threadIdx.x / 96, j = (threadIdx.x / 32) % 3, k = threadIdx.x & 31, q = n * NF_MD_NB + j;
        const float mean = sm[NF_MD_BNC + (q * 4 + 2) * 32 + k], var = sm[NF_MD_VAR + q * 32 + k];
        const float unb = N > 1 ? var * ((float)N / (float)(N - 1)) : var;
        rm_old = 0.9f * rm_old + 0.1f * mean;                                         // nn.BatchNorm1d momentum 0.1
        rv_old = 0.9f * rv_old + 0.1f * unb;
        p.rmean[n][j][k] = rm_old;
        p.rvar[n][j][k] = rv_old;
        if (MODE == 0) {
            save[(q * 2 + 0) * 32 + k] = mean;
            save[(q * 2 + 1) * 32 + k] = sm[NF_MD_BNC + (q * 4 + 3) * 32 + k];
        }
        if (k == 0 && p.nbt[n][j] != nullptr) p.nbt[n][j][0] += 1;
    }
    if (MODE == 2) {                                      // pass i fixes feature i (maf.py:112-115)
        float bs[8], bt[8];
        nf_fp_ldvec(sm + NF_MD_B + (0 * NF_MD_NL + 3) * 32, g, bs);
        nf_fp_ldvec(sm + NF_MD_B + (1 * NF_MD_NL + 3) * 32, g, bt);
        const float ca = sm[NF_MD_HEAD + 33], cc = sm[NF_MD_HEAD + 34];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c == pass && g == 0) {
                const float sv = tanhf(dv[0][c] + bs[c]) * ca + cc;
                zp[c] = (zp[c] - (dv[1][c] + bt[c])) * expf(-sv);
                inv_ld -= sv;
            }
        }
    }
    }   // passes
    if (MODE == 2) {                                      // perm^-1 (maf.py:118), then the flow BatchNorm's inverse (modules.py:309-322)
        if (rv && g == 0) {
            float dld = inv_ld;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < D) {
                    float u = 0.f;
#pragma unroll
                    for (int c = 0; c < 4; ++c) u = fmaf(zp[c], sm[NF_MD_HEAD + 16 + 4 * j + c], u);        // z @ perm^T
                    h.y[row * D + j] = (u - sm[NF_MD_HEAD + 12 + j]) / sm[NF_MD_HEAD + 8 + j] * sm[NF_MD_HEAD + 4 + j] + sm[NF_MD_HEAD + j];
                    dld -= sm[NF_MD_HEAD + 36 + j];
                }
            }
            h.ld[row] = ld_in + dld;
        }
        return;
    }
    // ---- affine transform over all features (maf.py:103-106) ----------------------------------------------------------------
    if (rv && g == 0) {
        float bs[8], bt[8], dld = 0.f;
        nf_fp_ldvec(sm + NF_MD_B + (0 * NF_MD_NL + 3) * 32, g, bs);
        nf_fp_ldvec(sm + NF_MD_B + (1 * NF_MD_NL + 3) * 32, g, bt);
        const float ca = sm[NF_MD_HEAD + 33], cc = sm[NF_MD_HEAD + 34];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < D) {
                const float sv = tanhf(dv[0][c] + bs[c]) * ca + cc;
                h.y[row * D + c] = zp[c] * expf(sv) + (dv[1][c] + bt[c]);
                dld += sv + sm[NF_MD_HEAD + 36 + c];
            }
        }
        h.ld[row] = ld_in + dld;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------
#define NF_MD_SLAB_Q 1056
#define NF_MD_SLAB_L (NF_MD_NKQ * NF_MD_SLAB_Q)
#define NF_MD_SLAB (2 * NF_MD_NL * NF_MD_SLAB_L)
static_assert(NF_MD_SLAB == NF_MAF_SLAB_WG_FLOATS && NF_MD_WAVES * 2 == NF_MAF_HEAD_REC_WG, "per-workgroup sizes in include/nfhip.h");
static_assert(NF_MD_SLAB * NF_MAF_MAX_BLOCKS == NF_MAF_BWD_SLAB_FLOATS, "slab workspace size in include/nfhip.h");
static_assert((NF_MD_ROUNDS * NF_MAF_MAX_BLOCKS * NF_MD_XW + NF_MD_ROUNDS * NF_MD_MAX_GROUPS * NF_MD_XW) * 2 + 64 == NF_MAF_WS_FLOATS,
              "exchange workspace size in include/nfhip.h");
static_assert(NF_MAF_MAX_BLOCKS * NF_MAF_ROWS_PER_BLOCK == NF_MAF_MAX_ROWS, "geometry in include/nfhip.h");

// wave w's share of g_W[n][L] and g_b[n][L]: output block (w & 1, (w >> 1) & 1) over the rows of waves 4 (w >> 2) .. + 3
__device__ __forceinline__ void nf_md_wgrad_job(const float* sm, float* slab, int n, int L, int lane, int wid) {
    const int c16 = lane & 15, g = lane >> 4;
    const int ob = wid & 1, ib = (wid >> 1) & 1, kq = wid >> 2;
    f32x4 d = nf_fp_zero4();
    float bs = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float* Gt = sm + NF_MD_TILES + ((1 + 2 * n) * NF_MD_WAVES + 4 * kq + q) * 16 * NF_FP_ST + 16 * ob + c16;
        const float* At = sm + NF_MD_TILES + ((2 + 2 * n) * NF_MD_WAVES + 4 * kq + q) * 16 * NF_FP_ST + 16 * ib + c16;
        float ga[4], av[4];
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) { ga[s2] = Gt[(4 * s2 + g) * NF_FP_ST]; av[s2] = At[(4 * s2 + g) * NF_FP_ST]; }
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            bs += ga[s2];
            d = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s2], av[s2], d, 0, 0, 0);
        }
    }
    float* sl = slab + (n * NF_MD_NL + L) * NF_MD_SLAB_L + kq * NF_MD_SLAB_Q;     // [i][o]: the fold walks columns contiguously
    *(float4*)(sl + (16 * ib + c16) * 32 + 16 * ob + 4 * g) = make_float4(d[0], d[1], d[2], d[3]);
    bs = nf_fp_rowsum(bs);
    if (ib == 0 && g == 0) sl[1024 + 16 * ob + c16] = bs;
}

template <int L>
__device__ __forceinline__ void nf_md_bwd_layer(float* sm, const float (&xa)[8], const float (&a)[2][NF_MD_NB][8], float (&G)[2][8],
                                                float* slab, unsigned long long* slots, bool rv, int64_t N, int lane, int wid) {
    const int c16 = lane & 15, g = lane >> 4;
    float* TS = sm + NF_MD_TILES + wid * 16 * NF_FP_ST;
    float t[2][8];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        float* TG = sm + NF_MD_TILES + ((1 + 2 * n) * NF_MD_WAVES + wid) * 16 * NF_FP_ST;
        float* TA = sm + NF_MD_TILES + ((2 + 2 * n) * NF_MD_WAVES + wid) * 16 * NF_FP_ST;
        float act[8];
        if (L == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) act[k] = xa[k];
        } else {
            nf_md_activate(sm, n, L > 0 ? L - 1 : 0, a[n][L > 0 ? L - 1 : 0], act, g);
        }
        nf_fp_store_rows(G[n], TG, c16, g);
        nf_fp_store_rows(act, TA, c16, g);
        f32x4 acc[2] = {nf_fp_zero4(), nf_fp_zero4()};
        nf_fp_gemm_d<2>(sm + NF_MD_W + (n * NF_MD_NL + L) * 32 * NF_FP_ST, NF_FP_ST, 0, G[n], acc, c16, g);
#pragma unroll
        for (int k = 0; k < 8; ++k) t[n][k] = acc[k >> 2][k & 3];
    }
    if (L == 0) {                                         // G <- gradient of the conditioner input (both nets)
        __syncthreads();
        nf_md_wgrad_job(sm, slab, 0, L, lane, wid);
        nf_md_wgrad_job(sm, slab, 1, L, lane, wid);
#pragma unroll
        for (int k = 0; k < 8; ++k) { G[0][k] = t[0][k]; G[1][k] = t[1][k]; }
        return;
    }
    constexpr int J = L > 0 ? L - 1 : 0;
    float sc[2][8], gn[2][8], xh[2][8];
    float* red = sm + NF_MD_RED + wid * NF_MD_XW;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        float sh[8], mean[8], invstd[8], gnx[8], s1[2], s2[2], dum[2];
        const int q = n * NF_MD_NB + J;
        nf_fp_ldvec(sm + NF_MD_BNC + (q * 4 + 0) * 32, g, sc[n]);
        nf_fp_ldvec(sm + NF_MD_BNC + (q * 4 + 1) * 32, g, sh);
        nf_fp_ldvec(sm + NF_MD_BNC + (q * 4 + 2) * 32, g, mean);
        nf_fp_ldvec(sm + NF_MD_BNC + (q * 4 + 3) * 32, g, invstd);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            gn[n][k] = fmaf(a[n][J][k], sc[n][k], sh[k]) > 0.f ? t[n][k] : 0.f;    // ReLU mask; rows beyond N carry G = 0
            xh[n][k] = (a[n][J][k] - mean[k]) * invstd[k];
            gnx[k] = gn[n][k] * xh[n][k];
        }
        nf_md_colsums<false>(gn[n], TS, s1, dum, c16, g);
        nf_md_colsums<false>(gnx, TS, s2, dum, c16, g);
        if (g == 0) {
            red[n * 64 + c16] = s1[0]; red[n * 64 + 16 + c16] = s1[1];
            red[n * 64 + 32 + c16] = s2[0]; red[n * 64 + 48 + c16] = s2[1];
        }
    }
    nf_md_publish(sm, slots, NF_MD_NB - 1 - J, (unsigned)(NF_MD_NB - J));     // its barrier also covers the parked tiles
    nf_md_wgrad_job(sm, slab, 0, L, lane, wid);                              // runs while the partial sums travel
    nf_md_wgrad_job(sm, slab, 1, L, lane, wid);
    const float* tot = nf_md_collect(sm, slots, NF_MD_NB - 1 - J, (unsigned)(NF_MD_NB - J));
    if (threadIdx.x < NF_MD_XW) sm[NF_MD_GB + J * NF_MD_XW + threadIdx.x] = tot[threadIdx.x];
    const float invN = 1.f / (float)N;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        float mg[8], mgx[8];
        nf_fp_ldvec(tot + n * 64, g, mg);
        nf_fp_ldvec(tot + n * 64 + 32, g, mgx);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float v = sc[n][k] * (gn[n][k] - mg[k] * invN - xh[n][k] * (mgx[k] * invN));     // sc = gamma * invstd
            G[n][k] = rv ? v : 0.f;
        }
    }
}

__global__ void __launch_bounds__(NF_MD_THREADS) k_maf_step_bwd(NfMadeP p, NfMafV h, const float* __restrict__ save, NfMadeG gr,
                                                                float* ws, float* __restrict__ slabs, int64_t N,
                                                                float* __restrict__ head_rec) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, c16 = lane & 15, g = lane >> 4;
    const int64_t row = ((int64_t)blockIdx.x * NF_MD_WAVES + wid) * 16 + c16;
    const bool rv = row < N;
    const int D = h.D;
    float zr[4], gy[4], gld = 0.f, bn_mean = 0.f, bn_invstd = 0.f, fm = 0.f, fv = 1.f;
    nf_md_load_row(h.z, row, rv, D, zr);
    nf_md_load_row(h.g_y, row, rv, D, gy);
    if (h.g_ld != nullptr && rv) gld = h.g_ld[row];
    if (threadIdx.x < 2 * NF_MD_NB * 32) {
        const int q = threadIdx.x >> 5, k = threadIdx.x & 31;
        bn_mean = save[(q * 2 + 0) * 32 + k];
        bn_invstd = save[(q * 2 + 1) * 32 + k];
    }
    if (threadIdx.x < 4) { fm = save[2 * NF_MD_NB * 2 * 32 + threadIdx.x]; fv = save[2 * NF_MD_NB * 2 * 32 + 4 + threadIdx.x]; }
    if (threadIdx.x >= 64 && threadIdx.x < 80) {
        const int e = threadIdx.x - 64;
        sm[NF_MD_HEAD + 16 + e] = ((e >> 2) < D && (e & 3) < D) ? h.perm[(e >> 2) * D + (e & 3)] : 0.f;
    }
    if (threadIdx.x == 80) { sm[NF_MD_HEAD + 33] = h.a[0]; sm[NF_MD_HEAD + 34] = h.c[0]; }
    nf_md_stage(p, sm, D);
    __syncthreads();
    if (threadIdx.x < 2 * NF_MD_NB * 32) nf_md_bn_consts(sm, threadIdx.x / 96, (threadIdx.x / 32) % 3, threadIdx.x & 31, bn_mean, bn_invstd);
    if (threadIdx.x < 4) nf_md_head_consts(sm, h, threadIdx.x, (int)threadIdx.x < D ? fm : 0.f, (int)threadIdx.x < D ? fv : 1.f);
    __syncthreads();
    unsigned long long* slots = (unsigned long long*)ws;
    float* slab = slabs + (size_t)blockIdx.x * NF_MD_SLAB;

    // ---- forward once more, pre-activations kept -------------------------------------------------------------------------------
    float zp[4], xa[8], a[2][NF_MD_NB][8], out[2][4];
    nf_md_head_row(sm, zr, D, zp);
#pragma unroll
    for (int j = 0; j < 8; ++j) xa[j] = 0.f;
    if (g == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) xa[c] = zp[c];
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        float av[8], dv[8], bias[8];
        nf_md_linear(sm, n, 0, xa, dv, c16, g);
#pragma unroll
        for (int l = 0; l < NF_MD_NB; ++l) {
            nf_fp_ldvec(sm + NF_MD_B + (n * NF_MD_NL + l) * 32, g, bias);
#pragma unroll
            for (int k = 0; k < 8; ++k) a[n][l][k] = dv[k] + bias[k];
            nf_md_activate(sm, n, l, a[n][l], av, g);
            nf_md_linear(sm, n, l + 1, av, dv, c16, g);
        }
        nf_fp_ldvec(sm + NF_MD_B + (n * NF_MD_NL + 3) * 32, g, bias);
#pragma unroll
        for (int c = 0; c < 4; ++c) out[n][c] = dv[c] + bias[c];          // meaningful in the g = 0 lane
    }
    // ---- affine transform backward (lane g = 0 of each row), maf.py:103-106 ----------------------------------------------------
    float G[2][8], gzp[4] = {0.f, 0.f, 0.f, 0.f}, sum_gsv = 0.f, sum_gsvth = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { G[0][j] = 0.f; G[1][j] = 0.f; }
    if (g == 0 && rv) {
        const float ca = sm[NF_MD_HEAD + 33], cc = sm[NF_MD_HEAD + 34];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < D) {
                const float th = tanhf(out[0][c]);
                const float ev = expf(th * ca + cc);
                const float gsv = gy[c] * zp[c] * ev + gld;                   // ld += s: the log-det gradient enters here
                gzp[c] = gy[c] * ev;
                G[0][c] = gsv * ca * (1.f - th * th);                         // net s: gradient of s_raw
                G[1][c] = gy[c];                                              // net t
                sum_gsv += gsv;
                sum_gsvth += gsv * th;
            }
        }
    }
    nf_md_bwd_layer<3>(sm, xa, a, G, slab, slots, rv, N, lane, wid);
    nf_md_bwd_layer<2>(sm, xa, a, G, slab, slots, rv, N, lane, wid);
    nf_md_bwd_layer<1>(sm, xa, a, G, slab, slots, rv, N, lane, wid);
    nf_md_bwd_layer<0>(sm, xa, a, G, slab, slots, rv, N, lane, wid);
    // ---- permutation and flow BatchNorm backward (statistics are constants for autograd, appendix B4) ------------------------
    if (g == 0 && rv) {
#pragma unroll
        for (int c = 0; c < 4; ++c) gzp[c] += G[0][c] + G[1][c];              // conditioner inputs = z'
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < D) {
                float gh = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) gh = fmaf(gzp[j], sm[NF_MD_HEAD + 16 + 4 * c + j], gh);
                h.g_z[row * D + c] = gh / sm[NF_MD_HEAD + 4 + c] * sm[NF_MD_HEAD + 8 + c];
            }
        }
    }
    // Deferred fold (head_rec != nullptr: nf_maf_step_bwd_partial): nothing downstream of this launch waits for the parameter
    // gradients, yet the fenced exchange + fold below end every launch (and at 128 workgroups they are its slowest phases).
    // Every wave leaves its two scalar sums, the workgroup its slab; k_maf_fold_all folds all steps of a flow in one launch.
    const float* htot = nullptr;
    if (head_rec != nullptr) {
        const float s0 = nf_wave_sum(sum_gsv), s1 = nf_wave_sum(sum_gsvth);
        if (lane == 0) {
            head_rec[((size_t)blockIdx.x * NF_MD_WAVES + wid) * 2 + 0] = s0;
            head_rec[((size_t)blockIdx.x * NF_MD_WAVES + wid) * 2 + 1] = s1;
        }
    } else {
        // ---- final exchange: the two scalar sums, and (fenced) the grid barrier in front of the fold -----------------------------------
        {
            float* red = sm + NF_MD_RED + wid * NF_MD_XW;
            const float s0 = nf_wave_sum(sum_gsv), s1 = nf_wave_sum(sum_gsvth);
            __syncthreads();                                  // the last weight-gradient jobs are done with RED? (RED is not theirs) -- tiles
            red[lane] = 0.f; red[64 + lane] = 0.f;
            nf_fp_wsync();
            if (lane == 0) { red[0] = s0; red[1] = s1; }
            __syncthreads();
            if (threadIdx.x == 0) __threadfence();            // release: the slabs of every wave (cumulative through the barrier)
        }
        nf_md_publish(sm, slots, NF_MD_NB, (unsigned)(NF_MD_NB + 1));
        htot = nf_md_collect(sm, slots, NF_MD_NB, (unsigned)(NF_MD_NB + 1));
        if (threadIdx.x == 0) __threadfence();                // acquire
        __syncthreads();
        // ---- fold: half wave per (net, layer, column | bias, eighth of the workgroups); partial sums meet by atomics ---------------------
        {
            const int o = threadIdx.x & 31;
            const int G_ = gridDim.x;
            // deterministic mode: ONE half wave sums all slabs of a column in workgroup order (no eighths that meet by atomics)
            const int parts = nf_det_on(nf_mdc_det) ? 1 : 8;
            const int chunk = (G_ + parts - 1) / parts;
            constexpr int HW = NF_MD_THREADS / 32;
            for (int u = blockIdx.x * HW + (threadIdx.x >> 5); u < 2 * NF_MD_NL * 33 * parts; u += G_ * HW) {
                const int part = u % parts, uu = u / parts;
                const int nl = uu / 33, i = uu - nl * 33;     // i == 32: the bias
                const int n = nl >> 2, l = nl & 3;
                const int I = l == 0 ? D : 32, O = l == NF_MD_NL - 1 ? D : 32;
                const int b_lo = part * chunk, b_hi = min(G_, b_lo + chunk);
                if ((i < 32 && i >= I) || b_lo >= b_hi) continue;
                float tsum = 0.f;
                const float* base = slabs + (size_t)nl * NF_MD_SLAB_L + i * 32 + o;
                for (int b0 = b_lo; b0 < b_hi; b0 += 8) {
                    float v[8][NF_MD_NKQ];
    #pragma unroll
                    for (int q8 = 0; q8 < 8; ++q8) {
                        const int b = b0 + q8 < b_hi ? b0 + q8 : b_hi - 1;
    #pragma unroll
                        for (int q = 0; q < NF_MD_NKQ; ++q) v[q8][q] = base[(size_t)b * NF_MD_SLAB + q * NF_MD_SLAB_Q];
                    }
    #pragma unroll
                    for (int q8 = 0; q8 < 8; ++q8)
    #pragma unroll
                        for (int q = 0; q < NF_MD_NKQ; ++q)
                            if (b0 + q8 < b_hi) tsum += v[q8][q];
                }
                if (o < O) {
                    if (i == 32) atomicAdd(gr.b[n][l] + o, tsum);
                    else atomicAdd(gr.w[n][l] + o * I + i, tsum * p.m[n][l][o * I + i]);          // maf.py:54: d(W * M) = g * M
                }
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < 2 * NF_MD_NB * 32) {
        const int n = threadIdx.x / 96, j = (threadIdx.x / 32) % 3, k = threadIdx.x & 31;
        atomicAdd(gr.beta[n][j] + k, sm[NF_MD_GB + j * NF_MD_XW + n * 64 + k]);
        atomicAdd(gr.gamma[n][j] + k, sm[NF_MD_GB + j * NF_MD_XW + n * 64 + 32 + k]);
    }
    if (htot != nullptr && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        atomicAdd(h.g_c, htot[0]);                        // d/d s_bias
        atomicAdd(h.g_a, htot[1]);                        // d/d s_log_scale
    }
}

static int nf_maf_ok(int64_t N, int D) { return D >= 1 && D <= 4 && N <= NF_MAF_MAX_ROWS; }

static void nf_maf_head(const void* const* t, NfMafV& h) {
    h.log_gamma = (const float*)t[0]; h.bn_beta = (const float*)t[1]; h.bmean = (float*)t[2]; h.bvar = (float*)t[3];
    h.rmean = (float*)t[4]; h.rvar = (float*)t[5]; h.perm = (const float*)t[6]; h.a = (const float*)t[7]; h.c = (const float*)t[8];
}

extern "C" int nf_maf_step_fwd(const float* z, float* y, float* ld, const void* const* head, const void* const* made_params,
                               float* save_stats, float* ws_zero, int64_t N, int D, float flow_bn_eps, float flow_bn_momentum,
                               float bn_eps, nf_stream_t stream) {
    if (z == nullptr || y == nullptr || ld == nullptr || head == nullptr || made_params == nullptr || save_stats == nullptr ||
        ws_zero == nullptr || !nf_maf_ok(N, D))
        return NF_E_BADARG;
    if (N <= 0) return N == 0 ? 0 : NF_E_BADARG;
    NfMadeP p;
    nf_made_unpack(made_params, p);
    NfMafV h{};
    nf_maf_head(head, h);
    h.z = z; h.y = y; h.ld = ld; h.D = D; h.eps = flow_bn_eps; h.mom = flow_bn_momentum;
    const unsigned grid = (unsigned)((N + NF_MAF_ROWS_PER_BLOCK - 1) / NF_MAF_ROWS_PER_BLOCK);
    const size_t lds = nf_md_lds_bytes(1);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_maf_step_fwd<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute((const void*)k_maf_step_fwd<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    if (flow_bn_momentum == NF_FBN_RUNNING)             // evaluation mode: running statistics everywhere, nothing updated
        hipLaunchKernelGGL(k_maf_step_fwd<1>, dim3(grid), dim3(NF_MD_THREADS), lds, (hipStream_t)stream, p, h, save_stats, ws_zero, N,
                           bn_eps, 0);
    else
        hipLaunchKernelGGL(k_maf_step_fwd<0>, dim3(grid), dim3(NF_MD_THREADS), lds, (hipStream_t)stream, p, h, save_stats, ws_zero, N,
                           bn_eps, 1);
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_maf_step_inv(const float* y, float* z, float* ld, const void* const* head, const void* const* made_params,
                               float* ws_zero, int64_t N, int D, int training, float bn_eps, nf_stream_t stream) {
    if (y == nullptr || z == nullptr || ld == nullptr || head == nullptr || made_params == nullptr || ws_zero == nullptr ||
        !nf_maf_ok(N, D))
        return NF_E_BADARG;
    if (N <= 0) return N == 0 ? 0 : NF_E_BADARG;
    NfMadeP p;
    nf_made_unpack(made_params, p);
    NfMafV h{};
    nf_maf_head(head, h);
    h.z = y; h.y = z; h.ld = ld; h.D = D;
    const unsigned grid = (unsigned)((N + NF_MAF_ROWS_PER_BLOCK - 1) / NF_MAF_ROWS_PER_BLOCK);
    const size_t lds = nf_md_lds_bytes(1);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_maf_step_fwd<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_maf_step_fwd<2>, dim3(grid), dim3(NF_MD_THREADS), lds, (hipStream_t)stream, p, h, (float*)nullptr, ws_zero, N,
                       bn_eps, training ? 1 : 0);
    NF_CHECK_LAUNCH();
    return 0;
}

static int nf_maf_launch_bwd(const float* z, const float* g_y, const float* g_ld, float* g_z, const void* const* head,
                             const void* const* made_params, const float* save_stats, void* const* made_grads, float* g_s_log_scale,
                             float* g_s_bias, float* ws_zero, float* slabs, float* head_rec, int64_t N, int D, nf_stream_t stream) {
    if (z == nullptr || g_y == nullptr || g_z == nullptr || head == nullptr || made_params == nullptr || save_stats == nullptr ||
        made_grads == nullptr || ws_zero == nullptr || slabs == nullptr || !nf_maf_ok(N, D))
        return NF_E_BADARG;
    if (N <= 0) return N == 0 ? 0 : NF_E_BADARG;
    NfMadeP p;
    nf_made_unpack(made_params, p);
    NfMadeG g;
    for (int n = 0; n < 2; ++n) {
        void* const* q = made_grads + n * (2 * NF_MD_NL + 2 * NF_MD_NB);
        for (int l = 0; l < NF_MD_NL; ++l) { g.w[n][l] = (float*)q[2 * l]; g.b[n][l] = (float*)q[2 * l + 1]; }
        for (int j = 0; j < NF_MD_NB; ++j) { g.gamma[n][j] = (float*)q[2 * NF_MD_NL + 2 * j]; g.beta[n][j] = (float*)q[2 * NF_MD_NL + 2 * j + 1]; }
    }
    NfMafV h{};
    nf_maf_head(head, h);
    h.z = z; h.g_y = g_y; h.g_ld = g_ld; h.g_z = g_z; h.D = D; h.g_a = g_s_log_scale; h.g_c = g_s_bias;
    const unsigned grid = (unsigned)((N + NF_MAF_ROWS_PER_BLOCK - 1) / NF_MAF_ROWS_PER_BLOCK);
    const size_t lds = nf_md_lds_bytes(5);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_maf_step_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_maf_step_bwd, dim3(grid), dim3(NF_MD_THREADS), lds, (hipStream_t)stream, p, h, save_stats, g, ws_zero, slabs, N,
                       head_rec);
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_maf_step_bwd(const float* z, const float* g_y, const float* g_ld, float* g_z, const void* const* head,
                               const void* const* made_params, const float* save_stats, void* const* made_grads, float* g_s_log_scale,
                               float* g_s_bias, float* ws_zero, float* slabs, int64_t N, int D, nf_stream_t stream) {
    if (g_s_log_scale == nullptr || g_s_bias == nullptr) return NF_E_BADARG;
    return nf_maf_launch_bwd(z, g_y, g_ld, g_z, head, made_params, save_stats, made_grads, g_s_log_scale, g_s_bias, ws_zero, slabs,
                             nullptr, N, D, stream);
}

extern "C" int nf_maf_step_bwd_partial(const float* z, const float* g_y, const float* g_ld, float* g_z, const void* const* head,
                                       const void* const* made_params, const float* save_stats, void* const* made_grads,
                                       float* ws_zero, float* slabs_step, float* head_rec_step, int64_t N, int D,
                                       nf_stream_t stream) {
    if (head_rec_step == nullptr) return NF_E_BADARG;
    return nf_maf_launch_bwd(z, g_y, g_ld, g_z, head, made_params, save_stats, made_grads, nullptr, nullptr, ws_zero, slabs_step,
                             head_rec_step, N, D, stream);
}

// the folds of up to NF_MD_FOLD_STEPS deferred steps in one launch: half wave per (step, net, layer, column | bias, part of the
// workgroups' slabs); the parts meet by atomics in the gradient (+= semantics, as in the in-kernel fold)
#define NF_MD_FOLD_STEPS 16
#define NF_MD_FOLD_PARTS 4
struct NfMafFoldStep { const float* m[2][NF_MD_NL]; float* w[2][NF_MD_NL]; float* b[2][NF_MD_NL]; float* g_a; float* g_c; };
struct NfMafFoldArgs { NfMafFoldStep st[NF_MD_FOLD_STEPS]; };
static_assert(sizeof(NfMafFoldArgs) <= 3584, "fold descriptors travel in the kernel argument segment");

__global__ void __launch_bounds__(NF_MD_THREADS) k_maf_fold_all(NfMafFoldArgs args, const float* __restrict__ slabs_all,
                                                                const float* __restrict__ head_rec, int G, int D) {
    const NfMafFoldStep& st = args.st[blockIdx.y];
    const float* slabs = slabs_all + (size_t)blockIdx.y * G * NF_MD_SLAB;
    const int o = threadIdx.x & 31;
    // deterministic mode: ONE half wave sums all G slabs of a column in workgroup order (no parts that meet by atomics)
    const int parts = nf_det_on(nf_mdc_det) ? 1 : NF_MD_FOLD_PARTS;
    const int chunk = (G + parts - 1) / parts;
    constexpr int HW = NF_MD_THREADS / 32;
    for (int u = blockIdx.x * HW + (threadIdx.x >> 5); u < 2 * NF_MD_NL * 33 * parts; u += gridDim.x * HW) {
        const int part = u % parts, uu = u / parts;
        const int nl = uu / 33, i = uu - nl * 33;         // i == 32: the bias
        const int n = nl >> 2, l = nl & 3;
        const int I = l == 0 ? D : 32, O = l == NF_MD_NL - 1 ? D : 32;
        const int b_lo = part * chunk, b_hi = min(G, b_lo + chunk);
        if ((i < 32 && i >= I) || b_lo >= b_hi) continue;
        float tsum = 0.f;
        const float* base = slabs + (size_t)nl * NF_MD_SLAB_L + i * 32 + o;
        for (int b0 = b_lo; b0 < b_hi; b0 += 16) {        // 32 independent loads in flight
            float v[16][NF_MD_NKQ];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int b = b0 + q < b_hi ? b0 + q : b_hi - 1;
#pragma unroll
                for (int k = 0; k < NF_MD_NKQ; ++k) v[q][k] = base[(size_t)b * NF_MD_SLAB + k * NF_MD_SLAB_Q];
            }
#pragma unroll
            for (int q = 0; q < 16; ++q)
#pragma unroll
                for (int k = 0; k < NF_MD_NKQ; ++k)
                    if (b0 + q < b_hi) tsum += v[q][k];
        }
        if (o < O) {
            if (i == 32) atomicAdd(st.b[n][l] + o, tsum);
            else atomicAdd(st.w[n][l] + o * I + i, tsum * st.m[n][l][o * I + i]);             // maf.py:54: d(W * M) = g * M
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) {            // the transform's scalar gradients: sum over workgroups and waves
        const float* r = head_rec + (size_t)blockIdx.y * G * NF_MD_WAVES * 2;
        float s0 = 0.f, s1 = 0.f;
        for (int e = threadIdx.x; e < G * NF_MD_WAVES; e += 64) { s0 += r[2 * e]; s1 += r[2 * e + 1]; }
        s0 = nf_wave_sum(s0); s1 = nf_wave_sum(s1);
        if (threadIdx.x == 0) {                           // (one writer per address and launch: += into the gradient, no race)
            atomicAdd(st.g_c, s0);                        // d/d s_bias
            atomicAdd(st.g_a, s1);                        // d/d s_log_scale
        }
    }
}

extern "C" int nf_maf_fold_all(const void* const* made_params_all, void* const* made_grads_all, float* const* g_s_log_scale_all,
                               float* const* g_s_bias_all, int S, const float* slabs_all, const float* head_rec_all, int blocks,
                               int D, nf_stream_t stream) {
    if (made_params_all == nullptr || made_grads_all == nullptr || g_s_log_scale_all == nullptr || g_s_bias_all == nullptr || S < 1 ||
        slabs_all == nullptr || head_rec_all == nullptr || blocks < 1 || blocks > NF_MAF_MAX_BLOCKS || D < 1 || D > 4)
        return NF_E_BADARG;
    constexpr int jobs = 2 * NF_MD_NL * 33 * NF_MD_FOLD_PARTS, HW = NF_MD_THREADS / 32;
    for (int s0 = 0; s0 < S; s0 += NF_MD_FOLD_STEPS) {
        const int ns = S - s0 < NF_MD_FOLD_STEPS ? S - s0 : NF_MD_FOLD_STEPS;
        NfMafFoldArgs a{};
        for (int k = 0; k < ns; ++k) {
            NfMadeP p;
            nf_made_unpack(made_params_all + (size_t)(s0 + k) * NF_MAF_PARAM_PTRS, p);
            void* const* q0 = made_grads_all + (size_t)(s0 + k) * NF_MAF_GRAD_PTRS;
            for (int n = 0; n < 2; ++n) {
                void* const* q = q0 + n * (2 * NF_MD_NL + 2 * NF_MD_NB);
                for (int l = 0; l < NF_MD_NL; ++l) {
                    a.st[k].m[n][l] = p.m[n][l]; a.st[k].w[n][l] = (float*)q[2 * l]; a.st[k].b[n][l] = (float*)q[2 * l + 1];
                }
            }
            a.st[k].g_a = g_s_log_scale_all[s0 + k]; a.st[k].g_c = g_s_bias_all[s0 + k];
        }
        hipLaunchKernelGGL(k_maf_fold_all, dim3((jobs + HW - 1) / HW, ns), dim3(NF_MD_THREADS), 0, (hipStream_t)stream, a,
                           slabs_all + (size_t)s0 * blocks * NF_MD_SLAB, head_rec_all + (size_t)s0 * blocks * NF_MD_WAVES * 2, blocks, D);
    }
    NF_CHECK_LAUNCH();
    return 0;
}

NF_PERSIST_HOST_API(nf_md)

__attribute__((visibility("hidden"))) int nf_md_persist_capacity(int* blocks) {
    int dev = 0, cus = 0, per_cu = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const size_t lds = nf_md_lds_bytes(5);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_maf_step_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_maf_step_bwd, NF_MD_THREADS, lds);
    if (e != hipSuccess) return (int)e;
    *blocks = per_cu * cus;
    return 0;
}
