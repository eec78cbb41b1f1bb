// The RealNVP density flow (flows/realnvp.py:49-53: S x [BatchNorm(affine=False), AffineCoupling around the MLP conditioner of
// flows/modules.py:391-413]) for batches of N <= 256 rows with D = 2 -- config 1 of BASELINE.json -- as ONE workgroup per direction.
//
// Why: at 256 rows the grid kernels of mlp_chain.hip are two workgroups that meet in global memory twelve times per flow step (six
// training-mode BatchNorm reductions forward, six backward): 1.5 us per meeting, 0.58 of the 1.49 ms of a train step.  One workgroup
// needs no global meeting at all -- if the whole batch fits its registers.  It does in the TRANSPOSED form:
//   * eight waves, two per SIMD (256 registers each: one hides the other's LDS / shuffle / matrix latencies); wave w owns the 32
//     batch columns 32 w .. 32 w + 31;
//   * an activation is features x batch: lane (c32, hs) holds column 32 w + c32 (both halves of a wave the same 32 columns), its sixteen registers the features
//     fm(r, hs) = (r & 3) + 8 (r >> 2) + 4 hs -- the C / D layout of v_mfma_f32_32x32x2_f32;
//   * a linear layer is D = W x act with the WEIGHTS as the A operand: K step r contracts the features fm(r, 0), fm(r, 1), so the B
//     operand of K step r is register r of the previous layer's result, as it lies -- layers chain register to register, forward
//     (A = v[o][fm]) and for the data gradient (A = v[fm][i]); no LDS between two layers;
//   * sums over the batch (BatchNorm statistics, BatchNorm-backward sums) go through the wave's LDS transposition tile (so_transpose) and
//     one LDS meeting of the eight waves: two barriers, no global traffic; statistics are merged as (count, mean, M2) with a per-wave
//     centre, so one pass is as exact as two;
//   * weight gradients contract over the batch: both operands are transposed through a wave-private LDS tile (32 writes + 8 reads of
//     16 bytes per tensor), every wave leaves its own partial product in the slab format of mlp_chain.hip (wave w = region w >> 1, row
//     group w & 1 there: NF_FLOW_SOLO_REGIONS = 4 regions per step), so the deferred fold (k_glow_fold_all) and the weight-norm backward are shared unchanged.
// Numerics are those of mlp_chain.hip / linear_bn.hip: fp32 MFMA products, biased variance for normalisation, unbiased for the running
// estimate, weight-norm as a scale of the activation column (weight_norm.py:40), flow-BatchNorm statistics as buffers (no gradient
// through them, modules.py:285-296).
#include <cstdlib>

#include "nf_common.h"

#include "nf_conv_core.h"

#include "nf_flow_rec.h"

// phase stamps of thread 0 (tools/probes/solo_prof.py builds this file with -DNF_SO_PROF=1; 100 MHz wall clock): slots 0 .. 31 of the
// second forward step, 32 .. 63 of the backward's second step
#ifdef NF_SO_PROF
__device__ long long nf_so_prof[64];
#define NF_SO_STAMP(on, i)                                                              \
    do {                                                                                \
        if ((on) && threadIdx.x == 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); nf_so_prof[i] = wall_clock64(); } \
    } while (0)
extern "C" int nf_so_prof_read(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(nf_so_prof), sizeof(long long) * 64); }
#else
#define NF_SO_STAMP(on, i)
#endif

#define NF_SO_WAVES 8
#define NF_SO_THREADS (NF_SO_WAVES * NF_WAVE)
#define NF_SO_ST 33                                   // row stride of a staged 32 x 32 weight matrix (odd: conflict-free both ways)
#define NF_SO_TS 36                                   // row stride of a transposition tile (16-byte rows, conflict-free b128 reads)
typedef float f32x4s __attribute__((ext_vector_type(4)));

// LDS map (floats)
#define SO_W 0                                        // [4][32 * 33] v of the hidden linears 1 .. 4
#define SO_V0 (SO_W + 4 * 32 * NF_SO_ST)              // [32] v_0[o][0]
#define SO_V5 (SO_V0 + 32)                            // [2][32] v_5
#define SO_WS (SO_V5 + 64)                            // [6][32] weight-norm column scales g_i / (||v[:, i]|| + eps)
#define SO_G (SO_WS + 192)                            // [6][32] gains
#define SO_B (SO_G + 192)                             // [6][32] biases
#define SO_GA (SO_B + 192)                            // [5][32] BatchNorm gamma
#define SO_BE (SO_GA + 160)                           // [5][32] BatchNorm beta
#define SO_BNC (SO_BE + 160)                          // [5][4][32] scale, shift, mean, invstd
#define SO_HD (SO_BNC + 640)                          // head: [0..1] exp(log_gamma) [2..3] beta [4..5] mean [6..7] sqrt(var) [8] log-det [9] a [10] c [12..15] raw ls, bs
#define SO_PAR (SO_HD + 32)                           // floats of ONE set of parameter tables (a multiple of 4: 16-byte words).  The forward
                                                      // leaves every step's set in memory as it used it (weight-norm scales, BatchNorm constants
                                                      // and flow-BatchNorm statistics included): the backward's data path loads that IMAGE straight
                                                      // into LDS (global_load_lds), a step ahead, into the set it is not reading -- no staging pass
#define SO_GROUPS (2 * NF_SO_WAVES)                   // a meeting's partials: one per wave half = 16 columns
#define SO_RED (2 * SO_PAR)                           // [16 groups][3][32] partial sums of a meeting (sum, squares / second sum, centre)
#define SO_TOT (SO_RED + SO_GROUPS * 3 * 32)                  // [3][32] totals of a meeting
#define SO_REC (SO_TOT + 96)                          // [2][record words as floats x 2]
#define SO_REC_WORDS ((int)((sizeof(NfGlowFlowStep) + 7) / 8))
#define SO_TILES (SO_REC + 2 * 2 * SO_REC_WORDS)      // [8 waves][2][32 * 36] transposition tiles (forward: the first of a wave's pair only)
static_assert((SO_PAR & 3) == 0 && SO_PAR == NF_FLOW_SOLO_TAB_FLOATS, "the table image: 16-byte words, the size of include/nfhip.h");
static_assert(SO_REC_WORDS <= NF_SO_THREADS, "one 8-byte word of a step record per thread");
static inline size_t nf_so_lds_bytes(bool bwd) { (void)bwd; return sizeof(float) * (size_t)(SO_TILES + NF_SO_WAVES * 2 * 32 * NF_SO_TS); }

__device__ __forceinline__ int so_fm(int r, int hs) { return (r & 3) + 8 * (r >> 2) + 4 * hs; }
// the lane's sixteen features of a 32-vector in LDS: four 16-byte reads
__device__ __forceinline__ void so_ldvec(const float* base, int hs, float (&out)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4s v = *(const f32x4s*)(base + 8 * q + 4 * hs);
#pragma unroll
        for (int k = 0; k < 4; ++k) out[4 * q + k] = v[k];
    }
}
// A operands from the staged matrix of hidden linear l (1 .. 4): forward v[o = c32][fm(r, hs)], data gradient v[fm(r, hs)][i = c32]
__device__ __forceinline__ void so_ld_afwd(const float* sm, int l, int c32, int hs, float (&a)[16]) {
    const float* W = sm + SO_W + (l - 1) * 32 * NF_SO_ST + c32 * NF_SO_ST;
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = W[so_fm(r, hs)];
}
__device__ __forceinline__ void so_ld_abwd(const float* sm, int l, int c32, int hs, float (&a)[16]) {
    const float* W = sm + SO_W + (l - 1) * 32 * NF_SO_ST + c32;
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = W[so_fm(r, hs) * NF_SO_ST];
}
// acc = A x B over the sixteen K steps; B is the lane's own registers (one accumulator: back-to-back products into the same
// registers issue without a gap, and a second one costs sixteen registers the backward does not have)
__device__ __forceinline__ void so_gemm(const float (&a)[16], const float (&bv)[16], f32x16& acc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], bv[r], acc, 0, 0, 0);
}
// The workgroup's meetings order LDS traffic only (nothing in a step is handed from wave to wave through global memory): a barrier
// that waits for the LDS counter alone, written out so that it stays one whatever fence a later toolchain attaches to
// __syncthreads() (hipcc of ROCm 7.2 emits exactly this pair for it on gfx950 -- checked on the ISA: no vmcnt wait -- so the
// parameter prefetch and the statistics / slab stores in flight are not waited for at a meeting either way).
__device__ __forceinline__ void so_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// sum over the 32 lanes of a wave half (every lane ends with the total)
__device__ __forceinline__ float so_half_sum(float v) {
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, NF_WAVE);
    return v;
}

// ---- what a thread prefetches of a step's parameters (all loads of a step in flight at once, one step ahead) ----------------------
struct SoParams {
    float w[4][2];                                    // v_1 .. v_4, elements t + 512 e (consecutive lanes, consecutive addresses: no alignment demand)
    float g, b;                                       // t < 192: gain / bias of linear t >> 5, index t & 31
    float ga, be, rm, rv;                             // t < 160: BatchNorm t >> 5
    float x0, x1;                                     // 192 <= t < 224: v_0[t - 192]; 224 <= t < 256: v_5[0 .. 1][t - 224]; 160 <= t < 192: head scalars
    long long nbt;                                    // t = 32 j: num_batches_tracked of BatchNorm j
};
// Every slot gets exactly ONE unconditional load from a selected address (a slot that is first zeroed and then conditionally loaded
// makes the compiler drain the vector-memory counter before the zeroing write: the step then waits for its own prefetch);
// threads outside a table's range read a valid dummy element that so_stage ignores.
__device__ __forceinline__ void so_prefetch(SoParams& P, const NfGlowFlowStep& st, int t, const void* dummy8) {
    const int k = t & 31;
#pragma unroll
    for (int l = 0; l < 4; ++l)
#pragma unroll
        for (int e = 0; e < 2; ++e) P.w[l][e] = st.p.v[l + 1][t + NF_SO_THREADS * e];
    {
        const int l = min(t >> 5, 5);
        const int I = l == 0 ? 1 : 32, O = l == 5 ? 2 : 32;
        P.g = st.p.g[l][k < I ? k : 0];
        P.b = st.p.b[l][k < O ? k : 0];
    }
    {
        const int j = min(t >> 5, 4);
        P.ga = st.p.gamma[j][k]; P.be = st.p.beta[j][k]; P.rm = st.p.rmean[j][k]; P.rv = st.p.rvar[j][k];
        const NF_G int64_t* np = st.p.nbt[j] != nullptr ? st.p.nbt[j] : (const NF_G int64_t*)dummy8;
        P.nbt = np[0];
    }
    {
        const int q = t - 160;
        const NF_G float* p0 = st.h.a;
        const NF_G float* p1 = st.h.a;
        if (q >= 0 && q < 2) p0 = st.h.ls + q;
        else if (q >= 2 && q < 4) p0 = st.h.bs + (q - 2);
        else if (q == 5) p0 = st.h.c;
        else if (q >= 6 && q < 8) p0 = st.h.rmean + (q - 6);
        else if (q >= 8 && q < 10) p0 = st.h.rvar + (q - 8);
        else if (t >= 192 && t < 224) p0 = st.p.v[0] + (t - 192);
        else if (t >= 224 && t < 256) { p0 = st.p.v[5] + (t - 224); p1 = st.p.v[5] + 32 + (t - 224); }
        P.x0 = p0[0];
        P.x1 = p1[0];
    }
}
// registers -> LDS tables; the caller brackets it with barriers
__device__ __forceinline__ void so_stage(float* sm, const SoParams& P, int t) {
    const int k = t & 31;
#pragma unroll
    for (int l = 0; l < 4; ++l)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int idx = t + NF_SO_THREADS * e;
            sm[SO_W + l * 32 * NF_SO_ST + (idx >> 5) * NF_SO_ST + (idx & 31)] = P.w[l][e];
        }
    if (t < 192) { sm[SO_G + t] = P.g; sm[SO_B + t] = P.b; }
    if (t < 160) { sm[SO_GA + t] = P.ga; sm[SO_BE + t] = P.be; }
    else if (t < 192) {
        const int q = t - 160;
        if (q < 2) { sm[SO_HD + q] = expf(P.x0); sm[SO_HD + 12 + q] = P.x0; }
        else if (q < 4) sm[SO_HD + q] = P.x0;
        else if (q == 4) sm[SO_HD + 9] = P.x0;
        else if (q == 5) sm[SO_HD + 10] = P.x0;
    } else if (t < 224) sm[SO_V0 + k] = P.x0;
    else if (t < 256) { sm[SO_V5 + k] = P.x0; sm[SO_V5 + 32 + k] = P.x1; }
}
// the weight-norm column scales of the six linears (threads 256 .. 447: the upper waves, which stage nothing else; after the
// staging barrier)
__device__ __forceinline__ void so_weight_norm(float* sm, int t, float wn_eps) {
    t -= 256;
    if (t < 0 || t >= 192) return;
    const int l = t >> 5, k = t & 31;
    float ss = 0.f;
    if (l == 0) {
#pragma unroll 8
        for (int o = 0; o < 32; ++o) ss = fmaf(sm[SO_V0 + o], sm[SO_V0 + o], ss);
    } else if (l == 5) {
        ss = sm[SO_V5 + k] * sm[SO_V5 + k] + sm[SO_V5 + 32 + k] * sm[SO_V5 + 32 + k];
    } else {
        const float* W = sm + SO_W + (l - 1) * 32 * NF_SO_ST + k;
#pragma unroll 8
        for (int o = 0; o < 32; ++o) ss = fmaf(W[o * NF_SO_ST], W[o * NF_SO_ST], ss);
    }
    const bool live = l != 0 || k == 0;
    sm[SO_WS + t] = live ? sm[SO_G + t] / (sqrtf(ss) + wn_eps) : 0.f;
}

// ---- meetings of the eight waves -------------------------------------------------------------------------------------------------
// Sums over the batch go through the wave's transposition tile instead of cross-lane shuffles: sixteen 4-byte writes (feature rows,
// the lane's column), four 16-byte reads (the lane's FEATURE c32, columns 16 hs .. + 15), fifteen adds -- a butterfly over the lanes
// was ~100 vector instructions + 16 LDS permutes per array, and the step was bound by vector-instruction issue.  A wave half
// (16 columns) is one GROUP of a meeting; the totals thread / the statistics merge sums the sixteen groups.
__device__ __forceinline__ void so_transpose(float* tile, const float (&v)[16], float (&x)[16], int c32, int hs) {
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[so_fm(r, hs) * NF_SO_TS + c32] = v[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f32x4s q = *(const f32x4s*)(tile + c32 * NF_SO_TS + 16 * hs + 4 * k);
#pragma unroll
        for (int e = 0; e < 4; ++e) x[4 * k + e] = q[e];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// columns of group g (wave g >> 1, half g & 1) inside the batch
__device__ __forceinline__ int so_group_rows(int N, int g) { return min(max(N - 16 * g, 0), 16); }
// NQ per-lane feature arrays -> SO_TOT[q][feature]: transposition, LDS, two barriers
template <int NQ>
__device__ __forceinline__ void so_colsums(float* sm, const float (&p)[NQ][16], int c32, int hs, int wid) {
    float* tile = sm + SO_TILES + (wid * 2) * 32 * NF_SO_TS;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        float x[16];
        so_transpose(tile, p[q], x, c32, hs);
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; i += 4) t += (x[i] + x[i + 1]) + (x[i + 2] + x[i + 3]);
        sm[SO_RED + ((2 * wid + hs) * 3 + q) * 32 + c32] = t;
    }
    so_barrier();
    if (threadIdx.x < NQ * 32) {
        const int q = threadIdx.x >> 5, f = threadIdx.x & 31;
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < SO_GROUPS; ++g) t += sm[SO_RED + (g * 3 + q) * 32 + f];
        sm[SO_TOT + q * 32 + f] = t;
    }
    so_barrier();
}
// (count, mean, M2) of the batch from the groups' centred sums (slot 0: sum (x - c_g), 1: sum (x - c_g)^2, 2: c_g), feature / channel f:
//   mean_g = c_g + S1_g / n_g,  mean = sum n_g mean_g / N,  M2 = sum (S2_g - S1_g^2 / n_g) + sum n_g (mean_g - mean)^2
// -- no sequential merge, one division
template <int GROUPS, int ROWS>
__device__ __forceinline__ void so_merge_stats(const float* sm, int f, int N, float& mean, float& M2) {
    float mw[GROUPS], nw[GROUPS], acc = 0.f, m2 = 0.f;
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) {
        nw[g] = (float)min(max(N - ROWS * g, 0), ROWS);
        const float inv = nw[g] > 0.f ? __frcp_rn(nw[g]) : 0.f;
        const int slot = (ROWS == 16 ? g : 2 * g) * 3;
        const float s1 = sm[SO_RED + (slot + 0) * 32 + f], s2 = sm[SO_RED + (slot + 1) * 32 + f], cw = sm[SO_RED + (slot + 2) * 32 + f];
        mw[g] = nw[g] > 0.f ? cw + s1 * inv : 0.f;
        acc = fmaf(nw[g], mw[g], acc);
        m2 += nw[g] > 0.f ? fmaxf(s2 - s1 * s1 * inv, 0.f) : 0.f;
    }
    mean = acc / (float)N;
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) m2 = fmaf(nw[g] * (mw[g] - mean), mw[g] - mean, m2);
    M2 = m2;
}

// training-mode BatchNorm j over a (= the producing linear's output incl. bias and residual): statistics of the whole batch, constants
// -> SO_BNC, bookkeeping (running statistics, saved mean / invstd) by the finalising threads 32 j .. 32 j + 31, which hold the old
// running statistics
__device__ __forceinline__ void so_batchnorm_train(float* sm, int j, const float (&a)[16], int N, float eps, float mom, float rm_old,
                                                   float rv_old, long long nbt_old, const NfGlowFlowStep& st, float* save, int c32, int hs,
                                                   int wid) {
    {
        // the lane's feature over its group's 16 columns, centred at the group's first column (which exists whenever the group has any)
        float x[16];
        so_transpose(sm + SO_TILES + (wid * 2) * 32 * NF_SO_TS, a, x, c32, hs);
        const int nv = so_group_rows(N, 2 * wid + hs);
        const float c = x[0];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 1; i < 16; ++i) {
            const float d = i < nv ? x[i] - c : 0.f;
            s1 += d;
            s2 = fmaf(d, d, s2);
        }
        float* red = sm + SO_RED + ((2 * wid + hs) * 3) * 32 + c32;
        red[0] = s1; red[32] = s2; red[64] = c;
    }
    so_barrier();
    const int t = threadIdx.x;
    if ((t >> 5) == j) {                              // one half wave finalises
        const int f = t & 31;
        float mean, M2;
        so_merge_stats<SO_GROUPS, 16>(sm, f, N, mean, M2);
        const float var = M2 / (float)N;               // biased, as BatchNorm normalises
        const float invstd = 1.f / sqrtf(var + eps);
        const float sc = sm[SO_GA + j * 32 + f] * invstd;
        sm[SO_BNC + (4 * j + 0) * 32 + f] = sc;
        sm[SO_BNC + (4 * j + 1) * 32 + f] = sm[SO_BE + j * 32 + f] - mean * sc;
        sm[SO_BNC + (4 * j + 2) * 32 + f] = mean;
        sm[SO_BNC + (4 * j + 3) * 32 + f] = invstd;
        const float unb = N > 1 ? var * ((float)N / (float)(N - 1)) : var;
        st.p.rmean[j][f] = (1.f - mom) * rm_old + mom * mean;
        st.p.rvar[j][f] = (1.f - mom) * rv_old + mom * unb;
        save[(2 * j + 0) * 32 + f] = mean;
        save[(2 * j + 1) * 32 + f] = invstd;
        if (f == 0 && st.p.nbt[j] != nullptr) st.p.nbt[j][0] = nbt_old + 1;      // (no load here: it would wait for the prefetch in flight)
    }
    so_barrier();
}

// the step's stash of BatchNorm inputs: [layer 0 .. 4][quad of registers 0 .. 3][thread] 16-byte words
__device__ __forceinline__ void so_stash_put(f32x4s* st4, int l, const float (&a)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) st4[(l * 4 + q) * NF_SO_THREADS] = f32x4s{a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
}
__device__ __forceinline__ void so_stash_get(const f32x4s* st4, int l, float (&a)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4s v = st4[(l * 4 + q) * NF_SO_THREADS];
#pragma unroll
        for (int k = 0; k < 4; ++k) a[4 * q + k] = v[k];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// forward, training mode
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(NF_SO_THREADS) k_solo_fwd(const NfGlowFlowStep* __restrict__ steps, int S, const float* __restrict__ z0,
                                                            float* __restrict__ ys, float* __restrict__ ld, float* __restrict__ saves,
                                                            int save_stride, f32x4s* __restrict__ stash, f32x4s* __restrict__ tabs, int N,
                                                            float eps, float mom, float wn_eps) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int t = threadIdx.x, lane = t & 63, wid = __builtin_amdgcn_readfirstlane(t >> 6), c32 = lane & 31, hs = lane >> 5;
    unsigned long long (*rec)[SO_REC_WORDS] = reinterpret_cast<unsigned long long (*)[SO_REC_WORDS]>(sm + SO_REC);
    const bool rt = t < SO_REC_WORDS;
    if (rt) {
        rec[0][t] = reinterpret_cast<const unsigned long long*>(steps)[t];
        if (S > 1) rec[1][t] = reinterpret_cast<const unsigned long long*>(steps + 1)[t];
    }
    const int col = 32 * wid + c32;
    const bool cv = col < N;
    const int cc = cv ? col : 0;
    float z[2], ldv;
    z[0] = cv ? z0[2 * cc] : 0.f;
    z[1] = cv ? z0[2 * cc + 1] : 0.f;
    ldv = cv ? ld[cc] : 0.f;
    so_barrier();
    SoParams P;
    so_prefetch(P, *reinterpret_cast<const NfGlowFlowStep*>(rec[0]), t, steps);
#pragma unroll 1
    for (int s = 0; s < S; ++s) {
        const NfGlowFlowStep& st = *reinterpret_cast<const NfGlowFlowStep*>(rec[s & 1]);
        float* save = saves + (size_t)s * save_stride;
        NF_SO_STAMP(s == 1, 0);
        so_stage(sm, P, t);
        const float rm_old = P.rm, rv_old = P.rv, x0_old = P.x0;   // (the old running statistics of this step: the finalising threads need them)
        const long long nbt_old = P.nbt;
        const float frv_all = __shfl(P.x0, 40 + (t & 1), NF_WAVE);   // wave 2: lanes 38, 39 take the head's running variance of channel 0, 1
        // flow-BatchNorm partial sums of z around the wave's first column, in the same meeting as the staging barrier
        {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float cen = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, z[c]), 0));
                const float d = cv ? z[c] - cen : 0.f;
                const float q1 = so_half_sum(d), q2 = so_half_sum(d * d);
                if (lane == 0) {                       // (group slot 2 wid: one partial per wave here)
                    sm[SO_RED + (2 * wid * 3 + 0) * 32 + c] = q1;
                    sm[SO_RED + (2 * wid * 3 + 1) * 32 + c] = q2;
                    sm[SO_RED + (2 * wid * 3 + 2) * 32 + c] = cen;
                }
            }
        }
        so_barrier();
        unsigned long long nxt = 0;
        if (rt && s + 2 < S) nxt = reinterpret_cast<const unsigned long long*>(steps + s + 2)[t];
        if (s + 1 < S) so_prefetch(P, *reinterpret_cast<const NfGlowFlowStep*>(rec[(s + 1) & 1]), t, steps);
        so_weight_norm(sm, t, wn_eps);
        if (t == 166 || t == 167) {                   // flow BatchNorm (modules.py:283-296): batch statistics are buffers; these two threads
            const int c = t - 166;                     // hold the old running mean of channel c (their neighbours' lanes the variance)
            float mean, M2;
            so_merge_stats<NF_SO_WAVES, 32>(sm, c, N, mean, M2);
            const float var = M2 / (float)N + st.h.fbn_eps;                      // biased, eps inside (modules.py:287)
            sm[SO_HD + 4 + c] = mean;
            sm[SO_HD + 6 + c] = sqrtf(var);
            sm[SO_HD + 16 + c] = sm[SO_HD + 12 + c] - 0.5f * logf(var);           // this channel's log-det
            st.h.bmean[c] = mean; st.h.bvar[c] = var;
            st.h.rmean[c] = x0_old * (1.f - st.h.fbn_mom) + mean * st.h.fbn_mom;   // modules.py:291-294
            st.h.rvar[c] = frv_all * (1.f - st.h.fbn_mom) + var * st.h.fbn_mom;
            save[2 * NF_MC_NB * 32 + c] = mean; save[2 * NF_MC_NB * 32 + 4 + c] = var;
        }
        so_barrier();
        NF_SO_STAMP(s == 1, 1);
        // ---- head: h = exp(log_gamma) (z - mean) / sqrt(var) + beta; x = the conditioning feature ----
        const int sel = st.h.odd ? 1 : 0;             // the transformed feature; the other one conditions (squeeze.py:68-69)
        const float dld = sm[SO_HD + 16] + sm[SO_HD + 17];
        float h[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float zn = (z[c] - sm[SO_HD + 4 + c]) / sm[SO_HD + 6 + c];
            h[c] = fmaf(sm[SO_HD + c], zn, 0.f) + sm[SO_HD + 2 + c];
        }
        // ---- the conditioner ----
        float a[5][16];
        {
            float w0[16], b0[16];
            so_ldvec(sm + SO_V0, hs, w0);
            so_ldvec(sm + SO_B, hs, b0);
            const float x = (sel ? h[0] : h[1]) * sm[SO_WS];
#pragma unroll
            for (int r = 0; r < 16; ++r) a[0][r] = fmaf(w0[r], x, 0.f) + b0[r];
        }
        // the five BatchNorm inputs of the step are kept for the backward (no recompute there): 16-byte stores in register order,
        // consecutive lanes consecutive addresses (160 KB per step, resident in L2 / the Infinity Cache until the backward reads them)
        f32x4s* const st4 = stash + (size_t)s * (5 * 4 * NF_SO_THREADS) + t;
        so_stash_put(st4, 0, a[0]);
        NF_SO_STAMP(s == 1, 2);
        so_batchnorm_train(sm, 0, a[0], N, eps, mom, rm_old, rv_old, nbt_old, st, save, c32, hs, wid);
        NF_SO_STAMP(s == 1, 3);
#pragma unroll
        for (int l = 1; l < 5; ++l) {
            __builtin_amdgcn_sched_barrier(0);         // (no hoisting of a later layer's operand reads: the registers are full)
            float sc[16], sh[16], ws[16], bias[16], A[16], act[16];
            so_ldvec(sm + SO_BNC + (4 * (l - 1) + 0) * 32, hs, sc);
            so_ldvec(sm + SO_BNC + (4 * (l - 1) + 1) * 32, hs, sh);
            so_ldvec(sm + SO_WS + l * 32, hs, ws);
            so_ldvec(sm + SO_B + l * 32, hs, bias);
            so_ld_afwd(sm, l, c32, hs, A);
#pragma unroll
            for (int r = 0; r < 16; ++r) act[r] = fmaxf(fmaf(a[l - 1][r], sc[r], sh[r]), 0.f) * ws[r];
            f32x16 acc;
            so_gemm(A, act, acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[r];
                if ((l & 1) == 0) v += a[l - 2][r];
                a[l][r] = v + bias[r];
            }
            so_stash_put(st4, l, a[l]);
            NF_SO_STAMP(s == 1, 2 + 2 * l);
            so_batchnorm_train(sm, l, a[l], N, eps, mom, rm_old, rv_old, nbt_old, st, save, c32, hs, wid);
            NF_SO_STAMP(s == 1, 3 + 2 * l);
        }
        // ---- linear 5 (32 -> t | s_raw) on the vector ALU, the affine coupling (coupling.py:104-113) ----
        {
            float sc[16], sh[16], ws[16], v50[16], v51[16];
            so_ldvec(sm + SO_BNC + (4 * 4 + 0) * 32, hs, sc);
            so_ldvec(sm + SO_BNC + (4 * 4 + 1) * 32, hs, sh);
            so_ldvec(sm + SO_WS + 5 * 32, hs, ws);
            so_ldvec(sm + SO_V5, hs, v50);
            so_ldvec(sm + SO_V5 + 32, hs, v51);
            const float b50 = sm[SO_B + 5 * 32], b51 = sm[SO_B + 5 * 32 + 1], ca = sm[SO_HD + 9], cc2 = sm[SO_HD + 10];
            float tp = 0.f, sp = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float av = fmaxf(fmaf(a[4][r], sc[r], sh[r]), 0.f) * ws[r];
                tp = fmaf(v50[r], av, tp);
                sp = fmaf(v51[r], av, sp);
            }
            tp += __shfl_xor(tp, 32, NF_WAVE);
            sp += __shfl_xor(sp, 32, NF_WAVE);
            const float tt = tp + b50, sraw = sp + b51;
            const float sv = tanhf(sraw) * ca + cc2;
            const float h0 = sel ? h[1] : h[0], h1 = sel ? h[0] : h[1];
            const float y0 = h0 * expf(sv) + tt;
            z[sel] = y0;
            z[1 - sel] = h1;
            ldv += sv + dld;
            if (cv && hs == 0) {
                float* yr = ys + ((size_t)s * N + col) * 2;
                yr[0] = z[0]; yr[1] = z[1];
            }
        }
        // the step's table set as it stands -- parameters, weight-norm scales, the BatchNorm constants the finalising threads wrote, the flow
        // BatchNorm's statistics -- for the backward's data path (three 16-byte words per thread; every writer of the set passed a barrier)
        {
            f32x4s* const img = tabs + (size_t)s * (SO_PAR / 4);
            for (int i = t; i < SO_PAR / 4; i += NF_SO_THREADS) img[i] = reinterpret_cast<const f32x4s*>(sm)[i];
        }
        NF_SO_STAMP(s == 1, 12);
        so_barrier();                               // the tables are restaged by the next step; every reader of record s is through
        NF_SO_STAMP(s == 1, 13);
        if (rt) rec[s & 1][t] = nxt;                   // its slot takes step s + 2 (read behind the next step's first barrier)
    }
    if (cv && hs == 0) ld[col] = ldv;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// backward (training mode, deferred fold): forward recomputed from the step's input and the saved statistics, then the chain of
//   G_J -> weight / bias gradient partials of linear J (per wave, slab) -> data gradient -> ReLU mask -> two batch sums -> BatchNorm backward
// ---------------------------------------------------------------------------------------------------------------------------------
// a step's saved statistics and (accumulate) the old values of its BatchNorm gradient sinks: unconditional loads from clamped
// addresses, as in so_prefetch
__device__ __forceinline__ void so_saved(float& sv_mean, float& sv_inv, float& sv_h, float& gb_old, float& gg_old, const NfGlowFlowStep& st,
                                         const float* save, int t, int accumulate) {
    const int j = min(t >> 5, 4), f = t & 31;
    sv_mean = save[(2 * j) * 32 + f];
    sv_inv = save[(2 * j + 1) * 32 + f];
    const int q = t - 160;
    sv_h = save[2 * NF_MC_NB * 32 + ((q >= 0 && q < 2) ? q : ((q >= 2 && q < 4) ? 4 + q - 2 : 0))];
    const float b0 = st.g.beta[j][f], g0 = st.g.gamma[j][f];
    gb_old = accumulate ? b0 : 0.f;
    gg_old = accumulate ? g0 : 0.f;
}

// transposes G and act (sixteen registers each) through the wave's tiles and leaves the wave's partial of g_Weff[l] ([i][o]) + bias sums
__device__ __forceinline__ void so_wgrad(float* sm, float* slab_l, const float (&act)[16], const float (&G)[16], int c32, int hs, int wid) {
#ifdef NF_SO_NO_WGRAD                                  // (tools/probes/solo_prof.py --no-wgrad: what the backward costs WITHOUT its six
    if (slab_l != nullptr) return;                     //  weight-gradient products -- the bound of handing them to a second workgroup)
#endif
    float* TA = sm + SO_TILES + (wid * 2 + 0) * 32 * NF_SO_TS;
    float* TG = sm + SO_TILES + (wid * 2 + 1) * 32 * NF_SO_TS;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        TA[so_fm(r, hs) * NF_SO_TS + c32] = act[r];
        TG[so_fm(r, hs) * NF_SO_TS + c32] = G[r];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bsum = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {                      // K step (k, e): column 16 hs + 4 k + e of the wave, the same in both operands
        const f32x4s av = *(const f32x4s*)(TA + c32 * NF_SO_TS + 16 * hs + 4 * k);
        const f32x4s gv = *(const f32x4s*)(TG + c32 * NF_SO_TS + 16 * hs + 4 * k);
        bsum += (gv[0] + gv[1]) + (gv[2] + gv[3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], gv[e], acc, 0, 0, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();                   // (the tiles are rewritten by the next layer)
#pragma unroll
    for (int r = 0; r < 16; ++r) slab_l[so_fm(r, hs) * 32 + c32] = acc[r];       // D[i = fm][o = c32]
    bsum += __shfl_xor(bsum, 32, NF_WAVE);
    if (hs == 0) slab_l[1024 + c32] = bsum;
}

// the backward's table set of a step: the image the forward left in memory (SO_PAR floats), straight into LDS -- lane i's 16 bytes land
// at base + 16 i, one KiB per wave instruction, 23 of them over the eight waves; completion: s_waitcnt vmcnt(0) + a barrier
__device__ __forceinline__ void so_tables_dma(float* set, const float* __restrict__ img, int wid, int lane) {
    constexpr int CH = SO_PAR / 256;                  // whole KiB chunks ...
    for (int c = wid; c < CH; c += NF_SO_WAVES)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(img + c * 256 + lane * 4),
                                         (__attribute__((address_space(3))) void*)(set + c * 256), 16, 0, 0);
    if (wid == NF_SO_WAVES - 1 && lane < (SO_PAR - CH * 256) / 4)      // ... and the tail
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(img + CH * 256 + lane * 4),
                                         (__attribute__((address_space(3))) void*)(set + CH * 256), 16, 0, 0);
}
// (accumulate) the old values of a step's BatchNorm gradient sinks, threads 32 j + f: unconditional loads from clamped addresses
__device__ __forceinline__ void so_sinks_old(float& gb_old, float& gg_old, const NfGlowFlowStep& st, int t, int accumulate) {
    const int j = min(t >> 5, 4), f = t & 31;
    const float b0 = st.g.beta[j][f], g0 = st.g.gamma[j][f];
    gb_old = accumulate ? b0 : 0.f;
    gg_old = accumulate ? g0 : 0.f;
}

// ---- the weight-gradient workgroup of the backward ------------------------------------------------------------------------------------
// Wave w forms, step by step, the six products g_Weff[l] = sum over ITS 32 columns of act_l (x) G_l that wave w of the data path used
// to form between its meetings: G_0 .. G_4, the two rows of G_5 and the conditioner's input x arrive through gbuf (written by the data
// path's wave w, announced per step through flags[w]), what the linears multiplied with -- ReLU(BatchNorm(a_{l-1})) -- is recomputed
// from the forward's stash and the saved statistics.  The BatchNorm constants of ALL steps are staged once (S x 5 x [scale | shift] x
// 32 floats: 40 KB at S = 32), so the waves never meet: each one only follows its counterpart, a whole step behind at most by choice
// of nobody -- the data path never waits for this workgroup.  Slabs and bias sums in the format the data path wrote (k_glow_fold_all).
#define NF_SO_B_BLOCK 8                               // workgroups are dealt round-robin to the 8 XCDs: block 8 shares block 0's L2
#define NF_SO_XCC_REG ((3 << 11) | 20)                // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, size 4): the XCD this wave runs on
#define SO_WK_BN 0                                    // worker LDS: [S][5][2][32] BatchNorm scale | shift, then the transposition tiles
#define NF_SO_WK_MAX_STEPS 48                         // ... which bounds the run length the pair of workgroups takes (61 KB of tables)
NF_PERSIST_STATE(nf_so)                               // (a worker wave that gave up waiting for its counterpart: counted and reported like
NF_PERSIST_HOST_API(nf_so)                            //  every other bounded spin loop of the library, nf_persistent_timeouts)
__device__ __forceinline__ void so_wgrad_worker(float* sm, const NfGlowFlowStep* __restrict__ steps, int S, const f32x4s* __restrict__ stash,
                                                const f32x4s* __restrict__ gbuf, const unsigned* __restrict__ flags,
                                                const float* __restrict__ saves, int save_stride, float* __restrict__ slabs, int N) {
    const int t = threadIdx.x, lane = t & 63, wid = __builtin_amdgcn_readfirstlane(t >> 6), c32 = lane & 31, hs = lane >> 5;
    constexpr int regions = NF_FLOW_SOLO_REGIONS;
    float* const bn = sm + SO_WK_BN;
    float* const tiles = sm + SO_WK_BN + S * 5 * 64;                           // so_wgrad addresses sm + SO_TILES: rebased below
    for (int e = t; e < S * 160; e += NF_SO_THREADS) {                        // (step, BatchNorm j, feature f)
        const int s = e / 160, jf = e - s * 160, j = jf >> 5, f = jf & 31;
        const float* save = saves + (size_t)s * save_stride;
        const float sc = steps[s].p.gamma[j][f] * save[(2 * j + 1) * 32 + f];
        bn[(s * 5 + j) * 64 + f] = sc;
        bn[(s * 5 + j) * 64 + 32 + f] = steps[s].p.beta[j][f] - save[(2 * j) * 32 + f] * sc;
    }
    if (t == 0) {
        // both workgroups must share an XCD (its L2 is the only point of coherence the hand-over uses): the hardware deals workgroups to
        // the XCDs round-robin, so block 8 sits where block 0 does -- a launch for which that does not hold fails LOUDLY (the sticky error
        // word every persistent kernel of the library reports through), it does not return stale numbers
        unsigned a = 0, spins = 0;
        while (((a = __hip_atomic_load(flags + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 0x100u) == 0u) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > nf_so_spin_limit) break;
        }
        if ((a & 0x100u) == 0u || (a & 15u) != (__builtin_amdgcn_s_getreg(NF_SO_XCC_REG) & 15u)) NF_PERSIST_GIVE_UP(nf_so);
    }
    so_barrier();
    const int col = 32 * wid + c32;
    const bool cv = col < N;
    float* const smt = tiles - SO_TILES;                // so_wgrad(smt, ...) finds this wave's tiles at smt + SO_TILES
#pragma unroll 1
    for (int s = S - 1; s >= 0; --s) {
        {                                               // the data path's wave of the same columns is through with step s
            unsigned spins = 0;
            while (__hip_atomic_load(flags + wid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(S - s)) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > nf_so_spin_limit) { NF_PERSIST_GIVE_UP(nf_so); break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");      // (orders the loads below behind the flag for the compiler; every
                                                                        //  address is read once per launch: no stale line in this CU's L1)
        }
        const f32x4s* const st4 = stash + (size_t)s * (5 * 4 * NF_SO_THREADS) + t;
        const f32x4s* const gb4 = gbuf + (size_t)s * (6 * 4 * NF_SO_THREADS) + t;
        float* slab = slabs + ((size_t)s * regions + (wid >> 1)) * NF_MC_SLAB + (wid & 1) * NF_MC_SLAB_Q;   // + l * NF_MC_SLAB_L
        const f32x4s g5x = gb4[(5 * 4) * NF_SO_THREADS];
        float G[16], act[16], araw[16];
        // linear 5: rows o = 0 (t), 1 (s_raw) of G, the rest zero; it multiplied with ReLU(BatchNorm 4)
        so_stash_get(st4, 4, araw);
        {
            float sc[16], sh[16];
            so_ldvec(bn + (s * 5 + 4) * 64, hs, sc);
            so_ldvec(bn + (s * 5 + 4) * 64 + 32, hs, sh);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                act[r] = fmaxf(fmaf(araw[r], sc[r], sh[r]), 0.f);
                G[r] = (cv && hs == 0 && r == 0) ? g5x[0] : ((cv && hs == 0 && r == 1) ? g5x[1] : 0.f);
            }
        }
        so_wgrad(smt, slab + 5 * NF_MC_SLAB_L, act, G, c32, hs, wid);
#pragma unroll
        for (int J = 4; J >= 1; --J) {
            so_stash_get(gb4, J, G);
            so_stash_get(st4, J - 1, araw);
            float sc[16], sh[16];
            so_ldvec(bn + (s * 5 + J - 1) * 64, hs, sc);
            so_ldvec(bn + (s * 5 + J - 1) * 64 + 32, hs, sh);
#pragma unroll
            for (int r = 0; r < 16; ++r) act[r] = fmaxf(fmaf(araw[r], sc[r], sh[r]), 0.f);
            so_wgrad(smt, slab + J * NF_MC_SLAB_L, act, G, c32, hs, wid);
        }
        // linear 0 (1 -> 32): row i = 0 of [i][o], the transposed product with x in feature row 0
        so_stash_get(gb4, 0, G);
#pragma unroll
        for (int r = 0; r < 16; ++r) act[r] = (hs == 0 && r == 0) ? g5x[2] : 0.f;
        so_wgrad(smt, slab + 0 * NF_MC_SLAB_L, act, G, c32, hs, wid);
    }
}

__global__ void __launch_bounds__(NF_SO_THREADS) k_solo_bwd(const NfGlowFlowStep* __restrict__ steps, int S, const float* __restrict__ z0,
                                                            const float* __restrict__ ys, const float* __restrict__ g_y,
                                                            const float* __restrict__ g_ld, float* __restrict__ gzs,
                                                            const float* __restrict__ saves, int save_stride,
                                                            const f32x4s* __restrict__ stash, const float* __restrict__ tabs,
                                                            f32x4s* __restrict__ gbuf, unsigned* __restrict__ flags, int accumulate,
                                                            float* __restrict__ slabs, float* __restrict__ head_rec, int N, float wn_eps) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int t = threadIdx.x, lane = t & 63, wid = __builtin_amdgcn_readfirstlane(t >> 6), c32 = lane & 31, hs = lane >> 5;
    // TWO workgroups share the backward (round 5): block 0 walks the data path -- the serial chain of meetings and data-gradient
    // products --, block NF_SO_B_BLOCK forms the six weight-gradient products of every step from what block 0 leaves in memory
    // (so_wgrad_worker).  Nothing on the data path waits for a weight gradient: handing them over took 8.5 of the 24.7 us of a step off
    // the chain (tools/probes/solo_prof.py --no-wgrad).  The blocks in between exit at once.
    if (blockIdx.x == NF_SO_B_BLOCK) {
        so_wgrad_worker(sm, steps, S, stash, gbuf, flags, saves, save_stride, slabs, N);
        return;
    }
    if (blockIdx.x != 0) return;
    if (t == 0) __hip_atomic_store(flags + 8, 0x100u | (__builtin_amdgcn_s_getreg(NF_SO_XCC_REG) & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long (*rec)[SO_REC_WORDS] = reinterpret_cast<unsigned long long (*)[SO_REC_WORDS]>(sm + SO_REC);
    const bool rt = t < SO_REC_WORDS;
    constexpr int regions = NF_FLOW_SOLO_REGIONS;       // slab / head-sum regions per step: wave w writes region w >> 1, row group w & 1
    if (rt) {
        rec[(S - 1) & 1][t] = reinterpret_cast<const unsigned long long*>(steps + S - 1)[t];
        if (S > 1) rec[(S - 2) & 1][t] = reinterpret_cast<const unsigned long long*>(steps + S - 2)[t];
    }
    const int col = 32 * wid + c32;
    const bool cv = col < N;
    const int cc = cv ? col : 0;
    float gy[2], gld, zin[2];
    gy[0] = cv ? g_y[2 * cc] : 0.f;
    gy[1] = cv ? g_y[2 * cc + 1] : 0.f;
    gld = (cv && g_ld != nullptr) ? g_ld[cc] : 0.f;
    {
        const float* zr = S == 1 ? z0 : ys + (size_t)(S - 2) * N * 2;
        zin[0] = cv ? zr[2 * cc] : 0.f;
        zin[1] = cv ? zr[2 * cc + 1] : 0.f;
    }
    so_barrier();
    // the BatchNorm gradient sinks' old values (accumulate): threads 32 J + f, a step ahead like everything else
    float gb_old = 0.f, gg_old = 0.f;
    so_sinks_old(gb_old, gg_old, *reinterpret_cast<const NfGlowFlowStep*>(rec[(S - 1) & 1]), t, accumulate);
    // the last step's table image -> set (S - 1) & 1; every later one is requested a step ahead (below)
    so_tables_dma(sm + ((S - 1) & 1) * SO_PAR, tabs + (size_t)(S - 1) * SO_PAR, wid, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const float invN = 1.f / (float)N;
    so_barrier();
#pragma unroll 1
    for (int s = S - 1; s >= 0; --s) {
        const NfGlowFlowStep& st = *reinterpret_cast<const NfGlowFlowStep*>(rec[s & 1]);
        // this step's parameter tables, weight-norm scales, BatchNorm constants and flow-BatchNorm statistics: the image the forward left,
        // loaded straight into this set during the previous step (no staging pass, no prefetch registers, no weight-norm pass: the
        // step's top was 3.1 of its 17 us)
        float* const tb = sm + (s & 1) * SO_PAR;
        // the step's five BatchNorm inputs, as the forward left them (no recompute), through a SLIDING WINDOW: stage J of the data path
        // needs a[J] only (ReLU mask, xhat), so a[4] is requested here and a[J - 1] at the start of stage J: two of the five arrays are
        // live at any time (32 registers instead of 80)
        const f32x4s* const st4 = stash + (size_t)s * (5 * 4 * NF_SO_THREADS) + t;
        f32x4s* const gb4 = gbuf + (size_t)s * (6 * 4 * NF_SO_THREADS) + t;      // what the worker reads of this step: G_0 .. G_4, (g5, x)
        float a[5][16];
        so_stash_get(st4, 4, a[4]);
        NF_SO_STAMP(s == S - 2, 32);
        // next step's record, table image, sinks and input rows: in flight under this step
        unsigned long long nxt = 0;
        if (rt && s >= 2) nxt = reinterpret_cast<const unsigned long long*>(steps + s - 2)[t];
        float zn_[2] = {0.f, 0.f};
        const float gb_cur = gb_old, gg_cur = gg_old;
        if (s >= 1) {
            const NfGlowFlowStep& stn = *reinterpret_cast<const NfGlowFlowStep*>(rec[(s - 1) & 1]);
            so_tables_dma(sm + ((s - 1) & 1) * SO_PAR, tabs + (size_t)(s - 1) * SO_PAR, wid, lane);   // (that set's readers left at the last barrier)
            so_sinks_old(gb_old, gg_old, stn, t, accumulate);
            const float* zr = s == 1 ? z0 : ys + (size_t)(s - 2) * N * 2;
            zn_[0] = cv ? zr[2 * cc] : 0.f;
            zn_[1] = cv ? zr[2 * cc + 1] : 0.f;
        }
        NF_SO_STAMP(s == S - 2, 33);
        // ---- head (recomputed: two values per lane); the conditioner's activations come from the stash ----
        const int sel = st.h.odd ? 1 : 0;
        float h[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float zn = (zin[c] - tb[SO_HD + 4 + c]) / tb[SO_HD + 6 + c];
            h[c] = fmaf(tb[SO_HD + c], zn, 0.f) + tb[SO_HD + 2 + c];
        }
        const float xw = sel ? h[0] : h[1];
        NF_SO_STAMP(s == S - 2, 34);
        // ---- linear 5 and the coupling's backward (coupling.py:104-113) -> G of the conditioner output, gradient of h ----
        float tg[16];                                  // in turn: the gradient of a ReLU output, its masked form gn, G_J
        float Gh[2], hsum[2];                          // head sums: sum g_s, sum g_s tanh(s_raw)
        {
            // (three short passes over the lane's sixteen features instead of one that keeps eight arrays alive: the five
            // activations already hold 80 of the lane's registers)
            const float b51 = tb[SO_B + 5 * 32 + 1], ca = tb[SO_HD + 9], cc2 = tb[SO_HD + 10];
            float sp = 0.f;
            {
                float sc[16], sh[16], ws[16], v51[16];
                so_ldvec(tb + SO_BNC + (4 * 4 + 0) * 32, hs, sc);
                so_ldvec(tb + SO_BNC + (4 * 4 + 1) * 32, hs, sh);
                so_ldvec(tb + SO_WS + 5 * 32, hs, ws);
                so_ldvec(tb + SO_V5 + 32, hs, v51);
#pragma unroll
                for (int r = 0; r < 16; ++r) sp = fmaf(v51[r], fmaxf(fmaf(a[4][r], sc[r], sh[r]), 0.f) * ws[r], sp);
            }
            sp += __shfl_xor(sp, 32, NF_WAVE);
            const float th = tanhf(sp + b51);
            const float ev = expf(th * ca + cc2);
            const float h0 = sel ? h[1] : h[0];
            const float gy0 = cv ? (sel ? gy[1] : gy[0]) : 0.f, gy1 = cv ? (sel ? gy[0] : gy[1]) : 0.f;
            const float gsv = cv ? gy0 * h0 * ev + gld : 0.f;               // ld += s: the log-det gradient enters here
            const float gsraw = gsv * ca * (1.f - th * th);
            hsum[0] = gsv;
            hsum[1] = gsv * th;
            Gh[sel] = gy0 * ev;
            Gh[1 - sel] = gy1;
            __builtin_amdgcn_sched_barrier(0);
            {
                float ws[16], v50[16], v51[16];
                so_ldvec(tb + SO_WS + 5 * 32, hs, ws);
                so_ldvec(tb + SO_V5, hs, v50);
                so_ldvec(tb + SO_V5 + 32, hs, v51);
#pragma unroll
                for (int r = 0; r < 16; ++r) tg[r] = (v50[r] * gy0 + v51[r] * gsraw) * ws[r];
            }
            __builtin_amdgcn_sched_barrier(0);
            // linear 5's and linear 0's weight gradients are the worker's: it needs the two rows of G_5 (t, s_raw) and the conditioner's
            // input of this column
            gb4[(5 * 4) * NF_SO_THREADS] = f32x4s{gy0, gsraw, xw, 0.f};
        }
        NF_SO_STAMP(s == S - 2, 35);
        // ---- BatchNorm J backward, then linear J: J = 4 .. 0 ----
        float Gs[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) Gs[r] = 0.f;
        float gx = 0.f;
#pragma unroll
        for (int J = 4; J >= 0; --J) {
            __builtin_amdgcn_sched_barrier(0);
            if (J >= 1) so_stash_get(st4, J - 1, a[J - 1]);    // (needed one stage on: a stage is ~3 us, an L2 round trip well under 2)
            __builtin_amdgcn_sched_barrier(0);
            {
                // the meeting: sum gn, sum gn xhat over the batch (+ the head sums of the step on its first round); one array at a time
                float* tile = sm + SO_TILES + (wid * 2) * 32 * NF_SO_TS;
                float* red = sm + SO_RED + ((2 * wid + hs) * 3) * 32 + c32;
                if (J == 4) {
                    const float h0 = so_half_sum(hsum[0]), h1 = so_half_sum(hsum[1]);
                    if (lane == 0) { sm[SO_RED + (2 * wid * 3 + 2) * 32 + 0] = h0; sm[SO_RED + (2 * wid * 3 + 2) * 32 + 1] = h1; }
                }
                {
                    float sc[16], sh[16], x[16];
                    so_ldvec(tb + SO_BNC + (4 * J + 0) * 32, hs, sc);
                    so_ldvec(tb + SO_BNC + (4 * J + 1) * 32, hs, sh);
#pragma unroll
                    for (int r = 0; r < 16; ++r) tg[r] = (cv && fmaf(a[J][r], sc[r], sh[r]) > 0.f) ? tg[r] : 0.f;      // = gn
                    so_transpose(tile, tg, x, c32, hs);
                    float tsum = 0.f;
#pragma unroll
                    for (int i = 0; i < 16; i += 4) tsum += (x[i] + x[i + 1]) + (x[i + 2] + x[i + 3]);
                    red[0] = tsum;
                }
                __builtin_amdgcn_sched_barrier(0);
                {
                    float mean[16], invstd[16], gx_[16], x[16];
                    so_ldvec(tb + SO_BNC + (4 * J + 2) * 32, hs, mean);
                    so_ldvec(tb + SO_BNC + (4 * J + 3) * 32, hs, invstd);
#pragma unroll
                    for (int r = 0; r < 16; ++r) gx_[r] = tg[r] * ((a[J][r] - mean[r]) * invstd[r]);
                    so_transpose(tile, gx_, x, c32, hs);
                    float tsum = 0.f;
#pragma unroll
                    for (int i = 0; i < 16; i += 4) tsum += (x[i] + x[i + 1]) + (x[i + 2] + x[i + 3]);
                    red[32] = tsum;
                }
                NF_SO_STAMP(s == S - 2, 36 + 4 * (4 - J));
                so_barrier();
                if (t < 64) {
                    const int q = t >> 5, f = t & 31;
                    float tot = 0.f;
#pragma unroll
                    for (int g = 0; g < SO_GROUPS; ++g) tot += sm[SO_RED + (g * 3 + q) * 32 + f];
                    sm[SO_TOT + q * 32 + f] = tot;
                }
                so_barrier();
                NF_SO_STAMP(s == S - 2, 37 + 4 * (4 - J));
            }
            if ((t >> 5) == J) {                       // BatchNorm parameter gradients (g_beta = sum g, g_gamma = sum g xhat): the half wave
                const int f = t & 31;                  // that holds the sinks' old values stores (a load here would be a global round trip)
                st.g.beta[J][f] = gb_cur + sm[SO_TOT + f];
                st.g.gamma[J][f] = gg_cur + sm[SO_TOT + 32 + f];
            } else if (J == 4 && t >= 256 && t < 320) {  // the step's head sums in the format of k_glow_fold_all: region 0 carries them
                const int q = t - 256;
                float v = 0.f;
                if (q == 24 || q == 26) {
                    const int e = q == 24 ? 0 : 1;
#pragma unroll
                    for (int w = 0; w < NF_SO_WAVES; ++w) v += sm[SO_RED + (2 * w * 3 + 2) * 32 + e];
                }
                head_rec[((size_t)s * regions) * 64 + q] = v;
#pragma unroll
                for (int g = 1; g < regions; ++g) head_rec[((size_t)s * regions + g) * 64 + q] = 0.f;
            }
            {
                float sc[16], mean[16], invstd[16], mg[16], mgx[16];
                so_ldvec(tb + SO_BNC + (4 * J + 0) * 32, hs, sc);
                so_ldvec(tb + SO_BNC + (4 * J + 2) * 32, hs, mean);
                so_ldvec(tb + SO_BNC + (4 * J + 3) * 32, hs, invstd);
                so_ldvec(sm + SO_TOT, hs, mg);
                so_ldvec(sm + SO_TOT + 32, hs, mgx);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float xh = (a[J][r] - mean[r]) * invstd[r];
                    float v = sc[r] * (tg[r] - mg[r] * invN - xh * (mgx[r] * invN));
                    if (J == 0 || J == 2) v += Gs[r];  // a_J also feeds the residual connection two linears on
                    v = cv ? v : 0.f;
                    tg[r] = v;                         // = G_J
                    if (J == 2 || J == 4) Gs[r] = v;
                }
            }
            so_stash_put(gb4, J, tg);                          // G_J for the worker (16-byte stores in register order; nothing waits for them)

            NF_SO_STAMP(s == S - 2, 38 + 4 * (4 - J));
            __builtin_amdgcn_sched_barrier(0);
            if (J >= 1) {
                // linear J's data gradient (its weight gradient is the worker's)
                NF_SO_STAMP(s == S - 2, 39 + 4 * (4 - J));
                __builtin_amdgcn_sched_barrier(0);
                float ws[16], A[16];
                so_ld_abwd(tb, J, c32, hs, A);
                so_ldvec(tb + SO_WS + J * 32, hs, ws);
                f32x16 acc;
                so_gemm(A, tg, acc);
#pragma unroll
                for (int r = 0; r < 16; ++r) tg[r] = acc[r] * ws[r];
            } else {
                // linear 0 (1 -> 32): g_x on the vector ALU
                float w0[16];
                so_ldvec(tb + SO_V0, hs, w0);
                float gp = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) gp = fmaf(tg[r], w0[r], gp);
                gp += __shfl_xor(gp, 32, NF_WAVE);
                gx = gp * tb[SO_WS];
            }
        }
        NF_SO_STAMP(s == S - 2, 56);
        // ---- head backward: the statistics are buffers (modules.py:285-296): g_z = exp(log_gamma) g_h / sqrt(var) ----
        Gh[1 - sel] += gx;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float gzn = fmaf(tb[SO_HD + c], Gh[c], 0.f);
            gy[c] = cv ? gzn / tb[SO_HD + 6 + c] : 0.f;
        }
        if (cv && hs == 0) {
            float* gr = gzs + ((size_t)s * N + col) * 2;
            gr[0] = gy[0]; gr[1] = gy[1];
        }
        zin[0] = zn_[0]; zin[1] = zn_[1];
        // the step's G arrays are out: tell the worker's wave of the same columns (one flag per wave).  The two workgroups sit on the SAME
        // XCD (checked below), whose L2 is the point of coherence of both: this wave's stores only have to have arrived there (vmcnt) --
        // an agent-scope release would write the whole L2 back (buffer_wbl2), 3.5 us per step on the data path
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(flags + wid, (unsigned)(S - s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        NF_SO_STAMP(s == S - 2, 57);
        so_barrier();
        NF_SO_STAMP(s == S - 2, 58);
        if (rt) rec[s & 1][t] = nxt;                   // (slot of step s - 2: read behind the next step's first barrier)
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
static int nf_so_mode = -1;                           // bit 0: forward, bit 1: backward; the environment (NF_FLOW_SOLO) at first use, nf_flow_solo_config afterwards
static int nf_so_enabled() {
    if (nf_so_mode < 0) {
        // default 3 = both directions.  N = 256, S = 32: the one-workgroup forward takes 455 us against the two-workgroup grid kernel's
        // 650; the one-workgroup backward took 1035 against 796 while it recomputed the forward -- since round 5 it reads the five
        // BatchNorm inputs the forward stashed through a three-array sliding window (no recompute, 20 instead of 35 spilled registers):
        // ~780 us, C1 1.312 -> 1.298 ms per step
        const char* e = getenv("NF_FLOW_SOLO");
        nf_so_mode = e != nullptr ? (atoi(e) & 3) : 3;
    }
    return nf_so_mode;
}
// The stash / hand-over / table regions behind the S statistics records start on a 128-byte line (S x 328 floats is one only for
// S % 4 == 0): the backward's worker reads the hand-over area with plain L1-cached loads on the argument that every line is read once
// per launch -- a line shared by the regions of two waves (a misaligned base) would break that argument.  NF_FLOW_SOLO_PAD_FLOATS
// per step pays for the padding whatever S is.
#define NF_FLOW_SOLO_PAD_FLOATS 32
static inline size_t nf_so_stash_offset(int S, int save_stride) { return (((size_t)S * save_stride + 31) / 32) * 32; }
// sizes a caller needs for a RealNVP run of N rows x D features in training mode (whichever kernel serves it)
extern "C" int nf_realnvp_flow_save_floats(int64_t N, int D) {
    const bool solo_shape = D == 2 && N >= 1 && N <= NF_FLOW_SOLO_MAX_ROWS;
    return NF_REALNVP_SAVE_FLOATS +
           (solo_shape ? NF_FLOW_SOLO_PAD_FLOATS + NF_FLOW_SOLO_STASH_FLOATS + NF_FLOW_SOLO_GBUF_FLOATS + NF_FLOW_SOLO_TAB_FLOATS : 0);
}
extern "C" int nf_realnvp_flow_bwd_regions(int64_t N, int D) {
    const int grid = (int)((N + NF_MLP_ROWS_PER_BLOCK - 1) / NF_MLP_ROWS_PER_BLOCK);
    const bool solo_shape = D == 2 && N >= 1 && N <= NF_FLOW_SOLO_MAX_ROWS;
    return solo_shape && grid < NF_FLOW_SOLO_REGIONS ? NF_FLOW_SOLO_REGIONS : grid;
}
extern "C" int nf_flow_solo_config(int mode) {
    nf_so_enabled();
    if (mode >= 0) nf_so_mode = mode & 3;
    return 0;
}
// The two-workgroup backward needs blocks 0 and NF_SO_B_BLOCK of a launch on ONE XCD (its L2 is the hand-over's only point of coherence).
// Workgroups are dealt round-robin over the device's XCCs, so that holds exactly when the XCC count of the (partition of the) device
// divides NF_SO_B_BLOCK: 8 (MI355X / MI300X in SPX mode), 4, 2 or 1 (DPX / QPX / CPX partitions).  Asked once per device; a device that
// does not answer, or answers anything else (a 6-XCD part), runs the grid kernels' backward -- the in-kernel check stays as the backstop.
static int nf_so_pair_ok() {
    static int cache[64];                             // 0: not asked, 1: yes, 2: no
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cache[dev] == 0) {
        int xcc = 0;
        const bool got = hipDeviceGetAttribute(&xcc, hipDeviceAttributeNumberOfXccs, dev) == hipSuccess;
        cache[dev] = (got && xcc >= 1 && NF_SO_B_BLOCK % xcc == 0) ? 1 : 2;
    }
    return cache[dev] == 1;
}
int nf_solo_plan(int64_t N, int D, int backward) {
    // (the one-workgroup backward reads the activations the one-workgroup forward stashed: it needs the forward bit as well)
    const int m = nf_so_enabled();
    const bool on = backward ? ((m & 3) == 3 && nf_so_pair_ok()) : (m & 1) != 0;
    return on && D == 2 && N >= 1 && N <= NF_FLOW_SOLO_MAX_ROWS ? 1 : 0;
}
int nf_solo_bwd_steps_ok(int S) { return S >= 1 && S <= NF_SO_WK_MAX_STEPS; }
int nf_solo_fwd(const void* steps_dev, int S, const float* z0, float* ys, float* ld, float* saves, int save_stride, int64_t N, float bn_eps,
                float bn_momentum, float wn_eps, hipStream_t stream) {
    const size_t lds = nf_so_lds_bytes(false);
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)k_solo_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    // saves: S statistics records | S stashes of BatchNorm inputs | S hand-over areas of the backward | S table images
    // (include/nfhip.h: nf_realnvp_flow_save_floats)
    f32x4s* stash = reinterpret_cast<f32x4s*>(saves + nf_so_stash_offset(S, save_stride));
    if ((reinterpret_cast<uintptr_t>(stash) & 127) != 0) return NF_E_BADARG;          // (`saves` itself must sit on a 128-byte line)
    f32x4s* tabs = stash + (size_t)S * ((NF_FLOW_SOLO_STASH_FLOATS + NF_FLOW_SOLO_GBUF_FLOATS) / 4);
    hipLaunchKernelGGL(k_solo_fwd, dim3(1), dim3(NF_SO_THREADS), lds, stream, (const NfGlowFlowStep*)steps_dev, S, z0, ys, ld, saves, save_stride,
                       stash, tabs, (int)N, bn_eps, bn_momentum, wn_eps);
    NF_CHECK_LAUNCH();
    return 0;
}
int nf_solo_bwd(const void* steps_dev, int S, const float* z0, const float* ys, const float* g_y, const float* g_ld, float* gzs,
                const float* saves, int save_stride, int accumulate, float* ws_zero, float* slabs_all, float* head_rec, int64_t N,
                float wn_eps, hipStream_t stream) {
    if (!nf_solo_bwd_steps_ok(S) || ws_zero == nullptr) return NF_E_BADARG;
    // the data path's tables, or the worker's (BatchNorm constants of all steps + the transposition tiles): whichever is larger
    size_t lds = nf_so_lds_bytes(true);
    const size_t lds_w = sizeof(float) * ((size_t)S * 5 * 64 + (size_t)NF_SO_WAVES * 2 * 32 * NF_SO_TS);
    if (lds_w > lds) lds = lds_w;
    static size_t attr = 0;
    if (lds > attr) {
        hipError_t e = hipFuncSetAttribute((const void*)k_solo_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr = lds;
    }
    // saves: S statistics records | S stashes of BatchNorm inputs (forward) | S hand-over areas data path -> worker (this launch);
    // flags: the first eight words of the zeroed exchange workspace
    const f32x4s* stash = reinterpret_cast<const f32x4s*>(saves + nf_so_stash_offset(S, save_stride));
    f32x4s* gbuf = const_cast<f32x4s*>(stash) + (size_t)S * (NF_FLOW_SOLO_STASH_FLOATS / 4);
    // every worker wave's 1 KB of a hand-over array starts on a cache line of its own (see nf_so_stash_offset)
    static_assert((NF_FLOW_SOLO_STASH_FLOATS * 4) % 128 == 0 && (NF_FLOW_SOLO_GBUF_FLOATS * 4) % 128 == 0, "regions: whole 128-byte lines");
    if ((reinterpret_cast<uintptr_t>(gbuf) & 127) != 0) return NF_E_BADARG;
    const float* tabs = reinterpret_cast<const float*>(gbuf + (size_t)S * (NF_FLOW_SOLO_GBUF_FLOATS / 4));
    hipLaunchKernelGGL(k_solo_bwd, dim3(NF_SO_B_BLOCK + 1), dim3(NF_SO_THREADS), lds, stream, (const NfGlowFlowStep*)steps_dev, S, z0, ys, g_y,
                       g_ld, gzs, saves, save_stride, stash, tabs, gbuf, reinterpret_cast<unsigned*>(ws_zero), accumulate, slabs_all,
                       head_rec, (int)N, wn_eps);
    NF_CHECK_LAUNCH();
    return 0;
}
