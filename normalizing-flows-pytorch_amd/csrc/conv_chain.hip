// The image conditioner ConvNet (flows/modules.py:416-438: WN(conv3x3) -> 2 x [BN, ReLU, WN(conv3x3), BN, ReLU, WN(conv3x3), + skip]
// -> BN, ReLU, WN(conv1x1)) as ONE persistent launch -- the convolutional twin of mlp_chain.hip.
//
// Why: at the reference's batch (64 samples per GPU) a conv + BatchNorm launch is pure latency: 0.3 GFLOP over <= 128
// workgroups, ~3 us of MFMA work inside ~19 us of dependent memory round trips (conv_bn.hip), six launches per
// conditioner, 161 conditioners per step.  Here a workgroup owns WHOLE SAMPLES (one 16 x 16 sample, two 8 x 8, eight 4 x 4:
// no halo between workgroups), keeps the activations of consecutive layers in two zero-padded LDS frames (normalise + ReLU on
// the way in, so the nine taps read finished values), and the only traffic between two layers is the grid-wide exchange of the
// BatchNorm statistics that training mode imposes (one memory round trip, mlp_chain.hip's publish / collect protocol).  The
// pre-BatchNorm outputs of every layer are still written to global memory -- the backward pass needs them -- but nothing waits
// for those stores.
//
//   16 waves = NPB pixel blocks (32 pixels) x NKQ splits of the (tap, channel) axis:  <8, 2> for 256-pixel tiles (16 x 16),
//   <4, 4> for 128-pixel tiles (8 x 8, 4 x 4).  GEMMs as in conv_bn.hip: out^T[oc][pixel] on v_mfma_f32_32x32x2_f32 (exact fp32),
//   the K splits meet in LDS, every wave finishes 16 / NKQ output channels of its pixel block: bias, residual (kept in
//   registers: the lane that finishes (channel, pixel) of layer l also finishes it for layer l + 2), store, statistics.
//   Batch statistics are (sum, M2) pairs merged by the parallel-variance rule: half wave -> workgroup -> grid (no E[x^2] - E[x]^2).
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_set>
#include "nf_conv_core.h"

#define NF_CC_NL 6
#define NF_CC_NB 5
#define NF_CC_MAX_BLOCKS NF_CONVNET_MAX_BLOCKS

NF_PERSIST_STATE(nf_cc)
NF_PERSIST_HOST_API(nf_cc)

// phase stamps of workgroup 0 (tools/probes/chain_prof.py builds this file with -DNF_CC_PROF=1; 100 MHz wall clock)
#ifdef NF_CC_PROF
__device__ long long nf_cc_prof[64];
__device__ long long nf_cc_arrive[128];
extern "C" int nf_cc_arrive_read(long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(nf_cc_arrive), sizeof(long long) * 128);
}
#define NF_CC_STAMP(i)                                                                 \
    do {                                                                               \
        if (blockIdx.x == 0 && threadIdx.x == 0) nf_cc_prof[i] = wall_clock64();       \
    } while (0)
extern "C" int nf_cc_prof_read(long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(nf_cc_prof), sizeof(long long) * 64);
}
#else
#define NF_CC_STAMP(i)
#endif

// LDS layouts ("4-packed": the MFMA K index runs over groups of 8 = two channel quads, a lane's four consecutive K values are one
// float4, so a group costs TWO ds_read_b128 instead of eight ds_read_b32):
//   frame  F4[cg][f][4]            channel quad cg = c >> 2, frame position f, c & 3            quad stride CS4 = 4 * CS
//   weight W4[tap][cg][oc][4]      W[oc][4 cg + j][tap]: quad stride RSW = 4 * WC + 4, tap stride TS = NCG * RSW + 4 (WC = 32 * OCB
//                                  output columns; the + 4 keep the staging stores of neighbouring taps / quads on different banks)
#define NF_CC_RSW(WC) (4 * (WC) + 4)
#define NF_CC_TS(NCG, WC) ((NCG) * NF_CC_RSW(WC) + 4)
struct NfCcLds {            // offsets in floats
    int FA, FB, WL, RS, KC, KB, RED, TOT, BNV, total;
};
template <int NPB, int NKQ>
__host__ __device__ inline NfCcLds nf_cc_lds(int CS, int OCB) {
    NfCcLds L;
    const int w3 = 9 * NF_CC_TS(8, 32), w1 = NF_CC_TS(8, 32 * OCB);
    L.FA = 0;
    L.FB = L.FA + 32 * CS;
    L.WL = L.FB + 32 * CS;
    L.RS = L.WL + (w3 > w1 ? w3 : w1);
    int rs = NPB * (NKQ - 1) * 16 * NF_WAVE;
    if (rs < NF_CC_MAX_BLOCKS * 64) rs = NF_CC_MAX_BLOCKS * 64;       // the gather buffer of the grid exchange aliases RS
    L.KC = L.RS + rs;
    L.KB = L.KC + 128;                                 // kc[4][32]: scale, shift (forward) + mean, invstd (backward)
    L.RED = L.KB + 32;
    L.TOT = L.RED + 2 * NPB * 32;
    L.total = L.TOT + 64;
    // the backward kernel keeps the saved statistics and parameters of all five BatchNorms here when the geometry leaves room (every
    // level but 4 x 4): a global load at each layer's top is ~0.7 us of exposed latency on the serial chain
    L.BNV = (L.total + NF_CC_NB * 128) * (int)sizeof(float) <= 160 * 1024 ? L.total : -1;
    if (L.BNV >= 0) L.total += NF_CC_NB * 128;
    return L;
}

// the K loop: acc += sum over the groups [g0, g0 + gcount) of W4(tap, quads 2 q + hs)[oc] * F4(quads 2 q + hs)[pixel + tap offset];
// a tap has `gpt` groups (NCG / 2: 2 or 4, a power of two, lgg = log2).  Operand reads of group g + 1 are in flight under the MFMAs
// of group g (the one-past-the-end prefetch re-reads a valid group: branch-free body).
template <int T>
__device__ __forceinline__ void nf_cc_kloop(f32x16& acc, const float* W4, const float* F4, int TS, int RSW, int CS4, int FW, int lgg,
                                            int fpos, int col, int hs, int g0, int gcount) {
    const int gpt = 1 << lgg;
    const int glast = T * gpt - 1;
    const float* wbase = W4 + hs * RSW + 4 * col;
    const float* fbase = F4 + hs * CS4 + 4 * fpos;
    float4 a0, b0, a1, b1;
    int gi = g0;
#define NF_CC_LOAD(A_, B_)                                                                     \
    do {                                                                                       \
        const int gg = gi < glast ? gi : glast;                                                \
        const int tap = T == 1 ? 0 : gg >> lgg, q = gg - (tap << lgg);                         \
        const int dy = T == 9 ? tap / 3 - 1 : 0, dx = T == 9 ? tap - (tap / 3) * 3 - 1 : 0;    \
        A_ = *(const float4*)(wbase + tap * TS + 2 * q * RSW);                                 \
        B_ = *(const float4*)(fbase + 4 * (dy * FW + dx) + 2 * q * CS4);                       \
        ++gi;                                                                                  \
    } while (0)
#define NF_CC_MFMA(A_, B_)                                                                     \
    do {                                                                                       \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A_.x, B_.x, acc, 0, 0, 0);                  \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A_.y, B_.y, acc, 0, 0, 0);                  \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A_.z, B_.z, acc, 0, 0, 0);                  \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A_.w, B_.w, acc, 0, 0, 0);                  \
    } while (0)
    NF_CC_LOAD(a0, b0);
    for (int i = 0; i < gcount; i += 2) {
        NF_CC_LOAD(a1, b1);
        NF_CC_MFMA(a0, b0);
        NF_CC_LOAD(a0, b0);
        if (i + 1 < gcount) NF_CC_MFMA(a1, b1);
    }
#undef NF_CC_LOAD
#undef NF_CC_MFMA
}

// The same K loop for the 32 -> 32 channel 3 x 3 layers with the frame geometry known at compile time (FW, CS: one instantiation per
// pyramid level): every operand address is base + IMMEDIATE (the ds_read offset field), so a K group is two ds_read_b128 and four
// MFMAs with no address arithmetic in between; the generic loop above spends a third of the K phase on it (5.4 us against the 3.8 us
// MFMA floor of four waves per SIMD).  Straight-line, one group of operands in flight ahead of the MFMAs.
template <int FW, int CS, int G0, int GC>
__device__ __forceinline__ void nf_cc_kloop_fixed(f32x16& acc, const float* wbase, const float* fbase) {
    constexpr int TS = NF_CC_TS(8, 32), RSW = NF_CC_RSW(32), CS4 = 4 * CS;
    float4 a[2], b[2];
#define NF_CC_OFFA(gg) (((gg) >> 2) * TS + 2 * ((gg) & 3) * RSW)
#define NF_CC_OFFB(gg) (4 * (((((gg) >> 2) / 3) - 1) * FW + ((((gg) >> 2) % 3) - 1)) + 2 * ((gg) & 3) * CS4)
    a[0] = *(const float4*)(wbase + NF_CC_OFFA(G0));
    b[0] = *(const float4*)(fbase + NF_CC_OFFB(G0));
#pragma unroll
    for (int i = 0; i < GC; ++i) {
        if (i + 1 < GC) {
            a[(i + 1) & 1] = *(const float4*)(wbase + NF_CC_OFFA(G0 + i + 1));
            b[(i + 1) & 1] = *(const float4*)(fbase + NF_CC_OFFB(G0 + i + 1));
        }
        const float4 A_ = a[i & 1], B_ = b[i & 1];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A_.x, B_.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A_.y, B_.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A_.z, B_.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A_.w, B_.w, acc, 0, 0, 0);
    }
#undef NF_CC_OFFA
#undef NF_CC_OFFB
}
template <int FW, int CS, int NKQ>
__device__ __forceinline__ void nf_cc_kloop_level(f32x16& acc, const float* W4, const float* F4, int fpos, int col, int hs, int kq) {
    constexpr int GC = 36 / NKQ;
    const float* wbase = W4 + hs * NF_CC_RSW(32) + 4 * col;
    const float* fbase = F4 + hs * 4 * CS + 4 * fpos;
    if (kq == 0) nf_cc_kloop_fixed<FW, CS, 0, GC>(acc, wbase, fbase);             // wave-uniform
    else if (kq == 1) nf_cc_kloop_fixed<FW, CS, GC, GC>(acc, wbase, fbase);
    else if (NKQ > 2 && kq == 2) nf_cc_kloop_fixed<FW, CS, (NKQ > 2 ? 2 : 0) * GC, GC>(acc, wbase, fbase);
    else if (NKQ > 2) nf_cc_kloop_fixed<FW, CS, (NKQ > 2 ? 3 : 0) * GC, GC>(acc, wbase, fbase);
}

// 3x3 weights of one chunk of IC (padded ICP, NCG = ICP / 4 quads) input channels, global (32, I, 3, 3) -> W4: lane entries r = ic * 9 + tap
// (contiguous in global memory for every oc), wave w takes oc = w, w + 16
struct NfCcW { float v[5][NF_CV_CU]; };
__device__ __forceinline__ void nf_cc_w_load(NfCcW& w, const float* __restrict__ weight, int I, int i0, int IC, int wid, int lane) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int r = lane + NF_WAVE * j;
#pragma unroll
        for (int u = 0; u < NF_CV_CU; ++u) w.v[j][u] = r < IC * 9 ? weight[((wid + u * NF_CV_WAVES) * I + i0) * 9 + r] : 0.f;
    }
}
__device__ __forceinline__ void nf_cc_w_store(const NfCcW& w, float* W4, int ICP, int wid, int lane) {
    const int RSW = NF_CC_RSW(32), TS = NF_CC_TS(ICP >> 2, 32);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int r = lane + NF_WAVE * j;
        const int ic = r / 9, tap = r - ic * 9;
        if (r < ICP * 9) {                              // entries of the padding channels [IC, ICP) were loaded as zeros
#pragma unroll
            for (int u = 0; u < NF_CV_CU; ++u) W4[tap * TS + (ic >> 2) * RSW + 4 * (wid + u * NF_CV_WAVES) + (ic & 3)] = w.v[j][u];
        }
    }
}

// K-split exchange: the NKQ waves of a pixel block each hold a partial 32 x 32 accumulator; wave kq ends up with the TOTAL of
// registers [OWN kq, OWN kq + OWN).  RS: [pb][owner][slot][OWN][64].  Callers sync before (RS readers of the previous round done).
template <int NKQ>
__device__ __forceinline__ void nf_cc_ksplit_exchange(float (&own)[16 / NKQ], const f32x16& acc, float* RS, int pb, int kq, int lane) {
    constexpr int OWN = 16 / NKQ;
#pragma unroll
    for (int o = 0; o < NKQ; ++o)
        if (o != kq) {                                 // wave-uniform
            const int slot = kq < o ? kq : kq - 1;
#pragma unroll
            for (int rr = 0; rr < OWN; ++rr) RS[(((pb * NKQ + o) * (NKQ - 1) + slot) * OWN + rr) * NF_WAVE + lane] = acc[OWN * o + rr];
        } else {
#pragma unroll
            for (int rr = 0; rr < OWN; ++rr) own[rr] = acc[OWN * o + rr];
        }
    __syncthreads();
#pragma unroll
    for (int slot = 0; slot < NKQ - 1; ++slot)
#pragma unroll
        for (int rr = 0; rr < OWN; ++rr) own[rr] += RS[(((pb * NKQ + kq) * (NKQ - 1) + slot) * OWN + rr) * NF_WAVE + lane];
}

__device__ __forceinline__ int nf_cc_valid_px(int64_t Npx, int64_t first, int count) {
    const int64_t left = Npx - first;
    return (int)(left < count ? (left > 0 ? left : 0) : count);
}

// ---- (sum, M2) of OWN per-lane values over the 32 lanes of a wave half, by PAIRWISE merging (Chan et al.) ----------------------------
// Two blocks of m pixels each with sums S, S' and squared deviations M2, M2' about their own means merge into
//     S + S',   M2 + M2' + (S' - S)^2 / (2 m)
// -- a tree reduction, no mean needed in advance, no cancellation.  Halving butterfly: at the lane masks 1, 2 (, 4) a lane keeps
// half of its values and hands the other half to its partner, so a step's shuffles halve as well (14 shuffles for OWN = 8, 9 for
// OWN = 4, against 10 per value for two plain reductions: 80 / 40, which were 3.8 / 2.0 us per layer); the remaining masks merge the
// one value left.  Pixels beyond the batch come in whole samples, i.e. whole 16-lane groups (a map has >= 16 pixels), and hold
// zeros: inside a group the rule above is exact as it stands; at mask 16 an empty group is skipped (lo_ok / hi_ok).
// Result: lane c32 holds S and M2 of value index `which` (a function of its low bits) over the half's valid pixels.
__device__ __forceinline__ void nf_cc_merge(float& S, float& M2, float So, float Mo, float half_inv_m) {
    const float dl = So - S;
    M2 = (M2 + Mo) + dl * dl * half_inv_m;
    S += So;
}
template <int OWN>
__device__ __forceinline__ void nf_cc_half_stats(const float (&v)[OWN], bool lo_ok, bool hi_ok, int c32, float& S, float& M2,
                                                 int& which) {
    static_assert(OWN == 4 || OWN == 8, "16 / NKQ values per lane");
    float s4[4], q4[4];
    int w = 0, m = 1;                                   // m: pixels merged so far
    if (OWN == 8) {
        const bool up = c32 & 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float keep = up ? v[k + 4] : v[k], send = up ? v[k] : v[k + 4];
            s4[k] = keep; q4[k] = 0.f;
            nf_cc_merge(s4[k], q4[k], __shfl_xor(send, 1, NF_WAVE), 0.f, 0.5f);
        }
        w = up ? 4 : 0;
        m = 2;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) { s4[k] = v[k]; q4[k] = 0.f; }
    }
    float s2[2], q2[2];
    {
        const bool up = c32 & m;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float ks = up ? s4[k + 2] : s4[k], ss = up ? s4[k] : s4[k + 2];
            const float kq_ = up ? q4[k + 2] : q4[k], sq = up ? q4[k] : q4[k + 2];
            s2[k] = ks; q2[k] = kq_;
            const float os = __shfl_xor(ss, m, NF_WAVE);
            const float oq = OWN == 8 ? __shfl_xor(sq, m, NF_WAVE) : 0.f;      // OWN = 4: first merge, the partner's M2 is zero
            nf_cc_merge(s2[k], q2[k], os, oq, 0.5f / (float)m);
        }
        w += up ? 2 : 0;
        m *= 2;
    }
    {
        const bool up = c32 & m;
        const float ks = up ? s2[1] : s2[0], ss = up ? s2[0] : s2[1];
        const float kq_ = up ? q2[1] : q2[0], sq = up ? q2[0] : q2[1];
        S = ks; M2 = kq_;
        nf_cc_merge(S, M2, __shfl_xor(ss, m, NF_WAVE), __shfl_xor(sq, m, NF_WAVE), 0.5f / (float)m);
        w += up ? 1 : 0;
        m *= 2;
    }
    for (; m < 16; m *= 2) nf_cc_merge(S, M2, __shfl_xor(S, m, NF_WAVE), __shfl_xor(M2, m, NF_WAVE), 0.5f / (float)m);
    {   // the two 16-lane groups: either may be empty (beyond the batch)
        const float So = __shfl_xor(S, 16, NF_WAVE), Mo = __shfl_xor(M2, 16, NF_WAVE);
        const bool me_ok = (c32 & 16) ? hi_ok : lo_ok, ot_ok = (c32 & 16) ? lo_ok : hi_ok;
        if (me_ok && ot_ok) nf_cc_merge(S, M2, So, Mo, 0.5f / 16.f);
        else if (ot_ok) { S = So; M2 = Mo; }
    }
    which = w;
}

// gather the G x 64 published values of one exchange round into LDS (xs[workgroup][XS]: row stride 65 keeps the per-channel walks over
// workgroups off one bank); a slot is {generation : value}, four polled per
// trip.  A workgroup that never arrives (not co-resident, lost) ends the wait after nf_cc_spin_limit polls: sticky error word, loud on the host.
__device__ __forceinline__ void nf_cc_collect_slots(float* xs, const unsigned long long* rs, unsigned gen, int G, int XS) {
    for (int e0 = threadIdx.x; e0 < G * 64; e0 += 4 * NF_CV_THREADS) {
        unsigned long long v[4];
        unsigned spins = 0;
        bool ok;
        do {
            ok = true;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = e0 + k * NF_CV_THREADS;
                v[k] = __hip_atomic_load(rs + (e < G * 64 ? e : e0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) ok = ok && (unsigned)(v[k] >> 32) == gen;
            if (ok) break;
            if (++spins > nf_cc_spin_limit) { NF_PERSIST_GIVE_UP(nf_cc); break; }
            __builtin_amdgcn_s_sleep(1);
        } while (true);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = e0 + k * NF_CV_THREADS;
            if (e < G * 64) xs[(e >> 6) * XS + (e & 63)] = __uint_as_float((unsigned)v[k]);
        }
    }
}

__device__ __forceinline__ float nf_cc_sum32(float v) {      // over the 32 lanes of a wave half, fixed order
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) v += __shfl_xor(v, off, NF_WAVE);
    return v;
}

// grid-wide (sum, M2) of 32 channels: red[0][pb][c] = sums, red[1][pb][c] = M2 about the pixel block's mean -> tot[c], tot[32 + c].
// The merges are spread over the whole workgroup (a 64-iteration loop of divisions on 64 threads cost 7.4 us at 64 workgroups):
// thread (channel i = t & 31, part p = t >> 5) takes the workgroups p, p + 32, ...; parts meet in `part` (aliases Wl, idle here).
template <int NPB>
__device__ __forceinline__ const float* nf_cc_stats_exchange(float* sm, const NfCcLds& L, unsigned long long* slots, int round,
                                                             int64_t Npx, int PXW) {
    float* red = sm + L.RED;
    float* xs = sm + L.RS;
    float* tot = sm + L.TOT;
    const int G = gridDim.x;
    const int i = threadIdx.x & 31;
    __syncthreads();                                    // red complete; RS no longer read by anybody
    if (round == 1) NF_CC_STAMP(56);
    if (threadIdx.x < 32) {
        const int nb = nf_cc_valid_px(Npx, (int64_t)blockIdx.x * PXW, PXW);
        float S, M2;
        if (nb == PXW) {                                // every pixel block full: pairwise tree, no division
            float sv[NPB], mv[NPB];
#pragma unroll
            for (int q = 0; q < NPB; ++q) { sv[q] = red[q * 32 + i]; mv[q] = red[NPB * 32 + q * 32 + i]; }
#pragma unroll
            for (int w = 1; w < NPB; w *= 2)
#pragma unroll
                for (int q = 0; q < NPB; q += 2 * w) nf_cc_merge(sv[q], mv[q], sv[q + w], mv[q + w], 0.5f / (float)(32 * w));
            S = sv[0]; M2 = mv[0];
        } else {                                        // the last workgroup of a batch that does not fill it
            S = 0.f; M2 = 0.f;
            for (int q = 0; q < NPB; ++q) S += red[q * 32 + i];
            const float mb = S / (float)max(nb, 1);
            for (int q = 0; q < NPB; ++q) {
                const int np = nf_cc_valid_px(Npx, (int64_t)blockIdx.x * PXW + 32 * q, 32);
                const float dlt = red[q * 32 + i] / (float)max(np, 1) - mb;
                M2 += np > 0 ? fmaf((float)np * dlt, dlt, red[NPB * 32 + q * 32 + i]) : 0.f;
            }
        }
        if (G == 1) {
            tot[i] = S; tot[32 + i] = M2;
        } else {
            unsigned long long* dst = slots + ((size_t)round * NF_CC_MAX_BLOCKS + blockIdx.x) * 64 + i;
            __hip_atomic_store(dst, ((unsigned long long)(round + 1) << 32) | (unsigned long long)__float_as_uint(S), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst + 32, ((unsigned long long)(round + 1) << 32) | (unsigned long long)__float_as_uint(M2),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (G == 1) {
        __syncthreads();
        return tot;
    }
    const unsigned long long* rs = slots + (size_t)round * NF_CC_MAX_BLOCKS * 64;
    const unsigned gen = (unsigned)(round + 1);
    if (round == 1) NF_CC_STAMP(57);
#ifdef NF_CC_PROF
    if (round == 1 && threadIdx.x == 0) nf_cc_arrive[blockIdx.x] = wall_clock64();
#endif
    const int XS = G * 65 <= L.KC - L.RS ? 65 : 64;
    nf_cc_collect_slots(xs, rs, gen, G, XS);
    if (round == 1) NF_CC_STAMP(58);
    __syncthreads();
    if (round == 1) NF_CC_STAMP(59);
    // Thread (wave w, half h, lane l) reduces channel i = 2 w + h over the workgroups b = l, l + 32, ...: per-lane partials in a fixed
    // order, then a butterfly over the 32 lanes -- no further barrier until the totals are written (the version with two passes of
    // LDS partials took four more barriers).
    {
        const int lane = threadIdx.x & 63, ci = 2 * (threadIdx.x >> 6) + (lane >> 5), l = lane & 31;
        float ps = 0.f;
        for (int b = l; b < G; b += 32) ps += xs[b * XS + ci];
        const float S = nf_cc_sum32(ps);
        const float mean = S / (float)Npx;
        const float inv_full = 1.f / (float)PXW;
        float pm = 0.f;
        for (int b = l; b < G; b += 32) {
            const int nb = nf_cc_valid_px(Npx, (int64_t)b * PXW, PXW);
            const float dlt = xs[b * XS + ci] * (nb == PXW ? inv_full : 1.f / (float)max(nb, 1)) - mean;
            pm += fmaf((float)nb * dlt, dlt, xs[b * XS + 32 + ci]);
        }
        const float M2 = nf_cc_sum32(pm);
        if (l == 0) { tot[ci] = S; tot[32 + ci] = M2; }
    }
    __syncthreads();
    return tot;
}

// ---- the affine coupling around the conditioner (flows/coupling.py:104-122), fused into the chain kernels ------------------------------
// element (half channel m, half pixel q) of half `which` -> offset inside one sample of the full tensor (nf_half_to_full without the
// divisions: the half's width is a power of two here)
__device__ __forceinline__ int nf_cc_half_to_full(const NfSplit& s, int which, int m, int q, int lgw) {
    const int sel = which ^ s.odd;
    if (s.mode == NF_SPLIT_CHANNEL) return (m + sel * s.Ch) * (s.h * s.w) + q;
    const int i = q >> lgw, j = q & (s.w - 1);
    const int k = sel == 0 ? (m < s.C ? m : m + 2 * s.C) : (m + s.C);
    const int c = k >> 2, dy = (k >> 1) & 1, dx = k & 1;
    return (c * s.H + 2 * i + dy) * s.W + 2 * j + dx;
}

// ---- halo rows between the workgroups of one sample (16 x 16 maps on 128-pixel tiles) ---------------------------------------------------
// A tile owns TH image rows; the 3 x 3 taps of its first / last row read one row of the neighbouring tile.  The owner of a boundary row
// publishes its 32 x W values of the layer as 64-bit {generation : value} slots addressed to the neighbour -- [layer][destination tile]
// [slot s][channel][x], s = 0: the row above the destination's first row, s = 1: the row below its last -- BEFORE the layer's grid-wide
// statistics exchange, which the hand-over hides behind; the receiver polls its own slots after the exchange (they are there by then) and
// finishes the values exactly as its own pixels.  Self-synchronising: no ordering between these stores and the statistics slots is assumed.
#define NF_CC_STAT_SLOTS (NF_CC_NB * NF_CC_MAX_BLOCKS * 64)                 // 64-bit slots of the statistics exchanges (before the halo slots)
#define NF_CC_HALO_SLOTS(W) (2 * 32 * (W))                                  // per layer and destination tile
template <int OWN>
__device__ __forceinline__ void nf_cc_halo_publish(unsigned long long* hslots, int layer, const NfCvGeo& g, int tile, int y0, int px, int kq,
                                                   int hs, const float (&v)[OWN]) {
    const int row = px >> g.lgW, x = px & (g.W - 1);
    const bool first = row == 0 && y0 > 0;                       // our first row is the row BELOW the previous tile's last row
    const bool last = row == g.TH - 1 && y0 + g.TH < g.H;        // our last row is the row ABOVE the next tile's first row
    if (!(first || last)) return;
    const int dst = first ? tile - 1 : tile + 1, s = first ? 1 : 0;
    unsigned long long* base = hslots + ((size_t)layer * NF_CC_MAX_BLOCKS + dst) * NF_CC_HALO_SLOTS(g.W) + s * 32 * g.W + x;
    const unsigned long long gen = (unsigned long long)(layer + 1) << 32;
#pragma unroll
    for (int rr = 0; rr < OWN; ++rr) {
        const int c = nf_cv_cd_row(OWN * kq + rr, hs);
        __hip_atomic_store(base + c * g.W, gen | (unsigned long long)__float_as_uint(v[rr]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// thread t < 32 W: channel t / W, column t % W of BOTH halo rows (s = 0, 1; an image border has none).  Returns the frame position of
// the value this call delivers in `v`, or -1; call once per row s.
__device__ __forceinline__ int nf_cc_halo_poll(const unsigned long long* hslots, int layer, const NfCvGeo& g, int tile, int y0, int s, float& v) {
    const bool has = s == 0 ? y0 > 0 : y0 + g.TH < g.H;
    if (!has) return -1;
    const int t = threadIdx.x, x = t & (g.W - 1);
    const unsigned long long* p = hslots + ((size_t)layer * NF_CC_MAX_BLOCKS + tile) * NF_CC_HALO_SLOTS(g.W) + s * 32 * g.W + t;
    const unsigned gen = (unsigned)(layer + 1);
    unsigned long long w;
    unsigned spins = 0;
    do {
        w = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(w >> 32) == gen) break;
        if (++spins > nf_cc_spin_limit) { NF_PERSIST_GIVE_UP(nf_cc); break; }
        __builtin_amdgcn_s_sleep(1);
    } while (true);
    v = __uint_as_float((unsigned)w);
    return (s == 0 ? 0 : (g.TH + 1) * g.FW) + x + 1;             // frame row 0 / TH + 1, column x + halo
}

// LDS: two frames F4 [8][CS][4] | W4 | RS (K-split exchange; gather buffer of the grid exchange) | kc[4][32] | kb[32] | red[2][NPB][32] | tot[64]
template <int NPB, int NKQ, bool HALO, bool CPL, int FWc, int CSc>
__global__ void __launch_bounds__(NF_CV_THREADS) k_convnet_chain_fwd(nf_convnet_desc d, NfCvGeo g, int I0, int O_out, int training,
                                                                     float eps, float mom, NfSplit cs) {
    static_assert(NPB * NKQ == NF_CV_WAVES, "sixteen waves");
    constexpr int OWN = 16 / NKQ, PXW = 32 * NPB;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int OCB = (O_out + 31) / 32;
    const NfCcLds L = nf_cc_lds<NPB, NKQ>(g.CS, OCB);
    float* Wl = sm + L.WL;
    float* RS = sm + L.RS;
    float* kc = sm + L.KC;
    float* kb = sm + L.KB;
    float* red = sm + L.RED;
    const int CS4 = FWc > 0 ? 4 * CSc : 4 * g.CS;       // (a literal in the per-level instantiations)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, c32 = lane & 31, hs = lane >> 5;
    const int pb = wid % NPB, kq = wid / NPB;
    const int64_t Npx = g.B * g.HW;
    const int64_t tile = blockIdx.x;
    const int64_t b0 = (tile * PXW) >> g.lgHW;          // first sample of the tile
    // A sample larger than the tile (16 x 16 on 128-pixel tiles) is split over HW / PXW workgroups, rows y0 .. y0 + TH - 1 each: the
    // frame's halo rows then belong to the neighbours, which hand them over layer by layer (nf_cc_halo_*, below).
    constexpr bool halo = HALO;                         // compile-time: the whole-sample variants carry none of its registers
    const int y0 = halo ? (int)((tile * PXW) & (g.HW - 1)) >> g.lgW : 0;
    unsigned long long* hslots = halo ? (unsigned long long*)d.ws_zero + NF_CC_STAT_SLOTS : nullptr;
    const int px = pb * 32 + c32;
    const int fpos = nf_cv_frame_of(g, px);
    const int64_t P = tile * PXW + px;
    const bool pv = P < Npx;
    const int64_t b = pv ? P >> g.lgHW : 0;
    const int64_t q = pv ? P & (g.HW - 1) : 0;
    const int npb = nf_cc_valid_px(Npx, tile * PXW + 32 * pb, 32);
    unsigned long long* slots = (unsigned long long*)d.ws_zero;

    NF_CC_STAMP(0);
    // ---- zero both frames (halo and padding stay zero for the whole launch) ------------------------------------------------
    for (int e = threadIdx.x; e < 2 * 32 * g.CS; e += NF_CV_THREADS) sm[L.FA + e] = 0.f;
    float* Fin = sm + L.FA;
    float* Fout = sm + L.FB;
    constexpr bool cpl = CPL;                           // the coupling rides the epilogue of the output convolution (d.cp_z != NULL)
    if (cpl) {
        // y <- z for this workgroup's (contiguous) samples, whole 16-byte vectors, no index arithmetic; the transformed half is
        // overwritten by the epilogue at the far end of the launch (same workgroup, barriers in between).  Nothing waits for it.
        if (!halo) {
            const int nsamp = g.HW < PXW ? PXW >> g.lgHW : 1;
            const int64_t left = g.B - b0;
            const int n4 = (int)(left < nsamp ? (left > 0 ? left : 0) : nsamp) * (cs.n_full >> 2);
            const float4* src = reinterpret_cast<const float4*>(d.cp_z + b0 * cs.n_full);
            float4* dst = reinterpret_cast<float4*>(d.cp_y + b0 * cs.n_full);
            for (int idx = threadIdx.x; idx < n4; idx += NF_CV_THREADS) dst[idx] = src[idx];
        } else if (b0 < g.B) {
            // this workgroup's share of its sample: the rows of every channel plane that its pixels map to -- one contiguous chunk of
            // PXW / HW of the plane per channel (both split maps keep conditioner rows in order)
            const int plane4 = (cs.H * cs.W) >> 2, chunk4 = (int)(((int64_t)plane4 * PXW) >> g.lgHW);
            const int part = (int)((tile * PXW) & (g.HW - 1)) / PXW;
            const float4* src = reinterpret_cast<const float4*>(d.cp_z + b0 * cs.n_full);
            float4* dst = reinterpret_cast<float4*>(d.cp_y + b0 * cs.n_full);
            for (int idx = threadIdx.x; idx < cs.C * chunk4; idx += NF_CV_THREADS) {
                const int c = idx / chunk4, r = idx - c * chunk4;
                dst[c * plane4 + part * chunk4 + r] = src[c * plane4 + part * chunk4 + r];
            }
        }
    }

    float stream[OWN], own[OWN];
#pragma unroll
    for (int rr = 0; rr < OWN; ++rr) stream[rr] = 0.f;

    // ---- convolution 0: input from global memory, chunks of 32 channels --------------------------------------------------------
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    {
        const int nchunks = (I0 + 31) / 32;
        const float* in0 = d.x + b0 * I0 * g.HW;
        for (int ch = 0; ch < nchunks; ++ch) {
            const int i0 = 32 * ch, IC = min(32, I0 - i0), ICP = (IC + 15) & ~15;
            NfCcW wv;
            nf_cc_w_load(wv, d.w[0], I0, i0, IC, wid, lane);
            __syncthreads();                            // the previous chunk's readers of Wl / Fin are done (and the zero fill)
            nf_cc_w_store(wv, Wl, ICP, wid, lane);
#pragma unroll 1
            for (int jj = 0; jj < g.nfj; ++jj) {       // one frame position per lane and trip (register budget, see conv_bn.hip)
                const int f = lane + NF_WAVE * jj;
                const int t = nf_cv_decode(g, b0, y0, f);
                const int sp = t >= 0 ? NF_CV_SP(t) : 0, sg_ = t >= 0 ? NF_CV_SEG(t) : 0;
                float xa[NF_CV_CU];
#pragma unroll
                for (int u = 0; u < NF_CV_CU; ++u) {
                    const int c = wid + u * NF_CV_WAVES;
                    xa[u] = (t >= 0 && c < IC) ? in0[(sg_ * I0 + i0 + c) * g.HW + sp] : 0.f;
                }
                if (f < g.FSZ) {
#pragma unroll
                    for (int u = 0; u < NF_CV_CU; ++u) {
                        const int c = wid + u * NF_CV_WAVES;
                        if (c < ICP) Fin[(c >> 2) * CS4 + 4 * f + (c & 3)] = xa[u];
                    }
                }
            }
            __syncthreads();
            const int lgg = ICP == 32 ? 2 : 1;          // groups of 8 channels per tap
            const int ng = 9 << lgg;
            const int g0 = (kq * ng) / NKQ, g1 = ((kq + 1) * ng) / NKQ;
            if (FWc > 0 && ICP == 32) nf_cc_kloop_level<FWc, CSc, NKQ>(acc, Wl, Fin, fpos, c32, hs, kq);   // (block-uniform)
            else nf_cc_kloop<9>(acc, Wl, Fin, NF_CC_TS(ICP >> 2, 32), NF_CC_RSW(32), CS4, g.FW, lgg, fpos, c32, hs, g0, g1 - g0);
        }
    }

    // ---- layers 0 .. 4: finish, statistics, normalise into the other frame, next 3x3 convolution ---------------------------------
    // (register budget: 128 VGPRs at sixteen waves -- the per-layer vectors live in LDS, the weight prefetch keeps values only)
    NF_CC_STAMP(1);
    // bias, gamma, beta of a layer are loaded by threads < 32 BEFORE the K loop that precedes their use (a global load at the point of
    // use is ~0.7 us of exposed latency on the serial chain, twice per layer)
    float nb_ = 0.f, ng_ = 0.f, nbe_ = 0.f;
    if (threadIdx.x < 32) { nb_ = d.b[0][threadIdx.x]; ng_ = d.gamma[0][threadIdx.x]; nbe_ = d.beta[0][threadIdx.x]; }
#pragma unroll 1
    for (int l = 0; l < NF_CC_NB; ++l) {
        // next layer's 32 x 32 x 9 weights: loads in flight under the exchanges of this layer
        constexpr bool PREFETCH_W = OWN <= 4;           // OWN = 8 has no registers to spare: it loads at the point of use
        NfCcW wv;
        if (PREFETCH_W && l < NF_CC_NB - 1) nf_cc_w_load(wv, d.w[l + 1], 32, 0, 32, wid, lane);
        const float cg_ = ng_, cbe_ = nbe_;
        if (threadIdx.x < 32) kb[threadIdx.x] = nb_;
        __syncthreads();                                // every wave is done with Wl / Fin of this layer; kb is written
        NF_CC_STAMP(2 + 8 * l);
        nf_cc_ksplit_exchange<NKQ>(own, acc, RS, pb, kq, lane);
        NF_CC_STAMP(3 + 8 * l);
        float* act = d.acts[l];
#pragma unroll
        for (int rr = 0; rr < OWN; ++rr) own[rr] = pv ? own[rr] + (((l & 1) == 0 && l > 0) ? stream[rr] : 0.f) : 0.f;   // + block input
        if (training) {                                 // statistics of the pre-bias output: pairwise (sum, M2) over the wave half
            float S, M2;
            int which;
            nf_cc_half_stats<OWN>(own, npb > 0, npb > 16, c32, S, M2, which);
            if (c32 < OWN) {
                const int oc = nf_cv_cd_row(OWN * kq + which, hs);
                red[pb * 32 + oc] = S;
                red[NPB * 32 + pb * 32 + oc] = M2;
            }
        }
#pragma unroll
        for (int rr = 0; rr < OWN; ++rr) {
            const int oc = nf_cv_cd_row(OWN * kq + rr, hs);
            const float a = own[rr] + kb[oc];
            own[rr] = a;
            if (pv) act[(b * 32 + oc) * g.HW + q] = a;
            if ((l & 1) == 0) stream[rr] = a;           // acts[0], acts[2] are the residual stream
        }
        if (halo) nf_cc_halo_publish<OWN>(hslots, l, g, (int)tile, y0, px, kq, hs, own);
        NF_CC_STAMP(4 + 8 * l);
        if (training) {
            const float* tot = nf_cc_stats_exchange<NPB>(sm, L, slots, l, Npx, PXW);
            NF_CC_STAMP(5 + 8 * l);
            if (threadIdx.x < 32) {
                const int k = threadIdx.x;
                const float invN = 1.f / (float)Npx;
                const float mean = kb[k] + tot[k] * invN;                     // statistics of the pre-bias output
                const float var = tot[32 + k] * invN;                         // biased, as BatchNorm normalises
                const float invstd = 1.f / sqrtf(var + eps);
                const float sc = cg_ * invstd;
                kc[k] = sc;
                kc[32 + k] = cbe_ - mean * sc;
                if (blockIdx.x == 0) {
                    d.save_mean[l][k] = mean;
                    d.save_invstd[l][k] = invstd;
                    const float unb = Npx > 1 ? var * ((float)Npx / (float)(Npx - 1)) : var;
                    d.rmean[l][k] = (1.f - mom) * d.rmean[l][k] + mom * mean;
                    d.rvar[l][k] = (1.f - mom) * d.rvar[l][k] + mom * unb;
                    if (k == 0 && d.nbt[l] != nullptr) d.nbt[l][0] += 1;
                }
            }
        } else {
            __syncthreads();                            // RS readers done before the frames are written
            if (threadIdx.x < 32) {
                const int k = threadIdx.x;
                const float mean = d.rmean[l][k], invstd = 1.f / sqrtf(d.rvar[l][k] + eps);
                const float sc = cg_ * invstd;
                kc[k] = sc;
                kc[32 + k] = cbe_ - mean * sc;
                if (blockIdx.x == 0) {                  // what the backward kernels normalise with (constants in this mode)
                    d.save_mean[l][k] = mean;
                    d.save_invstd[l][k] = invstd;
                }
            }
        }
        __syncthreads();
        NF_CC_STAMP(6 + 8 * l);
        if (halo && threadIdx.x < 32 * g.W) {           // the neighbours' boundary rows, normalised like our own pixels
            const int c = threadIdx.x >> g.lgW;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float v = 0.f;
                const int f = nf_cc_halo_poll(hslots, l, g, (int)tile, y0, s2, v);
                if (f >= 0) Fout[(c >> 2) * CS4 + 4 * f + (c & 3)] = fmaxf(fmaf(v, kc[c], kc[32 + c]), 0.f);
            }
        }
        // normalise + ReLU into the other frame (a lane's values are whole channel quads: one 16-byte store each); next weights into Wl
#pragma unroll
        for (int j = 0; j < OWN / 4; ++j) {
            const int c0 = nf_cv_cd_row(OWN * kq + 4 * j, hs);          // channels c0 .. c0 + 3
            float4 v;
            v.x = pv ? fmaxf(fmaf(own[4 * j + 0], kc[c0 + 0], kc[32 + c0 + 0]), 0.f) : 0.f;
            v.y = pv ? fmaxf(fmaf(own[4 * j + 1], kc[c0 + 1], kc[32 + c0 + 1]), 0.f) : 0.f;
            v.z = pv ? fmaxf(fmaf(own[4 * j + 2], kc[c0 + 2], kc[32 + c0 + 2]), 0.f) : 0.f;
            v.w = pv ? fmaxf(fmaf(own[4 * j + 3], kc[c0 + 3], kc[32 + c0 + 3]), 0.f) : 0.f;
            *(float4*)(Fout + (c0 >> 2) * CS4 + 4 * fpos) = v;
        }
        { float* t = Fin; Fin = Fout; Fout = t; }
        if (l < NF_CC_NB - 1) {
            if (threadIdx.x < 32) { nb_ = d.b[l + 1][threadIdx.x]; ng_ = d.gamma[l + 1][threadIdx.x]; nbe_ = d.beta[l + 1][threadIdx.x]; }
            if (!PREFETCH_W) nf_cc_w_load(wv, d.w[l + 1], 32, 0, 32, wid, lane);
            nf_cc_w_store(wv, Wl, 32, wid, lane);
            __syncthreads();
            NF_CC_STAMP(7 + 8 * l);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const int ng = 9 * 4;
            const int g0 = (kq * ng) / NKQ;
            int gcount = ng / NKQ;
            // opaque trip count: fully unrolled, the K loop's LDS addresses become loop invariants of the LAYER loop that the
            // compiler keeps in (and spills from) VGPRs
            asm volatile("" : "+s"(gcount));
            if (FWc > 0) nf_cc_kloop_level<FWc, CSc, NKQ>(acc, Wl, Fin, fpos, c32, hs, kq);        // geometry known at compile time
            else nf_cc_kloop<9>(acc, Wl, Fin, NF_CC_TS(8, 32), NF_CC_RSW(32), CS4, g.FW, 2, fpos, c32, hs, g0, gcount);
            NF_CC_STAMP(8 + 8 * l);
        }
    }
    NF_CC_STAMP(50);

    // ---- the 1 x 1 output convolution: wave (pb, kq) takes output blocks kq, kq + NKQ, ... ------------------------------------
    // With the coupling fused, the rows are staged interleaved -- row 2 p of block ob = shift channel m = 16 ob + p, row 2 p + 1 =
    // scale channel Ch + m -- so that a lane holds both parameters of the elements it transforms.
    const int WC = 32 * OCB, RSW5 = NF_CC_RSW(WC);
    const int Ch = O_out >> 1;
    for (int e = threadIdx.x; e < O_out * 32; e += NF_CV_THREADS) {
        const int oc = e >> 5, ic = e & 31;
        int row = oc;
        if (cpl) {
            const int m = oc < Ch ? oc : oc - Ch;
            row = 32 * (m >> 4) + 2 * (m & 15) + (oc < Ch ? 0 : 1);
        }
        Wl[(ic >> 2) * RSW5 + 4 * row + (ic & 3)] = d.w[5][e];
    }
    __syncthreads();
    float ls = 0.f;                                     // this lane's part of its sample's sum of scales
    const float ca = cpl ? d.cp_a[0] : 0.f, cc = cpl ? d.cp_c[0] : 0.f;
    const int lgw = cpl ? 31 - __clz(cs.w) : 0;
    for (int ob = kq; ob < OCB; ob += NKQ) {            // wave-uniform
        f32x16 a5;
#pragma unroll
        for (int r = 0; r < 16; ++r) a5[r] = 0.f;
        nf_cc_kloop<1>(a5, Wl, Fin, 0, RSW5, CS4, g.FW, 2, fpos, 32 * ob + c32, hs, 0, 4);
        if (!cpl) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int oc = ob * 32 + nf_cv_cd_row(r, hs);
                if (pv && oc < O_out) d.out[(b * O_out + oc) * g.HW + q] = a5[r] + d.b[5][oc];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int m = 16 * ob + (nf_cv_cd_row(r, hs) >> 1);
                if (pv && m < Ch) {
                    const float t = a5[r] + d.b[5][m], sr = a5[r + 1] + d.b[5][Ch + m];
                    const float th = tanhf(sr), sv = th * ca + cc;
                    const float es = expf(d.cp_inverse ? -sv : sv);
                    // what the backward needs of the conditioner's output is exp(s) and tanh(raw): `out` is private to the fused
                    // coupling, so they are what it keeps (8 of 256 compute units redo 12 k transcendentals otherwise)
                    d.out[(b * O_out + m) * g.HW + q] = es;
                    d.out[(b * O_out + Ch + m) * g.HW + q] = th;
                    const int64_t o = b * cs.n_full + nf_cc_half_to_full(cs, 0, m, (int)q, lgw);
                    const float z0 = d.cp_z[o];
                    d.cp_y[o] = d.cp_inverse ? es * (z0 - t) : z0 * es + t;
                    ls += sv;
                }
            }
        }
    }
    if (cpl) {                                          // log-det: per-sample sums inside the workgroup (it owns whole samples), fixed order
        __syncthreads();
        RS[(kq * 2 + hs) * PXW + px] = ls;
        __syncthreads();
        if (threadIdx.x < PXW) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 2 * NKQ; ++k) v += RS[k * PXW + threadIdx.x];
            const int seg = g.HW < NF_WAVE ? g.HW : NF_WAVE;
            for (int off = seg >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, NF_WAVE);
            if (g.HW <= NF_WAVE) {
                const int64_t bb = b0 + ((int)threadIdx.x >> g.lgHW);
                if ((threadIdx.x & (g.HW - 1)) == 0 && bb < g.B) atomicAdd(d.cp_ld + bb, d.cp_inverse ? -v : v);   // (no return value: nothing waits)
            } else if (lane == 0) {
                red[wid] = v;
            }
        }
        if (g.HW > NF_WAVE) {                           // block-uniform: one sample per workgroup
            __syncthreads();
            if (threadIdx.x == 0 && b0 < g.B) {
                float v = 0.f;
                for (int k = 0; k < (min(g.HW, PXW) >> 6); ++k) v += red[k];
                if (halo) {
                    // a sample's tiles add their parts in a fixed order (bit-reproducible log-det): the later tiles hand theirs to
                    // the first one through never-used layer-0 halo slots of that tile (its upper border has no neighbour)
                    const int nparts = g.HW / PXW, part = (int)((tile * PXW) & (g.HW - 1)) / PXW;
                    unsigned long long* hs0 = hslots + (size_t)(tile - part) * NF_CC_HALO_SLOTS(g.W);
                    const unsigned long long gen = 7ull << 32;
                    if (part > 0) {
                        __hip_atomic_store(hs0 + part, gen | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        v = 0.f;
                    } else {
                        for (int p2 = 1; p2 < nparts; ++p2) {
                            unsigned long long w;
                            unsigned spins = 0;
                            do {
                                w = __hip_atomic_load(hs0 + p2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if ((w >> 32) == 7ull) break;
                                if (++spins > nf_cc_spin_limit) { NF_PERSIST_GIVE_UP(nf_cc); break; }
                                __builtin_amdgcn_s_sleep(1);
                            } while (true);
                            v += __uint_as_float((unsigned)w);
                        }
                    }
                    if (part == 0) atomicAdd(d.cp_ld + b0, d.cp_inverse ? -v : v);
                } else {
                    atomicAdd(d.cp_ld + b0, d.cp_inverse ? -v : v);
                }
            }
        }
    }
    NF_CC_STAMP(51);
}

// =====================================================================================================================================
// The backward twin: the DATA gradient of the whole conditioner in one persistent launch.  What a layer's data pass hands to the
// next (nf_conv_bn_bwd with g_weff == NULL, six launches): gn_l = (W_{l+1}^T * G_{l+1}) . [ReLU mask of BatchNorm l], its batch sums
// sum gn_l and sum gn_l xhat_l, and G_l = BNbwd_l(gn_l) (+ G_{l+2} on the residual stream) -- here G_l never leaves the workgroup: it
// is written into the LDS frame the next transposed convolution reads.  The grid-wide exchange per layer carries the two PLAIN sums
// (fixed order: this pass is deterministic, where the per-layer kernels add into replicas with atomics).  Global memory receives
// exactly what the deferred weight-gradient launches (nf_conv_bn_wgrad_multi, unchanged) read afterwards: gn_0 .. gn_4, the sums
// (totals in replica 0 of the zero-initialised replica arrays), G_4 and G_2 (the g_skip operands), and the input gradient.
// Transposed convolutions reuse the forward K loop: the weights are staged as W4T[8 - tap][oc quad][ic][4].
// =====================================================================================================================================
__device__ __forceinline__ void nf_cc_wT_store(const NfCcW& w, float* W4, int wid, int lane) {
    const int RSW = NF_CC_RSW(32), TS = NF_CC_TS(8, 32);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int r = lane + NF_WAVE * j;
        const int ic = r / 9, tap = r - ic * 9;
        if (r < 32 * 9) {                               // rows beyond the chunk's channels were loaded as zeros
#pragma unroll
            for (int u = 0; u < NF_CV_CU; ++u) {
                const int oc = wid + u * NF_CV_WAVES;
                W4[(8 - tap) * TS + (oc >> 2) * RSW + 4 * ic + (oc & 3)] = w.v[j][u];
            }
        }
    }
}

// plain sums of two sets of OWN per-lane values over the 32 lanes of a wave half (halving butterfly, see nf_cc_half_stats)
template <int OWN>
__device__ __forceinline__ void nf_cc_half_sums2(const float (&u)[OWN], const float (&v)[OWN], int c32, float& S1, float& S2, int& which) {
    static_assert(OWN == 4 || OWN == 8, "16 / NKQ values per lane");
    float a4[4], b4[4];
    int w = 0, m = 1;
    if (OWN == 8) {
        const bool up = c32 & 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a4[k] = (up ? u[k + 4] : u[k]) + __shfl_xor(up ? u[k] : u[k + 4], 1, NF_WAVE);
            b4[k] = (up ? v[k + 4] : v[k]) + __shfl_xor(up ? v[k] : v[k + 4], 1, NF_WAVE);
        }
        w = up ? 4 : 0;
        m = 2;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) { a4[k] = u[k]; b4[k] = v[k]; }
    }
    float a2[2], b2[2];
    {
        const bool up = c32 & m;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            a2[k] = (up ? a4[k + 2] : a4[k]) + __shfl_xor(up ? a4[k] : a4[k + 2], m, NF_WAVE);
            b2[k] = (up ? b4[k + 2] : b4[k]) + __shfl_xor(up ? b4[k] : b4[k + 2], m, NF_WAVE);
        }
        w += up ? 2 : 0;
        m *= 2;
    }
    {
        const bool up = c32 & m;
        S1 = (up ? a2[1] : a2[0]) + __shfl_xor(up ? a2[0] : a2[1], m, NF_WAVE);
        S2 = (up ? b2[1] : b2[0]) + __shfl_xor(up ? b2[0] : b2[1], m, NF_WAVE);
        w += up ? 1 : 0;
        m *= 2;
    }
    for (; m < 32; m *= 2) {
        S1 += __shfl_xor(S1, m, NF_WAVE);
        S2 += __shfl_xor(S2, m, NF_WAVE);
    }
    which = w;
}

// grid-wide plain sums of 2 x 32 values: red[h][pb][c] -> tot[32 h + c]; fixed summation order
template <int NPB>
__device__ __forceinline__ const float* nf_cc_sum_exchange(float* sm, const NfCcLds& L, unsigned long long* slots, int round) {
    float* red = sm + L.RED;
    float* xs = sm + L.RS;
    float* tot = sm + L.TOT;
    const int G = gridDim.x;
    __syncthreads();                                    // red complete; RS / Wl no longer read by anybody
    if (threadIdx.x < 64) {
        const int i = threadIdx.x & 31, h = threadIdx.x >> 5;
        float sv[NPB];
#pragma unroll
        for (int q = 0; q < NPB; ++q) sv[q] = red[(h * NPB + q) * 32 + i];
#pragma unroll
        for (int w = 1; w < NPB; w *= 2)
#pragma unroll
            for (int q = 0; q < NPB; q += 2 * w) sv[q] += sv[q + w];
        if (G == 1) tot[threadIdx.x] = sv[0];
        else
            __hip_atomic_store(slots + ((size_t)round * NF_CC_MAX_BLOCKS + blockIdx.x) * 64 + threadIdx.x,
                               ((unsigned long long)(round + 1) << 32) | (unsigned long long)__float_as_uint(sv[0]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
    if (G == 1) {
        __syncthreads();
        return tot;
    }
    const int XS = G * 65 <= L.KC - L.RS ? 65 : 64;
    nf_cc_collect_slots(xs, slots + (size_t)round * NF_CC_MAX_BLOCKS * 64, (unsigned)(round + 1), G, XS);
    __syncthreads();
    {
        const int lane = threadIdx.x & 63, ci = 2 * (threadIdx.x >> 6) + (lane >> 5), l = lane & 31;
        float pa = 0.f, pb = 0.f;
        for (int b = l; b < G; b += 32) { pa += xs[b * XS + ci]; pb += xs[b * XS + 32 + ci]; }
        const float A = nf_cc_sum32(pa), Bs = nf_cc_sum32(pb);
        if (l == 0) { tot[ci] = A; tot[32 + ci] = Bs; }
    }
    __syncthreads();
    return tot;
}

template <int NPB, int NKQ, bool HALO, bool CPL, int FWc, int CSc>
__global__ void __launch_bounds__(NF_CV_THREADS) k_convnet_chain_bwd(nf_convnet_bwd_desc d, NfCvGeo g, int I0, int O_out, int training,
                                                                     NfSplit cs) {
    static_assert(NPB * NKQ == NF_CV_WAVES, "sixteen waves");
    constexpr int OWN = 16 / NKQ, PXW = 32 * NPB;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const NfCcLds L = nf_cc_lds<NPB, NKQ>(g.CS, 1);
    float* Wl = sm + L.WL;
    float* RS = sm + L.RS;
    float* kc = sm + L.KC;
    float* red = sm + L.RED;
    const int CS4 = FWc > 0 ? 4 * CSc : 4 * g.CS;       // (a literal in the per-level instantiations)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, c32 = lane & 31, hs = lane >> 5;
    const int pb = wid % NPB, kq = wid / NPB;
    const int64_t Npx = g.B * g.HW;
    const int64_t tile = blockIdx.x;
    const int px = pb * 32 + c32;
    const int fpos = nf_cv_frame_of(g, px);
    const int64_t P = tile * PXW + px;
    const bool pv = P < Npx;
    const int64_t b = pv ? P >> g.lgHW : 0;
    const int64_t q = pv ? P & (g.HW - 1) : 0;
    const float invN = 1.f / (float)Npx;
    unsigned long long* slots = (unsigned long long*)d.ws_zero;
    const int64_t b0 = (tile * PXW) >> g.lgHW;
    constexpr bool halo = HALO;                         // compile-time: the whole-sample variants carry none of its registers                       // a sample split over several workgroups (see the forward kernel)
    const int y0 = halo ? (int)((tile * PXW) & (g.HW - 1)) >> g.lgW : 0;
    unsigned long long* hslots = halo ? slots + NF_CC_STAT_SLOTS : nullptr;
    float gstream_h[2] = {0.f, 0.f};                    // halo rows of the residual stream's gradient (threads < 32 W)

    for (int e = threadIdx.x; e < 2 * 32 * g.CS; e += NF_CV_THREADS) sm[L.FA + e] = 0.f;
    float* Fin = sm + L.FA;
    float* Fout = sm + L.FB;
    const bool bnv = L.BNV >= 0;                        // block-uniform
    if (bnv && threadIdx.x < NF_CC_NB * 128) {          // [layer][mean | invstd | gamma | beta][32]; first read after the barrier below
        const int l2 = threadIdx.x >> 7, j = (threadIdx.x >> 5) & 3, k = threadIdx.x & 31;
        const float* src = j == 0 ? d.save_mean[l2] : (j == 1 ? d.save_invstd[l2] : (j == 2 ? d.gamma[l2] : d.beta[l2]));
        sm[L.BNV + threadIdx.x] = src[k];
    }

    // ---- the 1 x 1 output convolution, transposed: acc[ic][pixel] = sum_oc W5[oc][ic] g_out[oc][pixel].  No halo: the B operand comes
    //      straight from global memory (a lane's own pixel; 4 x 128-byte segments per K group), K = oc split over the NKQ waves ----
    constexpr bool cpl = CPL;                           // (d.cp_g_y != NULL)
    const int Ch = O_out >> 1;
    const int lgw = cpl ? 31 - __clz(cs.w) : 0;
    const float ca = cpl ? d.cp_a[0] : 0.f;
    float acc_a = 0.f, acc_c = 0.f;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    {
        const int ng = (O_out + 7) >> 3;
        for (int e = threadIdx.x; e < ng * 8 * 32; e += NF_CV_THREADS) {
            const int oc = e >> 5, ic = e & 31;
            Wl[(oc >> 2) * NF_CC_RSW(32) + 4 * ic + (oc & 3)] = oc < O_out ? d.w[5][e] : 0.f;
        }
        __syncthreads();
        const int g0 = (kq * ng) / NKQ, g1 = ((kq + 1) * ng) / NKQ;
        const float* go = cpl ? nullptr : d.g_out + b * O_out * g.HW + q;
        const float gl = (cpl && pv) ? d.cp_g_ld[b] : 0.f;
#pragma unroll 1
        for (int gi = g0; gi < g1; gi += 4) {
            float bv[4][4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int oc = 8 * (gi + k) + 4 * hs + j;
                    const bool ok = gi + k < g1 && pv && oc < O_out;
                    float v = 0.f;
                    if (!cpl) {
                        v = ok ? go[(int64_t)oc * g.HW] : 0.f;
                    } else if (ok) {                    // the coupling's backward produces the conditioner's output gradient on the fly
                        const bool is_s = oc >= Ch;
                        const int m = is_s ? oc - Ch : oc;
                        const int64_t o = b * cs.n_full + nf_cc_half_to_full(cs, 0, m, (int)q, lgw);
                        const float gy0 = d.cp_g_y[o];
                        v = gy0;                        // gradient of the shift
                        if (is_s) {                     // cp_out = [exp(s) | tanh(raw)], left by the forward launch
                            const float th = d.cp_out[(b * O_out + oc) * g.HW + q];
                            const float es = d.cp_out[(b * O_out + m) * g.HW + q];
                            d.cp_g_z[o] = gy0 * es;
                            const float gs = gy0 * d.cp_z[o] * es + gl;
                            v = gs * ca * (1.f - th * th);
                            acc_a += gs * th;
                            acc_c += gs;
                        }
                        d.cp_g_out[(b * O_out + oc) * g.HW + q] = v;   // the deferred weight-gradient pass of the 1 x 1 convolution reads it
                    }
                    bv[k][j] = v;
                }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (gi + k < g1) {                      // wave-uniform
                    const float4 a = *(const float4*)(Wl + (2 * (gi + k) + hs) * NF_CC_RSW(32) + 4 * c32);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bv[k][0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bv[k][1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bv[k][2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bv[k][3], acc, 0, 0, 0);
                }
        }
        if (cpl) {                                      // gradients of the coupling's two scalars: wave sums now, one atomic pair per workgroup below
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                acc_a += __shfl_xor(acc_a, off, NF_WAVE);
                acc_c += __shfl_xor(acc_c, off, NF_WAVE);
            }
            if (lane == 0) { red[wid] = acc_a; red[NF_CV_WAVES + wid] = acc_c; }
        }
    }

    float gstream[OWN], own[OWN];
#pragma unroll
    for (int rr = 0; rr < OWN; ++rr) gstream[rr] = 0.f;

#pragma unroll 1
    for (int l = NF_CC_NB - 1; l >= 0; --l) {
        // BatchNorm l (input of the convolution whose transpose just ran): scale / shift exactly as the forward kernel computed them
        if (threadIdx.x < 32) {
            const int k = threadIdx.x;
            const float* v = sm + L.BNV + 128 * l;
            const float mean = bnv ? v[k] : d.save_mean[l][k], invstd = bnv ? v[32 + k] : d.save_invstd[l][k];
            const float sc = (bnv ? v[64 + k] : d.gamma[l][k]) * invstd;
            kc[k] = sc;
            kc[32 + k] = (bnv ? v[96 + k] : d.beta[l][k]) - mean * sc;
            kc[64 + k] = mean;
            kc[96 + k] = invstd;
        }
        float xh[OWN];                                  // the forward activation, then its normalised value
        {
            const float* act = d.acts[l];
#pragma unroll
            for (int rr = 0; rr < OWN; ++rr) {
                const int oc = nf_cv_cd_row(OWN * kq + rr, hs);
                xh[rr] = pv ? act[(b * 32 + oc) * g.HW + q] : 0.f;
            }
        }
        float a_h[2] = {0.f, 0.f};                      // forward activations at our halo pixels (threads < 32 W; loads in flight early)
        if (halo && threadIdx.x < 32 * g.W) {
            const int c = threadIdx.x >> g.lgW, x = threadIdx.x & (g.W - 1);
            if (y0 > 0) a_h[0] = d.acts[l][(b0 * 32 + c) * g.HW + (y0 - 1) * g.W + x];
            if (y0 + g.TH < g.H) a_h[1] = d.acts[l][(b0 * 32 + c) * g.HW + (y0 + g.TH) * g.W + x];
        }
        __syncthreads();                                // every wave is done with Wl / the frames of the K loop; kc is written
        if (cpl && l == NF_CC_NB - 1 && threadIdx.x == 0) {
            float ta = 0.f, tc = 0.f;
            for (int k = 0; k < NF_CV_WAVES; ++k) { ta += red[k]; tc += red[NF_CV_WAVES + k]; }
            atomicAdd(d.cp_g_a, ta);
            atomicAdd(d.cp_g_c, tc);
        }
        nf_cc_ksplit_exchange<NKQ>(own, acc, RS, pb, kq, lane);
        float* gn = d.gn[l];
#pragma unroll
        for (int rr = 0; rr < OWN; ++rr) {
            const int oc = nf_cv_cd_row(OWN * kq + rr, hs);
            const float a = xh[rr];
            const bool keep = pv && fmaf(a, kc[oc], kc[32 + oc]) > 0.f;
            const float v = keep ? own[rr] : 0.f;
            own[rr] = v;
            xh[rr] = (a - kc[64 + oc]) * kc[96 + oc];
            if (pv) gn[(b * 32 + oc) * g.HW + q] = v;
        }
        if (halo) nf_cc_halo_publish<OWN>(hslots, l, g, (int)tile, y0, px, kq, hs, own);   // gn of our boundary rows, to the neighbours
        float mg[OWN], mgx[OWN];
        {   // batch sums of gn and gn * xhat: the gradients of beta and gamma in either mode, the mean terms of the BatchNorm backward
            // in training mode (evaluation mode normalises with constants: no mean terms)
            float p2[OWN], S1, S2;
            int which;
#pragma unroll
            for (int rr = 0; rr < OWN; ++rr) p2[rr] = own[rr] * xh[rr];
            nf_cc_half_sums2<OWN>(own, p2, c32, S1, S2, which);
            if (c32 < OWN) {
                const int oc = nf_cv_cd_row(OWN * kq + which, hs);
                red[pb * 32 + oc] = S1;
                red[NPB * 32 + pb * 32 + oc] = S2;
            }
            const float* tot = nf_cc_sum_exchange<NPB>(sm, L, slots, l);
            if (blockIdx.x == 0 && threadIdx.x < 64) {
                (threadIdx.x < 32 ? d.sum_g[l] : d.sum_gx[l])[threadIdx.x & 31] = tot[threadIdx.x];
                // the sums ARE the gradients of beta (sum gn) and gamma (sum gn xhat): straight into the caller's accumulators
                float* sink = threadIdx.x < 32 ? d.g_beta[l] : d.g_gamma[l];
                if (sink != nullptr) sink[threadIdx.x & 31] += tot[threadIdx.x];
            }
#pragma unroll
            for (int rr = 0; rr < OWN; ++rr) {
                const int oc = nf_cv_cd_row(OWN * kq + rr, hs);
                mg[rr] = training ? tot[oc] * invN : 0.f;
                mgx[rr] = training ? tot[32 + oc] * invN : 0.f;
            }
            if (halo && threadIdx.x < 32 * g.W) {       // G_l at the neighbours' boundary rows: the same per-pixel formula, their gn
                const int c = threadIdx.x >> g.lgW;
                const float mgc = training ? tot[c] * invN : 0.f, mgxc = training ? tot[32 + c] * invN : 0.f;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    float v = 0.f;
                    const int f = nf_cc_halo_poll(hslots, l, g, (int)tile, y0, s2, v);
                    if (f >= 0) {
                        const float xhh = (a_h[s2] - kc[64 + c]) * kc[96 + c];
                        float G = kc[c] * (v - mgc - xhh * mgxc);
                        if ((l & 1) == 0) { G += gstream_h[s2]; gstream_h[s2] = G; }
                        Fout[(c >> 2) * CS4 + 4 * f + (c & 3)] = G;
                    }
                }
            }
        }
        // G_l = BatchNorm backward (+ the residual stream's gradient), into the other frame
        const bool on_stream = (l & 1) == 0;
        float* gs = (l == 4) ? d.g_store[0] : (l == 2 ? d.g_store[1] : nullptr);
#pragma unroll
        for (int rr = 0; rr < OWN; ++rr) {
            const int oc = nf_cv_cd_row(OWN * kq + rr, hs);
            float G = kc[oc] * (own[rr] - mg[rr] - xh[rr] * mgx[rr]);
            if (on_stream) G += gstream[rr];
            G = pv ? G : 0.f;
            own[rr] = G;
            if (on_stream) gstream[rr] = G;
            if (gs != nullptr && pv) gs[(b * 32 + oc) * g.HW + q] = G;
        }
#pragma unroll
        for (int j = 0; j < OWN / 4; ++j) {
            const int c0 = nf_cv_cd_row(OWN * kq + 4 * j, hs);
            *(float4*)(Fout + (c0 >> 2) * CS4 + 4 * fpos) = make_float4(own[4 * j], own[4 * j + 1], own[4 * j + 2], own[4 * j + 3]);
        }
        { float* t = Fin; Fin = Fout; Fout = t; }
        if (l >= 1) {                                   // transposed 3 x 3 convolution l: G_l -> layer l - 1
            NfCcW wv;
            nf_cc_w_load(wv, d.w[l], 32, 0, 32, wid, lane);
            nf_cc_wT_store(wv, Wl, wid, lane);
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const int ng = 9 * 4;
            const int g0 = (kq * ng) / NKQ;
            int gcount = ng / NKQ;
            asm volatile("" : "+s"(gcount));            // see the forward kernel
            if (FWc > 0) nf_cc_kloop_level<FWc, CSc, NKQ>(acc, Wl, Fin, fpos, c32, hs, kq);        // geometry known at compile time
            else nf_cc_kloop<9>(acc, Wl, Fin, NF_CC_TS(8, 32), NF_CC_RSW(32), CS4, g.FW, 2, fpos, c32, hs, g0, gcount);
        }
    }

    // ---- gradient of the conditioner's input: convolution 0 transposed, 32 input channels per pass ----
    if (d.g_x != nullptr || cpl) {
        for (int i0 = 0; i0 < I0; i0 += 32) {
            const int IC = min(32, I0 - i0);
            NfCcW wv;
            nf_cc_w_load(wv, d.w[0], I0, i0, IC, wid, lane);
            __syncthreads();                            // readers of Wl / RS of the previous pass are done
            nf_cc_wT_store(wv, Wl, wid, lane);
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const int ng = 9 * 4;
            const int g0 = (kq * ng) / NKQ;
            int gcount = ng / NKQ;
            asm volatile("" : "+s"(gcount));
            float gyv[OWN];                             // pass-through gradient of the untouched half: in flight under the K loop
            if (cpl) {
#pragma unroll
                for (int rr = 0; rr < OWN; ++rr) {
                    const int ic = nf_cv_cd_row(OWN * kq + rr, hs);
                    gyv[rr] = (pv && ic < IC) ? d.cp_g_y[b * cs.n_full + nf_cc_half_to_full(cs, 1, i0 + ic, (int)q, lgw)] : 0.f;
                }
            }
            if (FWc > 0) nf_cc_kloop_level<FWc, CSc, NKQ>(acc, Wl, Fin, fpos, c32, hs, kq);        // geometry known at compile time
            else nf_cc_kloop<9>(acc, Wl, Fin, NF_CC_TS(8, 32), NF_CC_RSW(32), CS4, g.FW, 2, fpos, c32, hs, g0, gcount);
            __syncthreads();
            nf_cc_ksplit_exchange<NKQ>(own, acc, RS, pb, kq, lane);
#pragma unroll
            for (int rr = 0; rr < OWN; ++rr) {
                const int ic = nf_cv_cd_row(OWN * kq + rr, hs);
                if (pv && ic < IC) {
                    if (!cpl) d.g_x[(b * I0 + i0 + ic) * g.HW + q] = own[rr];
                    else      // the untouched half: its pass-through gradient + the conditioner's input gradient
                        d.cp_g_z[b * cs.n_full + nf_cc_half_to_full(cs, 1, i0 + ic, (int)q, lgw)] = gyv[rr] + own[rr];
                }
            }
        }
    }
}

template <typename K>
static inline int nf_cc_optin(K kernel) {
    static std::mutex mu;
    static std::unordered_set<const void*> done;
    const void* key = reinterpret_cast<const void*>(kernel);
    std::lock_guard<std::mutex> lock(mu);
    if (done.find(key) == done.end()) {
        hipError_t e = hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        done.insert(key);
    }
    return 0;
}

// tile of whole samples: 256 pixels for 16 x 16 (one sample), 128 for smaller maps (two 8 x 8, eight 4 x 4 ...)
static int nf_cc_capacity();
// 16 x 16 maps split a sample over two 128-pixel tiles (halo rows handed over between the two workgroups: 2 B <= 128 workgroups on twice
// the compute units) when the grid fits, else one 256-pixel tile per sample
static int nf_cc_halo_on() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("NF_CONV_HALO"); on = (e == nullptr || e[0] != '0') ? 1 : 0; }
    return on;
}
static inline int nf_cc_tile_px(int64_t B, int H, int W) {
    if (H * W < 256) return 128;
    if (H * W == 256 && W == 16 && nf_cc_halo_on() && 2 * B <= NF_CC_MAX_BLOCKS && 2 * B <= nf_cc_capacity()) return 128;
    return 256;
}

static int nf_cc_capacity() {
    static int cap = -1;
    if (cap < 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        cap = cus;                                      // one 1024-thread workgroup with > 80 KB of LDS per compute unit
    }
    return cap;
}

template <int NPB, int NKQ>
static inline size_t nf_cc_lds_bytes(const NfCvGeo& g, int OCB) { return sizeof(float) * (size_t)nf_cc_lds<NPB, NKQ>(g.CS, OCB).total; }

extern "C" int nf_convnet_chain_usable(int64_t B, int I0, int O_out, int H, int W) {
    NfCvGeo g;
    if (B < 1 || I0 < 1 || I0 > NF_CV_MAX_I || O_out < 1 || O_out > NF_CV_MAX_O) return 0;
    const int PX = nf_cc_tile_px(B, H, W);
    if (H * W > PX && !(H * W == 256 && PX == 128)) return 0;   // whole samples per workgroup, or the two-tile split of 16 x 16
    if (!nf_cv_geometry(g, B, H, W, 3, PX)) return 0;
    if (g.tiles > NF_CC_MAX_BLOCKS || g.tiles > nf_cc_capacity()) return 0;
    if (!(B * 192 * (int64_t)H * W < (int64_t)1 << 31)) return 0;
    const int OCB = (O_out + 31) / 32;
    if ((PX == 256 ? nf_cc_lds_bytes<8, 2>(g, OCB) : nf_cc_lds_bytes<4, 4>(g, OCB)) > 160 * 1024) return 0;
    return 1;
}

// the split map of the coupling fused into a chain launch (the conditioner runs on the half's shape (Ch, H, W))
static bool nf_cc_coupling_split(NfSplit& cs, int mode, int odd, int C, int I0, int O_out, int H, int W) {
    if (mode != NF_SPLIT_CHANNEL && mode != NF_SPLIT_CHECKER) return false;
    const bool ck = mode == NF_SPLIT_CHECKER;
    if (!nf_make_split(cs, mode, odd, C, ck ? 2 * H : H, ck ? 2 * W : W)) return false;
    return cs.Ch == I0 && O_out == 2 * I0 && cs.h == H && cs.w == W;
}

extern "C" int nf_convnet_chain_ws_floats(int64_t B, int I0, int O_out, int H, int W) {
    if (!nf_convnet_chain_usable(B, I0, O_out, H, W)) return 0;
    const int PX = nf_cc_tile_px(B, H, W);
    int64_t slots = NF_CC_STAT_SLOTS;
    if (H * W > PX) slots += (int64_t)NF_CC_NB * NF_CC_MAX_BLOCKS * NF_CC_HALO_SLOTS(W);
    return (int)(2 * slots);
}

extern "C" int nf_convnet_chain_fwd(const nf_convnet_desc* desc, int64_t B, int I0, int O_out, int H, int W, int training,
                                    float bn_eps, float bn_momentum, nf_stream_t stream) {
    if (desc == nullptr || !nf_convnet_chain_usable(B, I0, O_out, H, W)) return NF_E_BADARG;
    NfSplit cs;
    memset(&cs, 0, sizeof(cs));
    if (desc->cp_z != nullptr) {
        if (desc->cp_y == nullptr || desc->cp_ld == nullptr || desc->cp_a == nullptr || desc->cp_c == nullptr) return NF_E_BADARG;
        if (!nf_cc_coupling_split(cs, desc->cp_mode, desc->cp_odd, desc->cp_C, I0, O_out, H, W)) return NF_E_BADARG;
    }
    NfCvGeo g;
    const int PX = nf_cc_tile_px(B, H, W);
    if (!nf_cv_geometry(g, B, H, W, 3, PX)) return NF_E_BADARG;
    if ((training || H * W > PX) && desc->ws_zero == nullptr) return NF_E_BADARG;     // exchange slots (statistics; halo rows)
    const int OCB = (O_out + 31) / 32;
    hipStream_t st = (hipStream_t)stream;
    int rc = 0;
    const bool cp = desc->cp_z != nullptr;
#define NF_CC_FWD(NPB_, NKQ_, HALO_, CP_, FW_, CS_)                                                                                         \
    do {                                                                                                                                    \
        rc = nf_cc_optin(k_convnet_chain_fwd<NPB_, NKQ_, HALO_, CP_, FW_, CS_>);                                                            \
        if (rc == 0)                                                                                                                        \
            hipLaunchKernelGGL((k_convnet_chain_fwd<NPB_, NKQ_, HALO_, CP_, FW_, CS_>), dim3((unsigned)g.tiles), dim3(NF_CV_THREADS),       \
                               (nf_cc_lds_bytes<NPB_, NKQ_>(g, OCB)), st, *desc, g, I0, O_out, training, bn_eps, bn_momentum, cs);          \
    } while (0)
#define NF_CC_FWD2(NPB_, NKQ_, HALO_, FW_, CS_)                                                                                             \
    do {                                                                                                                                    \
        if (cp) NF_CC_FWD(NPB_, NKQ_, HALO_, true, FW_, CS_);                                                                               \
        else NF_CC_FWD(NPB_, NKQ_, HALO_, false, FW_, CS_);                                                                                 \
    } while (0)
    // one instantiation per level of the CIFAR pyramid (frame width / channel stride as compile-time constants), a generic one for the rest
    if (PX == 256) NF_CC_FWD2(8, 2, false, 0, 0);
    else if (H * W > PX) {                              // a sample over several workgroups: the variant with the halo hand-over
        if (g.FW == 18 && g.CS == 181) NF_CC_FWD2(4, 4, true, 18, 181);
        else NF_CC_FWD2(4, 4, true, 0, 0);
    } else if (g.FW == 10 && g.CS == 201) NF_CC_FWD2(4, 4, false, 10, 201);
    else if (g.FW == 6 && g.CS == 289) NF_CC_FWD2(4, 4, false, 6, 289);
    else NF_CC_FWD2(4, 4, false, 0, 0);
#undef NF_CC_FWD2
#undef NF_CC_FWD
    if (rc) return rc;
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_convnet_chain_bwd(const nf_convnet_bwd_desc* desc, int64_t B, int I0, int O_out, int H, int W, int training,
                                    nf_stream_t stream) {
    if (desc == nullptr || !nf_convnet_chain_usable(B, I0, O_out, H, W)) return NF_E_BADARG;
    NfSplit cs;
    memset(&cs, 0, sizeof(cs));
    if (desc->cp_g_y != nullptr) {
        if (desc->cp_g_ld == nullptr || desc->cp_z == nullptr || desc->cp_out == nullptr || desc->cp_a == nullptr || desc->cp_c == nullptr ||
            desc->cp_g_z == nullptr || desc->cp_g_out == nullptr || desc->cp_g_a == nullptr || desc->cp_g_c == nullptr)
            return NF_E_BADARG;
        if (!nf_cc_coupling_split(cs, desc->cp_mode, desc->cp_odd, desc->cp_C, I0, O_out, H, W)) return NF_E_BADARG;
    } else if (desc->g_out == nullptr) {
        return NF_E_BADARG;
    }
    NfCvGeo g;
    const int PX = nf_cc_tile_px(B, H, W);
    if (!nf_cv_geometry(g, B, H, W, 3, PX)) return NF_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    int rc = 0;
    const bool cp = desc->cp_g_y != nullptr;
#define NF_CC_BWD(NPB_, NKQ_, HALO_, CP_, FW_, CS_)                                                                                         \
    do {                                                                                                                                    \
        rc = nf_cc_optin(k_convnet_chain_bwd<NPB_, NKQ_, HALO_, CP_, FW_, CS_>);                                                            \
        if (rc == 0)                                                                                                                        \
            hipLaunchKernelGGL((k_convnet_chain_bwd<NPB_, NKQ_, HALO_, CP_, FW_, CS_>), dim3((unsigned)g.tiles), dim3(NF_CV_THREADS),       \
                               (nf_cc_lds_bytes<NPB_, NKQ_>(g, 1)), st, *desc, g, I0, O_out, training, cs);                                 \
    } while (0)
#define NF_CC_BWD2(NPB_, NKQ_, HALO_, FW_, CS_)                                                                                             \
    do {                                                                                                                                    \
        if (cp) NF_CC_BWD(NPB_, NKQ_, HALO_, true, FW_, CS_);                                                                               \
        else NF_CC_BWD(NPB_, NKQ_, HALO_, false, FW_, CS_);                                                                                 \
    } while (0)
    if (PX == 256) NF_CC_BWD2(8, 2, false, 0, 0);
    else if (H * W > PX) {
        if (g.FW == 18 && g.CS == 181) NF_CC_BWD2(4, 4, true, 18, 181);
        else NF_CC_BWD2(4, 4, true, 0, 0);
    } else if (g.FW == 10 && g.CS == 201) NF_CC_BWD2(4, 4, false, 10, 201);
    else if (g.FW == 6 && g.CS == 289) NF_CC_BWD2(4, 4, false, 6, 289);
    else NF_CC_BWD2(4, 4, false, 0, 0);
#undef NF_CC_BWD2
#undef NF_CC_BWD
    if (rc) return rc;
    NF_CHECK_LAUNCH();
    return 0;
}
