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
#include "nf_det.h"

NF_DET_STATE(nf_ccd)
NF_DET_HOST_API(nf_ccd)
#include "nf_bf16x3.h"
#include "nf_small_plu.h"

#define NF_CC_NL 6
#define NF_CC_NB 5
#define NF_CC_MAX_BLOCKS NF_CONVNET_MAX_BLOCKS

#define NF_CC_WG_SLOTS (NF_CC_NB * NF_CC_MAX_BLOCKS * 64)                   // 64-bit slots of the statistics exchanges, one row per workgroup
#define NF_CC_GROUP 16                                                      // two-level exchange above NF_CC_FLAT_MAX workgroups: groups of 16
#define NF_CC_MAX_GROUPS ((NF_CC_MAX_BLOCKS + NF_CC_GROUP - 1) / NF_CC_GROUP)
// Measured (tools/probes/chain_prof.py, B = 64): two levels cost 7 - 9 us per exchange at 128 workgroups and 4.4 at 64, against 4.3 and
// 2.5 us for all-to-all polling -- the second hop is a second full memory round trip behind the slowest group leader.  So the flat
// exchange serves every grid the kernels admit today; the two-level path only exists for grids beyond it (compiled out otherwise).
#define NF_CC_FLAT_MAX 128
#define NF_CC_STAT_SLOTS (NF_CC_WG_SLOTS + NF_CC_NB * NF_CC_MAX_GROUPS * 64) // + one row per group (before the halo slots)
static_assert(2 * NF_CC_STAT_SLOTS == NF_CONVNET_WS_FLOATS, "exchange workspace size in include/nfhip.h");

NF_PERSIST_STATE(nf_cc)
NF_PERSIST_HOST_API(nf_cc)

// phase stamps of workgroup 0 (tools/probes/chain_prof.py builds this file with -DNF_CC_PROF=1; 100 MHz wall clock)
#ifdef NF_CC_PROF
__device__ long long nf_cc_prof[128];
__device__ long long nf_cc_arrive[128];
extern "C" int nf_cc_arrive_read(long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(nf_cc_arrive), sizeof(long long) * 128);
}
#define NF_CC_STAMP(i)                                                                 \
    do {                                                                               \
        if (blockIdx.x == 0 && threadIdx.x == 0) nf_cc_prof[i] = wall_clock64();       \
    } while (0)
extern "C" int nf_cc_prof_read(long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(nf_cc_prof), sizeof(long long) * 128);
}
#else
#define NF_CC_STAMP(i)
#endif

// ---- fp32 convolutions on the bf16 matrix pipe: the three-way split ------------------------------------------------------------------
// A 32 -> 32 channel 3 x 3 layer over a 128-pixel tile is 36 x v_mfma_f32_32x32x2_f32 per wave, 3.9 us per layer with four waves per
// SIMD (the fp32 MFMA runs at the fp32 VECTOR rate: 1/16 of the bf16 rate) -- the largest single phase of a layer.  An fp32 value is
// EXACTLY the sum of three bf16 values (x = h + m + l: 3 x 8 significand bits, h = rne(x), m = rne(x - h), l = rne(x - h - m)), a
// product of two bf16 values is exact in fp32, and the bf16 MFMA accumulates in fp32.  So
//       a b  =  ah bh + (ah bm + am bh) + (ah bl + am bm + al bh)  +  O(2^-24 |a b|)        (the three dropped terms)
// is an fp32 product to fp32 accuracy on SIX bf16 MFMAs of K = 16 each: 6 x 32 cycles per 16 K against 8 x 64 for fp32 -- 0.375 of
// the matrix-pipe time.  Measured (tools/probes/bf16x3_probe.hip, K = 288 conditioner-like operands): max error against float64
// 6.5e-7 for the split form, 9.3e-7 for v_mfma_f32_32x32x2_f32 itself; K loop of a layer 3.94 -> 1.77 us per workgroup.
// Operands are split ONCE where they are written to LDS (an activation is read by nine taps), into three planes:
//   frame   F8[plane][octet o][frame position f][8 bf16]     channel c = 8 o + j; 16 B per (o, f): one ds_read_b128 per operand
//   weights W8[plane][slot][row][8 bf16]                     slot = (tap, octet of the K channel), row = output channel (forward) or
//                                                            input channel (transposed); slot stride 32 rows x 16 B
// v_mfma_f32_32x32x16_bf16: lane l supplies row / column l & 31 and the K indices 8 (l >> 5) + [0, 8): the two wave halves take the
// two slots of a PAIR (2 p, 2 p + 1) -- always the same tap, neighbouring octets, so every operand address is base(hs) + literal.

struct NfCcLds {            // offsets in floats
    int FA, WL, RS, KC, KB, RED, TOT, BNV, total, WPs;
};
template <int NPB, int NKQ>
__host__ __device__ inline NfCcLds nf_cc_lds(int CS, int OCB) {
    NfCcLds L;
    L.FA = 0;                                          // ONE frame, rewritten in place: a layer's output is written after every wave
    L.WL = L.FA + 3 * NF_CC_FP(CS);                    // has left the K loop that read its input (the exchanges' barriers lie between)
    L.WPs = NF_CC_WSLOTS * NF_CC_WSLOT;                // one weight image = NF_CONV_PACK_IMAGE_FLOATS: what nf_conv_weight_pack writes
    // the K-split exchange and the gather buffer of the grid exchange have their own region: the NEXT layer's weight image streams
    // into WL (direct global -> LDS loads) while they run
    L.RS = L.WL + 3 * L.WPs;
    int rs = NPB * (NKQ - 1) * 16 * NF_WAVE;
    if (rs < NF_CC_MAX_BLOCKS * 65) rs = NF_CC_MAX_BLOCKS * 65;
    L.KC = L.RS + rs;
    L.KB = L.KC + 128;                                 // kc[4][32]: scale, shift (forward) + mean, invstd (backward)
    L.RED = L.KB + 96;                                 // kb[3][32]: bias | gamma | beta of the layer under way
    L.TOT = L.RED + 2 * NPB * 32;
    L.total = L.TOT + 64;
    // the backward kernel keeps the saved statistics and parameters of all five BatchNorms here when the geometry leaves room (every
    // level but 4 x 4): a global load at each layer's top is ~0.7 us of exposed latency on the serial chain
    L.BNV = (L.total + NF_CC_NB * 128) * (int)sizeof(float) <= 160 * 1024 ? L.total : -1;
    if (L.BNV >= 0) L.total += NF_CC_NB * 128;
    (void)OCB;
    return L;
}
static_assert(3 * NF_CC_WSLOTS * NF_CC_WSLOT == NF_CONV_PACK_IMAGE_FLOATS, "image size of include/nfhip.h");

// one packed weight image (nf_conv_weight_pack) global -> LDS without passing through registers: 54 wave-instructions of 1 KiB
// (global_load_lds_dwordx4: lane i's 16 bytes land at base + 16 i), 3 or 4 per wave.  Completion: nf_cc_dma_wait + a barrier.
__device__ __forceinline__ void nf_cc_dma_image(float* W8, const float* __restrict__ img, int wid, int lane) {
    constexpr int CH = NF_CONV_PACK_IMAGE_FLOATS / 256;
    for (int c = wid; c < CH; c += NF_CV_WAVES)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(img + c * 256 + lane * 4),
                                         (__attribute__((address_space(3))) void*)(W8 + c * 256), 16, 0, 0);
}
__device__ __forceinline__ void nf_cc_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// four consecutive channels c0 .. c0 + 3 (c0 a multiple of 4) of frame position f: one 8-byte store per plane
__device__ __forceinline__ void nf_cc_frame_store4(float* F, int CS, int c0, int f, float v0, float v1, float v2, float v3) {
    bf16x2 h0, m0, l0, h1, m1, l1;
    nf_cc_split2(f32x2{v0, v1}, h0, m0, l0);
    nf_cc_split2(f32x2{v2, v3}, h1, m1, l1);
    float* p = F + (c0 >> 3) * 4 * CS + 4 * f + ((c0 >> 2) & 1) * 2;
    *(bf16x4*)(p) = bf16x4{h0[0], h0[1], h1[0], h1[1]};
    *(bf16x4*)(p + NF_CC_FP(CS)) = bf16x4{m0[0], m0[1], m1[0], m1[1]};
    *(bf16x4*)(p + 2 * NF_CC_FP(CS)) = bf16x4{l0[0], l0[1], l1[0], l1[1]};
}
// two consecutive channels c0, c0 + 1 (c0 even) of frame position f: one 4-byte store per plane
__device__ __forceinline__ void nf_cc_frame_store2(float* F, int CS, int c0, int f, float v0, float v1) {
    bf16x2 h, m, l;
    nf_cc_split2(f32x2{v0, v1}, h, m, l);
    float* p = F + (c0 >> 3) * 4 * CS + 4 * f + ((c0 & 7) >> 1);
    *(bf16x2*)(p) = h;
    *(bf16x2*)(p + NF_CC_FP(CS)) = m;
    *(bf16x2*)(p + 2 * NF_CC_FP(CS)) = l;
}
// one channel c of frame position f: three 2-byte stores
__device__ __forceinline__ void nf_cc_frame_store1(float* F, int CS, int c, int f, float v) {
    __bf16 h, m, l;
    nf_cc_split1(v, h, m, l);
    __bf16* p = (__bf16*)(F + (c >> 3) * 4 * CS + 4 * f) + (c & 7);
    p[0] = h;
    p[2 * NF_CC_FP(CS)] = m;
    p[4 * NF_CC_FP(CS)] = l;
}
// one weight: row r, K channel k of slot `slot` (plane stride WPs floats)
__device__ __forceinline__ void nf_cc_w_put(float* W8, int WPs, int slot, int row, int k8, float v) {
    __bf16 h, m, l;
    nf_cc_split1(v, h, m, l);
    __bf16* p = (__bf16*)(W8 + slot * NF_CC_WSLOT + 4 * row) + k8;
    p[0] = h;
    p[2 * WPs] = m;
    p[4 * WPs] = l;
}


// the generic K loop: acc += sum over the pairs [p0, p0 + np) of W8(slot)[row] * F8(tap offset, octet)[pixel]; slot = tap * noct + octet
// (noct octets of K channels per tap: 1 .. 4), a wave half takes slot 2 p + hs.  Slots >= T * noct read the zero slot of the weights
// (their B operand is any valid frame position).  Addresses are computed per pair: this version serves the first convolution's
// ragged chunks and the 1 x 1 output layer (T = 1); the hidden layers run nf_cc_kloop_fixed.
template <int T>
__device__ __forceinline__ void nf_cc_kloop(f32x16& acc, const float* W8, const float* F8, int WPs, int CS, int FW, int noct, int fpos, int row,
                                            int hs, int p0, int np) {
    const int nslots = T * noct;
    const int FPs = NF_CC_FP(CS);
    for (int p = p0; p < p0 + np; ++p) {
        const int slot = 2 * p + hs;
        const bool live = slot < nslots;
        const int sl = live ? slot : 0;
        const int tap = T == 1 ? 0 : sl / noct, o = sl - tap * noct;
        const int dy = T == 9 ? tap / 3 - 1 : 0, dx = T == 9 ? tap - (tap / 3) * 3 - 1 : 0;
        const float* wa = W8 + (live ? slot : nslots) * NF_CC_WSLOT + 4 * row;
        const float* fb = F8 + o * 4 * CS + 4 * (fpos + dy * FW + dx);
        const bf16x8 ah = *(const bf16x8*)(wa), am = *(const bf16x8*)(wa + WPs), al = *(const bf16x8*)(wa + 2 * WPs);
        const bf16x8 bh = *(const bf16x8*)(fb), bm = *(const bf16x8*)(fb + FPs), bl = *(const bf16x8*)(fb + 2 * FPs);
        NF_CC_MFMA6(ah, am, al, bh, bm, bl);
    }
}

// The K loop of the 32 -> 32 channel 3 x 3 layers with the frame geometry known at compile time (FW, CS: one instantiation per pyramid
// level): every operand address is base + IMMEDIATE (the ds_read offset field), six ds_read_b128 and six MFMAs per pair with no address
// arithmetic in between.  Pair p = slots (2 p, 2 p + 1) = tap p >> 1, octets 2 (p & 1) + hs.  Straight-line, one pair of operands in
// flight ahead of the MFMAs.
template <int FW, int CS, int P0, int NP>
__device__ __forceinline__ void nf_cc_kloop_fixed(f32x16& acc, const float* wbase, const float* fbase) {
    constexpr int WPs = NF_CC_WSLOTS * NF_CC_WSLOT, FPs = NF_CC_FP(CS);
    bf16x8 a[2][3], b[2][3];
#define NF_CC_OFFA(pp) (2 * (pp) * NF_CC_WSLOT)
#define NF_CC_OFFB(pp) (4 * (((((pp) >> 1) / 3) - 1) * FW + ((((pp) >> 1) % 3) - 1)) + 2 * ((pp) & 1) * 4 * CS)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        a[0][q] = *(const bf16x8*)(wbase + NF_CC_OFFA(P0) + q * WPs);
        b[0][q] = *(const bf16x8*)(fbase + NF_CC_OFFB(P0) + q * FPs);
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        if (i + 1 < NP) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                a[(i + 1) & 1][q] = *(const bf16x8*)(wbase + NF_CC_OFFA(P0 + i + 1) + q * WPs);
                b[(i + 1) & 1][q] = *(const bf16x8*)(fbase + NF_CC_OFFB(P0 + i + 1) + q * FPs);
            }
        }
        NF_CC_MFMA6(a[i & 1][0], a[i & 1][1], a[i & 1][2], b[i & 1][0], b[i & 1][1], b[i & 1][2]);
    }
#undef NF_CC_OFFA
#undef NF_CC_OFFB
}
// 18 pairs over the NKQ waves of a pixel block: 9 + 9, or 5 + 5 + 4 + 4 (a SIMD holds one wave of every quarter: 18 pairs per SIMD
// either way)
template <int FW, int CS, int NKQ>
__device__ __forceinline__ void nf_cc_kloop_level(f32x16& acc, const float* W8, const float* F8, int fpos, int row, int hs, int kq) {
    const float* wbase = W8 + hs * NF_CC_WSLOT + 4 * row;
    const float* fbase = F8 + hs * 4 * CS + 4 * fpos;
    if (NKQ == 2) {
        if (kq == 0) nf_cc_kloop_fixed<FW, CS, 0, 9>(acc, wbase, fbase);            // wave-uniform
        else nf_cc_kloop_fixed<FW, CS, 9, 9>(acc, wbase, fbase);
    } else if (NKQ == 16) {                             // 2 + 2 + 1 x 14 (wave-uniform switch)
        switch (kq) {
            case 0: nf_cc_kloop_fixed<FW, CS, 0, 2>(acc, wbase, fbase); break;
            case 1: nf_cc_kloop_fixed<FW, CS, 2, 2>(acc, wbase, fbase); break;
            case 2: nf_cc_kloop_fixed<FW, CS, 4, 1>(acc, wbase, fbase); break;
            case 3: nf_cc_kloop_fixed<FW, CS, 5, 1>(acc, wbase, fbase); break;
            case 4: nf_cc_kloop_fixed<FW, CS, 6, 1>(acc, wbase, fbase); break;
            case 5: nf_cc_kloop_fixed<FW, CS, 7, 1>(acc, wbase, fbase); break;
            case 6: nf_cc_kloop_fixed<FW, CS, 8, 1>(acc, wbase, fbase); break;
            case 7: nf_cc_kloop_fixed<FW, CS, 9, 1>(acc, wbase, fbase); break;
            case 8: nf_cc_kloop_fixed<FW, CS, 10, 1>(acc, wbase, fbase); break;
            case 9: nf_cc_kloop_fixed<FW, CS, 11, 1>(acc, wbase, fbase); break;
            case 10: nf_cc_kloop_fixed<FW, CS, 12, 1>(acc, wbase, fbase); break;
            case 11: nf_cc_kloop_fixed<FW, CS, 13, 1>(acc, wbase, fbase); break;
            case 12: nf_cc_kloop_fixed<FW, CS, 14, 1>(acc, wbase, fbase); break;
            case 13: nf_cc_kloop_fixed<FW, CS, 15, 1>(acc, wbase, fbase); break;
            case 14: nf_cc_kloop_fixed<FW, CS, 16, 1>(acc, wbase, fbase); break;
            default: nf_cc_kloop_fixed<FW, CS, 17, 1>(acc, wbase, fbase); break;
        }
    } else if (NKQ == 8) {                              // 3 + 3 + 2 x 6
        if (kq == 0) nf_cc_kloop_fixed<FW, CS, 0, 3>(acc, wbase, fbase);
        else if (kq == 1) nf_cc_kloop_fixed<FW, CS, 3, 3>(acc, wbase, fbase);
        else if (kq == 2) nf_cc_kloop_fixed<FW, CS, 6, 2>(acc, wbase, fbase);
        else if (kq == 3) nf_cc_kloop_fixed<FW, CS, 8, 2>(acc, wbase, fbase);
        else if (kq == 4) nf_cc_kloop_fixed<FW, CS, 10, 2>(acc, wbase, fbase);
        else if (kq == 5) nf_cc_kloop_fixed<FW, CS, 12, 2>(acc, wbase, fbase);
        else if (kq == 6) nf_cc_kloop_fixed<FW, CS, 14, 2>(acc, wbase, fbase);
        else nf_cc_kloop_fixed<FW, CS, 16, 2>(acc, wbase, fbase);
    } else {
        if (kq == 0) nf_cc_kloop_fixed<FW, CS, 0, 5>(acc, wbase, fbase);
        else if (kq == 1) nf_cc_kloop_fixed<FW, CS, 5, 5>(acc, wbase, fbase);
        else if (kq == 2) nf_cc_kloop_fixed<FW, CS, 10, 4>(acc, wbase, fbase);
        else nf_cc_kloop_fixed<FW, CS, 14, 4>(acc, wbase, fbase);
    }
}
// the same split for the generic loop: first pair and pair count of quarter kq out of `pairs`
__device__ __forceinline__ void nf_cc_pair_range(int pairs, int nkq, int kq, int& p0, int& np) {
    const int base = pairs / nkq, extra = pairs - base * nkq;
    p0 = __builtin_amdgcn_readfirstlane(kq * base + (kq < extra ? kq : extra));      // wave-uniform: scalar registers
    np = __builtin_amdgcn_readfirstlane(base + (kq < extra ? 1 : 0));
}

// 3x3 weights of one chunk of IC (padded to ICP = 8 noct) input channels, global (32, I, 3, 3), as W8 ITEMS: an item is the eight K
// values of one (slot, row) -- what one lane-half consumes per MFMA -- so staging is eight loads, a split and three 16-byte stores.
//   forward     K = input channel:  item (oc, octet o, tap) -> slot tap * noct + o, row oc;   values W[oc][i0 + 8 o + j][tap]
//   transposed  K = output channel: item (octet o, r = ic * 9 + tap) -> slot (8 - tap) * 4 + o, row ic;   values W[8 o + j][i0 + ic][tap]
//               (lanes run over r: contiguous in global memory for every output channel)
// 32 * 9 * noct items (<= 1152) over 1024 threads: threads < 128 hold a second one.  Channels beyond IC load zeros.
struct NfCcW { float v[2][8]; };
template <bool TR>
__device__ __forceinline__ void nf_cc_w_load(NfCcW& w, const float* __restrict__ weight, int I, int i0, int IC, int noct) {
    const int nitems = 32 * 9 * (TR ? 4 : noct);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int item = threadIdx.x + NF_CV_THREADS * t;
        if (TR) {
            const int o = item / 288, r = item - o * 288;
#pragma unroll
            for (int j = 0; j < 8; ++j) w.v[t][j] = (item < nitems && r < IC * 9) ? weight[((8 * o + j) * I + i0) * 9 + r] : 0.f;
        } else {
            const int per = 9 * noct;
            const int oc = item / per, rem = item - oc * per, o = rem / 9, tap = rem - o * 9;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                w.v[t][j] = (item < nitems && 8 * o + j < IC) ? weight[((oc * I + i0 + 8 * o + j) * 9) + tap] : 0.f;
        }
    }
}
template <bool TR>
__device__ __forceinline__ void nf_cc_w_store(const NfCcW& w, float* W8, int WPs, int noct) {
    const int nitems = 32 * 9 * (TR ? 4 : noct);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int item = threadIdx.x + NF_CV_THREADS * t;
        if (item < nitems) {
            if (TR) {
                const int o = item / 288, r = item - o * 288, ic = r / 9, tap = r - ic * 9;
                nf_cc_w_put8(W8, WPs, (8 - tap) * 4 + o, ic, w.v[t]);
            } else {
                const int per = 9 * noct;
                const int oc = item / per, rem = item - oc * per, o = rem / 9, tap = rem - o * 9;
                nf_cc_w_put8(W8, WPs, tap * noct + o, oc, w.v[t]);
            }
        }
    }
}
// slot `slot` of all three planes <- 0 (what the half of an odd last pair multiplies with)
__device__ __forceinline__ void nf_cc_w_zero_slot(float* W8, int WPs, int slot) {
    if (threadIdx.x < 3 * NF_CC_WSLOT) W8[(threadIdx.x / NF_CC_WSLOT) * WPs + slot * NF_CC_WSLOT + (threadIdx.x % NF_CC_WSLOT)] = 0.f;
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
    static_assert(OWN == 1 || OWN == 2 || OWN == 4 || OWN == 8, "16 / NKQ values per lane");
    if constexpr (OWN == 1) {                           // whole merges only
        S = v[0];
        M2 = 0.f;
        for (int m = 1; m < 16; m *= 2) nf_cc_merge(S, M2, __shfl_xor(S, m, NF_WAVE), __shfl_xor(M2, m, NF_WAVE), 0.5f / (float)m);
        const float So = __shfl_xor(S, 16, NF_WAVE), Mo = __shfl_xor(M2, 16, NF_WAVE);
        const bool me_ok = (c32 & 16) ? hi_ok : lo_ok, ot_ok = (c32 & 16) ? lo_ok : hi_ok;
        if (me_ok && ot_ok) nf_cc_merge(S, M2, So, Mo, 0.5f / 16.f);
        else if (ot_ok) { S = So; M2 = Mo; }
        which = 0;
        return;
    }
    if constexpr (OWN == 2) {                           // one halving step, then whole merges
        const bool up = c32 & 1;
        S = up ? v[1] : v[0];
        M2 = 0.f;
        nf_cc_merge(S, M2, __shfl_xor(up ? v[0] : v[1], 1, NF_WAVE), 0.f, 0.5f);
        for (int m = 2; m < 16; m *= 2) nf_cc_merge(S, M2, __shfl_xor(S, m, NF_WAVE), __shfl_xor(M2, m, NF_WAVE), 0.5f / (float)m);
        const float So = __shfl_xor(S, 16, NF_WAVE), Mo = __shfl_xor(M2, 16, NF_WAVE);
        const bool me_ok = (c32 & 16) ? hi_ok : lo_ok, ot_ok = (c32 & 16) ? lo_ok : hi_ok;
        if (me_ok && ot_ok) nf_cc_merge(S, M2, So, Mo, 0.5f / 16.f);
        else if (ot_ok) { S = So; M2 = Mo; }
        which = up ? 1 : 0;
        return;
    }
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
// (sum, M2 about the overall mean) of channel ci over n rows of the gather buffer -- row b = (sum, M2 about its own mean) of the rowpx
// pixels that start at pixel px0 + b rowpx of the batch (the last row may hold fewer, or none) -- by the 32 lanes l of a wave half:
// per-lane partials in a fixed order, then a butterfly.  Every lane returns the totals.
__device__ __forceinline__ void nf_cc_merge_rows(const float* xs, int XS, int n, int64_t Npx, int64_t px0, int rowpx, int ci, int l,
                                                 float& S, float& M2) {
    float ps = 0.f;
    for (int b = l; b < n; b += 32) ps += xs[b * XS + ci];
    S = nf_cc_sum32(ps);
    const int64_t left = Npx - px0;
    const int64_t ntot = left < (int64_t)n * rowpx ? left : (int64_t)n * rowpx;
    const float mean = S / (float)ntot;
    float pm = 0.f;
    if (ntot == (int64_t)n * rowpx) {                   // every row full (uniform): no per-row pixel counts
        const float inv_full = 1.f / (float)rowpx, fpx = (float)rowpx;
        for (int b = l; b < n; b += 32) {
            const float dlt = xs[b * XS + ci] * inv_full - mean;
            pm += fmaf(fpx * dlt, dlt, xs[b * XS + 32 + ci]);
        }
    } else {
        for (int b = l; b < n; b += 32) {
            const int nb = nf_cc_valid_px(Npx, px0 + (int64_t)b * rowpx, rowpx);
            const float dlt = xs[b * XS + ci] / (float)max(nb, 1) - mean;
            pm += nb > 0 ? fmaf((float)nb * dlt, dlt, xs[b * XS + 32 + ci]) : 0.f;
        }
    }
    M2 = nf_cc_sum32(pm);
}

// Returns true in the ONE lane per channel that holds the channel's totals (ci, S, M2) -- the caller finishes the BatchNorm constants
// there and then synchronises: no barrier between the merge and the constants.
template <int NPB>
__device__ __forceinline__ bool nf_cc_stats_exchange(float* sm, const NfCcLds& L, unsigned long long* slots, unsigned gbase, int round,
                                                     int64_t Npx, int PXW, int& ci_out, float& S_out, float& M2_out) {
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
            ci_out = i; S_out = S; M2_out = M2;
        } else {
            unsigned long long* dst = slots + ((size_t)round * NF_CC_MAX_BLOCKS + blockIdx.x) * 64 + i;
            __hip_atomic_store(dst, ((unsigned long long)(gbase + round + 1) << 32) | (unsigned long long)__float_as_uint(S), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst + 32, ((unsigned long long)(gbase + round + 1) << 32) | (unsigned long long)__float_as_uint(M2),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (G == 1) return threadIdx.x < 32;
    const unsigned long long* rs = slots + (size_t)round * NF_CC_MAX_BLOCKS * 64;
    const unsigned gen = gbase + (unsigned)(round + 1);
    if (round == 1) NF_CC_STAMP(57);
#ifdef NF_CC_PROF
    if (round == 1 && threadIdx.x == 0) nf_cc_arrive[blockIdx.x] = wall_clock64();
#endif
    const int XS = 65;
    const int lane = threadIdx.x & 63, ci = 2 * (threadIdx.x >> 6) + (lane >> 5), l = lane & 31;
    float S, M2;
    if (NF_CC_MAX_BLOCKS <= NF_CC_FLAT_MAX || G <= NF_CC_FLAT_MAX) {
        // every workgroup polls every row: thread (wave w, half h, lane l) reduces channel ci = 2 w + h over the workgroups b = l,
        // l + 32, ... in a fixed order, then a butterfly over the 32 lanes
        nf_cc_collect_slots(xs, rs, gen, G, XS);
        if (round == 1) NF_CC_STAMP(58);
        __syncthreads();
        if (round == 1) NF_CC_STAMP(59);
        nf_cc_merge_rows(xs, XS, G, Npx, 0, PXW, ci, l, S, M2);
    } else {
        // two levels (all-to-all polling moves G x G x 512 bytes per round: 8 MB at 128 workgroups): the first workgroup of every
        // group of 16 merges its group's rows and publishes the group's (sum, M2 about the group mean); everybody polls the <= 16
        // group rows.  Fixed order on both levels: deterministic.
        const int grp = blockIdx.x / NF_CC_GROUP, NG = (G + NF_CC_GROUP - 1) / NF_CC_GROUP;
        unsigned long long* gs = slots + NF_CC_WG_SLOTS + (size_t)round * NF_CC_MAX_GROUPS * 64;
        if (blockIdx.x % NF_CC_GROUP == 0) {            // block-uniform
            const int nmem = min(NF_CC_GROUP, G - grp * NF_CC_GROUP);
            nf_cc_collect_slots(xs, rs + (size_t)grp * NF_CC_GROUP * 64, gen, nmem, XS);
            __syncthreads();
            float Sg, Mg;
            nf_cc_merge_rows(xs, XS, nmem, Npx, (int64_t)grp * NF_CC_GROUP * PXW, PXW, ci, l, Sg, Mg);
            if (l == 0) {
                __hip_atomic_store(gs + grp * 64 + ci, ((unsigned long long)gen << 32) | (unsigned long long)__float_as_uint(Sg),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gs + grp * 64 + 32 + ci, ((unsigned long long)gen << 32) | (unsigned long long)__float_as_uint(Mg),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();                            // the group's rows are read before the gather buffer is reused
        }
        nf_cc_collect_slots(xs, gs, gen, NG, XS);
        if (round == 1) NF_CC_STAMP(58);
        __syncthreads();
        if (round == 1) NF_CC_STAMP(59);
        nf_cc_merge_rows(xs, XS, NG, Npx, 0, NF_CC_GROUP * PXW, ci, l, S, M2);
    }
    ci_out = ci; S_out = S; M2_out = M2;
    (void)tot;
    return l == 0;
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

#define NF_CC_HALO_SLOTS(W) (2 * 32 * (W))                                  // per layer and destination tile
template <int OWN>
__device__ __forceinline__ void nf_cc_halo_publish(unsigned long long* hslots, unsigned gbase, int layer, const NfCvGeo& g, int tile, int y0,
                                                   int px, int kq, int hs, const float (&v)[OWN]) {
    const int row = px >> g.lgW, x = px & (g.W - 1);
    const bool first = row == 0 && y0 > 0;                       // our first row is the row BELOW the previous tile's last row
    const bool last = row == g.TH - 1 && y0 + g.TH < g.H;        // our last row is the row ABOVE the next tile's first row
    if (!(first || last)) return;
    const int dst = first ? tile - 1 : tile + 1, s = first ? 1 : 0;
    unsigned long long* base = hslots + ((size_t)layer * NF_CC_MAX_BLOCKS + dst) * NF_CC_HALO_SLOTS(g.W) + s * 32 * g.W + x;
    const unsigned long long gen = (unsigned long long)(gbase + layer + 1) << 32;
#pragma unroll
    for (int rr = 0; rr < OWN; ++rr) {
        const int c = nf_cv_cd_row(OWN * kq + rr, hs);
        __hip_atomic_store(base + c * g.W, gen | (unsigned long long)__float_as_uint(v[rr]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// thread t < 32 W: channel t / W, column t % W of BOTH halo rows (s = 0, 1; an image border has none).  Returns the frame position of
// the value this call delivers in `v`, or -1; call once per row s.
__device__ __forceinline__ int nf_cc_halo_poll(const unsigned long long* hslots, unsigned gbase, int layer, const NfCvGeo& g, int tile, int y0,
                                               int s, float& v) {
    const bool has = s == 0 ? y0 > 0 : y0 + g.TH < g.H;
    if (!has) return -1;
    const int t = threadIdx.x, x = t & (g.W - 1);
    const unsigned long long* p = hslots + ((size_t)layer * NF_CC_MAX_BLOCKS + tile) * NF_CC_HALO_SLOTS(g.W) + s * 32 * g.W + t;
    const unsigned gen = gbase + (unsigned)(layer + 1);
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

// ---- the head of an image Glow step in the prologue of the forward launch ------------------------------------------------------------
// h = W ((x - bias) / exp(ls)) per full-resolution pixel (ActNorm, flows/modules.py:246-249; invertible 1 x 1 convolution with its weight
// assembled, :471-480), the arithmetic of k_glow_head_w_fwd (csrc/glow_head_mfma.hip): 16-pixel blocks, one per wave at a time, on
// v_mfma_f32_16x16x4_f32 with the pixels as columns -- A[i = li][k = lk] = W[16 rt + li][4 q + lk] from LDS, B[k = lk][j = li] = the
// normalised input -- and the reference's own rounding of the normalisation (a true division).
// The workgroup computes the pixels it OWNS (whole samples; with the halo hand-over its rows of the sample) -> h (= cp_z, kept for the
// backward) and y (= cp_y: the coupling's epilogue overwrites the transformed half), plus the rows its layer-0 frame reads from the
// NEIGHBOUR tile (full rows fr_lo .. fr_hi - 1 around the owned fo_lo .. fo_hi - 1): of those only the conditioning half x1 is produced.
// x1 goes to memory (the weight-gradient pass of convolution 0 reads it) AND into an LDS buffer X1 [sample][channel][half pixel - sp0]
// from which the layer-0 frame is built: no store -> load round trip, no workgroup waits for another before its first convolution.
// What made a first version cost as much as the launch it replaced (24.37 ms per C4 step either way) was its latency chain -- W and x
// requested one after the other, the frame read back from memory: here the first block's pixels, W and the ActNorm vectors are
// requested up front (nf_cc_head_request, before the frame is zeroed) and arrive under that work.
// LDS (region RS, free until the first K-split exchange): Ws [64][65] | An [2][64] | X1.
#define NF_CC_HD_WS 0
#define NF_CC_HD_AN (64 * 65)
#define NF_CC_HD_X1 (NF_CC_HD_AN + 128)
#define NF_CC_HD_X1_MAX 3840                             // floats: the 4 x 4 level holds 2 samples x 96 channels x 16 pixels
static_assert(NF_CC_HD_X1 + NF_CC_HD_X1_MAX <= NF_CC_MAX_BLOCKS * 65, "the head's LDS fits the exchange region");
struct NfCcHeadReq {                                    // what a thread has in flight for the head
    float w[3];                                         // W elements t, t + 1024, t + 2048
    float an0, an1;                                     // threads < 64: bias, log_scale of channel t
    float xv[16];                                       // the wave's first block: channel 4 q + lk at pixel li
};
// Work items: (row tile of 16 output channels, 16-pixel block), item = rt * nblk + blk, a wave takes items wid, wid + 16, ...: at the
// CIFAR levels a wave's items share their block (nblk = 8 | 16 | 40), whose pixels are loaded and normalised once.  One accumulator
// tile per item: the form with all row tiles of a block in registers at once (RT x 4 accumulators, RT x KQ weights, two copies of the
// pixels) cost the WHOLE kernel ~50 spilled registers -- 1 us per launch even with the head elsewhere (round 6, same-box A/B).
// KQ = K steps of four input channels (template parameter: the operand registers); the row tile is a run-time index.
template <int KQ>
__device__ __forceinline__ void nf_cc_head_request(NfCcHeadReq& R, const nf_convnet_desc& d, const NfSplit& cs, int64_t b0, int fr_lo,
                                                   int per, int nblk, int64_t B) {
    const int C = cs.C, P = cs.H * cs.W;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int e = threadIdx.x + u * NF_CV_THREADS;
        R.w[u] = e < C * C ? d.hd_W[e] : 0.f;
    }
    {
        const int c = threadIdx.x < C ? threadIdx.x : 0;
        R.an0 = d.hd_bias[c];
        R.an1 = d.hd_ls[c];
    }
    {
        const int blk = wid % nblk, sidx = blk / per, rem = blk - sidx * per;      // (item wid: row tile wid / nblk of block wid % nblk)
        const int64_t b = (b0 + sidx) < B ? b0 + sidx : b0;
        const float* xb = d.hd_x + b * cs.n_full + fr_lo * cs.W + (rem << 4) + li;
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const int c = 4 * q + lk;
            R.xv[q] = xb[(int64_t)(c < C ? c : 0) * P];
        }
    }
}
template <int KQ>
__device__ __forceinline__ void nf_cc_head_fwd(const NfCcHeadReq& R, const nf_convnet_desc& d, const NfSplit& cs, float* hl, int64_t b0,
                                               int nsamp, int fr_lo, int fo_lo, int fo_hi, int per, int nblk, int sp0, int np1,
                                               int64_t B, bool add_ld) {
    const int C = cs.C, P = cs.H * cs.W, RT = (C + 15) >> 4;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, li = lane & 15, lk = lane >> 4;
    float* const hout = const_cast<float*>(d.cp_z);
    float* const Ws = hl + NF_CC_HD_WS;
    float* const An = hl + NF_CC_HD_AN;
    float* const X1 = hl + NF_CC_HD_X1;
    const int lgWf = 31 - __clz(cs.W);
    {
        const float rc = 1.f / (float)C;               // e / C for e < 4096 by reciprocal multiplication (the half-integer offset keeps the
#pragma unroll                                          // quotient away from every integer)
        for (int u = 0; u < 3; ++u) {
            const int e = threadIdx.x + u * NF_CV_THREADS;
            const int er = (int)(((float)e + 0.5f) * rc);
            if (e < C * C) Ws[er * 65 + (e - er * C)] = R.w[u];
        }
    }
    if (threadIdx.x < 64) {
        An[threadIdx.x] = threadIdx.x < C ? R.an0 : 0.f;
        An[64 + threadIdx.x] = threadIdx.x < C ? expf(R.an1) : 1.f;
    }
    NF_CC_STAMP(108);
    if (wid == 1 && add_ld) {                           // log-det of the two layers: P (sum log_s - sum ls), one add per owned sample
        float sl = lane < C ? d.hd_log_s[lane] - d.hd_ls[lane] : 0.f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sl += __shfl_xor(sl, off, NF_WAVE);
        if (lane < nsamp && b0 + lane < B) atomicAdd(d.cp_ld + b0 + lane, (float)P * sl);      // (one writer per sample: this workgroup)
    }
    __syncthreads();
    NF_CC_STAMP(109);
    const int nitem = nblk * RT;
    int have = -1;                                      // the block whose normalised pixels bv holds
    float bv[KQ];
#pragma unroll 1
    for (int item = wid; item < nitem; item += NF_CV_WAVES) {
        const int rt = item / nblk, blk = item - rt * nblk;
        const int sidx = blk / per, rem = blk - sidx * per;
        const int64_t b = b0 + sidx;
        if (b >= B) continue;                           // (wave-uniform)
        const int p = fr_lo * cs.W + (rem << 4) + li;
        const int row = p >> lgWf, xx = p & (cs.W - 1);          // (power-of-two maps: no division anywhere in the element loop -- an integer
        const bool own = row >= fo_lo && row < fo_hi;            //  division is ~40 instructions, two per output element were 8 us per launch)
        if (blk != have) {                              // (wave-uniform) the block's pixels: the first item's came with the request
            const float* xb = d.hd_x + b * cs.n_full + p;
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                const int c = 4 * q + lk, cc = min(c, C - 1);
                const float x = item == wid ? R.xv[q] : xb[(int64_t)cc * P];
                const float v = (x - An[cc]) / An[64 + cc];                  // (the reference's own rounding: a true division)
                bv[q] = c < C ? v : 0.f;
            }
            have = blk;
        }
        // every LDS read unconditional (clamped index, then a select): behind a branch each is a round trip of its own in front of its MFMA
        const int ra = 16 * rt + li;
        const float* wrow = Ws + min(ra, C - 1) * 65;
        float av[KQ];
#pragma unroll
        for (int q = 0; q < KQ; ++q) av[q] = wrow[min(4 * q + lk, C - 1)];
        NF_CC_STAMP(111);
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const float a = (ra < C && 4 * q + lk < C) ? av[q] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[q], acc, 0, 0, 0);
        }
        NF_CC_STAMP(112);
        float* hb = hout + b * cs.n_full + p;
        float* yb = d.cp_y + b * cs.n_full + p;
        float* zb = d.hd_x1 + b * cs.n_half;
#pragma unroll
        for (int j = 0; j < 4; ++j) {                   // D: column = li (pixel), row = 4 lk + j
            const int r = 16 * rt + 4 * lk + j;
            if (r < C) {
                const float v = acc[j];
                if (own) { hb[(int64_t)r * P] = v; yb[(int64_t)r * P] = v; }
                // full element (channel r, pixel (row, xx)) -> half `which`, half channel m, half pixel sp  (squeeze.py:5-10, 32-44)
                int which, m, sp;
                if (cs.mode == NF_SPLIT_CHANNEL) {
                    const int hc = C >> 1, sel = r >= hc ? 1 : 0;
                    which = sel ^ cs.odd; m = r - sel * hc; sp = p;
                } else {
                    const int k = 4 * r + 2 * (row & 1) + (xx & 1);
                    const int qd = (k >= C ? 1 : 0) + (k >= 2 * C ? 1 : 0) + (k >= 3 * C ? 1 : 0);
                    const int sel = (qd == 1 || qd == 2) ? 1 : 0;
                    which = sel ^ cs.odd; m = sel ? k - C : (qd == 0 ? k : k - 2 * C);
                    sp = (row >> 1) * cs.w + (xx >> 1);
                }
                if (which == 1) {
                    zb[m * (cs.h * cs.w) + sp] = v;
                    X1[(sidx * cs.Ch + m) * np1 + (sp - sp0)] = v;
                }
            }
        }
    }
    NF_CC_STAMP(110);
    __syncthreads();                                    // X1 is complete (LDS); the stores to memory travel on their own
}

// The same for a head of 2 .. 4 channels (the first level of an image Glow: (3, 32, 32) under the checkerboard split): a thread per
// pixel with the arithmetic of the stand-alone kernel (k_glow_head_fwd<CT>, csrc/glow_head.hip: the weight assembled from its PLU factors
// by every thread, a true division, fmaf over c ascending, the log-det term summed in channel order) -- bit-identical.  The assembled
// weight is saved for the backward pass by thread 0 of workgroup 0 (hs_Wout).
template <int CT>
__device__ __forceinline__ void nf_cc_head_small_fwd(const nf_convnet_desc& d, const NfSplit& cs, float* hl, int64_t b0, int nsamp,
                                                     int fr_lo, int fr_hi, int fo_lo, int fo_hi, int sp0, int np1, int64_t B, bool add_ld) {
    const int P = cs.H * cs.W, npx = (fr_hi - fr_lo) * cs.W;
    float* const hout = const_cast<float*>(d.cp_z);
    float* const X1 = hl + NF_CC_HD_X1;
    const int lgWf = 31 - __clz(cs.W);
    int idx = threadIdx.x, sidx = 0, p = 0;
    int64_t base = 0;
    auto place = [&](int i) {
        sidx = i / npx;
        p = fr_lo * cs.W + (i - sidx * npx);
        base = (b0 + sidx) * cs.n_full + p;
    };
    bool ok = idx < nsamp * npx;
    if (ok) {
        place(idx);
        ok = b0 + sidx < B;
    }
    float zr[CT];                                       // (the first pixel's operands are requested before the constants)
#pragma unroll
    for (int c = 0; c < CT; ++c) zr[c] = ok ? d.hd_x[base + (int64_t)c * P] : 0.f;
    float Wm[CT][CT], es[CT], bb[CT];
    nf_small_plu<CT>(d.hs_P, d.hs_L, d.hs_U, d.hs_Lm, d.hs_Um, d.hs_sign, d.hd_log_s, Wm);
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        es[c] = expf(d.hd_ls[c]);
        bb[c] = d.hd_bias[c];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && d.hs_Wout != nullptr) {
#pragma unroll
        for (int r = 0; r < CT; ++r)
#pragma unroll
            for (int c = 0; c < CT; ++c) d.hs_Wout[r * CT + c] = Wm[r][c];
    }
    if (add_ld && threadIdx.x >= NF_WAVE && threadIdx.x < NF_WAVE + nsamp && b0 + (threadIdx.x - NF_WAVE) < B) {
        float dld = 0.f;
#pragma unroll
        for (int c = 0; c < CT; ++c) dld += d.hd_log_s[c] - d.hd_ls[c];                       // modules.py:249, :480
        atomicAdd(d.cp_ld + b0 + (threadIdx.x - NF_WAVE), dld * (float)P);                     // (one writer per sample: this workgroup)
    }
    while (idx < nsamp * npx) {
        if (ok) {
            const int row = p >> lgWf, xx = p & (cs.W - 1);
            const bool own = row >= fo_lo && row < fo_hi;
            float zn[CT];
#pragma unroll
            for (int c = 0; c < CT; ++c) zn[c] = (zr[c] - bb[c]) / es[c];                      // modules.py:246
            float* zb = d.hd_x1 + (b0 + sidx) * cs.n_half;
#pragma unroll
            for (int r = 0; r < CT; ++r) {
                float a = 0.f;
#pragma unroll
                for (int c = 0; c < CT; ++c) a = fmaf(Wm[r][c], zn[c], a);                     // modules.py:477
                if (own) { hout[base + (int64_t)r * P] = a; d.cp_y[base + (int64_t)r * P] = a; }
                int which, m, sp;                       // (squeeze.py:5-10, 32-44)
                if (cs.mode == NF_SPLIT_CHANNEL) {
                    const int hc = CT >> 1, sel = r >= hc ? 1 : 0;
                    which = sel ^ cs.odd; m = r - sel * hc; sp = p;
                } else {
                    const int k = 4 * r + 2 * (row & 1) + (xx & 1);
                    const int qd = (k >= CT ? 1 : 0) + (k >= 2 * CT ? 1 : 0) + (k >= 3 * CT ? 1 : 0);
                    const int sel = (qd == 1 || qd == 2) ? 1 : 0;
                    which = sel ^ cs.odd; m = sel ? k - CT : (qd == 0 ? k : k - 2 * CT);
                    sp = (row >> 1) * cs.w + (xx >> 1);
                }
                if (which == 1) {
                    zb[m * (cs.h * cs.w) + sp] = a;
                    X1[(sidx * cs.Ch + m) * np1 + (sp - sp0)] = a;
                }
            }
        }
        idx += NF_CV_THREADS;
        ok = idx < nsamp * npx;
        if (ok) {
            place(idx);
            ok = b0 + sidx < B;
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) zr[c] = ok ? d.hd_x[base + (int64_t)c * P] : 0.f;
    }
    __syncthreads();                                    // X1 is complete (LDS); the stores to memory travel on their own
}

// ---- the data gradient of the NEXT step's head in the prologue of the backward launch --------------------------------------------------
// Backward order: chain launch of step k -> g_h (gradient at the output of head k) -> head k transposed -> g_y of coupling k - 1 ->
// chain launch of step k - 1.  The middle link was a launch of its own (k_glow_head_w_bwd<PART = 1>: 129 per C4 step, ~6 us each on the
// serial chain); here the chain launch of step k - 1 computes ITS OWN g_y first: g_y[c][p] = (sum_r W[r][c] g_h[r][p]) / exp(ls[c]) for
// the full-resolution pixels its tile owns (the same set the forward prologue owns), with the arithmetic of the stand-alone kernel
// (v_mfma_f32_16x16x4_f32, K ascending, one multiplication by 1 / exp(ls) at the end: bit-identical), written to the cp_g_y buffer
// and read back by the same workgroup after a barrier.  The parameter sums of head k (g_W, g_log_scale, g_bias) read g_h and x only and
// stay where they were (k_glow_head_w_params_multi at the end of the pass).
struct NfCcHeadBwdReq {                                 // what a thread has in flight for the head
    float w[3];                                         // W elements t, t + 1024, t + 2048
    float ls;                                           // threads < 64: log_scale of channel t
    float gv[2][16];                                    // the wave's first TWO work items (all there are at the CIFAR levels): g_h of channel
};                                                      // 4 q + lk at pixel li -- a load inside the item loop is ~2 us of exposed latency
// Work items: (16-pixel block, row tile of 16 output channels) -- a 48-channel head on the 4 x 4 level is 8 blocks x 3 row tiles over the
// sixteen waves, not 8 waves with three tiles each: the fp32 MFMA is the slow pipe (32 cycles per 16 x 16 x 4), four waves share a
// SIMD's.  KQ = K steps of four input channels (a template parameter: the operand registers), the row tile is a run-time index.
template <int KQ>
__device__ __forceinline__ void nf_cc_head_bwd_request(NfCcHeadBwdReq& R, const nf_convnet_bwd_desc& d, const NfSplit& cs, int64_t b0,
                                                       int fo_lo, int per, int nblk, int64_t B) {
    const int C = cs.C, P = cs.H * cs.W, RT = (C + 15) >> 4;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int e = threadIdx.x + u * NF_CV_THREADS;
        R.w[u] = e < C * C ? d.hd_W[e] : 0.f;
    }
    R.ls = d.hd_ls[threadIdx.x < C ? threadIdx.x : 0];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int it = wid + u * NF_CV_WAVES;
        const int item = it < nblk * RT ? it : 0, blk = item / RT, sidx = blk / per, rem = blk - sidx * per;
        const int64_t b = (b0 + sidx) < B ? b0 + sidx : b0;
        const float* gb = d.hd_g_h + b * cs.n_full + fo_lo * cs.W + (rem << 4) + li;
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const int c = 4 * q + lk;
            R.gv[u][q] = gb[(int64_t)(c < C ? c : 0) * P];
        }
    }
}
// (Carrying the results in registers to store them behind the first K loop's loads -- on gfx9 a store counts in the in-order vmcnt, so
// later waits for loads include its acknowledgement -- measured nothing here, and eight more live registers through the launch's first
// phase cost every launch 1.5 us in spills: the stores are issued where the values are produced.)
template <int KQ>
__device__ __forceinline__ void nf_cc_head_bwd(const NfCcHeadBwdReq& R, const nf_convnet_bwd_desc& d, const NfSplit& cs, float* hl,
                                               int64_t b0, int fo_lo, int per, int nblk, int64_t B, int PXW, int HWh, int sp0) {
    const int C = cs.C, P = cs.H * cs.W, RT = (C + 15) >> 4;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, li = lane & 15, lk = lane >> 4;
    float* const gx = const_cast<float*>(d.cp_g_y);
    float* const Ws = hl + NF_CC_HD_WS;                 // W TRANSPOSED: Ws[c][r] = W[r][c]
    float* const An = hl + NF_CC_HD_AN;
    float* const G0 = hl + NF_CC_HD_X1;                 // the TRANSFORMED half of the result, [half channel m][tile pixel]: what the first
    const int lgWf = 31 - __clz(cs.W);                  // transposed convolution reads (the other half is read at the launch's far end)
    {
        const float rc = 1.f / (float)C;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int e = threadIdx.x + u * NF_CV_THREADS;
            const int er = (int)(((float)e + 0.5f) * rc);
            if (e < C * C) Ws[(e - er * C) * 65 + er] = R.w[u];
        }
    }
    if (threadIdx.x < 64) An[64 + threadIdx.x] = threadIdx.x < C ? 1.f / expf(R.ls) : 1.f;
    NF_CC_STAMP(102);
    __syncthreads();
    NF_CC_STAMP(103);
    const int nitem = nblk * RT;
    // one work item: gv = its operand column (g_h of channel 4 q + lk at pixel li).  Every LDS read is unconditional (clamped index, then a select): behind a branch each read was a round trip of
    // its own in front of its MFMA -- twelve in a row, 0.7 us per item.
    auto run = [&](int item, const float* gv) {
        const int blk = item / RT, rt = item - blk * RT;
        const int sidx = blk / per, rem = blk - sidx * per;
        const int64_t b = b0 + sidx;
        const int p = fo_lo * cs.W + (rem << 4) + li;
        const int ra = 16 * rt + li;                    // A[i = li][k = lk] = W[c][r]
        const float* wrow = Ws + min(ra, C - 1) * 65;
        float av[KQ];
#pragma unroll
        for (int q = 0; q < KQ; ++q) av[q] = wrow[min(4 * q + lk, C - 1)];
        float sc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) sc[j] = An[64 + min(16 * rt + 4 * lk + j, C - 1)];
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const int c = 4 * q + lk;
            const float bv = c < C ? gv[q] : 0.f;       // B[k = lk][j = li] = g_h[c][pixel]
            const float a = (ra < C && c < C) ? av[q] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc, 0, 0, 0);
        }
        float* gxb = gx + b * cs.n_full + p;
        const int row = p >> lgWf, xx = p & (cs.W - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 16 * rt + 4 * lk + j;
            const float v = acc[j] * sc[j];
            if (r < C) gxb[(int64_t)r * P] = v;
            int which, m, sp;                           // (the forward prologue's map: squeeze.py:5-10, 32-44)
            if (cs.mode == NF_SPLIT_CHANNEL) {
                const int hc = C >> 1, sel = r >= hc ? 1 : 0;
                which = sel ^ cs.odd; m = r - sel * hc; sp = p;
            } else {
                const int k = 4 * r + 2 * (row & 1) + (xx & 1);
                const int qd = (k >= C ? 1 : 0) + (k >= 2 * C ? 1 : 0) + (k >= 3 * C ? 1 : 0);
                const int sel = (qd == 1 || qd == 2) ? 1 : 0;
                which = sel ^ cs.odd; m = sel ? k - C : (qd == 0 ? k : k - 2 * C);
                sp = (row >> 1) * cs.w + (xx >> 1);
            }
            if (r < C && which == 0) G0[m * PXW + sidx * HWh + (sp - sp0)] = v;
        }
    };
#pragma unroll
    for (int u = 0; u < 2; ++u) {                       // the two items whose operands came with the request
        const int item = wid + u * NF_CV_WAVES;
        if (item < nitem && b0 + (item / RT) / per < B) run(item, R.gv[u]);             // (wave-uniform)
    }
#pragma unroll 1
    for (int item = wid + 2 * NF_CV_WAVES; item < nitem; item += NF_CV_WAVES) {           // (none at the CIFAR levels)
        const int blk = item / RT, sidx = blk / per, rem = blk - sidx * per;
        const int64_t b = b0 + sidx;
        if (b >= B) continue;                           // (wave-uniform)
        const float* gb = d.hd_g_h + b * cs.n_full + fo_lo * cs.W + (rem << 4) + li;
        float gv[KQ];
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const int c = 4 * q + lk;
            gv[q] = gb[(int64_t)(c < C ? c : 0) * P];
        }
        run(item, gv);
    }
    NF_CC_STAMP(104);
    // The transformed half travels through LDS: waiting for the stores and reading them back from L2 was ~2 us of the launch's serial
    // chain.  The other half IS read back from memory, ~40 us later, by the workgroup that wrote it: the barriers' workgroup-scope
    // release / acquire is all that takes (one CU, one L1).  NOT __threadfence(): an agent-scope release writes the XCD's whole L2
    // back -- +20 us per launch, measured.  Nor does this barrier wait for the stores (0.6 .. 1.0 us): the LDS writes only.
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// The same for a head of 2 .. 4 channels (the first level of an image Glow: (3, 32, 32) under the checkerboard split): no matrix pipe,
// a thread per pixel with the arithmetic of the stand-alone kernel (k_glow_head_bwd<CT, 1>, csrc/glow_head.hip: fmaf over r ascending, one
// multiplication by 1 / exp(ls)) -- bit-identical.  hd_W is the weight the forward assembled from the PLU factors and saved.
template <int CT>
__device__ __forceinline__ void nf_cc_head_small_bwd(const nf_convnet_bwd_desc& d, const NfSplit& cs, float* hl, int64_t b0, int nsamp,
                                                     int fo_lo, int fo_hi, int64_t B, int PXW, int HWh, int sp0) {
    const int P = cs.H * cs.W, npx = (fo_hi - fo_lo) * cs.W;
    float* const gx = const_cast<float*>(d.cp_g_y);
    float* const G0 = hl + NF_CC_HD_X1;
    const int lgWf = 31 - __clz(cs.W);
    // (the first pixel's operands are requested before the constants: one round trip for both)
    float G[CT];
    int idx = threadIdx.x;
    auto place = [&](int i, int& sidx, int& p, int64_t& base) {
        sidx = i / npx;
        p = fo_lo * cs.W + (i - sidx * npx);
        base = (b0 + sidx) * cs.n_full + p;
    };
    int sidx = 0, p = 0;
    int64_t base = 0;
    bool ok = idx < nsamp * npx;
    if (ok) {
        place(idx, sidx, p, base);
        ok = b0 + sidx < B;
    }
#pragma unroll
    for (int r = 0; r < CT; ++r) G[r] = ok ? d.hd_g_h[base + (int64_t)r * P] : 0.f;
    float Wm[CT][CT], es[CT];
#pragma unroll
    for (int r = 0; r < CT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) Wm[r][c] = d.hd_W[r * CT + c];
#pragma unroll
    for (int c = 0; c < CT; ++c) es[c] = 1.f / expf(d.hd_ls[c]);
    while (idx < nsamp * npx) {                          // (block-uniform trip count up to the ragged end)
        if (ok) {
            const int row = p >> lgWf, xx = p & (cs.W - 1);
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                float a = 0.f;
#pragma unroll
                for (int r = 0; r < CT; ++r) a = fmaf(Wm[r][c], G[r], a);
                const float v = a * es[c];
                gx[base + (int64_t)c * P] = v;
                int which, m, sp;                       // (the forward prologue's map: squeeze.py:5-10, 32-44)
                if (cs.mode == NF_SPLIT_CHANNEL) {
                    const int hc = CT >> 1, sel = c >= hc ? 1 : 0;
                    which = sel ^ cs.odd; m = c - sel * hc; sp = p;
                } else {
                    const int k = 4 * c + 2 * (row & 1) + (xx & 1);
                    const int qd = (k >= CT ? 1 : 0) + (k >= 2 * CT ? 1 : 0) + (k >= 3 * CT ? 1 : 0);
                    const int sel = (qd == 1 || qd == 2) ? 1 : 0;
                    which = sel ^ cs.odd; m = sel ? k - CT : (qd == 0 ? k : k - 2 * CT);
                    sp = (row >> 1) * cs.w + (xx >> 1);
                }
                if (which == 0) G0[m * PXW + sidx * HWh + (sp - sp0)] = v;
            }
        }
        idx += NF_CV_THREADS;
        ok = idx < nsamp * npx;
        if (ok) {
            place(idx, sidx, p, base);
            ok = b0 + sidx < B;
        }
#pragma unroll
        for (int r = 0; r < CT; ++r) G[r] = ok ? d.hd_g_h[base + (int64_t)r * P] : 0.f;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (the LDS writes; see nf_cc_head_bwd)
}

// LDS: ONE frame F8 (three bf16 planes) | W8 (three planes; aliased by the K-split exchange and the gather buffer of the grid exchange) |
//      kc[4][32] | kb[32] | red[2][NPB][32] | tot[64]
template <int NPB, int NKQ, bool HALO, bool CPL, int FWc, int CSc, bool PK>
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
    const int CSr = FWc > 0 ? CSc : g.CS;               // (a literal in the per-level instantiations)
    const int WPs = L.WPs;
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
    // generation tags of this launch's slots: 8 ws_gen + (1 .. 7).  ws_gen = 0: the slots are fresh zeros; > 0: every launch of a train
    // step shares ONE slot buffer, zeroed where the step begins -- a launch never matches what an earlier one left (round 6: the fresh
    // buffers were 795 MB of memset per C4 step, 0.17 ms)
    const unsigned gbase = 8u * (unsigned)d.ws_gen;

    constexpr bool packed = PK;                         // compile-time: the weights arrive as LDS images (nf_conv_weight_pack)
    constexpr bool cpl = CPL;                           // the coupling rides the epilogue of the output convolution (d.cp_z != NULL)
    NF_CC_STAMP(0);
    // ---- the step's head (ActNorm + 1 x 1 convolution) rides the prologue: its operands are requested before anything else ----------
#ifdef NF_CC_NO_HEAD_FWD                                 // (probe builds: what the prologue's code costs the rest of the kernel in registers)
    const bool headed = false;
#else
    const bool headed = cpl && d.hd_x != nullptr;       // (block-uniform)
#endif
    // the full-resolution rows of this workgroup: whole samples, or (halo hand-over) its rows of sample b0 plus the rows of the
    // neighbours that its layer-0 frame reads (one half-map row each side: two full rows under the checkerboard split)
    const int hf = cs.mode == NF_SPLIT_CHECKER ? 2 : 1;
    const int h_ns = halo ? 1 : (g.HW < PXW ? PXW >> g.lgHW : 1);
    const int h_olo = halo ? hf * y0 : 0, h_ohi = halo ? hf * (y0 + g.TH) : cs.H;
    const int h_rlo = halo ? max(h_olo - hf, 0) : 0, h_rhi = halo ? min(h_ohi + hf, cs.H) : cs.H;
    const int h_per = ((h_rhi - h_rlo) * cs.W) >> 4, h_nblk = h_ns * h_per;
    const int h_sp0 = (h_rlo / hf) * g.W, h_np1 = ((h_rhi - h_rlo) / hf) * g.W;        // X1: half-map pixels sp0 .. sp0 + np1 - 1 per channel
    const bool hsmall = headed && cs.C <= 4;            // the thread-per-pixel form of a 2 .. 4 channel head
    NfCcHeadReq hreq;
    if (headed && !hsmall) {
        if (cs.C <= 16) nf_cc_head_request<4>(hreq, d, cs, b0, h_rlo, h_per, h_nblk, g.B);
        else if (cs.C <= 24) nf_cc_head_request<6>(hreq, d, cs, b0, h_rlo, h_per, h_nblk, g.B);
        else if (cs.C <= 48) nf_cc_head_request<12>(hreq, d, cs, b0, h_rlo, h_per, h_nblk, g.B);
        else nf_cc_head_request<16>(hreq, d, cs, b0, h_rlo, h_per, h_nblk, g.B);
    }
    NF_CC_STAMP(105);
    // ---- zero the frame (halo and padding stay zero for the whole launch) ---------------------------------------------------
    for (int e = threadIdx.x; e < 3 * NF_CC_FP(g.CS); e += NF_CV_THREADS) sm[L.FA + e] = 0.f;
    float* Fr = sm + L.FA;
    NF_CC_STAMP(106);
    if (hsmall) {
        const bool add_ld = !halo || y0 == 0;
        if (cs.C == 3) nf_cc_head_small_fwd<3>(d, cs, RS, b0, h_ns, h_rlo, h_rhi, h_olo, h_ohi, h_sp0, h_np1, g.B, add_ld);
        else if (cs.C == 4) nf_cc_head_small_fwd<4>(d, cs, RS, b0, h_ns, h_rlo, h_rhi, h_olo, h_ohi, h_sp0, h_np1, g.B, add_ld);
        else nf_cc_head_small_fwd<2>(d, cs, RS, b0, h_ns, h_rlo, h_rhi, h_olo, h_ohi, h_sp0, h_np1, g.B, add_ld);
    } else if (headed) {
        const bool add_ld = !halo || y0 == 0;
        if (cs.C <= 16) nf_cc_head_fwd<4>(hreq, d, cs, RS, b0, h_ns, h_rlo, h_olo, h_ohi, h_per, h_nblk, h_sp0, h_np1, g.B, add_ld);
        else if (cs.C <= 24) nf_cc_head_fwd<6>(hreq, d, cs, RS, b0, h_ns, h_rlo, h_olo, h_ohi, h_per, h_nblk, h_sp0, h_np1, g.B, add_ld);
        else if (cs.C <= 48) nf_cc_head_fwd<12>(hreq, d, cs, RS, b0, h_ns, h_rlo, h_olo, h_ohi, h_per, h_nblk, h_sp0, h_np1, g.B, add_ld);
        else nf_cc_head_fwd<16>(hreq, d, cs, RS, b0, h_ns, h_rlo, h_olo, h_ohi, h_per, h_nblk, h_sp0, h_np1, g.B, add_ld);
    }
    NF_CC_STAMP(107);
    if (cpl && !headed) {
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
            const int i0 = 32 * ch, IC = min(32, I0 - i0), ICP = (IC + 7) & ~7, noct = ICP >> 3;
            NfCcW wv;
            if (!packed) nf_cc_w_load<false>(wv, d.w[0], I0, i0, IC, noct);
            __syncthreads();                            // the previous chunk's readers of Wl / the frame are done (and the zero fill)
            if (packed) {
                nf_cc_dma_image(Wl, d.wpk[0] + (size_t)ch * NF_CONV_PACK_IMAGE_FLOATS, wid, lane);   // lands under the frame staging
            } else {
                nf_cc_w_store<false>(wv, Wl, WPs, noct);
                if ((9 * noct) & 1) nf_cc_w_zero_slot(Wl, WPs, 9 * noct);
            }
            // the chunk's input frame as ITEMS: the eight channels of octet o at frame position f -- eight loads in flight per item (one
            // memory round trip per chunk), a split, three 16-byte stores (a position outside the image / batch stores zeros)
            for (int item = threadIdx.x; item < noct * g.FSZ; item += NF_CV_THREADS) {
                const int o = item / g.FSZ, f = item - o * g.FSZ;
                const int t = nf_cv_decode(g, b0, y0, f);
                const int sp = t >= 0 ? NF_CV_SP(t) : 0, sg_ = t >= 0 ? NF_CV_SEG(t) : 0;
                float v[8];
                if (headed) {                           // (block-uniform) the conditioning half as the head left it in LDS
                    const float* X1 = RS + NF_CC_HD_X1 + (sg_ * I0 + i0 + 8 * o) * h_np1 + (sp - h_sp0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (t >= 0 && 8 * o + j < IC) ? X1[j * h_np1] : 0.f;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (t >= 0 && 8 * o + j < IC) ? in0[(sg_ * I0 + i0 + 8 * o + j) * g.HW + sp] : 0.f;
                }
                nf_cc_frame_store8(Fr, CSr, o, f, v);
            }
            if (packed) nf_cc_dma_wait();
            __syncthreads();
            if (FWc > 0 && ICP == 32) nf_cc_kloop_level<FWc, CSc, NKQ>(acc, Wl, Fr, fpos, c32, hs, kq);   // (block-uniform)
            else {
                int p0, np;
                nf_cc_pair_range((9 * noct + 1) >> 1, NKQ, kq, p0, np);
                nf_cc_kloop<9>(acc, Wl, Fr, WPs, CSr, g.FW, noct, fpos, c32, hs, p0, np);
            }
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
        if (!packed && PREFETCH_W && l < NF_CC_NB - 1) nf_cc_w_load<false>(wv, d.w[l + 1], 32, 0, 32, 4);
        if (threadIdx.x < 32) { kb[threadIdx.x] = nb_; kb[32 + threadIdx.x] = ng_; kb[64 + threadIdx.x] = nbe_; }
        __syncthreads();                                // every wave is done with Wl / the frame of this layer; kb is written
        // the next layer's weight image streams into Wl under this layer's exchanges (the 1 x 1's under the last layer's)
        if (packed && (l < NF_CC_NB - 1 || cpl)) nf_cc_dma_image(Wl, d.wpk[l + 1], wid, lane);   // (the 1 x 1 image has the coupling's row order)
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
        if (halo) nf_cc_halo_publish<OWN>(hslots, gbase, l, g, (int)tile, y0, px, kq, hs, own);
        NF_CC_STAMP(4 + 8 * l);
        if (training) {
            int k;
            float tS, tM2;
            const bool fin = nf_cc_stats_exchange<NPB>(sm, L, slots, gbase, l, Npx, PXW, k, tS, tM2);
            NF_CC_STAMP(5 + 8 * l);
            if (fin) {                                  // the lane that holds channel k's totals finishes its constants
                const float invN = 1.f / (float)Npx;
                const float mean = kb[k] + tS * invN;                         // statistics of the pre-bias output
                const float var = tM2 * invN;                                 // biased, as BatchNorm normalises
                const float invstd = 1.f / sqrtf(var + eps);
                const float sc = kb[32 + k] * invstd;
                kc[k] = sc;
                kc[32 + k] = kb[64 + k] - mean * sc;
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
                const float sc = kb[32 + k] * invstd;
                kc[k] = sc;
                kc[32 + k] = kb[64 + k] - mean * sc;
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
                const int f = nf_cc_halo_poll(hslots, gbase, l, g, (int)tile, y0, s2, v);
                if (f >= 0) nf_cc_frame_store1(Fr, CSr, c, f, fmaxf(fmaf(v, kc[c], kc[32 + c]), 0.f));
            }
        }
        // normalise + ReLU back into the frame, split into the three bf16 planes (a lane's values are whole channel quads: one 8-byte
        // store per plane); next weights into Wl
#pragma unroll
        for (int j = 0; j < OWN / 4; ++j) {
            const int c0 = nf_cv_cd_row(OWN * kq + 4 * j, hs);          // channels c0 .. c0 + 3
            float4 v;
            v.x = pv ? fmaxf(fmaf(own[4 * j + 0], kc[c0 + 0], kc[32 + c0 + 0]), 0.f) : 0.f;
            v.y = pv ? fmaxf(fmaf(own[4 * j + 1], kc[c0 + 1], kc[32 + c0 + 1]), 0.f) : 0.f;
            v.z = pv ? fmaxf(fmaf(own[4 * j + 2], kc[c0 + 2], kc[32 + c0 + 2]), 0.f) : 0.f;
            v.w = pv ? fmaxf(fmaf(own[4 * j + 3], kc[c0 + 3], kc[32 + c0 + 3]), 0.f) : 0.f;
            nf_cc_frame_store4(Fr, CSr, c0, fpos, v.x, v.y, v.z, v.w);
        }
        if constexpr (OWN == 1) {
            const int c0 = nf_cv_cd_row(kq, hs);
            nf_cc_frame_store1(Fr, CSr, c0, fpos, pv ? fmaxf(fmaf(own[0], kc[c0], kc[32 + c0]), 0.f) : 0.f);
        }
        if constexpr (OWN == 2) {
            const int c0 = nf_cv_cd_row(OWN * kq, hs);                  // channels c0, c0 + 1
            nf_cc_frame_store2(Fr, CSr, c0, fpos, pv ? fmaxf(fmaf(own[0], kc[c0], kc[32 + c0]), 0.f) : 0.f,
                               pv ? fmaxf(fmaf(own[1], kc[c0 + 1], kc[32 + c0 + 1]), 0.f) : 0.f);
        }
        if (l < NF_CC_NB - 1) {
            if (threadIdx.x < 32) { nb_ = d.b[l + 1][threadIdx.x]; ng_ = d.gamma[l + 1][threadIdx.x]; nbe_ = d.beta[l + 1][threadIdx.x]; }
            if (!packed) {
                if (!PREFETCH_W) nf_cc_w_load<false>(wv, d.w[l + 1], 32, 0, 32, 4);
                nf_cc_w_store<false>(wv, Wl, WPs, 4);
            } else {
                nf_cc_dma_wait();
            }
            __syncthreads();
            NF_CC_STAMP(7 + 8 * l);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            if (FWc > 0) nf_cc_kloop_level<FWc, CSc, NKQ>(acc, Wl, Fr, fpos, c32, hs, kq);        // geometry known at compile time
            else {
                int p0, np;
                nf_cc_pair_range(18, NKQ, kq, p0, np);
                // opaque trip count: fully unrolled, the K loop's LDS addresses become loop invariants of the LAYER loop that the
                // compiler keeps in (and spills from) VGPRs
                asm volatile("" : "+s"(np));
                nf_cc_kloop<9>(acc, Wl, Fr, WPs, CSr, g.FW, 4, fpos, c32, hs, p0, np);
            }
            NF_CC_STAMP(8 + 8 * l);
        }
    }
    NF_CC_STAMP(50);

    // ---- the 1 x 1 output convolution: wave (pb, kq) takes output blocks kq, kq + NKQ, ... ------------------------------------
    // With the coupling fused, the rows are staged interleaved -- row 2 p of block ob = shift channel m = 16 ob + p, row 2 p + 1 =
    // scale channel Ch + m -- so that a lane holds both parameters of the elements it transforms.
    // W8 of the 1 x 1: slot = 4 (output block) + (octet of the input channel), rows = the block's 32 output channels; rows beyond
    // O_out multiply with zeros
    const int Ch = O_out >> 1;
    const bool packed5 = packed && cpl;                 // the packed 1 x 1 image has the row order of the fused coupling
    if (packed5) nf_cc_dma_wait();
    for (int e = threadIdx.x; e < (packed5 ? 0 : 32 * OCB * 4); e += NF_CV_THREADS) {
        const int row = e >> 2, o = e & 3;              // staged row -> the output channel it holds; octet o of its 32 input channels
        int oc = row;
        if (cpl) {
            const int m = 16 * (row >> 5) + ((row & 31) >> 1);
            oc = (row & 1) ? Ch + m : m;
            if (m >= Ch) oc = O_out;
        }
        float v[8];
        const float4 lo = oc < O_out ? *(const float4*)(d.w[5] + oc * 32 + 8 * o) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 hi = oc < O_out ? *(const float4*)(d.w[5] + oc * 32 + 8 * o + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
        nf_cc_w_put8(Wl, WPs, 4 * (row >> 5) + o, row & 31, v);
    }
    __syncthreads();
    float ls = 0.f;                                     // this lane's part of its sample's sum of scales
    const float ca = cpl ? d.cp_a[0] : 0.f, cc = cpl ? d.cp_c[0] : 0.f;
    const int lgw = cpl ? 31 - __clz(cs.w) : 0;
    for (int ob = kq; ob < OCB; ob += NKQ) {            // wave-uniform
        f32x16 a5;
#pragma unroll
        for (int r = 0; r < 16; ++r) a5[r] = 0.f;
        nf_cc_kloop<1>(a5, Wl + 4 * ob * NF_CC_WSLOT, Fr, WPs, CSr, g.FW, 4, fpos, c32, hs, 0, 2);
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
                    const unsigned long long gen = (unsigned long long)(gbase + 7u) << 32;
                    if (part > 0) {
                        __hip_atomic_store(hs0 + part, gen | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        v = 0.f;
                    } else {
                        for (int p2 = 1; p2 < nparts; ++p2) {
                            unsigned long long w;
                            unsigned spins = 0;
                            do {
                                w = __hip_atomic_load(hs0 + p2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if ((unsigned)(w >> 32) == gbase + 7u) break;
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
// plain sums of two sets of OWN per-lane values over the 32 lanes of a wave half (halving butterfly, see nf_cc_half_stats)
template <int OWN>
__device__ __forceinline__ void nf_cc_half_sums2(const float (&u)[OWN], const float (&v)[OWN], int c32, float& S1, float& S2, int& which) {
    static_assert(OWN == 1 || OWN == 2 || OWN == 4 || OWN == 8, "16 / NKQ values per lane");
    if constexpr (OWN == 1) {
        S1 = u[0];
        S2 = v[0];
        for (int m = 1; m < 32; m *= 2) {
            S1 += __shfl_xor(S1, m, NF_WAVE);
            S2 += __shfl_xor(S2, m, NF_WAVE);
        }
        which = 0;
        return;
    }
    if constexpr (OWN == 2) {
        const bool up = c32 & 1;
        S1 = (up ? u[1] : u[0]) + __shfl_xor(up ? u[0] : u[1], 1, NF_WAVE);
        S2 = (up ? v[1] : v[0]) + __shfl_xor(up ? v[0] : v[1], 1, NF_WAVE);
        for (int m = 2; m < 32; m *= 2) {
            S1 += __shfl_xor(S1, m, NF_WAVE);
            S2 += __shfl_xor(S2, m, NF_WAVE);
        }
        which = up ? 1 : 0;
        return;
    }
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
__device__ __forceinline__ const float* nf_cc_sum_exchange(float* sm, const NfCcLds& L, unsigned long long* slots, unsigned gbase, int round) {
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
                               ((unsigned long long)(gbase + round + 1) << 32) | (unsigned long long)__float_as_uint(sv[0]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
    if (G == 1) {
        __syncthreads();
        return tot;
    }
    const int XS = 65;
    const unsigned gen = gbase + (unsigned)(round + 1);
    const unsigned long long* rs = slots + (size_t)round * NF_CC_MAX_BLOCKS * 64;
    const int lane = threadIdx.x & 63, ci = 2 * (threadIdx.x >> 6) + (lane >> 5), l = lane & 31;
    int nrows = G;
    if (NF_CC_MAX_BLOCKS > NF_CC_FLAT_MAX && G > NF_CC_FLAT_MAX) {     // two levels, as in the forward kernel's statistics exchange
        const int grp = blockIdx.x / NF_CC_GROUP;
        unsigned long long* gs = slots + NF_CC_WG_SLOTS + (size_t)round * NF_CC_MAX_GROUPS * 64;
        if (blockIdx.x % NF_CC_GROUP == 0) {
            const int nmem = min(NF_CC_GROUP, G - grp * NF_CC_GROUP);
            nf_cc_collect_slots(xs, rs + (size_t)grp * NF_CC_GROUP * 64, gen, nmem, XS);
            __syncthreads();
            float pa = 0.f, pb = 0.f;
            for (int b = l; b < nmem; b += 32) { pa += xs[b * XS + ci]; pb += xs[b * XS + 32 + ci]; }
            const float A = nf_cc_sum32(pa), Bs = nf_cc_sum32(pb);
            if (l == 0) {
                __hip_atomic_store(gs + grp * 64 + ci, ((unsigned long long)gen << 32) | (unsigned long long)__float_as_uint(A),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gs + grp * 64 + 32 + ci, ((unsigned long long)gen << 32) | (unsigned long long)__float_as_uint(Bs),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
        }
        rs = gs;
        nrows = (G + NF_CC_GROUP - 1) / NF_CC_GROUP;
    }
    nf_cc_collect_slots(xs, rs, gen, nrows, XS);
    __syncthreads();
    {
        float pa = 0.f, pb = 0.f;
        for (int b = l; b < nrows; b += 32) { pa += xs[b * XS + ci]; pb += xs[b * XS + 32 + ci]; }
        const float A = nf_cc_sum32(pa), Bs = nf_cc_sum32(pb);
        if (l == 0) { tot[ci] = A; tot[32 + ci] = Bs; }
    }
    __syncthreads();
    return tot;
}

template <int NPB, int NKQ, bool HALO, bool CPL, int FWc, int CSc, bool PK>
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
    const int CSr = FWc > 0 ? CSc : g.CS;               // (a literal in the per-level instantiations)
    const int WPs = L.WPs;
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
    const unsigned gbase = 8u * (unsigned)d.ws_gen;     // (generation tags of this launch: see the forward kernel)
    const int64_t b0 = (tile * PXW) >> g.lgHW;
    constexpr bool halo = HALO;                         // compile-time: the whole-sample variants carry none of its registers                       // a sample split over several workgroups (see the forward kernel)
    const int y0 = halo ? (int)((tile * PXW) & (g.HW - 1)) >> g.lgW : 0;
    unsigned long long* hslots = halo ? slots + NF_CC_STAT_SLOTS : nullptr;
    float gstream_h[2] = {0.f, 0.f};                    // halo rows of the residual stream's gradient (threads < 32 W)

    NF_CC_STAMP(64);
    // ---- the head between this step and the next: its data gradient IS this launch's cp_g_y (nf_cc_head_bwd, above) ----
    const bool hbw = CPL && d.hd_g_h != nullptr;        // (block-uniform)
    const int hb_hf = cs.mode == NF_SPLIT_CHECKER ? 2 : 1;
    const int hb_ns = halo ? 1 : (g.HW < PXW ? PXW >> g.lgHW : 1);
    const int hb_lo = halo ? hb_hf * y0 : 0, hb_hi = halo ? hb_hf * (y0 + g.TH) : cs.H;
    const int hb_per = ((hb_hi - hb_lo) * cs.W) >> 4, hb_nblk = hb_ns * hb_per;
    NfCcHeadBwdReq hbreq;
    const bool hbs = hbw && cs.C <= 4;                  // the thread-per-pixel form of a 2 .. 4 channel head
    if (hbw && !hbs) {
        if (cs.C <= 16) nf_cc_head_bwd_request<4>(hbreq, d, cs, b0, hb_lo, hb_per, hb_nblk, g.B);
        else if (cs.C <= 24) nf_cc_head_bwd_request<6>(hbreq, d, cs, b0, hb_lo, hb_per, hb_nblk, g.B);
        else if (cs.C <= 48) nf_cc_head_bwd_request<12>(hbreq, d, cs, b0, hb_lo, hb_per, hb_nblk, g.B);
        else nf_cc_head_bwd_request<16>(hbreq, d, cs, b0, hb_lo, hb_per, hb_nblk, g.B);
    }
    for (int e = threadIdx.x; e < 3 * NF_CC_FP(g.CS); e += NF_CV_THREADS) sm[L.FA + e] = 0.f;
    float* Fr = sm + L.FA;                              // ONE frame: G_l is written over G_{l+1} after every wave has left the K loop
    const bool bnv = L.BNV >= 0;                        // block-uniform
    constexpr bool packed = PK;                         // compile-time: transposed weight images (nf_conv_weight_pack)
    const int nch0 = (I0 + 31) / 32;                    // a 3 x 3 layer's buffer: its forward images, then the transposed ones
    if (packed) nf_cc_dma_image(Wl, d.wpk[5] + NF_CONV_PACK_IMAGE_FLOATS, wid, lane);
    if (bnv && threadIdx.x < NF_CC_NB * 128) {          // [layer][mean | invstd | gamma | beta][32]; first read after the barrier below
        const int l2 = threadIdx.x >> 7, j = (threadIdx.x >> 5) & 3, k = threadIdx.x & 31;
        const float* src = j == 0 ? d.save_mean[l2] : (j == 1 ? d.save_invstd[l2] : (j == 2 ? d.gamma[l2] : d.beta[l2]));
        sm[L.BNV + threadIdx.x] = src[k];
    }

    // ---- the 1 x 1 output convolution, transposed: acc[ic][pixel] = sum_oc W5[oc][ic] g_out[oc][pixel].  No halo: the B operand comes
    //      straight from global memory (a lane's own pixel; 4 x 128-byte segments per K group), K = oc split over the NKQ waves ----
    constexpr bool cpl = CPL;                           // (d.cp_g_y != NULL)
    NF_CC_STAMP(100);
    if (hbs) {
        const int hb_sp0 = halo ? y0 * g.W : 0;
        if (cs.C == 3) nf_cc_head_small_bwd<3>(d, cs, RS, b0, hb_ns, hb_lo, hb_hi, g.B, PXW, g.HW, hb_sp0);
        else if (cs.C == 4) nf_cc_head_small_bwd<4>(d, cs, RS, b0, hb_ns, hb_lo, hb_hi, g.B, PXW, g.HW, hb_sp0);
        else nf_cc_head_small_bwd<2>(d, cs, RS, b0, hb_ns, hb_lo, hb_hi, g.B, PXW, g.HW, hb_sp0);
    } else if (hbw) {                                   // (the weight image and the BatchNorm vectors requested above are in flight)
        const int hb_sp0 = halo ? y0 * g.W : 0;
        if (cs.C <= 16) nf_cc_head_bwd<4>(hbreq, d, cs, RS, b0, hb_lo, hb_per, hb_nblk, g.B, PXW, g.HW, hb_sp0);
        else if (cs.C <= 24) nf_cc_head_bwd<6>(hbreq, d, cs, RS, b0, hb_lo, hb_per, hb_nblk, g.B, PXW, g.HW, hb_sp0);
        else if (cs.C <= 48) nf_cc_head_bwd<12>(hbreq, d, cs, RS, b0, hb_lo, hb_per, hb_nblk, g.B, PXW, g.HW, hb_sp0);
        else nf_cc_head_bwd<16>(hbreq, d, cs, RS, b0, hb_lo, hb_per, hb_nblk, g.B, PXW, g.HW, hb_sp0);
    }
    NF_CC_STAMP(101);
    const int Ch = O_out >> 1;
    const int lgw = cpl ? 31 - __clz(cs.w) : 0;
    const float ca = cpl ? d.cp_a[0] : 0.f;
    float acc_a = 0.f, acc_c = 0.f;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    {
        const int npair = (O_out + 15) >> 4;            // pairs of octets of the K axis (= output channels of the 1 x 1)
        if (packed) nf_cc_dma_wait();
        for (int e = threadIdx.x; e < (packed ? 0 : npair * 2 * 32); e += NF_CV_THREADS) {
            const int o = e >> 5, ic = e & 31;          // W8T[plane][octet o of the output channel][ic][oc & 7]
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 8 * o + j < O_out ? d.w[5][(8 * o + j) * 32 + ic] : 0.f;
            nf_cc_w_put8(Wl, WPs, o, ic, v);
        }
        __syncthreads();
        int p0, np;
        nf_cc_pair_range(npair, NKQ, kq, p0, np);
        const float* go = cpl ? nullptr : d.g_out + b * O_out * g.HW + q;
        const float gl = (cpl && pv) ? d.cp_g_ld[b] : 0.f;
#pragma unroll 1
        for (int pi = p0; pi < p0 + np; ++pi) {
            float bv[8];                                // this lane's eight K values: output channels 16 pi + 8 hs + j of its pixel
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int oc = 16 * pi + 8 * hs + j;
                const bool ok = pv && oc < O_out;
                float v = 0.f;
                if (!cpl) {
                    v = ok ? go[(int64_t)oc * g.HW] : 0.f;
                } else if (ok) {                        // the coupling's backward produces the conditioner's output gradient on the fly
                    const bool is_s = oc >= Ch;
                    const int m = is_s ? oc - Ch : oc;
                    const int64_t o = b * cs.n_full + nf_cc_half_to_full(cs, 0, m, (int)q, lgw);
                    const float gy0 = hbw ? RS[NF_CC_HD_X1 + m * PXW + px] : d.cp_g_y[o];
                    v = gy0;                            // gradient of the shift
                    if (is_s) {                         // cp_out = [exp(s) | tanh(raw)], left by the forward launch
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
                bv[j] = v;
            }
            bf16x8 bh, bm, bl;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                bf16x2 h2, m2, l2;
                nf_cc_split2(f32x2{bv[j], bv[j + 1]}, h2, m2, l2);
                bh[j] = h2[0]; bh[j + 1] = h2[1];
                bm[j] = m2[0]; bm[j + 1] = m2[1];
                bl[j] = l2[0]; bl[j + 1] = l2[1];
            }
            const float* wa = Wl + (2 * pi + hs) * NF_CC_WSLOT + 4 * c32;
            const bf16x8 ah = *(const bf16x8*)(wa), am = *(const bf16x8*)(wa + WPs), al = *(const bf16x8*)(wa + 2 * WPs);
            NF_CC_MFMA6(ah, am, al, bh, bm, bl);
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
    NF_CC_STAMP(65);

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
        // the transposed weights of the convolution at the end of this iteration stream into Wl under the exchange
        if (packed) {
            if (l >= 1) nf_cc_dma_image(Wl, d.wpk[l] + NF_CONV_PACK_IMAGE_FLOATS, wid, lane);
            else if (d.g_x != nullptr || cpl) nf_cc_dma_image(Wl, d.wpk[0] + (size_t)nch0 * NF_CONV_PACK_IMAGE_FLOATS, wid, lane);
        }
        NF_CC_STAMP(66 + 6 * (4 - l));
        if (cpl && l == NF_CC_NB - 1 && threadIdx.x == 0) {
            float ta = 0.f, tc = 0.f;
            for (int k = 0; k < NF_CV_WAVES; ++k) { ta += red[k]; tc += red[NF_CV_WAVES + k]; }
            // (deterministic mode: the workgroups add in block order.  Safe inside the persistent loop: a workgroup with a smaller
            // index needs nothing from this one to get here -- every grid exchange before this point has been published by all)
            // Round 6: the LAST workgroup to arrive adds the workgroups' pairs in block order (nf_det_fold_add) -- nobody waits; the turnstile
            // (128 workgroups one after the other: 389 us per launch against 72) only for a grid beyond the fold's slab.
            if (nf_det_on(nf_ccd_det)) {
                const float v2[2] = {ta, tc};
                float* const dst2[2] = {d.cp_g_a, d.cp_g_c};
                if (!nf_det_fold_add<2>(nf_ccd_det_slab, nf_ccd_det_cnt, v2, dst2)) {
                    nf_det_wait(nf_ccd_det);
                    atomicAdd(d.cp_g_a, ta);
                    atomicAdd(d.cp_g_c, tc);
                    nf_det_pass(nf_ccd_det);
                }
            } else {
                atomicAdd(d.cp_g_a, ta);
                atomicAdd(d.cp_g_c, tc);
            }
        }
        nf_cc_ksplit_exchange<NKQ>(own, acc, RS, pb, kq, lane);
        NF_CC_STAMP(67 + 6 * (4 - l));
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
        if (halo) nf_cc_halo_publish<OWN>(hslots, gbase, l, g, (int)tile, y0, px, kq, hs, own);   // gn of our boundary rows, to the neighbours
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
            const float* tot = nf_cc_sum_exchange<NPB>(sm, L, slots, gbase, l);
            NF_CC_STAMP(68 + 6 * (4 - l));
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
                    const int f = nf_cc_halo_poll(hslots, gbase, l, g, (int)tile, y0, s2, v);
                    if (f >= 0) {
                        const float xhh = (a_h[s2] - kc[64 + c]) * kc[96 + c];
                        float G = kc[c] * (v - mgc - xhh * mgxc);
                        if ((l & 1) == 0) { G += gstream_h[s2]; gstream_h[s2] = G; }
                        nf_cc_frame_store1(Fr, CSr, c, f, G);
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
            nf_cc_frame_store4(Fr, CSr, c0, fpos, own[4 * j], own[4 * j + 1], own[4 * j + 2], own[4 * j + 3]);
        }
        if constexpr (OWN == 2) nf_cc_frame_store2(Fr, CSr, nf_cv_cd_row(OWN * kq, hs), fpos, own[0], own[1]);
        if constexpr (OWN == 1) nf_cc_frame_store1(Fr, CSr, nf_cv_cd_row(kq, hs), fpos, own[0]);
        NF_CC_STAMP(69 + 6 * (4 - l));
        if (l >= 1) {                                   // transposed 3 x 3 convolution l: G_l -> layer l - 1
            if (!packed) {
                NfCcW wv;
                nf_cc_w_load<true>(wv, d.w[l], 32, 0, 32, 4);
                nf_cc_w_store<true>(wv, Wl, WPs, 4);
            } else {
                nf_cc_dma_wait();
            }
            __syncthreads();
            NF_CC_STAMP(70 + 6 * (4 - l));
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            if (FWc > 0) nf_cc_kloop_level<FWc, CSc, NKQ>(acc, Wl, Fr, fpos, c32, hs, kq);        // geometry known at compile time
            else {
                int p0, np;
                nf_cc_pair_range(18, NKQ, kq, p0, np);
                asm volatile("" : "+s"(np));            // see the forward kernel
                nf_cc_kloop<9>(acc, Wl, Fr, WPs, CSr, g.FW, 4, fpos, c32, hs, p0, np);
            }
            NF_CC_STAMP(71 + 6 * (4 - l));
        }
    }
    NF_CC_STAMP(96);

    // ---- gradient of the conditioner's input: convolution 0 transposed, 32 input channels per pass ----
    if (d.g_x != nullptr || cpl) {
        for (int i0 = 0; i0 < I0; i0 += 32) {
            const int IC = min(32, I0 - i0);
            if (!packed) {
                NfCcW wv;
                nf_cc_w_load<true>(wv, d.w[0], I0, i0, IC, 4);
                __syncthreads();                        // readers of Wl / RS of the previous pass are done
                nf_cc_w_store<true>(wv, Wl, WPs, 4);
            } else {
                nf_cc_dma_wait();                       // (chunk 0 was requested in the l = 0 iteration, the others behind the previous K loop)
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            float gyv[OWN];                             // pass-through gradient of the untouched half: in flight under the K loop
            if (cpl) {
#pragma unroll
                for (int rr = 0; rr < OWN; ++rr) {
                    const int ic = nf_cv_cd_row(OWN * kq + rr, hs);
                    gyv[rr] = (pv && ic < IC) ? d.cp_g_y[b * cs.n_full + nf_cc_half_to_full(cs, 1, i0 + ic, (int)q, lgw)] : 0.f;
                }
            }
            if (FWc > 0) nf_cc_kloop_level<FWc, CSc, NKQ>(acc, Wl, Fr, fpos, c32, hs, kq);        // geometry known at compile time
            else {
                int p0, np;
                nf_cc_pair_range(18, NKQ, kq, p0, np);
                asm volatile("" : "+s"(np));
                nf_cc_kloop<9>(acc, Wl, Fr, WPs, CSr, g.FW, 4, fpos, c32, hs, p0, np);
            }
            __syncthreads();
            if (packed && i0 + 32 < I0) nf_cc_dma_image(Wl, d.wpk[0] + (size_t)(nch0 + (i0 >> 5) + 1) * NF_CONV_PACK_IMAGE_FLOATS, wid, lane);
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
    NF_CC_STAMP(97);
}

// =====================================================================================================================================
// nf_conv_weight_pack: effective weights -> LDS images of the split form (layouts in the header of this file / include/nfhip.h)
// =====================================================================================================================================
struct NfCcPackArgs { nf_conv_pack_desc d[NF_CONV_PACK_MAX_LAYERS]; };
__host__ __device__ static inline int nf_cc_pack_images(int O, int I, int ksize) {
    if (ksize == 3 && O == 32 && I >= 1 && I <= NF_CV_MAX_I) return 2 * ((I + 31) / 32);
    if (ksize == 1 && I == 32 && O >= 2 && O <= NF_CV_MAX_O && (O & 1) == 0) return 2;
    return 0;
}
// grid (images of the largest layer, layers), 256 threads: an image is 36 x 32 items of eight K values
__global__ void __launch_bounds__(256) k_conv_weight_pack(NfCcPackArgs a) {
    const nf_conv_pack_desc d = a.d[blockIdx.y];
    const int nimg = nf_cc_pack_images(d.O, d.I, d.ksize);
    const int img = blockIdx.x;
    if (img >= nimg) return;
    float* dst = d.dst + (size_t)img * NF_CONV_PACK_IMAGE_FLOATS;
    constexpr int WPs = NF_CC_WSLOTS * NF_CC_WSLOT;
    const bool tr = img >= nimg / 2;
    const int ch = tr ? img - nimg / 2 : img;
    for (int item = threadIdx.x; item < NF_CC_WSLOTS * 32; item += 256) {
        const int slot = item >> 5, row = item & 31;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        if (d.ksize == 3) {
            const int i0 = 32 * ch, IC = min(32, d.I - i0);
            if (!tr) {                                  // slot = tap * noct + o, row = oc, K = input channel i0 + 8 o + j
                const int noct = (IC + 7) >> 3;
                const int tap = slot / noct, o = slot - tap * noct;
                if (tap < 9)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (8 * o + j < IC) v[j] = d.w[((size_t)row * d.I + i0 + 8 * o + j) * 9 + tap];
            } else {                                    // slot = (8 - tap) * 4 + o, row = ic, K = output channel 8 o + j
                const int tap = 8 - (slot >> 2), o = slot & 3;
                if (row < IC)
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = d.w[((size_t)(8 * o + j) * d.I + i0 + row) * 9 + tap];
            }
        } else if (!tr) {                               // 1 x 1 forward, the fused coupling's row order: slot = 4 ob + o
            const int ob = slot >> 2, o = slot & 3, Ch = d.O >> 1;
            const int m = 16 * ob + (row >> 1), oc = (row & 1) ? Ch + m : m;
            if (m < Ch)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = d.w[(size_t)oc * 32 + 8 * o + j];
        } else {                                        // 1 x 1 transposed: slot = octet of the output channel, row = ic
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (8 * slot + j < d.O) v[j] = d.w[(size_t)(8 * slot + j) * 32 + row];
        }
        nf_cc_w_put8(dst, WPs, slot, row, v);
    }
}

extern "C" int nf_conv_weight_pack_images(int O, int I, int ksize) { return nf_cc_pack_images(O, I, ksize); }

extern "C" int nf_conv_weight_pack(const nf_conv_pack_desc* descs, int n, nf_stream_t stream) {
    if (descs == nullptr || n < 1 || n > NF_CONV_PACK_MAX_LAYERS) return NF_E_BADARG;
    NfCcPackArgs a;
    int most = 0;
    for (int i = 0; i < n; ++i) {
        const int k = nf_cc_pack_images(descs[i].O, descs[i].I, descs[i].ksize);
        if (k == 0 || descs[i].w == nullptr || descs[i].dst == nullptr) return NF_E_BADARG;
        if (k > most) most = k;
        a.d[i] = descs[i];
    }
    hipLaunchKernelGGL(k_conv_weight_pack, dim3((unsigned)most, (unsigned)n), dim3(256), 0, (hipStream_t)stream, a);
    NF_CHECK_LAUNCH();
    return 0;
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
static int nf_cc_halo_on() { return 1; }
// maps below 16 x 16 run on 64-pixel tiles (<2, 8>: two pixel blocks x eight K splits) where that doubles the workgroups within the
// co-residency limits -- the prologue / epilogue phases of a launch (input frame, 1 x 1 convolution, the coupling and its backward) are
// per-pixel work of 8 .. 32 compute units at the 4 x 4 and 8 x 8 levels
static int nf_cc_tile64_on() { return 1; }
static int nf_cc_tile32_on() { return 1; }
static inline int nf_cc_tile_px(int64_t B, int H, int W) {
    if (H * W <= 64 && nf_cc_tile64_on()) {
        const int64_t t64 = (B * H * W + 63) / 64, t32 = (B * H * W + 31) / 32;
        // 32-pixel tiles (<1, 16>) where 64-pixel ones leave fewer than 32 workgroups (the 4 x 4 level at B = 64: 16 -> 32)
        if (H * W <= 32 && t64 < 32 && t32 <= NF_CC_MAX_BLOCKS && t32 <= nf_cc_capacity() && nf_cc_tile32_on()) return 32;
        if (t64 <= NF_CC_MAX_BLOCKS && t64 <= nf_cc_capacity()) return 64;
    }
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
    if ((PX == 256 ? nf_cc_lds_bytes<8, 2>(g, OCB)
                   : (PX == 64 ? nf_cc_lds_bytes<2, 8>(g, OCB) : (PX == 32 ? nf_cc_lds_bytes<1, 16>(g, OCB) : nf_cc_lds_bytes<4, 4>(g, OCB)))) >
        160 * 1024)
        return 0;
    return 1;
}

// the split map of the coupling fused into a chain launch (the conditioner runs on the half's shape (Ch, H, W))
static bool nf_cc_coupling_split(NfSplit& cs, int mode, int odd, int C, int I0, int O_out, int H, int W) {
    if (mode != NF_SPLIT_CHANNEL && mode != NF_SPLIT_CHECKER) return false;
    const bool ck = mode == NF_SPLIT_CHECKER;
    if (!nf_make_split(cs, mode, odd, C, ck ? 2 * H : H, ck ? 2 * W : W)) return false;
    return cs.Ch == I0 && O_out == 2 * I0 && cs.h == H && cs.w == W;
}

extern "C" int nf_convnet_chain_blocks(int64_t B, int I0, int O_out, int H, int W) {
    if (!nf_convnet_chain_usable(B, I0, O_out, H, W)) return 0;
    const int PX = nf_cc_tile_px(B, H, W);
    return (int)((B * H * W + PX - 1) / PX);
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
    if (desc->hd_x != nullptr) {                        // the step's head in the prologue: needs the coupling, 9 .. 64 channels, 16-pixel blocks
        const bool small = desc->cp_C >= 2 && desc->cp_C <= 4;      // thread per pixel, the weight assembled from its PLU factors
        if (desc->cp_z == nullptr || desc->cp_inverse || desc->hd_ls == nullptr || desc->hd_bias == nullptr ||
            desc->hd_log_s == nullptr || desc->hd_x1 == nullptr || desc->hd_x1 != desc->x || ((cs.H * cs.W) & 15) != 0)
            return NF_E_BADARG;
        if (small ? (desc->hs_P == nullptr || desc->hs_L == nullptr || desc->hs_U == nullptr || desc->hs_Lm == nullptr ||
                     desc->hs_Um == nullptr || desc->hs_sign == nullptr)
                  : (desc->hd_W == nullptr || desc->cp_C < 9 || desc->cp_C > 64))
            return NF_E_BADARG;
    }
    NfCvGeo g;
    const int PX = nf_cc_tile_px(B, H, W);
    if (!nf_cv_geometry(g, B, H, W, 3, PX)) return NF_E_BADARG;
    if ((training || H * W > PX) && desc->ws_zero == nullptr) return NF_E_BADARG;     // exchange slots (statistics; halo rows)
    if (desc->hd_x != nullptr) {
        // the conditioning half of a tile (+ its halo rows) must fit the head's LDS buffer; the halo hand-over is two tiles per sample
        const int rows = H * W > PX ? PX / W + 2 : H, ns = H * W > PX ? 1 : PX / (H * W);
        if ((int64_t)ns * I0 * rows * W > NF_CC_HD_X1_MAX || (H * W > PX && cs.W < 16)) return NF_E_BADARG;
    }
    const int OCB = (O_out + 31) / 32;
    hipStream_t st = (hipStream_t)stream;
    int rc = 0;
    const bool cp = desc->cp_z != nullptr;
    bool pk = desc->wpk[0] != nullptr;                  // all six images or none
    for (int i = 1; i < NF_CC_NL; ++i)
        if ((desc->wpk[i] != nullptr) != pk) return NF_E_BADARG;
#define NF_CC_FWD(NPB_, NKQ_, HALO_, CP_, FW_, CS_, PK_)                                                                                    \
    do {                                                                                                                                    \
        rc = nf_cc_optin(k_convnet_chain_fwd<NPB_, NKQ_, HALO_, CP_, FW_, CS_, PK_>);                                                       \
        if (rc == 0)                                                                                                                        \
            hipLaunchKernelGGL((k_convnet_chain_fwd<NPB_, NKQ_, HALO_, CP_, FW_, CS_, PK_>), dim3((unsigned)g.tiles), dim3(NF_CV_THREADS),  \
                               (nf_cc_lds_bytes<NPB_, NKQ_>(g, OCB)), st, *desc, g, I0, O_out, training, bn_eps, bn_momentum, cs);          \
    } while (0)
#define NF_CC_FWD2(NPB_, NKQ_, HALO_, FW_, CS_)                                                                                             \
    do {                                                                                                                                    \
        if (cp && pk) NF_CC_FWD(NPB_, NKQ_, HALO_, true, FW_, CS_, true);                                                                   \
        else if (cp) NF_CC_FWD(NPB_, NKQ_, HALO_, true, FW_, CS_, false);                                                                   \
        else if (pk) NF_CC_FWD(NPB_, NKQ_, HALO_, false, FW_, CS_, true);                                                                   \
        else NF_CC_FWD(NPB_, NKQ_, HALO_, false, FW_, CS_, false);                                                                          \
    } while (0)
    // one instantiation per level of the CIFAR pyramid (frame width / channel stride as compile-time constants), a generic one for the rest
    if (PX == 256) NF_CC_FWD2(8, 2, false, 0, 0);
    else if (PX == 32) {
        if (g.FW == 6 && g.CS == 73) NF_CC_FWD2(1, 16, false, 6, 73);
        else NF_CC_FWD2(1, 16, false, 0, 0);
    } else if (PX == 64) {
        if (g.FW == 10 && g.CS == 101) NF_CC_FWD2(2, 8, false, 10, 101);
        else if (g.FW == 6 && g.CS == 145) NF_CC_FWD2(2, 8, false, 6, 145);
        else NF_CC_FWD2(2, 8, false, 0, 0);
    } else if (H * W > PX) {                              // a sample over several workgroups: the variant with the halo hand-over
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
    if (desc->hd_g_h != nullptr) {                      // the next step's head transposed in the prologue (same shapes as the forward's)
        const bool small = desc->cp_C >= 2 && desc->cp_C <= 4;      // (thread per pixel; hd_W = the weight the forward saved)
        if (desc->cp_g_y == nullptr || desc->hd_W == nullptr || desc->hd_ls == nullptr || (!small && (desc->cp_C < 9 || desc->cp_C > 64)) ||
            ((cs.H * cs.W) & 15) != 0 || (H * W > PX && cs.W < 16) || (int64_t)(O_out >> 1) * PX > NF_CC_HD_X1_MAX)
            return NF_E_BADARG;
    }
    hipStream_t st = (hipStream_t)stream;
    int rc = 0;
    const bool cp = desc->cp_g_y != nullptr;
    bool pk = desc->wpk[0] != nullptr;
    for (int i = 1; i < NF_CC_NL; ++i)
        if ((desc->wpk[i] != nullptr) != pk) return NF_E_BADARG;
#define NF_CC_BWD(NPB_, NKQ_, HALO_, CP_, FW_, CS_, PK_)                                                                                    \
    do {                                                                                                                                    \
        rc = nf_cc_optin(k_convnet_chain_bwd<NPB_, NKQ_, HALO_, CP_, FW_, CS_, PK_>);                                                       \
        if (rc == 0)                                                                                                                        \
            hipLaunchKernelGGL((k_convnet_chain_bwd<NPB_, NKQ_, HALO_, CP_, FW_, CS_, PK_>), dim3((unsigned)g.tiles), dim3(NF_CV_THREADS),  \
                               (nf_cc_lds_bytes<NPB_, NKQ_>(g, 1)), st, *desc, g, I0, O_out, training, cs);                                 \
    } while (0)
#define NF_CC_BWD2(NPB_, NKQ_, HALO_, FW_, CS_)                                                                                             \
    do {                                                                                                                                    \
        if (cp && pk) NF_CC_BWD(NPB_, NKQ_, HALO_, true, FW_, CS_, true);                                                                   \
        else if (cp) NF_CC_BWD(NPB_, NKQ_, HALO_, true, FW_, CS_, false);                                                                   \
        else if (pk) NF_CC_BWD(NPB_, NKQ_, HALO_, false, FW_, CS_, true);                                                                   \
        else NF_CC_BWD(NPB_, NKQ_, HALO_, false, FW_, CS_, false);                                                                          \
    } while (0)
    if (PX == 256) NF_CC_BWD2(8, 2, false, 0, 0);
    else if (PX == 32) {
        if (g.FW == 6 && g.CS == 73) NF_CC_BWD2(1, 16, false, 6, 73);
        else NF_CC_BWD2(1, 16, false, 0, 0);
    } else if (PX == 64) {
        if (g.FW == 10 && g.CS == 101) NF_CC_BWD2(2, 8, false, 10, 101);
        else if (g.FW == 6 && g.CS == 145) NF_CC_BWD2(2, 8, false, 6, 145);
        else NF_CC_BWD2(2, 8, false, 0, 0);
    } else if (H * W > PX) {
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

// ---- self-test of the three-way split (tests/test_gpu_ops.py::test_bf16_split_is_no_precision_reduction) ------------------------------
// D[32][32] = A[32][K] * B[K][32] on ONE wave, K a multiple of 16: mode 0 = v_mfma_f32_32x32x2_f32 (the exact fp32 instruction),
// mode 1 = the six-product form of the chain kernels -- the SAME nf_cc_split2 and the SAME NF_CC_MFMA6 accumulation order they use.
__global__ void __launch_bounds__(64) k_selftest_gemm32(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ D,
                                                        int K, int mode) {
    const int lane = threadIdx.x, c32 = lane & 31, hs = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (mode == 0) {
        for (int k = 0; k < K; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[c32 * K + k + hs], B[(k + hs) * 32 + c32], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 16) {
            bf16x8 ah, am, al, bh, bm, bl;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                bf16x2 h2, m2, l2;
                nf_cc_split2(f32x2{A[c32 * K + k + 8 * hs + j], A[c32 * K + k + 8 * hs + j + 1]}, h2, m2, l2);
                ah[j] = h2[0]; ah[j + 1] = h2[1]; am[j] = m2[0]; am[j + 1] = m2[1]; al[j] = l2[0]; al[j + 1] = l2[1];
                nf_cc_split2(f32x2{B[(k + 8 * hs + j) * 32 + c32], B[(k + 8 * hs + j + 1) * 32 + c32]}, h2, m2, l2);
                bh[j] = h2[0]; bh[j + 1] = h2[1]; bm[j] = m2[0]; bm[j + 1] = m2[1]; bl[j] = l2[0]; bl[j + 1] = l2[1];
            }
            NF_CC_MFMA6(ah, am, al, bh, bm, bl);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * hs) * 32 + c32] = acc[r];
}

extern "C" int nf_selftest_gemm32(const float* A, const float* B, float* D, int K, int mode, nf_stream_t stream) {
    if (!A || !B || !D || K <= 0 || (K & 15) || (mode != 0 && mode != 1)) return NF_E_BADARG;
    hipLaunchKernelGGL(k_selftest_gemm32, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, D, K, mode);
    NF_CHECK_LAUNCH();
    return 0;
}
