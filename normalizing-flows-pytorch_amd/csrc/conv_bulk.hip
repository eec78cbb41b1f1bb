// Large-batch form of the 3x3 convolution + BatchNorm2d + ReLU building block (flows/modules.py:416-438; contract of conv_bn.hip:
// "normalise on load, statistics on store", BatchNorm backward finished by the producer on load) for the launches that are THROUGHPUT-
// bound: batches beyond the persistent chain (conv_chain.hip: <= 128 co-resident workgroups), e.g. config 4's literal batch 512 on one
// GPU, where a 32 -> 32 channel layer at 16 x 16 is 2.4 GFLOP per launch.  The per-layer kernels of conv_bn.hip are built for latency
// (sixteen waves share one 128-pixel tile, barriers between its phases, v_mfma_f32_32x32x2_f32); they reach 22 % of the fp32 matrix
// peak there because nothing overlaps a tile's global loads with another tile's matrix work.  Here:
//   * WAVES ARE INDEPENDENT.  A 256-thread workgroup (one per compute unit, persistent) is four waves, one per SIMD.  After the
//     prologue (weight image + BatchNorm constants in LDS, ONE barrier) a wave owns UNITS of 32 NBLK consecutive pixels: its own
//     zero-padded activation frame in LDS (halo rows re-read from global memory / L2), its own accumulators, no barrier in the loop;
//   * the unit's inputs for the NEXT unit are requested (<= 56 values per lane and tensor, all in flight) before the K loop of the
//     current one, and converted / split / stored into the frame after it: one wave per SIMD has 512 registers to hold them;
//   * the products run on the bf16 matrix pipe as three-way splits (nf_bf16x3.h, DESIGN.md 3.21: fp32 accuracy, 0.375 of the fp32
//     instruction's pipe time); the weights are read from ONE shared image (nf_conv_weight_pack, or split here when none is given);
//     NBLK = 2 pixel blocks per wave share every weight read (9 LDS reads per 12 matrix instructions);
//   * statistics / BatchNorm-backward sums are kept per lane over all units of the wave and leave once per workgroup.
// Forward: k_conv3_bulk_fwd (I <= 32 -> 32 channels).  Backward, data gradient only (the weight gradient is nf_conv_bn_wgrad_multi's):
// k_conv3_bulk_bwd (32 -> I <= 32 channels, transposed image).  Same results as conv_bn.hip's kernels to fp32 rounding.
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <unordered_set>

#include "nf_common.h"
#include "nf_det.h"

NF_DET_STATE(nf_cbk)
NF_DET_HOST_API(nf_cbk)

#include "nf_conv_core.h"

#include "nf_bf16x3.h"

// phase stamps of wave 0 of workgroup 0 (tools/probes/bulk_prof.py builds this file with -DNF_CB_PROF=1; 100 MHz wall clock)
#ifdef NF_CB_PROF
__device__ long long nf_cb_prof[64];
#define NF_CB_STAMP(i)                                                                 \
    do {                                                                               \
        if (blockIdx.x == 0 && threadIdx.x == 0 && (i) < 64) nf_cb_prof[i] = wall_clock64(); \
    } while (0)
extern "C" int nf_cb_prof_read(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(nf_cb_prof), sizeof(long long) * 64); }
#define NF_CBW_STAMP(T, i)                                                             \
    do {                                                                               \
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == (T) && (i) >= 0 && (i) < 64) nf_cb_prof[i] = wall_clock64(); \
    } while (0)
#else
#define NF_CB_STAMP(i)
#define NF_CBW_STAMP(T, i)
#endif


// Deterministic mode, the per-layer launches (one-dimensional grids of <= 256 persistent workgroups): wave 0 of every workgroup holds the
// workgroup's 2 x 32 batch sums (lane half hs = which sum, c32 = channel) for replica blockIdx.x % NF_STAT_REPL.  A chain per replica is 32
// hand-overs of ~5 us behind workgroups that all finish together (200 us per launch against 30); instead the sums go to a slab and the
// workgroup that arrives LAST adds every replica's members in block order -- nobody waits (the vector form of nf_det_fold_add).
#define NF_CBK_FOLD_MAX 256
__device__ float nf_cbk_vslab[NF_CBK_FOLD_MAX * 64];
__device__ unsigned nf_cbk_vcnt[1];
__device__ __forceinline__ bool nf_cbk_det_fold64(float t, bool valid, float* __restrict__ dst0, float* __restrict__ dst1) {
    if (gridDim.y != 1 || gridDim.z != 1 || gridDim.x > NF_CBK_FOLD_MAX) return false;        // (block-uniform)
    const int lane = threadIdx.x & 63, hs = lane >> 5, c32 = lane & 31;
    const unsigned n = gridDim.x;
    __hip_atomic_store(nf_cbk_vslab + blockIdx.x * 64 + lane, valid ? t : 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    unsigned old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(nf_cbk_vcnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    old = __shfl(old, 0, NF_WAVE);
    if (old + 1 == n) {
        __threadfence();
        for (unsigned r = 0; r < NF_STAT_REPL && r < n; ++r) {
            float acc = 0.f;
            for (unsigned b = r; b < n; b += NF_STAT_REPL) acc += __hip_atomic_load(nf_cbk_vslab + b * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (valid) atomicAdd((hs == 0 ? dst0 : dst1) + 32 * r + c32, acc);
        }
        if (lane == 0) __hip_atomic_store(nf_cbk_vcnt, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    return true;
}

#define NF_CB_WAVES 4
#define NF_CB_THREADS (NF_CB_WAVES * NF_WAVE)
#define NF_CB_MAXIT 7                                 // items (channel octet, frame position) per lane: ceil(4 * FSZ / 64), FSZ <= 112
#define NF_CB_WP (NF_CC_WSLOTS * NF_CC_WSLOT)         // floats per plane of a weight image

struct NfCbGeo {
    NfCvGeo g;          // geometry of a UNIT (nf_cv_geometry with PX = 32 NBLK pixels)
    int64_t units;
    int noct;           // channel octets of the frame (forward: ceil(I / 8); backward: 4)
    int nit;            // items per lane
    int C;              // channels of the tensors the frame is built from (forward: I, backward: O = 32)
};

// switch: the environment (NF_CONV_BULK=0: off) at first use; nf_conv_bulk_config afterwards (tests: threshold and block count too)
static int nf_cb_cfg_on = -1, nf_cb_cfg_nblk = -1;
static int64_t nf_cb_cfg_min_px = -1;
static void nf_cb_cfg_init() {
    if (nf_cb_cfg_on >= 0) return;
    const char* e = getenv("NF_CONV_BULK");
    nf_cb_cfg_min_px = 16384 + 1;                      // beyond 128 tiles of 128 pixels: where the persistent chain ends
    nf_cb_cfg_nblk = 0;
    nf_cb_cfg_on = (e == nullptr || e[0] != '0') ? 1 : 0;
}
static int nf_cb_on() { nf_cb_cfg_init(); return nf_cb_cfg_on; }
static int64_t nf_cb_min_px() { nf_cb_cfg_init(); return nf_cb_cfg_min_px; }
static int nf_cb_force_nblk() { nf_cb_cfg_init(); return nf_cb_cfg_nblk; }
extern "C" int nf_conv_bulk_config(int on, int64_t min_pixels, int nblk) {
    nf_cb_cfg_init();
    if (on >= 0) nf_cb_cfg_on = on ? 1 : 0;
    if (min_pixels >= 0) nf_cb_cfg_min_px = min_pixels;
    if (nblk >= 0) nf_cb_cfg_nblk = nblk <= 2 ? nblk : 0;
    return 0;
}

// LDS (floats): W8[3 planes][36 slots][128] | frames[4 waves][3 planes][16 CS] | consts[9][32] | red[2][4][32]
static inline size_t nf_cb_lds_floats(int CS) { return (size_t)3 * NF_CB_WP + (size_t)NF_CB_WAVES * 3 * NF_CC_FP(CS) + 9 * 32 + 2 * 4 * 32; }

// NBLK for a shape, 0 = the bulk kernels do not take it
static int nf_cb_plan(NfCbGeo& cg, int64_t B, int C, int H, int W, int noct, int force_nblk = 0) {
    for (int nblk = 2; nblk >= 1; --nblk) {
        if (force_nblk && nblk != force_nblk) continue;
        const int UPX = 32 * nblk;
        if (!nf_cv_geometry(cg.g, B, H, W, 3, UPX)) continue;
        if (noct * cg.g.FSZ > NF_CB_MAXIT * NF_WAVE) continue;
        if (sizeof(float) * nf_cb_lds_floats(cg.g.CS) > 160 * 1024) continue;
        cg.units = cg.g.tiles;
        // two pixel blocks per wave while that still gives every wave of the machine >= 1.5 units (256 CUs x 4 waves)
        if (!force_nblk && nblk == 2 && cg.units < 1536) continue;
        cg.noct = noct;
        cg.nit = (noct * cg.g.FSZ + NF_WAVE - 1) / NF_WAVE;
        cg.C = C;
        return nblk;
    }
    return 0;
}

// ---- per-lane description of the frame items, invariant over the units of a launch -------------------------------------------------
// All global traffic of the unit loop goes through buffer descriptors (base = a wave-uniform pointer per unit, 2 GB window): a lane's
// part of an address is a 32-bit byte offset that does not depend on the unit, an entry that does not exist for a unit (halo row
// outside the sample, sample outside the batch, channel beyond the tensor) gets the offset 0xffffffff -- out of range: the load
// returns 0, the store is dropped -- so there is no branch and no 64-bit address arithmetic in the loop.
#define NF_CB_OOB 0xffffffffu
#define NF_CB_WINDOW 0x7fffffff
struct NfCbItems {
    unsigned voff[NF_CB_MAXIT];  // byte offset of (channel 8 o, frame position) from the unit's base - (W + 1) elements (channel stride HW)
    unsigned lds[NF_CB_MAXIT];   // float offset of the item's 16 bytes inside a frame plane
    unsigned meta;               // 4 bits per item: class (0 never valid, 1 inside, 2 top halo row, 3 bottom halo row) | 4: the item exists
    unsigned seg;                // 4 bits per item: sample of the unit (units of several whole samples)
    unsigned nch;                // 4 bits per item: valid channels of the octet - 1 (0 .. 7)
    unsigned oct;                // 4 bits per item: channel octet
};
__device__ __forceinline__ void nf_cb_items(NfCbItems& it, const NfCbGeo& cg, int lane) {
    const NfCvGeo& g = cg.g;
    it.meta = 0u; it.seg = 0u; it.nch = 0u; it.oct = 0u;
    const int total = cg.noct * g.FSZ;
#pragma unroll
    for (int k = 0; k < NF_CB_MAXIT; ++k) {
        const int id = lane + NF_WAVE * k;
        unsigned off = 0u, lds = 0u, cls = 0u, sj = 0u, nch = 0u, oc_ = 0u;
        if (k < cg.nit && id < total) {
            const int o = id / g.FSZ, f = id - o * g.FSZ;
            const int sm = (int)(((float)f + 0.5f) * g.invFS), q = f - sm * g.FS;            // (as nf_cv_decode)
            const int fy = (int)(((float)q + 0.5f) * g.invFW), fx = q - fy * g.FW;
            const int gx = fx - 1, fyh = fy - 1;
            lds = (unsigned)(o * 4 * g.CS + 4 * f);
            oc_ = (unsigned)o;
            const int left = cg.C - 8 * o;
            nch = (unsigned)((left > 8 ? 8 : left) - 1);
            if (gx >= 0 && gx < g.W) {
                if (g.SEG == 1) cls = fyh < 0 ? 2u : (fyh >= g.TH ? 3u : 1u);
                else cls = (fyh >= 0 && fyh < g.TH) ? 1u : 0u;
                off = 4u * (unsigned)((sm * cg.C + 8 * o) * g.HW + (fyh + 1) * g.W + gx + 1);
                sj = (unsigned)sm;
            }
            cls |= 4u;
        }
        it.voff[k] = off;
        it.lds[k] = lds;
        it.meta |= cls << (4 * k);
        it.seg |= sj << (4 * k);
        it.nch |= nch << (4 * k);
        it.oct |= oc_ << (4 * k);
    }
}

struct NfCbUnit {            // where a unit lies (wave-uniform)
    int64_t base;            // element offset of (sample b0, channel 0, row y0, column 0) in a (B, C, H, W) tensor
    bool top, bottom;        // halo rows inside the sample (units of whole rows)
    int nsamp;               // samples of the unit inside the batch (units of whole samples), else 1
};
__device__ __forceinline__ NfCbUnit nf_cb_unit(const NfCbGeo& cg, int64_t u, int C, int UPX) {
    const NfCvGeo& g = cg.g;
    NfCbUnit un;
    const int64_t P0 = u * UPX;
    const int64_t b0 = P0 >> g.lgHW;
    const int y0 = g.SEG == 1 ? (int)(P0 & (g.HW - 1)) >> g.lgW : 0;
    un.base = b0 * C * g.HW + (int64_t)y0 * g.W;
    un.top = g.SEG == 1 && y0 > 0;
    un.bottom = g.SEG == 1 && y0 + g.TH < g.H;
    const int64_t left = g.B - b0;
    un.nsamp = g.SEG == 1 ? 1 : (int)(left < g.SEG ? left : g.SEG);
    return un;
}
// the items' byte offsets for a unit (0xffffffff where the entry does not exist) and the mask of the existing ones
struct NfCbVo { unsigned v[NF_CB_MAXIT]; unsigned ok; };
__device__ __forceinline__ void nf_cb_offsets(NfCbVo& vo, const NfCbItems& it, const NfCbGeo& cg, const NfCbUnit& un) {
    vo.ok = 0u;
#pragma unroll
    for (int k = 0; k < NF_CB_MAXIT; ++k) {
        const unsigned cls = (it.meta >> (4 * k)) & 3u;
        const int s = (int)((it.seg >> (4 * k)) & 15u);
        const bool ok = (cls == 1u || (cls == 2u && un.top) || (cls == 3u && un.bottom)) && s < un.nsamp;
        vo.v[k] = ok ? it.voff[k] : NF_CB_OOB;
        vo.ok |= (ok ? 1u : 0u) << k;
    }
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t nf_cb_rsrc(const float* p) {
    // (the pointer is wave-uniform -- it derives from the wave's unit -- but not provably so: pin it to scalar registers)
    const uint64_t a = (uint64_t)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, NF_CB_WINDOW, 0x00020000);
}
__device__ __forceinline__ float nf_cb_ld(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, soff, 0));
}
__device__ __forceinline__ void nf_cb_st(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, soff, 0);
}

// raw values of one tensor for a unit: <= 56 unconditional loads per lane, all in flight together.  RAGGED: the last octet holds
// fewer than eight channels (the first layer of a conditioner): those loads are switched off like the missing frame entries.
struct NfCbRaw { float v[NF_CB_MAXIT][8]; };
template <bool RAGGED>
__device__ __forceinline__ void nf_cb_issue(NfCbRaw& r, const float* __restrict__ t, const NfCbVo& vo, const NfCbItems& it, const NfCbGeo& cg,
                                            const NfCbUnit& un) {
    const __amdgpu_buffer_rsrc_t rs = nf_cb_rsrc(t + un.base - (cg.g.W + 1));
    const int cstride = 4 * cg.g.HW;
#pragma unroll
    for (int k = 0; k < NF_CB_MAXIT; ++k)
        if (k < cg.nit) {                              // wave-uniform
            const int nch = (int)((it.nch >> (4 * k)) & 15u);
#pragma unroll
            for (int j = 0; j < 8; ++j) r.v[k][j] = nf_cb_ld(rs, (RAGGED && j > nch) ? NF_CB_OOB : vo.v[k], j * cstride);
        }
}

// the weight image of the layer -> LDS: direct global -> LDS loads of a packed image, or split here from the (32, I, 3, 3) weights
// (forward: slot tap * noct + o, row oc, K = input channel; transposed: slot (8 - tap) * 4 + o, row ic, K = output channel)
template <bool TR>
__device__ __forceinline__ void nf_cb_weights(float* W8, const float* __restrict__ wpk, const float* __restrict__ w, int I, int noct) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (wpk != nullptr) {
        constexpr int CH = 3 * NF_CB_WP / 256;
        for (int c = wid; c < CH; c += NF_CB_WAVES)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wpk + c * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void*)(W8 + c * 256), 16, 0, 0);
        return;                                        // (the caller waits: s_waitcnt vmcnt(0) in front of its barrier)
    }
    for (int item = threadIdx.x; item < NF_CC_WSLOTS * 32; item += NF_CB_THREADS) {
        const int slot = item >> 5, row = item & 31;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        if (!TR) {
            const int tap = slot / noct, o = slot - tap * noct;
            if (tap < 9)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (8 * o + j < I) v[j] = w[((size_t)row * I + 8 * o + j) * 9 + tap];
        } else {
            const int tap = 8 - (slot >> 2), o = slot & 3;
            if (row < I)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = w[((size_t)(8 * o + j) * I + row) * 9 + tap];
        }
        nf_cc_w_put8(W8, NF_CB_WP, slot, row, v);
    }
}

// eight values -> the three bf16 planes of a frame entry (16-byte stores)
__device__ __forceinline__ void nf_cb_put8(float* p, int FPs, const float (&v)[8]) {
    bf16x8 h, m, l;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        bf16x2 h2, m2, l2;
        nf_cc_split2(f32x2{v[j], v[j + 1]}, h2, m2, l2);
        h[j] = h2[0]; h[j + 1] = h2[1]; m[j] = m2[0]; m[j + 1] = m2[1]; l[j] = l2[0]; l[j + 1] = l2[1];
    }
    *(bf16x8*)(p) = h;
    *(bf16x8*)(p + FPs) = m;
    *(bf16x8*)(p + 2 * FPs) = l;
}

// ---- the K loop of a unit: acc[nb] += W8(slot)[row] * F8(tap offset, octet)[pixel of block nb] over the pairs of slots (2 p, 2 p + 1) ----
// NOCT octets of K channels per tap; the wave half hs takes slot 2 p + hs; a dead second slot of an odd last pair reads the image's
// zero slot.  The operands of pair p + 1 are requested BEFORE the matrix instructions of pair p (scheduling barriers keep the
// compiler from sinking the reads to their use: with one wave per SIMD nothing else hides an LDS round trip); the two pixel blocks
// share every weight read and alternate in the matrix pipe (independent accumulators back to back).
//   wa = W8 + hs * 128 + 4 * row (floats);  fb[nb] = F8 + 4 * fpos[nb] (+ hs * 4 * CS when NOCT is even)
template <int NOCT, int NBLK>
__device__ __forceinline__ void nf_cb_kloop(f32x16 (&accs)[NBLK], const float* wa, const float* const (&fb)[NBLK], int CS, int FW, int hs) {
    constexpr int NSLOT = 9 * NOCT, NP = (NSLOT + 1) / 2;
    const int FPs = NF_CC_FP(CS);
    bf16x8 a[3][3], b[3][NBLK][3];
    // operand reads of a pair in two halves: the weights + the first pixel block's planes, then the remaining blocks
#define NF_CB_FO(P)                                                                                            \
    const int s0 = 2 * (P), s1 = 2 * (P) + 1;                                                                  \
    const int t0 = s0 / NOCT, o0 = s0 - t0 * NOCT;                                                             \
    const int t1 = s1 < NSLOT ? s1 / NOCT : 4, o1 = s1 < NSLOT ? s1 - (s1 / NOCT) * NOCT : 0;                 \
    const int d0 = 4 * ((t0 / 3 - 1) * FW + (t0 % 3 - 1)) + o0 * 4 * CS;                                       \
    const int d1 = 4 * ((t1 / 3 - 1) * FW + (t1 % 3 - 1)) + o1 * 4 * CS;                                       \
    const int fo = (NOCT % 2 == 0) ? d0 : (hs ? d1 : d0);
#define NF_CB_LOAD_A(BUF, P)                                                                                   \
    do {                                                                                                       \
        NF_CB_FO(P)                                                                                            \
        _Pragma("unroll") for (int q = 0; q < 3; ++q) a[BUF][q] = *(const bf16x8*)(wa + (P) * 2 * NF_CC_WSLOT + q * NF_CB_WP); \
        _Pragma("unroll") for (int q = 0; q < 3; ++q) b[BUF][0][q] = *(const bf16x8*)(fb[0] + fo + q * FPs);   \
    } while (0)
#define NF_CB_LOAD_B(BUF, P)                                                                                   \
    do {                                                                                                       \
        NF_CB_FO(P)                                                                                            \
        _Pragma("unroll") for (int nb = 1; nb < NBLK; ++nb)                                                    \
            _Pragma("unroll") for (int q = 0; q < 3; ++q) b[BUF][nb][q] = *(const bf16x8*)(fb[nb] + fo + q * FPs); \
    } while (0)
#define NF_CB_STEP(BUF, AQ, BQ)                                                                                              \
    _Pragma("unroll") for (int nb = 0; nb < NBLK; ++nb)                                                                      \
        accs[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[BUF][AQ], b[BUF][nb][BQ], accs[nb], 0, 0, 0)
    // The operands of pair p + 2 are requested in the middle of the matrix instructions of pair p: a wave issues in order, so reads
    // placed behind a pair's last matrix instruction would only start when that instruction has issued, and the next pair would wait
    // a full LDS round trip for them; two pairs ahead, every read has a pair's worth of matrix time (>= 190 cycles) to come back.
    NF_CB_LOAD_A(0, 0); NF_CB_LOAD_B(0, 0);
    if (NP > 1) { NF_CB_LOAD_A(1, 1); NF_CB_LOAD_B(1, 1); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        NF_CB_STEP(p % 3, 0, 2); NF_CB_STEP(p % 3, 2, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (p + 2 < NP) NF_CB_LOAD_A((p + 2) % 3, p + 2);
        __builtin_amdgcn_sched_barrier(0);
        NF_CB_STEP(p % 3, 1, 1); NF_CB_STEP(p % 3, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (p + 2 < NP && NBLK > 1) NF_CB_LOAD_B((p + 2) % 3, p + 2);
        __builtin_amdgcn_sched_barrier(0);
        NF_CB_STEP(p % 3, 1, 0); NF_CB_STEP(p % 3, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
#undef NF_CB_FO
#undef NF_CB_LOAD_A
#undef NF_CB_LOAD_B
#undef NF_CB_STEP
}

// per-lane byte offsets of the pixels a lane finishes (rows of the 32 x 32 result = channels, columns = the block's pixels), from the
// unit's base in a (B, C, H, W) tensor: pixel p of the unit -> (sample p >> lgHW) * C * HW + (p & (HW - 1)), + the lane half's rows
__device__ __forceinline__ unsigned nf_cb_pix_off(const NfCvGeo& g, int C, int p, int hs) {
    const int s = g.SEG == 1 ? 0 : p >> g.lgHW, q = g.SEG == 1 ? p : p & (g.HW - 1);
    return 4u * (unsigned)(s * C * g.HW + q + 4 * hs * g.HW);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------------------
template <int NOCT, int NBLK, bool RAGGED, bool HAS_BN, bool HAS_RES>
__global__ void __launch_bounds__(NF_CB_THREADS) k_conv3_bulk_fwd(nf_conv_desc d, NfCbGeo cg, int I, int training, float eps, float mom) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const NfCvGeo& g = cg.g;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), c32 = lane & 31, hs = lane >> 5;
    float* W8 = smem;
    float* F8 = W8 + 3 * NF_CB_WP + wid * 3 * NF_CC_FP(g.CS);
    float* kc = smem + 3 * NF_CB_WP + NF_CB_WAVES * 3 * NF_CC_FP(g.CS);      // [2][32] scale, shift (+ 7 unused rows)
    float* red = kc + 9 * 32;
    constexpr bool has_bn = HAS_BN, has_res = HAS_RES;
    const int64_t Npx = g.B * g.HW;
    const int FPs = NF_CC_FP(g.CS);
    constexpr int UPX = 32 * NBLK;

    NF_CB_STAMP(0);
    nf_cb_weights<false>(W8, d.wpk, d.weight, I, NOCT);      // (a packed image: asynchronous global -> LDS loads, waited for below)
    NfCbItems it;
    nf_cb_items(it, cg, lane);
    const int64_t stride = (int64_t)gridDim.x * NF_CB_WAVES;
    int64_t u = (int64_t)blockIdx.x * NF_CB_WAVES + wid;
    NfCbRaw raw;
    NfCbUnit un;
    NfCbVo vo;
    unsigned okm = 0u;                                 // existing items of the unit whose raw values are in flight
    if (u < cg.units) {                                // the first unit's requests travel under the weight staging
        un = nf_cb_unit(cg, u, I, UPX);
        nf_cb_offsets(vo, it, cg, un);
        okm = vo.ok;
        nf_cb_issue<RAGGED>(raw, d.in, vo, it, cg, un);
    }
    NF_CB_STAMP(1);
    nf_cv_bn_consts_fwd(kc, d, I, Npx, training, eps, mom);
    const float* wa = W8 + hs * NF_CC_WSLOT + 4 * c32;
    const float* fb[NBLK];
    unsigned poff[NBLK];
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb) {
        fb[nb] = F8 + 4 * nf_cv_frame_of(g, nb * 32 + c32) + ((NOCT % 2 == 0) ? hs * 4 * g.CS : 0);
        poff[nb] = nf_cb_pix_off(g, 32, nb * 32 + c32, hs);
    }
    float bias_r[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bias_r[r] = d.bias[nf_cv_cd_row(r, hs)];
    float s1[16], s2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { s1[r] = 0.f; s2[r] = 0.f; }
    const int rstride = 4 * g.HW;                      // bytes between channel planes
    NF_CB_STAMP(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the weight image (direct global -> LDS loads) has landed
    __syncthreads();                                   // weights and constants are in LDS; the only barrier of the launch
    NF_CB_STAMP(3);
#ifdef NF_CB_PROF
    int pu = 0;
#endif
    for (; u < cg.units; u += stride) {
#ifdef NF_CB_PROF
        NF_CB_STAMP(4 + 8 * pu);
#endif
        // ---- the unit's frame: BatchNorm + ReLU, zeros outside the image, split, three 16-byte stores per item ----
#pragma unroll
        for (int k = 0; k < NF_CB_MAXIT; ++k)
            if (k < cg.nit) {                          // wave-uniform
                if ((it.meta >> (4 * k)) & 4u) {
                    const bool ok = (okm >> k) & 1u;
                    const int o = (int)((it.oct >> (4 * k)) & 15u);
                    const int nch = (int)((it.nch >> (4 * k)) & 15u);
                    float v[8];
                    if (has_bn) {                      // (compile time) the octet's constants: four 16-byte reads, one wait
                        const f32x4 sc0 = *(const f32x4*)(kc + 8 * o), sc1 = *(const f32x4*)(kc + 8 * o + 4);
                        const f32x4 sh0 = *(const f32x4*)(kc + 32 + 8 * o), sh1 = *(const f32x4*)(kc + 32 + 8 * o + 4);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float x = fmaxf(fmaf(raw.v[k][j], j < 4 ? sc0[j & 3] : sc1[j & 3], j < 4 ? sh0[j & 3] : sh1[j & 3]), 0.f);
                            v[j] = (ok && (!RAGGED || j <= nch)) ? x : 0.f;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = raw.v[k][j];      // (0 where the entry does not exist)
                    }
                    nf_cb_put8(F8 + it.lds[k], FPs, v);
                }
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#ifdef NF_CB_PROF
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        NF_CB_STAMP(5 + 8 * pu);
#endif
        // ---- requests of the next unit, and this unit's residual, in flight under the K loop ----
        const int64_t obase = nf_cb_unit(cg, u, 32, UPX).base;
        const int64_t P0 = u * UPX;
        const int64_t un_next = u + stride;
        if (un_next < cg.units) {
            un = nf_cb_unit(cg, un_next, I, UPX);
            nf_cb_offsets(vo, it, cg, un);
            okm = vo.ok;
            nf_cb_issue<RAGGED>(raw, d.in, vo, it, cg, un);
        }
        unsigned po[NBLK];
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) po[nb] = (P0 + nb * 32 + c32 < Npx) ? poff[nb] : NF_CB_OOB;
        float res[NBLK][16];
        if (has_res) {
            const __amdgpu_buffer_rsrc_t rr = nf_cb_rsrc(d.residual + obase);
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) res[nb][r] = nf_cb_ld(rr, po[nb], ((r & 3) + 8 * (r >> 2)) * rstride);
        } else {
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) res[nb][r] = 0.f;
        }
        f32x16 acc[NBLK];
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
#ifdef NF_CB_PROF
        NF_CB_STAMP(6 + 8 * pu);
#endif
        nf_cb_kloop<NOCT, NBLK>(acc, wa, fb, g.CS, g.FW, hs);
        __builtin_amdgcn_wave_barrier();               // (the frame is rewritten only after the last operand read was issued)
#ifdef NF_CB_PROF
        if (acc[0][0] == 123.456f) bias_r[0] += 1.f;   // (the stamp waits for the accumulators)
        NF_CB_STAMP(7 + 8 * pu);
#endif
        // ---- bias, residual, store (32 consecutive pixels of a channel plane per instruction), batch sums shifted by the bias ----
        // (pixels outside the batch: their frame is zero, their residual reads 0 -> they add nothing to the sums; their store is dropped)
        const __amdgpu_buffer_rsrc_t ro = nf_cb_rsrc(d.out + obase);
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float dv = acc[nb][r] + res[nb][r];
                nf_cb_st(ro, po[nb], ((r & 3) + 8 * (r >> 2)) * rstride, dv + bias_r[r]);
                s1[r] += dv;
                s2[r] = fmaf(dv, dv, s2[r]);
            }
#ifdef NF_CB_PROF
        NF_CB_STAMP(8 + 8 * pu);
        ++pu;
#endif
    }
    NF_CB_STAMP(60);
    if (d.stat_sum != nullptr) {                       // block-uniform
        const float t1 = nf_cv_butterfly16(s1, c32), t2 = nf_cv_butterfly16(s2, c32);
        if ((c32 & 1) == 0) {
            const int oc = nf_cv_cd_row(c32 >> 1, hs);
            red[(0 * 4 + wid) * 32 + oc] = t1;
            red[(1 * 4 + wid) * 32 + oc] = t2;
        }
        __syncthreads();
        if (threadIdx.x < 64) {                        // wave 0 adds for the workgroup; half 0: sums, half 1: squares
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) t += red[(hs * 4 + w) * 32 + c32];
            if (!(nf_det_on(nf_cbk_det) && nf_cbk_det_fold64(t, true, d.stat_sum, d.stat_sqsum))) {
                NF_DET_REPL_CHAIN(NF_STAT_REPL);       // (deterministic mode beyond the fold's slab: a chain per replica)
                NF_DET_ENTER_WAVE_K(nf_cbk);
                const int rep = 32 * (blockIdx.x % NF_STAT_REPL);
                atomicAdd((hs == 0 ? d.stat_sum : d.stat_sqsum) + rep + c32, t);
                NF_DET_LEAVE_WAVE_K(nf_cbk);
            }
        }
    }
    NF_CB_STAMP(61);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// backward, data gradient: G = g_skip + BNbwd(gn_src; out) assembled into the frame (g_store = G at the pixels the unit owns),
// gn_out = (transposed convolution of G) * [act > 0] with its two batch sums, or the plain input gradient without an input BatchNorm
// ---------------------------------------------------------------------------------------------------------------------------------
template <int NBLK, bool SKIP, bool HAS_BN>
__global__ void __launch_bounds__(NF_CB_THREADS) k_conv3_bulk_bwd(nf_conv_bwd_desc d, NfCbGeo cg, int I) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const NfCvGeo& g = cg.g;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), c32 = lane & 31, hs = lane >> 5;
    float* W8 = smem;
    float* F8 = W8 + 3 * NF_CB_WP + wid * 3 * NF_CC_FP(g.CS);
    float* cb = smem + 3 * NF_CB_WP + NF_CB_WAVES * 3 * NF_CC_FP(g.CS);      // [5][32] consumer BatchNorm: c1, mean, invstd, mean g, mean g xhat
    float* kc = cb + 5 * 32;                                                   // [4][32] input BatchNorm: scale, shift, mean, invstd
    float* red = kc + 4 * 32;
    constexpr bool has_bn = HAS_BN;
    const int64_t Npx = g.B * g.HW;
    const float invN = 1.f / (float)Npx;
    const int FPs = NF_CC_FP(g.CS);
    constexpr int UPX = 32 * NBLK;

    nf_cb_weights<true>(W8, d.wpk, d.weight, I, 4);
    NfCbItems it;
    nf_cb_items(it, cg, lane);
    const int64_t stride = (int64_t)gridDim.x * NF_CB_WAVES;
    int64_t u = (int64_t)blockIdx.x * NF_CB_WAVES + wid;
    NfCbRaw rs, ro, rk;                                // gn_src, out, g_skip of the unit whose frame is built next
    NfCbUnit un;
    NfCbVo vo;
    unsigned okm = 0u;
    if (u < cg.units) {
        un = nf_cb_unit(cg, u, 32, UPX);
        nf_cb_offsets(vo, it, cg, un);
        okm = vo.ok;
        nf_cb_issue<false>(rs, d.gn_src, vo, it, cg, un);
        nf_cb_issue<false>(ro, d.out, vo, it, cg, un);
        if (SKIP) nf_cb_issue<false>(rk, d.g_skip, vo, it, cg, un);
    }
    if (threadIdx.x < 32) {
        const int oo = threadIdx.x;
        const float invstd = d.cbn_save_invstd[oo], mean = d.cbn_save_mean[oo];
        float mg = 0.f, mgx = 0.f;
        if (d.cbn_sum_g != nullptr) {
#pragma unroll
            for (int r = 0; r < NF_STAT_REPL; ++r) { mg += d.cbn_sum_g[32 * r + oo]; mgx += d.cbn_sum_gx[32 * r + oo]; }
            mg *= invN;
            mgx *= invN;
        }
        cb[oo] = d.cbn_gamma[oo] * invstd; cb[32 + oo] = mean; cb[64 + oo] = invstd; cb[96 + oo] = mg; cb[128 + oo] = mgx;
    } else if (threadIdx.x < 64) {
        const int k = threadIdx.x - 32;
        float sc = 1.f, sh = 0.f, mean = 0.f, invstd = 0.f;
        if (has_bn && k < I) {
            mean = d.bn_save_mean[k];
            invstd = d.bn_save_invstd[k];
            sc = d.bn_gamma[k] * invstd;
            sh = d.bn_beta[k] - mean * sc;
        }
        kc[k] = sc; kc[32 + k] = sh; kc[64 + k] = mean; kc[96 + k] = invstd;
    }
    const float* wa = W8 + hs * NF_CC_WSLOT + 4 * c32;
    const float* fb[NBLK];
    unsigned poff[NBLK];
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb) {
        fb[nb] = F8 + 4 * nf_cv_frame_of(g, nb * 32 + c32) + hs * 4 * g.CS;
        poff[nb] = nf_cb_pix_off(g, I, nb * 32 + c32, hs);
    }
    float sg[16], sgx[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { sg[r] = 0.f; sgx[r] = 0.f; }
    const int rstride = 4 * g.HW;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (; u < cg.units; u += stride) {
        const int64_t gbase = nf_cb_unit(cg, u, 32, UPX).base;               // this unit in the 32-channel tensors
        const __amdgpu_buffer_rsrc_t rg = nf_cb_rsrc(d.g_store != nullptr ? d.g_store + gbase - (g.W + 1) : d.out);
#pragma unroll
        for (int k = 0; k < NF_CB_MAXIT; ++k)
            if (k < cg.nit) {
                const unsigned meta = (it.meta >> (4 * k)) & 15u;
                if (meta & 4u) {
                    const bool ok = (okm >> k) & 1u;
                    const int o = (int)((it.oct >> (4 * k)) & 15u);
                    float v[8];
                    f32x4 q[5][2];                     // the octet's five constants: ten 16-byte reads, one wait
#pragma unroll
                    for (int a = 0; a < 5; ++a) { q[a][0] = *(const f32x4*)(cb + 32 * a + 8 * o); q[a][1] = *(const f32x4*)(cb + 32 * a + 8 * o + 4); }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float xh = (ro.v[k][j] - q[1][j >> 2][j & 3]) * q[2][j >> 2][j & 3];
                        float G = q[0][j >> 2][j & 3] * (rs.v[k][j] - q[3][j >> 2][j & 3] - xh * q[4][j >> 2][j & 3]);
                        if (SKIP) G = rk.v[k][j] + G;
                        v[j] = ok ? G : 0.f;
                    }
                    if (d.g_store != nullptr) {        // G at the pixels the unit owns (class 1: inside, never a halo row)
                        const unsigned so = (ok && (meta & 3u) == 1u) ? it.voff[k] : NF_CB_OOB;
#pragma unroll
                        for (int j = 0; j < 8; ++j) nf_cb_st(rg, so, j * rstride, v[j]);
                    }
                    nf_cb_put8(F8 + it.lds[k], FPs, v);
                }
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int64_t ibase = nf_cb_unit(cg, u, I, UPX).base;
        const int64_t P0 = u * UPX;
        const int64_t un_next = u + stride;
        if (un_next < cg.units) {
            un = nf_cb_unit(cg, un_next, 32, UPX);
            nf_cb_offsets(vo, it, cg, un);
            okm = vo.ok;
            nf_cb_issue<false>(rs, d.gn_src, vo, it, cg, un);
            nf_cb_issue<false>(ro, d.out, vo, it, cg, un);
            if (SKIP) nf_cb_issue<false>(rk, d.g_skip, vo, it, cg, un);
        }
        // forward input of the pixels this lane finishes (rows of the result = input channels)
        unsigned po[NBLK];
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) po[nb] = (P0 + nb * 32 + c32 < Npx) ? poff[nb] : NF_CB_OOB;
        float xin[NBLK][16];
        if (has_bn) {
            const __amdgpu_buffer_rsrc_t ri = nf_cb_rsrc(d.in + ibase);
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ic = nf_cv_cd_row(r, hs);
                    xin[nb][r] = nf_cb_ld(ri, ic < I ? po[nb] : NF_CB_OOB, ((r & 3) + 8 * (r >> 2)) * rstride);
                }
        } else {
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) xin[nb][r] = 0.f;
        }
        f32x16 acc[NBLK];
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
        nf_cb_kloop<4, NBLK>(acc, wa, fb, g.CS, g.FW, hs);
        __builtin_amdgcn_wave_barrier();
        if (d.gn_out != nullptr) {                     // block-uniform
            const __amdgpu_buffer_rsrc_t rn = nf_cb_rsrc(d.gn_out + ibase);
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ic = nf_cv_cd_row(r, hs);
                    float gn = acc[nb][r];
                    if (has_bn) {
                        const float x = xin[nb][r];
                        gn = fmaf(x, kc[ic], kc[32 + ic]) > 0.f ? gn : 0.f;
                        gn = po[nb] != NF_CB_OOB ? gn : 0.f;       // (a pixel outside the batch: x read 0, but shift alone may be > 0)
                        sg[r] += gn;
                        sgx[r] = fmaf(gn, (x - kc[64 + ic]) * kc[96 + ic], sgx[r]);
                    }
                    nf_cb_st(rn, ic < I ? po[nb] : NF_CB_OOB, ((r & 3) + 8 * (r >> 2)) * rstride, gn);
                }
        }
    }
    if (has_bn && d.sum_g != nullptr) {                // block-uniform
        const float t1 = nf_cv_butterfly16(sg, c32), t2 = nf_cv_butterfly16(sgx, c32);
        if ((c32 & 1) == 0) {
            const int ic = nf_cv_cd_row(c32 >> 1, hs);
            red[(0 * 4 + wid) * 32 + ic] = t1;
            red[(1 * 4 + wid) * 32 + ic] = t2;
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) t += red[(hs * 4 + w) * 32 + c32];
            if (!(nf_det_on(nf_cbk_det) && nf_cbk_det_fold64(t, c32 < I, d.sum_g, d.sum_gx))) {
                NF_DET_REPL_CHAIN(NF_STAT_REPL);       // (deterministic mode beyond the fold's slab: a chain per replica)
                NF_DET_ENTER_WAVE_K(nf_cbk);
                if (c32 < I) {
                    const int rep = 32 * (blockIdx.x % NF_STAT_REPL);
                    atomicAdd((hs == 0 ? d.sum_g : d.sum_gx) + rep + c32, t);
                }
                NF_DET_LEAVE_WAVE_K(nf_cbk);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// weight gradient: g_weff[tap][oc][ic] = sum_px G[oc][px] act[ic][px + tap], G and act as in the data pass.  The contraction runs
// over PIXELS, so both operands of v_mfma_f32_32x32x16_bf16 are eight consecutive pixels of one channel per lane: frames in LDS are
// bf16 planes [channel][row][pixel] (three planes per tensor: the three-way split).  A tap's dy is a row offset of the activation
// operand; its dx = +-1 would break the 16-byte groups, so the operand is read ONCE per (row, plane) with the dword on either side
// and the two shifted variants are formed in registers (v_alignbit_b32: 8 VALU per plane and row against 18 matrix instructions).
// A workgroup is eight waves with two jobs (one of each per SIMD, so the hardware overlaps them without any software pipelining):
//   waves 0..3 WALK: wave w owns K steps 2 w, 2 w + 1 (16 pixels each) of every 128-pixel tile for all nine taps -- nine 32 x 32
//     accumulators (144 registers), the G operand read once per K step and used by nine taps x six products;
//   waves 4..7 FILL: the next tile's loads (16-byte, coalesced: a lane takes four pixels of a channel), BatchNorm backward on G /
//     BatchNorm + ReLU on the activations, the split, 8-byte LDS stores into the other frame pair; the tile after that is requested
//     before the barrier, so its loads have a whole tile's matrix time to land.
// ONE barrier per tile.  The workgroup walks tiles blockIdx.x + k gridDim.x of layer blockIdx.y and leaves one slab (T, O, I) and
// its bias sums, exactly as k_conv_bn_wgrad_multi does (same slab count: nf_conv_wgrad_slabs), so nf_slab_sum is unchanged.
// ---------------------------------------------------------------------------------------------------------------------------------
#define NF_CBW_THREADS 512
#define NF_CBW_FILL 256                                // filling threads
#define NF_CBW_GCH 272                                 // bytes per channel of a G plane: 128 pixels x 2 B + 16 (17 x 16: b128 reads conflict-free)
#define NF_CBW_MAXR 5                                  // activation items (channel, four pixels) per filling thread: ceil(32 NQ / 256), NQ <= 40 (W = 8, 16)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
struct NfCbwGeo {
    int H, W, HW, lgW, lgHW, lgSP;
    int TH, SEG;                                       // rows per segment, segments (samples) per 128-pixel tile
    int RSB;                                           // bytes per activation frame row: 2 (W + 4) -- four zero pixels behind every row
    int ACH;                                           // bytes per channel of an activation plane: 8 + SEG (TH + 2) RSB (+ 8 if needed: 8 x odd)
    int NQ;                                            // (row, four-pixel) items per channel and tile: SEG (TH + 2) W / 4
    int nr;                                            // activation items per filling thread
    int64_t B, tiles;
};
struct NfCbwMulti { nf_conv_bwd_desc d[NF_CONV_WGRAD_MAX]; };
static inline size_t nf_cbw_buf_bytes(const NfCbwGeo& g) { return (size_t)3 * 32 * (g.ACH + NF_CBW_GCH); }
static inline size_t nf_cbw_lds_bytes(const NfCbwGeo& g) { return 2 * nf_cbw_buf_bytes(g) + 7 * 32 * sizeof(float); }
static bool nf_cbw_geometry(NfCbwGeo& g, int64_t B, int H, int W) {
    if (B < 1 || H < 1 || W < 8 || W > 64) return false;
    g.H = H; g.W = W; g.HW = H * W; g.B = B;
    g.lgW = nf_cv_log2(W); g.lgHW = nf_cv_log2(g.HW);
    if (g.lgW < 0 || g.lgHW < 0) return false;
    if (g.HW >= 128) { g.TH = 128 / W; g.SEG = 1; }
    else { if (g.HW < 16) return false; g.TH = H; g.SEG = 128 / g.HW; }
    g.lgSP = nf_cv_log2(g.TH * W);
    g.RSB = 2 * (W + 4);
    g.ACH = 8 + g.SEG * (g.TH + 2) * g.RSB;
    if (((g.ACH >> 3) & 1) == 0) g.ACH += 8;
    g.NQ = g.SEG * (g.TH + 2) * (W / 4);
    g.nr = (32 * g.NQ + NF_CBW_FILL - 1) / NF_CBW_FILL;
    if (g.nr > NF_CBW_MAXR) return false;
    g.tiles = (B * g.HW + 127) / 128;
    return nf_cbw_lds_bytes(g) <= 160 * 1024 && 2 * nf_cbw_buf_bytes(g) >= 9 * 1024 * sizeof(float);
}
// 16 bytes through a buffer descriptor (wave-uniform base in scalar registers, per-lane 32-bit byte offset, entries that do not exist
// read 0 at the offset 0xffffffff: no 64-bit address arithmetic, no clamping).  The vector is converted WHOLE: __builtin_bit_cast(float,
// v[j]) on an element of an integer vector reads element 0 with this toolchain -- one dword whose value fills all four pixels.
__device__ __forceinline__ f32x4 nf_cbw_ld128(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0));
}
// four consecutive pixels of one channel -> 8 bytes in each of the three planes
__device__ __forceinline__ void nf_cbw_put4(char* p, int plane_bytes, const float (&v)[4]) {
    bf16x2 h0, m0, l0, h1, m1, l1;
    nf_cc_split2(f32x2{v[0], v[1]}, h0, m0, l0);
    nf_cc_split2(f32x2{v[2], v[3]}, h1, m1, l1);
    *(bf16x4*)(p) = bf16x4{h0[0], h0[1], h1[0], h1[1]};
    *(bf16x4*)(p + plane_bytes) = bf16x4{m0[0], m0[1], m1[0], m1[1]};
    *(bf16x4*)(p + 2 * plane_bytes) = bf16x4{l0[0], l0[1], l1[0], l1[1]};
}

__global__ void __launch_bounds__(NF_CBW_THREADS) k_conv3_bulk_wgrad(NfCbwMulti m, const nf_conv_bwd_desc* __restrict__ tab, NfCbwGeo g, int I) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const nf_conv_bwd_desc& d = tab != nullptr ? tab[blockIdx.y] : m.d[blockIdx.y];     // (tab: the descriptor table of nf_conv_bn_wgrad_table)
    char* const lds = (char*)smem;
    const int PA = 32 * g.ACH, PG = 32 * NF_CBW_GCH;  // bytes per activation / G plane
    const int BUF = 3 * (PA + PG);                     // a frame pair: act planes h | m | l, G planes h | m | l
    float* cst = (float*)(lds + 2 * BUF);              // [5][32] consumer BatchNorm: c1, mean, invstd, mean g, mean g xhat | [2][32] input BatchNorm: scale, shift
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), c32 = lane & 31, hs = lane >> 5;
    const bool walker = wid < 4;
    const bool has_bn = d.bn_gamma != nullptr, has_src = d.gn_src != nullptr;
    const int64_t Npx = g.B * g.HW;

    // ---- prologue: zero the frames (the padding is never written again), constants ----
    for (int e = tid; e < 2 * BUF / 16; e += NF_CBW_THREADS) *(u32x4*)(lds + 16 * e) = u32x4{0u, 0u, 0u, 0u};
    if (tid < 32) {
        const int oo = tid;
        float c1 = 0.f, mean = 0.f, invstd = 0.f, mg = 0.f, mgx = 0.f;
        if (has_src) {
            invstd = d.cbn_save_invstd[oo]; mean = d.cbn_save_mean[oo]; c1 = d.cbn_gamma[oo] * invstd;
            if (d.cbn_sum_g != nullptr) {
#pragma unroll
                for (int r = 0; r < NF_STAT_REPL; ++r) { mg += d.cbn_sum_g[32 * r + oo]; mgx += d.cbn_sum_gx[32 * r + oo]; }
                const float invN = 1.f / (float)Npx;
                mg *= invN; mgx *= invN;
            }
        }
        cst[oo] = c1; cst[32 + oo] = mean; cst[64 + oo] = invstd; cst[96 + oo] = mg; cst[128 + oo] = mgx;
    } else if (tid < 64) {
        const int k = tid - 32;
        float sc = 1.f, sh = 0.f;
        if (has_bn && k < I) {
            const float mean = d.bn_save_mean[k], invstd = d.bn_save_invstd[k];
            sc = d.bn_gamma[k] * invstd;
            sh = d.bn_beta[k] - mean * sc;
        }
        cst[160 + k] = sc; cst[192 + k] = sh;
    }
    __syncthreads();

    const int64_t tile0 = blockIdx.x, tstep = gridDim.x;
    f32x16 acc[9];
    float gsum[4] = {0.f, 0.f, 0.f, 0.f};
    if (walker) {
        // =========================================== WALK ===========================================
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        // the wave's two K steps: pixels 16 k + 8 hs .. + 7 of the tile -> byte offsets of the operands inside a plane
        int aoff[2], boff[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pp = 16 * (2 * wid + j) + 8 * hs;
            const int sg = pp >> g.lgSP, q = pp & ((1 << g.lgSP) - 1), row = q >> g.lgW, x0 = q & (g.W - 1);
            aoff[j] = c32 * NF_CBW_GCH + 2 * pp;
            boff[j] = c32 * g.ACH + 8 + (sg * (g.TH + 2) + row + 1) * g.RSB + 2 * x0;
        }
        __syncthreads();                               // frame pair 0 is complete (the fillers' matching barrier follows their first convert)
        int it = 0;
        for (int64_t tile = tile0; tile < g.tiles; tile += tstep, ++it) {
            const char* fa = lds + (it & 1) * BUF;     // activation planes
            const char* fg = fa + 3 * PA;              // G planes
            NF_CBW_STAMP(0, it >= 2 && it < 10 ? 3 * (it - 2) : -1);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bf16x8 a[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) a[q] = *(const bf16x8*)(fg + q * PG + aoff[j]);
#pragma unroll
                for (int dy = -1; dy <= 1; ++dy) {
                    bf16x8 b[3][3];                    // [dx + 1][plane]
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const char* bp = fa + q * PA + boff[j] + dy * g.RSB;
                        const u32x2 lo = *(const u32x2*)(bp), hi = *(const u32x2*)(bp + 8);
                        const unsigned prev = *(const unsigned*)(bp - 4), next = *(const unsigned*)(bp + 16);
                        b[1][q] = __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
                        b[0][q] = __builtin_bit_cast(bf16x8, u32x4{__builtin_amdgcn_alignbit(lo[0], prev, 16), __builtin_amdgcn_alignbit(lo[1], lo[0], 16),
                                                                   __builtin_amdgcn_alignbit(hi[0], lo[1], 16), __builtin_amdgcn_alignbit(hi[1], hi[0], 16)});
                        b[2][q] = __builtin_bit_cast(bf16x8, u32x4{__builtin_amdgcn_alignbit(lo[1], lo[0], 16), __builtin_amdgcn_alignbit(hi[0], lo[1], 16),
                                                                   __builtin_amdgcn_alignbit(hi[1], hi[0], 16), __builtin_amdgcn_alignbit(next, hi[1], 16)});
                    }
                    // six products per tap (NF_CC_MFMA6's order), the three taps of the row interleaved: independent accumulators back to back
#define NF_CBW_STEP(AQ, BQ)                                                                                                                  \
    _Pragma("unroll") for (int dx = 0; dx < 3; ++dx)                                                                                         \
        acc[3 * (dy + 1) + dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[AQ], b[dx][BQ], acc[3 * (dy + 1) + dx], 0, 0, 0)
                    NF_CBW_STEP(0, 2); NF_CBW_STEP(2, 0); NF_CBW_STEP(1, 1); NF_CBW_STEP(0, 1); NF_CBW_STEP(1, 0); NF_CBW_STEP(0, 0);
#undef NF_CBW_STEP
                }
            }
#ifdef NF_CB_PROF
            if (acc[0][0] == 123.456f) acc[1][0] += 1.f;   // (the stamp waits for the accumulators)
            NF_CBW_STAMP(0, it >= 2 && it < 10 ? 3 * (it - 2) + 1 : -1);
#endif
            __syncthreads();
            NF_CBW_STAMP(0, it >= 2 && it < 10 ? 3 * (it - 2) + 2 : -1);
        }
    } else {
        // =========================================== FILL ===========================================
        const int ft = tid - NF_CBW_FILL;              // 0 .. 255
        // G items: round r -> channel 8 r + ft / 32, pixels 4 (ft % 32) .. + 3 of the tile
        const int gq4 = ft & 31, gch0 = ft >> 5;
        const int gpx = 4 * gq4;
        const int gsg = g.SEG == 1 ? 0 : gpx >> g.lgHW, gqq = g.SEG == 1 ? gpx : gpx & (g.HW - 1);
        const unsigned goff0 = 4u * (unsigned)((gsg * 32 + gch0) * g.HW + gqq);       // + r * 8 channels
        const unsigned gcstep = 4u * 8u * (unsigned)g.HW;
        const int glds0 = gch0 * NF_CBW_GCH + 8 * gq4;                                 // + r * 8 * GCH
        // activation items: i = 256 r + ft -> channel i / NQ, item j = i % NQ -> (segment, frame row, four-pixel group)
        unsigned xoff[NF_CBW_MAXR];
        int xlds[NF_CBW_MAXR];
        unsigned xchn = 0u;                            // 6 bits per item: channel (the BatchNorm constants are read from LDS per tile: registers hold two tiles of loads)
        unsigned xmeta = 0u;                           // 4 bits per item: class (0 never, 1 inside, 2 top halo row, 3 bottom halo row) | 4: exists; segment in xseg
        unsigned xseg = 0u;
        const int QW = g.W >> 2, RW = g.TH + 2;
#pragma unroll
        for (int r = 0; r < NF_CBW_MAXR; ++r) {
            const int i = NF_CBW_FILL * r + ft;
            unsigned off = 0u, cls = 0u, sg = 0u;
            int ldo = 0;
            unsigned chn = 0u;
            if (r < g.nr && i < 32 * g.NQ) {
                const int ch = i / g.NQ, j = i - ch * g.NQ;
                const int rw = j / QW, xq = j - rw * QW;
                const int sm = rw / RW, frow = rw - sm * RW;
                ldo = ch * g.ACH + 8 + rw * g.RSB + 8 * xq;
                cls = 4u;
                if (ch < I) {
                    cls |= frow == 0 ? 2u : (frow == g.TH + 1 ? 3u : 1u);
                    off = 4u * (unsigned)((sm * I + ch) * g.HW + frow * g.W + 4 * xq);   // from (sample b0, channel 0, row y0 - 1)
                    sg = (unsigned)sm;
                    chn = (unsigned)ch;
                }
            }
            xoff[r] = off; xlds[r] = ldo;
            xchn |= chn << (6 * r);
            xmeta |= cls << (4 * r);
            xseg |= sg << (4 * r);
        }
        // TWO register sets: the loads of tiles k + 1 and k + 2 are in flight while tile k is walked (one set was 68 KB per compute
        // unit in flight for one round trip per tile: 2.7 TB/s; the set index is a compile-time constant: the tile loop is unrolled by two)
        f32x4 rp[2][4], rs[2][4], ro[2][4], rx[2][NF_CBW_MAXR];       // plain gradient (g_direct or g_skip), gn_src, out, activations
        unsigned gok[2] = {0u, 0u}, xok[2] = {0u, 0u};                // validity of the tile in flight
        const float* const plain = d.g_direct != nullptr ? d.g_direct : d.g_skip;   // (the host routes layers with both elsewhere)
        auto issue = [&](auto SET, int64_t tile) {
            constexpr int S = decltype(SET)::value;
            const int64_t P0 = tile * 128;
            const int64_t b0 = P0 >> g.lgHW;
            const int q0 = g.SEG == 1 ? (int)(P0 & (g.HW - 1)) : 0;
            const int y0 = q0 >> g.lgW;
            const int64_t gbase = b0 * 32 * g.HW + q0;
            const bool pv = b0 + gsg < g.B;
            gok[S] = pv ? 1u : 0u;
            unsigned go[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) go[r] = pv ? goff0 + (unsigned)r * gcstep : NF_CB_OOB;
            if (plain != nullptr) {
                const __amdgpu_buffer_rsrc_t rr = nf_cb_rsrc(plain + gbase);
#pragma unroll
                for (int r = 0; r < 4; ++r) rp[S][r] = nf_cbw_ld128(rr, go[r]);
            }
            if (has_src) {
                const __amdgpu_buffer_rsrc_t r3 = nf_cb_rsrc(d.gn_src + gbase), r4 = nf_cb_rsrc(d.out + gbase);
#pragma unroll
                for (int r = 0; r < 4; ++r) { rs[S][r] = nf_cbw_ld128(r3, go[r]); ro[S][r] = nf_cbw_ld128(r4, go[r]); }
            }
            const __amdgpu_buffer_rsrc_t rxs = nf_cb_rsrc(d.in + b0 * I * g.HW + (int64_t)(y0 - 1) * g.W);
            const bool top = g.SEG == 1 && y0 > 0, bot = g.SEG == 1 && y0 + g.TH < g.H;
            unsigned okm = 0u;
#pragma unroll
            for (int r = 0; r < NF_CBW_MAXR; ++r)
                if (r < g.nr) {                        // uniform
                    const unsigned cls = (xmeta >> (4 * r)) & 3u;
                    const int64_t sm = (int64_t)((xseg >> (4 * r)) & 15u);
                    const bool ok = (cls == 1u || (cls == 2u && top) || (cls == 3u && bot)) && b0 + sm < g.B;
                    okm |= (ok ? 1u : 0u) << r;
                    rx[S][r] = nf_cbw_ld128(rxs, ok ? xoff[r] : NF_CB_OOB);
                }
            xok[S] = okm;
        };
        auto convert = [&](auto SET) {                 // register set S -> frame pair S
            constexpr int S = decltype(SET)::value;
            char* fa = lds + S * BUF;
            char* fg = fa + 3 * PA;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (plain != nullptr) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = rp[S][r][j];
                }
                if (has_src) {
                    const int c = 8 * r + gch0;
                    const float kc1 = cst[c], kmean = cst[32 + c], kinv = cst[64 + c], kmg = cst[96 + c], kmgx = cst[128 + c];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float xh = (ro[S][r][j] - kmean) * kinv;
                        v[j] += kc1 * (rs[S][r][j] - kmg - xh * kmgx);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = gok[S] ? v[j] : 0.f;
                gsum[r] += (v[0] + v[1]) + (v[2] + v[3]);
                nf_cbw_put4(fg + glds0 + r * 8 * NF_CBW_GCH, PG, v);
            }
#pragma unroll
            for (int r = 0; r < NF_CBW_MAXR; ++r)
                if (r < g.nr && ((xmeta >> (4 * r)) & 4u)) {
                    const bool ok = (xok[S] >> r) & 1u;
                    const int ch = (int)((xchn >> (6 * r)) & 63u);
                    const float xsc = cst[160 + ch], xsh = cst[192 + ch];
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float x = rx[S][r][j];
                        if (has_bn) x = fmaxf(fmaf(x, xsc, xsh), 0.f);
                        v[j] = ok ? x : 0.f;
                    }
                    nf_cbw_put4(fa + xlds[r], PA, v);
                }
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        const int64_t n = tile0 < g.tiles ? (g.tiles - tile0 + tstep - 1) / tstep : 0;      // tiles of this workgroup: k -> tile0 + k tstep, set k & 1
        if (n > 0) issue(S0{}, tile0);
        if (n > 1) issue(S1{}, tile0 + tstep);
        if (n > 0) convert(S0{});
        if (n > 2) issue(S0{}, tile0 + 2 * tstep);
        __syncthreads();                               // frame pair 0 is complete
        for (int64_t k = 0; k < n; k += 2) {           // while the walkers are on tile k (pair 0) / k + 1 (pair 1)
            NF_CBW_STAMP(256, k >= 2 && k < 10 ? 24 + 4 * (int)(k - 2) : -1);
            if (k + 1 < n) {
                convert(S1{});
#ifdef NF_CB_PROF
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                NF_CBW_STAMP(256, k >= 2 && k < 10 ? 24 + 4 * (int)(k - 2) + 1 : -1);
#endif
                if (k + 3 < n) issue(S1{}, tile0 + (k + 3) * tstep);
            }
            NF_CBW_STAMP(256, k >= 2 && k < 10 ? 24 + 4 * (int)(k - 2) + 2 : -1);
            __syncthreads();
            NF_CBW_STAMP(256, k >= 2 && k < 10 ? 24 + 4 * (int)(k - 2) + 3 : -1);
            if (k + 1 >= n) break;
            NF_CBW_STAMP(256, k + 1 >= 2 && k + 1 < 10 ? 24 + 4 * (int)(k + 1 - 2) : -1);
            if (k + 2 < n) {
                convert(S0{});
#ifdef NF_CB_PROF
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                NF_CBW_STAMP(256, k + 1 >= 2 && k + 1 < 10 ? 24 + 4 * (int)(k + 1 - 2) + 1 : -1);
#endif
                if (k + 4 < n) issue(S0{}, tile0 + (k + 4) * tstep);
            }
            NF_CBW_STAMP(256, k + 1 >= 2 && k + 1 < 10 ? 24 + 4 * (int)(k + 1 - 2) + 2 : -1);
            __syncthreads();
            NF_CBW_STAMP(256, k + 1 >= 2 && k + 1 < 10 ? 24 + 4 * (int)(k + 1 - 2) + 3 : -1);
        }
    }
    // (both roles have executed the same number of barriers: one before the loop, one per tile)
    // ---- the four walkers' tap tiles meet in LDS (fixed order), one slab per workgroup ----
    float* red = smem;
#pragma unroll 1
    for (int pass = 0; pass < 4; ++pass) {
        if (walker && wid == pass) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float* q = red + t * 1024 + nf_cv_cd_row(r, hs) * 32 + c32;
                    *q = pass == 0 ? acc[t][r] : *q + acc[t][r];
                }
        }
        __syncthreads();
    }
    float* slab = d.g_weff + (int64_t)blockIdx.x * 9 * 32 * I;
    for (int e = tid; e < 9 * 1024; e += NF_CBW_THREADS) {
        const int tap = e >> 10, oc = (e >> 5) & 31, ic = e & 31;
        if (ic < I) slab[(tap * 32 + oc) * I + ic] = red[e];
    }
    NF_DET_REPL_CHAIN(NF_STAT_REPL);                   // (blockIdx.y = the layer: chains per layer and replica)
    NF_DET_ENTER_ALL_K(nf_cbk);                        // (one thread per output channel and workgroup)
    if (!walker && d.g_bias != nullptr) {
        const int ft = tid - NF_CBW_FILL;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = gsum[r];
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, NF_WAVE);
            if ((ft & 31) == 0) atomicAdd(d.g_bias + 256 * (blockIdx.x % NF_STAT_REPL) + 8 * r + (ft >> 5), v);
        }
    }
    NF_DET_LEAVE_ALL_K(nf_cbk);
}

template <typename K>
static inline int nf_cb_optin(K kernel) {
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
static int nf_cb_cus() {
    static int cus = -1;
    if (cus < 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 256;
        cus = n > 0 ? n : 256;
    }
    return cus;
}
// ---------------------------------------------------------------------------------------------------------------------------------
// The 1x1 output convolution of the conditioner (flows/modules.py:429-434: BatchNorm2d, ReLU, 32 -> O channels) at large batches.
// No tap reaches a neighbour, so there is no frame and no LDS traffic at all: lane (pixel c32 of a 32-pixel block, K half hs) loads
// the eight channels of the octets 2 p + hs of ITS pixel straight into the B-operand layout (eight dword loads per octet, 32 lanes =
// 128 consecutive bytes each), applies BatchNorm + ReLU and the three-way split in registers; the weights are A operands held in
// registers for the whole launch.  A wave owns units of 64 pixels, the next unit's loads in flight under the current one's products.
// The launches move whole tensors for 1.2 GFLOP: HBM-bound (forward 23 MB, data gradient 40 MB at 16 x 16, B = 512).
//   k_conv1_bulk_fwd<OB>: out rows 32 ob .. of O <= 32 OB channels;  k_conv1_bulk_bwd<NP>: G = g_direct (O <= 16 NP channels),
//   gn_out = (W^T G) [act > 0] with its two batch sums.
// ---------------------------------------------------------------------------------------------------------------------------------
// eight fp32 values -> the three bf16x8 planes of an operand
__device__ __forceinline__ void nf_c1_split8(const float (&v)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        bf16x2 h2, m2, l2;
        nf_cc_split2(f32x2{v[j], v[j + 1]}, h2, m2, l2);
        h[j] = h2[0]; h[j + 1] = h2[1]; m[j] = m2[0]; m[j + 1] = m2[1]; l[j] = l2[0]; l[j + 1] = l2[1];
    }
}
#define NF_C1_MFMA6(ACC, A, B)                                                           \
    do {                                                                                 \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[2], ACC, 0, 0, 0);         \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[2], B[0], ACC, 0, 0, 0);         \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[1], ACC, 0, 0, 0);         \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[1], ACC, 0, 0, 0);         \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[0], ACC, 0, 0, 0);         \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[0], ACC, 0, 0, 0);         \
    } while (0)
struct NfC1Geo { int HW, lgHW; int64_t B, Npx, units; };
// byte offset of (sample of pixel p, channel 0, pixel) in a (B, C, H, W) tensor, or out of range
__device__ __forceinline__ unsigned nf_c1_off(const NfC1Geo& g, int64_t p, int C) {
    const int64_t b = p >> g.lgHW;
    const int q = (int)(p & (g.HW - 1));
    return p < g.Npx ? 4u * (unsigned)(b * C * g.HW + q) : NF_CB_OOB;
}

template <int OB>
__global__ void __launch_bounds__(NF_CB_THREADS) k_conv1_bulk_fwd(nf_conv_desc d, NfC1Geo g, int O, int training, float eps, float mom) {
    __shared__ float kc[64];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), c32 = lane & 31, hs = lane >> 5;
    // A operands: rows = output channels 32 ob + c32, K = input channels of the octets 2 p + hs (p = 0, 1)
    bf16x8 a[OB][2][3];
#pragma unroll
    for (int ob = 0; ob < OB; ++ob)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float v[8];
            const int oc = 32 * ob + c32;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = oc < O ? d.weight[(size_t)oc * 32 + 8 * (2 * p + hs) + j] : 0.f;
            nf_c1_split8(v, a[ob][p][0], a[ob][p][1], a[ob][p][2]);
        }
    nf_cv_bn_consts_fwd(kc, d, 32, g.Npx, training, eps, mom);
    float bias_r[OB][16];
#pragma unroll
    for (int ob = 0; ob < OB; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int oc = 32 * ob + nf_cv_cd_row(r, hs);
            bias_r[ob][r] = oc < O ? d.bias[oc] : 0.f;
        }
    __syncthreads();
    float sc[2][8], sh[2][8];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int j = 0; j < 8; ++j) { sc[p][j] = kc[8 * (2 * p + hs) + j]; sh[p][j] = kc[32 + 8 * (2 * p + hs) + j]; }
    const __amdgpu_buffer_rsrc_t ri = nf_cb_rsrc(d.in), ro = nf_cb_rsrc(d.out);
    const int cstride = 4 * g.HW;
    const int64_t stride = (int64_t)gridDim.x * NF_CB_WAVES;
    int64_t u = (int64_t)blockIdx.x * NF_CB_WAVES + wid;
    float raw[2][2][8];
    auto issue = [&](int64_t un) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const unsigned off = nf_c1_off(g, un * 64 + 32 * nb + c32, 32);
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int j = 0; j < 8; ++j) raw[nb][p][j] = nf_cb_ld(ri, off, (8 * (2 * p + hs) + j) * cstride);
        }
    };
    if (u < g.units) issue(u);
    for (; u < g.units; u += stride) {
        bf16x8 b[2][2][3];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaxf(fmaf(raw[nb][p][j], sc[p][j], sh[p][j]), 0.f);
                nf_c1_split8(v, b[nb][p][0], b[nb][p][1], b[nb][p][2]);
            }
        if (u + stride < g.units) issue(u + stride);
#pragma unroll
        for (int ob = 0; ob < OB; ++ob) {
            f32x16 acc[2];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) NF_C1_MFMA6(acc[nb], a[ob][p], b[nb][p]);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const unsigned off = nf_c1_off(g, u * 64 + 32 * nb + c32, O);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int oc = 32 * ob + nf_cv_cd_row(r, hs);
                    nf_cb_st(ro, oc < O ? off : NF_CB_OOB, oc * cstride, acc[nb][r] + bias_r[ob][r]);
                }
            }
        }
    }
}

template <int NP>
__global__ void __launch_bounds__(NF_CB_THREADS) k_conv1_bulk_bwd(nf_conv_bwd_desc d, NfC1Geo g, int O) {
    __shared__ float kc[4 * 32];
    __shared__ float red[2 * 4 * 32];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), c32 = lane & 31, hs = lane >> 5;
    // A operands: rows = input channels c32, K = output channels of the octets 2 p + hs (the transposed weight)
    bf16x8 a[NP][3];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int oc = 8 * (2 * p + hs) + j;
            v[j] = oc < O ? d.weight[(size_t)oc * 32 + c32] : 0.f;
        }
        nf_c1_split8(v, a[p][0], a[p][1], a[p][2]);
    }
    if (threadIdx.x < 32) {
        const int k = threadIdx.x;
        const float mean = d.bn_save_mean[k], invstd = d.bn_save_invstd[k];
        const float scv = d.bn_gamma[k] * invstd;
        kc[k] = scv; kc[32 + k] = d.bn_beta[k] - mean * scv; kc[64 + k] = mean; kc[96 + k] = invstd;
    }
    __syncthreads();
    float ksc[16], ksh[16], kmean[16], kinv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ic = nf_cv_cd_row(r, hs);
        ksc[r] = kc[ic]; ksh[r] = kc[32 + ic]; kmean[r] = kc[64 + ic]; kinv[r] = kc[96 + ic];
    }
    const __amdgpu_buffer_rsrc_t rg = nf_cb_rsrc(d.g_direct), ri = nf_cb_rsrc(d.in), rn = nf_cb_rsrc(d.gn_out);
    const int cstride = 4 * g.HW;
    const int64_t stride = (int64_t)gridDim.x * NF_CB_WAVES;
    int64_t u = (int64_t)blockIdx.x * NF_CB_WAVES + wid;
    float raw[2][NP][8], xin[2][16];
    auto issue = [&](int64_t un) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int64_t p0 = un * 64 + 32 * nb + c32;
            const unsigned og = nf_c1_off(g, p0, O), oi = nf_c1_off(g, p0, 32);
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int oc = 8 * (2 * p + hs) + j;
                    raw[nb][p][j] = nf_cb_ld(rg, oc < O ? og : NF_CB_OOB, oc * cstride);
                }
#pragma unroll
            for (int r = 0; r < 16; ++r) xin[nb][r] = nf_cb_ld(ri, oi, nf_cv_cd_row(r, hs) * cstride);
        }
    };
    float sg[16], sgx[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { sg[r] = 0.f; sgx[r] = 0.f; }
    if (u < g.units) issue(u);
    for (; u < g.units; u += stride) {
        bf16x8 b[2][NP][3];
        float x[2][16];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
            for (int p = 0; p < NP; ++p) nf_c1_split8(raw[nb][p], b[nb][p][0], b[nb][p][1], b[nb][p][2]);
#pragma unroll
            for (int r = 0; r < 16; ++r) x[nb][r] = xin[nb][r];
        }
        if (u + stride < g.units) issue(u + stride);
        f32x16 acc[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) NF_C1_MFMA6(acc[nb], a[p], b[nb][p]);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const unsigned off = nf_c1_off(g, u * 64 + 32 * nb + c32, 32);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float gn = fmaf(x[nb][r], ksc[r], ksh[r]) > 0.f ? acc[nb][r] : 0.f;
                gn = off != NF_CB_OOB ? gn : 0.f;
                sg[r] += gn;
                sgx[r] = fmaf(gn, (x[nb][r] - kmean[r]) * kinv[r], sgx[r]);
                nf_cb_st(rn, off, nf_cv_cd_row(r, hs) * cstride, gn);
            }
        }
    }
    if (d.sum_g != nullptr) {                          // block-uniform
        const float t1 = nf_cv_butterfly16(sg, c32), t2 = nf_cv_butterfly16(sgx, c32);
        if ((c32 & 1) == 0) {
            const int ic = nf_cv_cd_row(c32 >> 1, hs);
            red[(0 * 4 + wid) * 32 + ic] = t1;
            red[(1 * 4 + wid) * 32 + ic] = t2;
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) t += red[(hs * 4 + w) * 32 + c32];
            if (!(nf_det_on(nf_cbk_det) && nf_cbk_det_fold64(t, true, d.sum_g, d.sum_gx))) {
                NF_DET_REPL_CHAIN(NF_STAT_REPL);       // (deterministic mode beyond the fold's slab: a chain per replica)
                NF_DET_ENTER_WAVE_K(nf_cbk);
                const int rep = 32 * (blockIdx.x % NF_STAT_REPL);
                atomicAdd((hs == 0 ? d.sum_g : d.sum_gx) + rep + c32, t);
                NF_DET_LEAVE_WAVE_K(nf_cbk);
            }
        }
    }
}

static bool nf_c1_geometry(NfC1Geo& g, int64_t B, int H, int W) {
    g.HW = H * W; g.lgHW = nf_cv_log2(g.HW); g.B = B; g.Npx = B * g.HW;
    g.units = (g.Npx + 63) / 64;
    return g.lgHW >= 5 && B * 192 * (int64_t)g.HW < ((int64_t)1 << 29);      // whole 32-pixel blocks inside a sample; 32-bit byte offsets
}
static int nf_c1_on() { return 1; }
int nf_conv1_bulk_fwd_plan(const nf_conv_desc* d, int64_t B, int I, int O, int H, int W, int ksize) {
    NfC1Geo g;
    if (!nf_c1_on() || !nf_cb_on() || ksize != 1 || I != 32 || O < 1 || O > 64 || B * H * W < nf_cb_min_px()) return 0;
    if (d->bn_gamma == nullptr || d->residual != nullptr || d->stat_sum != nullptr) return 0;
    return nf_c1_geometry(g, B, H, W) ? 1 : 0;
}
int nf_conv1_bulk_fwd(const nf_conv_desc* desc, int64_t B, int O, int H, int W, int training, float eps, float mom, hipStream_t st) {
    NfC1Geo g;
    if (!nf_c1_geometry(g, B, H, W)) return NF_E_BADARG;
    const int64_t wgs = (g.units + NF_CB_WAVES - 1) / NF_CB_WAVES;
    const unsigned grid = (unsigned)(wgs < nf_cb_cus() ? wgs : nf_cb_cus());
    if (O <= 32) hipLaunchKernelGGL((k_conv1_bulk_fwd<1>), dim3(grid), dim3(NF_CB_THREADS), 0, st, *desc, g, O, training, eps, mom);
    else hipLaunchKernelGGL((k_conv1_bulk_fwd<2>), dim3(grid), dim3(NF_CB_THREADS), 0, st, *desc, g, O, training, eps, mom);
    NF_CHECK_LAUNCH();
    return 0;
}
int nf_conv1_bulk_bwd_plan(const nf_conv_bwd_desc* d, int64_t B, int I, int O, int H, int W, int ksize) {
    NfC1Geo g;
    if (!nf_c1_on() || !nf_cb_on() || ksize != 1 || I != 32 || O < 1 || O > 48 || B * H * W < nf_cb_min_px()) return 0;
    if (d->bn_gamma == nullptr || d->g_direct == nullptr || d->gn_src != nullptr || d->g_skip != nullptr || d->g_weff != nullptr ||
        d->g_bias != nullptr || d->g_store != nullptr || d->gn_out == nullptr)
        return 0;
    return nf_c1_geometry(g, B, H, W) ? 1 : 0;
}
int nf_conv1_bulk_bwd(const nf_conv_bwd_desc* desc, int64_t B, int O, int H, int W, hipStream_t st) {
    NfC1Geo g;
    if (!nf_c1_geometry(g, B, H, W)) return NF_E_BADARG;
    const int64_t wgs = (g.units + NF_CB_WAVES - 1) / NF_CB_WAVES;
    const unsigned grid = (unsigned)(wgs < nf_cb_cus() ? wgs : nf_cb_cus());
    const int np = (O + 15) / 16;
    if (np == 1) hipLaunchKernelGGL((k_conv1_bulk_bwd<1>), dim3(grid), dim3(NF_CB_THREADS), 0, st, *desc, g, O);
    else if (np == 2) hipLaunchKernelGGL((k_conv1_bulk_bwd<2>), dim3(grid), dim3(NF_CB_THREADS), 0, st, *desc, g, O);
    else hipLaunchKernelGGL((k_conv1_bulk_bwd<3>), dim3(grid), dim3(NF_CB_THREADS), 0, st, *desc, g, O);
    NF_CHECK_LAUNCH();
    return 0;
}

// does the large-batch kernel take this forward launch?  (called by nf_conv_bn_fwd; 0 = no, else NBLK)
int nf_conv_bulk_fwd_plan(const nf_conv_desc* d, int64_t B, int I, int O, int H, int W, int ksize) {
    if (!nf_cb_on() || ksize != 3 || O != 32 || I < 1 || I > 32 || B * H * W < nf_cb_min_px()) return 0;
    if (!(B * 32 * (int64_t)H * W < (int64_t)1 << 31)) return 0;
    NfCbGeo cg;
    return nf_cb_plan(cg, B, I, H, W, (I + 7) / 8, nf_cb_force_nblk());
}

int nf_conv_bulk_fwd(const nf_conv_desc* desc, int64_t B, int I, int H, int W, int training, float eps, float mom, hipStream_t st) {
    NfCbGeo cg;
    const int noct = (I + 7) / 8;
    const int nblk = nf_cb_plan(cg, B, I, H, W, noct, nf_cb_force_nblk());
    if (nblk == 0) return NF_E_BADARG;
    const size_t lds = sizeof(float) * nf_cb_lds_floats(cg.g.CS);
    const int64_t wgs = (cg.units + NF_CB_WAVES - 1) / NF_CB_WAVES;
    const unsigned grid = (unsigned)(wgs < nf_cb_cus() ? wgs : nf_cb_cus());
    const bool has_bn = desc->bn_gamma != nullptr, has_res = desc->residual != nullptr;
    int rc = 0;
#define NF_CB_FWD3(NOCT_, NBLK_, RG_, BN_, RS_)                                                                                         \
    do {                                                                                                                                \
        rc = nf_cb_optin(k_conv3_bulk_fwd<NOCT_, NBLK_, RG_, BN_, RS_>);                                                                \
        if (rc == 0)                                                                                                                    \
            hipLaunchKernelGGL((k_conv3_bulk_fwd<NOCT_, NBLK_, RG_, BN_, RS_>), dim3(grid), dim3(NF_CB_THREADS), lds, st, *desc, cg, I, training, eps, mom); \
    } while (0)
#define NF_CB_FWD2(NOCT_, NBLK_, RG_)                                                                                                   \
    do {                                                                                                                                \
        if (has_bn && has_res) NF_CB_FWD3(NOCT_, NBLK_, RG_, true, true);                                                               \
        else if (has_bn) NF_CB_FWD3(NOCT_, NBLK_, RG_, true, false);                                                                    \
        else if (has_res) NF_CB_FWD3(NOCT_, NBLK_, RG_, false, true);                                                                   \
        else NF_CB_FWD3(NOCT_, NBLK_, RG_, false, false);                                                                               \
    } while (0)
#define NF_CB_FWD(NOCT_, NBLK_)                                                                                                         \
    do {                                                                                                                                \
        if (I % 8) NF_CB_FWD2(NOCT_, NBLK_, true);                                                                                      \
        else NF_CB_FWD2(NOCT_, NBLK_, false);                                                                                           \
    } while (0)
    if (nblk == 2) {
        switch (noct) {
            case 1: NF_CB_FWD(1, 2); break;
            case 2: NF_CB_FWD(2, 2); break;
            case 3: NF_CB_FWD(3, 2); break;
            default: NF_CB_FWD(4, 2); break;
        }
    } else {
        switch (noct) {
            case 1: NF_CB_FWD(1, 1); break;
            case 2: NF_CB_FWD(2, 1); break;
            case 3: NF_CB_FWD(3, 1); break;
            default: NF_CB_FWD(4, 1); break;
        }
    }
#undef NF_CB_FWD
#undef NF_CB_FWD2
#undef NF_CB_FWD3
    if (rc) return rc;
    NF_CHECK_LAUNCH();
    return 0;
}

// weight-gradient pass of up to NF_CONV_WGRAD_MAX layers of one shape (called by nf_conv_bn_wgrad_multi; 0 = the kernels of conv_bn.hip)
int nf_conv_bulk_wgrad_plan(int64_t B, int I, int O, int H, int W, int ksize) {
    const int on = 1;
    // (the weight pass starts one tile count earlier than the data passes: at exactly 16 384 pixels -- config 4's per-GPU shard at
    // the 16 x 16 level, whose data passes run the persistent chain -- it measured 24.56 against 24.9 ms per step)
    // Round 6: from 4 096 pixels (config 4's 8 x 8 level) when the data passes' threshold is at its default -- the contraction over
    // pixels on the bf16 x 3 pipe against conv_bn.hip's fp32 MFMA: 20.94 -> 20.83 ms per step (tools/probes/ab_bulk_wgrad_8x8.sh).
    const int64_t min_px = nf_cb_min_px() == 16384 + 1 ? 4096 : nf_cb_min_px() - 1;
    if (!on || !nf_cb_on() || ksize != 3 || O != 32 || I < 1 || I > 32 || B * H * W < min_px) return 0;
    NfCbwGeo g;
    return nf_cbw_geometry(g, B, H, W) ? 1 : 0;
}
// descs (host, n <= NF_CONV_WGRAD_MAX) travel in the kernel arguments, or tab (device, any n; descs == NULL) is read by the workgroups;
// the caller has checked that every layer has ONE plain gradient tensor
int nf_conv_bulk_wgrad(const nf_conv_bwd_desc* descs, const nf_conv_bwd_desc* tab, int n, int64_t B, int I, int H, int W, int slabs,
                       hipStream_t st) {
    NfCbwGeo g;
    if (!nf_cbw_geometry(g, B, H, W) || n < 1 || slabs < 1 || (descs == nullptr) == (tab == nullptr)) return NF_E_BADARG;
    if (descs != nullptr && n > NF_CONV_WGRAD_MAX) return NF_E_BADARG;
    NfCbwMulti m{};
    if (descs != nullptr)
        for (int k = 0; k < n; ++k) m.d[k] = descs[k];
    const size_t lds = nf_cbw_lds_bytes(g);
    int rc = nf_cb_optin(k_conv3_bulk_wgrad);
    if (rc) return rc;
    const unsigned gx = (unsigned)(g.tiles < slabs ? g.tiles : slabs);
    if ((int)gx != slabs) return NF_E_BADARG;          // (every slab the caller sums must be written)
    hipLaunchKernelGGL(k_conv3_bulk_wgrad, dim3(gx, (unsigned)n), dim3(NF_CBW_THREADS), lds, st, m, tab, g, I);
    NF_CHECK_LAUNCH();
    return 0;
}

// backward data pass (g_weff == NULL): 3 x 3, O = 32, I <= 32, G = [g_skip +] BNbwd(gn_src)
int nf_conv_bulk_bwd_plan(const nf_conv_bwd_desc* d, int64_t B, int I, int O, int H, int W, int ksize) {
    if (!nf_cb_on() || ksize != 3 || O != 32 || I < 1 || I > 32 || B * H * W < nf_cb_min_px()) return 0;
    if (d->g_weff != nullptr || d->g_bias != nullptr || d->g_direct != nullptr || d->gn_src == nullptr || d->out == nullptr) return 0;
    if (!(B * 32 * (int64_t)H * W < (int64_t)1 << 31)) return 0;
    NfCbGeo cg;
    return nf_cb_plan(cg, B, 32, H, W, 4, nf_cb_force_nblk());
}

int nf_conv_bulk_bwd(const nf_conv_bwd_desc* desc, int64_t B, int I, int H, int W, hipStream_t st) {
    NfCbGeo cg;
    const int nblk = nf_cb_plan(cg, B, 32, H, W, 4, nf_cb_force_nblk());
    if (nblk == 0) return NF_E_BADARG;
    const size_t lds = sizeof(float) * nf_cb_lds_floats(cg.g.CS);
    const int64_t wgs = (cg.units + NF_CB_WAVES - 1) / NF_CB_WAVES;
    const unsigned grid = (unsigned)(wgs < nf_cb_cus() ? wgs : nf_cb_cus());
    const bool skip = desc->g_skip != nullptr;
    nf_conv_bwd_desc dd = *desc;
    if (dd.wpk != nullptr) dd.wpk += NF_CONV_PACK_IMAGE_FLOATS;    // I <= 32: image 0 = forward, image 1 = transposed
    desc = &dd;
    int rc = 0;
#define NF_CB_BWD(NBLK_, SKIP_)                                                                                                \
    do {                                                                                                                       \
        if (desc->bn_gamma != nullptr) {                                                                                       \
            rc = nf_cb_optin(k_conv3_bulk_bwd<NBLK_, SKIP_, true>);                                                            \
            if (rc == 0)                                                                                                       \
                hipLaunchKernelGGL((k_conv3_bulk_bwd<NBLK_, SKIP_, true>), dim3(grid), dim3(NF_CB_THREADS), lds, st, *desc, cg, I); \
        } else {                                                                                                               \
            rc = nf_cb_optin(k_conv3_bulk_bwd<NBLK_, SKIP_, false>);                                                           \
            if (rc == 0)                                                                                                       \
                hipLaunchKernelGGL((k_conv3_bulk_bwd<NBLK_, SKIP_, false>), dim3(grid), dim3(NF_CB_THREADS), lds, st, *desc, cg, I); \
        }                                                                                                                      \
    } while (0)
    if (nblk == 2) {
        if (skip) NF_CB_BWD(2, true);
        else NF_CB_BWD(2, false);
    } else {
        if (skip) NF_CB_BWD(1, true);
        else NF_CB_BWD(1, false);
    }
#undef NF_CB_BWD
    if (rc) return rc;
    NF_CHECK_LAUNCH();
    return 0;
}
