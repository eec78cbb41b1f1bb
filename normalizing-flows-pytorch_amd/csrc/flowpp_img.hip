// Flow++ conditioner for IMAGE data (flows/coupling.py:159-166, flows/modules.py:519-578):
//     Conv2d(I0, 32, 3) -> GatedConv2d(32) -> LayerNorm(32,H,W) -> GatedAttn(4 heads over the H*W positions) -> LayerNorm -> Conv2d(32, O, 3)
// with H = W in {4, 8, 16} (the mid shapes of Flowpp on 32 x 32 and 16 x 16 images, flows/flowpp.py:22-57) and O up to 1344.
//
// Nothing on this path couples samples (LayerNorm and the attention are per sample), so there is no grid exchange: the work is cut
// per sample and the kernels are plain launches.
//   k_fi_conv        3 x 3 convolution as a GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32
//                    accumulation).  A workgroup owns 256 pixels (1 / 4 / 16 whole samples) x 32 output channels: the input
//                    frame (32 channels with a zero halo) and the 32 x 288 weight block sit in LDS, each of the 8 waves finishes
//                    a 32-pixel block.  The same kernel is the data gradient (weight block read transposed and tap-flipped)
//                    and applies concat-ELU to its input while staging (the GatedConv2d input never exists in HBM).
//   k_fi_conv64      the same on 64-pixel tiles (2 pixel blocks x 4 K quarters per workgroup) for launches of at most 128 workgroups.
//   k_fi_conv_wgrad2 weight + bias gradient of the same convolutions: D[o][c] per tap, waves = 2 pixel halves x 4 tap groups, one LDS
//                    meeting of the halves; partial-sum slabs per workgroup, folded by nf_slab_sum (no atomics).
//   k_fi_mid         everything between the convolutions for one sample per workgroup, thread (head, position): gate, LayerNorm,
//                    1x1 projections, softmax attention (two sweeps over the keys held in LDS), gate, LayerNorm; the backward
//                    variant recomputes that from the two 4-byte inputs and walks it in reverse (no activation is saved).
// Algorithmic HBM bytes per sample and direction: (I0 + 2 * 32 + 32 + 32 + O) * H * W * 4 for the forward (inputs and outputs of the
// four launches) -- the activations between the layers of `mid` stay in LDS / registers.
#include <mutex>
#include <unordered_set>

#include "nf_common.h"
#include "nf_det.h"

NF_DET_STATE(nf_fpi)
NF_DET_HOST_API(nf_fpi)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define NF_FI_THREADS 512
#define NF_FI_WS 289            // row stride of the LDS weight block (288 = 32 channels x 9 taps, odd stride: conflict-free rows)
#define NF_FI_GS 257            // row stride of the LDS gradient tile of the weight-gradient kernel
#define NF_FI_LNEPS 1.0e-5f

template <int LGW>
struct NfFiGeo {
    static constexpr int W = 1 << LGW, N = W * W, S = 256 / N, PW = W + 2, FS = PW * PW, CS = (S * FS) | 1;
};

// Maps whose side V is no power of two live in the next power-of-two storage map (side W = 1 << LGW): pixel q = y W + x of the
// storage map belongs to the image iff y < V and x < V.  Everything outside is DEAD: never read as data (the convolutions load zero
// there, exactly as for their halo), left out of the LayerNorm statistics and the attention, its stored values are meaningless.
template <int LGW>
__device__ __forceinline__ bool nf_fi_live(int q, int V) { return (q >> LGW) < V && (q & ((1 << LGW) - 1)) < V; }

template <int LGW>
__device__ __forceinline__ int nf_fi_fpos(int p) {          // frame position of pixel p (0..255) of the tile
    using G = NfFiGeo<LGW>;
    const int s = p >> (2 * LGW), y = (p >> LGW) & (G::W - 1), x = p & (G::W - 1);
    return s * G::FS + (y + 1) * G::PW + x + 1;
}

__device__ __forceinline__ int nf_fi_cd_row(int r, int hs) { return (r & 3) + 8 * (r >> 2) + 4 * hs; }
__device__ __forceinline__ float nf_fi_elu(float v) { return v > 0.f ? v : expm1f(v); }
// inside the convolution staging: v_exp_f32 (absolute error ~1e-7 on an O(1) value that then enters a K = 576 dot product)
__device__ __forceinline__ float nf_fi_elu_fast(float v) { return v > 0.f ? v : __expf(v) - 1.f; }
__device__ __forceinline__ float nf_fi_elu_grad(float v) { return v > 0.f ? 1.f : expf(v); }
__device__ __forceinline__ float nf_fi_sigmoid(float v) { return 1.f / (1.f + expf(-v)); }

// 32 channels [c0, c0 + 32) of the tile's samples [b0, b0 + S) with a zero halo; INMODE 1: the convolution sees
// concat_elu(in) = elu([in, -in]) of a (B, Ci / 2, H, W) tensor (flows/modules.py:500-517)
// Staging.  The halo of the frame is zeroed ONCE per workgroup (nf_fi_zero_frame); a chunk then brings only the 32 x 256 interior
// values: thread t owns pixel t & 255 of the tile and the channels (t >> 8) + 2 u, u < 16 -- shifts only, coalesced along the pixels.
// Loading (global -> registers, all 16 + 18 loads of a chunk unconditional through clamped 32-bit offsets, so they are in flight
// together) is split from storing (registers -> LDS, where concat-ELU is applied): the loads of chunk i + 1 are issued before the
// matrix instructions of chunk i.
template <int LGW>
__device__ __forceinline__ void nf_fi_zero_frame(float* F) {
    for (int e = threadIdx.x; e < 32 * NfFiGeo<LGW>::CS; e += NF_FI_THREADS) F[e] = 0.f;
}

// INMODE 1: the convolution sees concat_elu(in) = elu([in, -in]) of a (B, Ci / 2, H, W) tensor (flows/modules.py:500-517)
template <int LGW, int INMODE>
__device__ __forceinline__ void nf_fi_frame_load(float (&tmp)[16], const float* __restrict__ in, int64_t b0, int64_t B, int Ci, int c0,
                                                 int V) {
    using G = NfFiGeo<LGW>;
    const int p = threadIdx.x & 255, ch = threadIdx.x >> 8;
    const int bs = (int)b0 + (p >> (2 * LGW)), q = p & (G::N - 1), Ch = Ci >> 1;
    const bool bok = bs < B && nf_fi_live<LGW>(q, V);        // (a dead pixel of the storage map reads as zero, like the halo)
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int cc = c0 + ch + 2 * u;
        const bool ok = bok && cc < Ci;
        const unsigned idx = INMODE == 0 ? (unsigned)((bs * Ci + cc) * G::N + q) : (unsigned)((bs * Ch + (cc < Ch ? cc : cc - Ch)) * G::N + q);
        const float v = in[ok ? idx : 0u];
        tmp[u] = ok ? v : 0.f;
    }
}

template <int LGW, int INMODE>
__device__ __forceinline__ void nf_fi_frame_store(float* F, const float (&tmp)[16], int Ci, int c0) {
    using G = NfFiGeo<LGW>;
    const int p = threadIdx.x & 255, ch = threadIdx.x >> 8;
    float* dst = F + nf_fi_fpos<LGW>(p);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        float v = tmp[u];
        if (INMODE == 1) v = nf_fi_elu_fast(c0 + ch + 2 * u < (Ci >> 1) ? v : -v);     // (elu(0) = 0: padding stays zero)
        dst[(ch + 2 * u) * G::CS] = v;
    }
}

// Wl[r][k]: row r = output channel o0 + r of this GEMM, k = 9 * (local input channel) + tap.
//   TR = false: weight (Co, Ci, 3, 3): thread t reads row t >> 4, k = (t & 15) + 16 u;
//   TR = true : weight (Ci, Co, 3, 3) read transposed with the taps flipped (data gradient): thread t reads local input channel t >> 4
//               and the 288 (output row, tap) values (t & 15) + 16 u of it
template <bool TR>
__device__ __forceinline__ void nf_fi_w_load(float (&tmp)[18], const float* __restrict__ w, int Ci, int Co, int o0, int c0) {
    const int cc = min(32, Ci - c0), r = threadIdx.x >> 4, l16 = threadIdx.x & 15;
#pragma unroll
    for (int u = 0; u < 18; ++u) {
        const int k = l16 + 16 * u;
        const bool ok = TR ? (r < cc && o0 + k / 9 < Co) : (o0 + r < Co && k < cc * 9);
        const unsigned idx = TR ? (unsigned)(((c0 + r) * Co + o0) * 9 + k) : (unsigned)(((o0 + r) * Ci + c0) * 9 + k);
        const float v = w[ok ? idx : 0u];
        tmp[u] = ok ? v : 0.f;
    }
}

template <bool TR>
__device__ __forceinline__ void nf_fi_w_store(float* Wl, const float (&tmp)[18]) {
    const int r = threadIdx.x >> 4, l16 = threadIdx.x & 15;
#pragma unroll
    for (int u = 0; u < 18; ++u) {
        const int k = l16 + 16 * u;
        if (!TR) Wl[r * NF_FI_WS + k] = tmp[u];
        else Wl[(k / 9) * NF_FI_WS + r * 9 + 8 - (k % 9)] = tmp[u];
    }
}

// one 32-channel chunk of the K axis: groups of 8 channels x 9 taps = 36 two-deep matrix instructions, half hs of the wave
// supplies k = 72 j + 36 hs + s (any fixed pairing of the k's is a valid order of the sum)
template <int LGW>
__device__ __forceinline__ void nf_fi_kchunk(f32x16& acc, const float* Wl, const float* F, int ngroups, int fpos, int r32, int hs) {
    using G = NfFiGeo<LGW>;
    const float* wp = Wl + r32 * NF_FI_WS + 36 * hs;
    const float* fp = F + 4 * hs * G::CS + fpos;
    for (int j = 0; j < ngroups; ++j) {
#pragma unroll
        for (int s = 0; s < 36; ++s) {
            const int t = s % 9, dy = t / 3 - 1, dx = t % 3 - 1;
            const float a = wp[72 * j + s];
            const float b = fp[(8 * j + s / 9) * G::CS + dy * G::PW + dx];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
}

template <int LGW, int INMODE, bool TR>
__global__ void __launch_bounds__(NF_FI_THREADS) k_fi_conv(const float* __restrict__ in, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ out, int64_t B, int Ci,
                                                           int Co, int V) {
    using G = NfFiGeo<LGW>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* F = smem;                       // [32][CS]
    float* Wl = smem + 32 * G::CS;         // [32][NF_FI_WS]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, r32 = lane & 31, hs = lane >> 5;
    const int64_t b0 = (int64_t)blockIdx.x * G::S;
    const int o0 = 32 * blockIdx.y;
    const int p = 32 * wid + r32;
    const int fpos = nf_fi_fpos<LGW>(p);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // K split: workgroup z of gridDim.z takes the 32-channel chunks [z * cps, (z + 1) * cps) and leaves its partial sums in slab z
    const int nchunks = (Ci + 31) >> 5, cps = (nchunks + (int)gridDim.z - 1) / (int)gridDim.z;
    const int ch0 = (int)blockIdx.z * cps, ch1 = min(nchunks, ch0 + cps);
    float tf[16], tw[18];
    if (ch0 < ch1) {
        nf_fi_frame_load<LGW, INMODE>(tf, in, b0, B, Ci, 32 * ch0, V);
        nf_fi_w_load<TR>(tw, w, Ci, Co, o0, 32 * ch0);
    }
    nf_fi_zero_frame<LGW>(F);                                // (under the first loads' latency)
    for (int ch = ch0; ch < ch1; ++ch) {
        const int c0 = 32 * ch;
        __syncthreads();
        nf_fi_frame_store<LGW, INMODE>(F, tf, Ci, c0);
        nf_fi_w_store<TR>(Wl, tw);
        __syncthreads();
        if (ch + 1 < ch1) {                                  // the next chunk's operands travel while this chunk's products issue
            nf_fi_frame_load<LGW, INMODE>(tf, in, b0, B, Ci, c0 + 32, V);
            nf_fi_w_load<TR>(tw, w, Ci, Co, o0, c0 + 32);
        }
        nf_fi_kchunk<LGW>(acc, Wl, F, (min(32, Ci - c0) + 7) >> 3, fpos, r32, hs);
    }
    const int64_t b = b0 + (p >> (2 * LGW));
    const int q = p & (G::N - 1);
    float* slab = out + (int64_t)blockIdx.z * B * Co * G::N;
    const bool add_bias = bias != nullptr && blockIdx.z == 0;
    if (b < B) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int oc = o0 + nf_fi_cd_row(r, hs);
            if (oc < Co) slab[(b * Co + oc) * G::N + q] = acc[r] + (add_bias ? bias[oc] : 0.f);
        }
    }
}

// The same convolution on 64-PIXEL tiles when 256-pixel tiles leave the chip nearly empty (B = 64: 64 / 16 / 4 tiles per 32 output
// channels): a workgroup owns four rows of a 16 x 16 sample (the halo rows are read from the same tensor) or 1 / 4 whole samples, its eight waves are 2 pixel blocks x 4 K quarters (wave (pb, kq) multiplies
// the 8-channel group kq of every chunk), the quarters meet in LDS once, at the end.  Four times the workgroups, a quarter of the
// matrix instructions per wave.
template <int LGW>
struct NfFiGeo64 {
    // N <= 64: S = 64 / N whole samples per tile.  16 x 16 maps: a tile is ROWS = 4 rows of ONE sample (TPS = 4 tiles per sample); the
    // rows above and below come from the same global tensor (zero outside the image)
    static constexpr int W = 1 << LGW, N = W * W, PW = W + 2;
    static constexpr int S = N <= 64 ? 64 / N : 1, ROWS = N <= 64 ? W : 64 / W, LGR = N <= 64 ? LGW : 2, TPS = N <= 64 ? 1 : N / 64;
    static constexpr int FS = (ROWS + 2) * PW, CS = (S * FS) | 1;
    static_assert(LGW >= 2 && LGW <= 4, "4 x 4, 8 x 8 or 16 x 16 maps");
};

template <int LGW, int INMODE, bool TR>
__global__ void __launch_bounds__(NF_FI_THREADS) k_fi_conv64(const float* __restrict__ in, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ out, int64_t B, int Ci,
                                                             int Co, int V) {
    using G = NfFiGeo64<LGW>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* F = smem;                       // [32][CS]
    float* Wl = smem + 32 * G::CS;         // [32][NF_FI_WS]
    float* RED = Wl + 32 * NF_FI_WS;       // [3][2][16][64]: the K quarters 1..3 of both pixel blocks
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, r32 = lane & 31, hs = lane >> 5;
    const int pb = wid & 1, kq = wid >> 1;
    const int64_t b0 = (int64_t)(blockIdx.x / G::TPS) * G::S;
    const int rb = blockIdx.x % G::TPS;                                    // row block of the sample (16 x 16 maps)
    const int o0 = 32 * blockIdx.y;
    const int p = 32 * pb + r32;                                           // this lane's pixel of the tile (matrix phase)
    const int fpos = (p >> (LGW + G::LGR)) * G::FS + (((p >> LGW) & (G::ROWS - 1)) + 1) * G::PW + (p & (G::W - 1)) + 1;
    // staging: thread t owns pixel t & 63 and the channels (t >> 6) + 8 u, u < 4
    const int sp = threadIdx.x & 63, sch = threadIdx.x >> 6;
    const int sbs = (int)b0 + (sp >> (LGW + G::LGR)), sq = G::TPS == 1 ? (sp & (G::N - 1)) : 64 * rb + sp, Ch = Ci >> 1;
    const int sfpos = (sp >> (LGW + G::LGR)) * G::FS + (((sp >> LGW) & (G::ROWS - 1)) + 1) * G::PW + (sp & (G::W - 1)) + 1;
    // 16 x 16: the two halo rows -- thread t owns halo pixel t & 31 (x = t & 15 of the row above (t & 16 == 0) or below) and the channels
    // (t >> 5) + 16 u, u < 2
    const int hx = threadIdx.x & 15, hbot = (threadIdx.x >> 4) & 1, hch = threadIdx.x >> 5;
    const int hy = hbot ? G::ROWS * rb + G::ROWS : G::ROWS * rb - 1;
    const bool hrow = G::TPS > 1 && hy >= 0 && hy < V && hx < V;      // (V <= W: inside the image)
    const bool slive = nf_fi_live<LGW>(sq, V);
    const int hfpos = (hbot ? G::ROWS + 1 : 0) * G::PW + hx + 1;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nchunks = (Ci + 31) >> 5, cps = (nchunks + (int)gridDim.z - 1) / (int)gridDim.z;
    const int ch0 = (int)blockIdx.z * cps, ch1 = min(nchunks, ch0 + cps);
    float tf[4], th[2], tw[18];
    auto frame_load = [&](int c0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cc = c0 + sch + 8 * u;
            const bool ok = sbs < B && cc < Ci && slive;
            const unsigned idx = INMODE == 0 ? (unsigned)((sbs * Ci + cc) * G::N + sq) : (unsigned)((sbs * Ch + (cc < Ch ? cc : cc - Ch)) * G::N + sq);
            const float v = in[ok ? idx : 0u];
            tf[u] = ok ? v : 0.f;
        }
        if (G::TPS > 1) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int cc = c0 + hch + 16 * u;
                const bool ok = hrow && cc < Ci;
                const int hq = hy * G::W + hx;
                const unsigned idx = INMODE == 0 ? (unsigned)(((int)b0 * Ci + cc) * G::N + hq) : (unsigned)(((int)b0 * Ch + (cc < Ch ? cc : cc - Ch)) * G::N + hq);
                const float v = in[ok ? idx : 0u];
                th[u] = ok ? v : 0.f;
            }
        }
    };
    if (ch0 < ch1) {
        frame_load(32 * ch0);
        nf_fi_w_load<TR>(tw, w, Ci, Co, o0, 32 * ch0);
    }
    for (int e = threadIdx.x; e < 32 * G::CS; e += NF_FI_THREADS) F[e] = 0.f;
    for (int ch = ch0; ch < ch1; ++ch) {
        const int c0 = 32 * ch;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float v = tf[u];
            if (INMODE == 1) v = nf_fi_elu_fast(c0 + sch + 8 * u < Ch ? v : -v);
            F[(sch + 8 * u) * G::CS + sfpos] = v;
        }
        if (G::TPS > 1) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float v = th[u];
                if (INMODE == 1) v = nf_fi_elu_fast(c0 + hch + 16 * u < Ch ? v : -v);
                F[(hch + 16 * u) * G::CS + hfpos] = v;
            }
        }
        nf_fi_w_store<TR>(Wl, tw);
        __syncthreads();
        if (ch + 1 < ch1) {
            frame_load(c0 + 32);
            nf_fi_w_load<TR>(tw, w, Ci, Co, o0, c0 + 32);
        }
        if (8 * kq < min(32, Ci - c0)) {                                   // this wave's 8-channel group of the chunk (wave-uniform)
            const float* wp = Wl + r32 * NF_FI_WS + 72 * kq + 36 * hs;
            const float* fp = F + (8 * kq + 4 * hs) * G::CS + fpos;
#pragma unroll
            for (int s2 = 0; s2 < 36; ++s2) {
                const int t = s2 % 9, dy = t / 3 - 1, dx = t % 3 - 1;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wp[s2], fp[(s2 / 9) * G::CS + dy * G::PW + dx], acc, 0, 0, 0);
            }
        }
    }
    __syncthreads();
    if (kq > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) RED[(((kq - 1) * 2 + pb) * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (kq == 0) {
        const int64_t b = b0 + (p >> (LGW + G::LGR));
        const int q = G::TPS == 1 ? (p & (G::N - 1)) : 64 * rb + p;
        float* slab = out + (int64_t)blockIdx.z * B * Co * G::N;
        const bool add_bias = bias != nullptr && blockIdx.z == 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = acc[r] + RED[((0 * 2 + pb) * 16 + r) * 64 + lane] + RED[((1 * 2 + pb) * 16 + r) * 64 + lane] +
                            RED[((2 * 2 + pb) * 16 + r) * 64 + lane];
            const int oc = o0 + nf_fi_cd_row(r, hs);
            if (b < B && oc < Co) slab[(b * Co + oc) * G::N + q] = v + (add_bias ? bias[oc] : 0.f);
        }
    }
}

// slab_w[blockIdx.x][t][o][c] = sum over this workgroup's tiles of  g[b][o][p] * act[b][c][p + off(t)],  slab_b[blockIdx.x][o] = sum g
// (tap-major slabs: whole 128-byte runs per (tap, o); nf_slab_sum folds them in its taps mode -- no atomics).  Workgroup (x, y, z) walks
// the pixel tiles x, x + gridDim.x, ... for its (32 output, 32 input) channel block.  The first form of this kernel gave every wave 32
// pixels and all nine tap tiles (144 accumulator registers) and met in nine LDS rounds over eight waves: 22 - 25 us per launch (33 with
// cold caches) at every map size.
// The waves are cut by TAP: wave (half ph, tap group tg) walks half of the tile's pixels for the taps
// {0,1,2} / {3,4} / {5,6} / {7,8} -- three or two accumulator tiles per wave instead of nine, and ONE meeting of the two halves in LDS
// instead of nine rounds over eight waves.  Tiles are 256 pixels on 16 x 16 maps and 64 pixels (1 / 4 whole samples) below, where
// 256-pixel tiles leave 4 - 16 workgroups per channel block at B = 64.
template <int LGW>
struct NfFiGeoW {
    static constexpr int W = 1 << LGW, N = W * W, PW = W + 2, TPX = N <= 64 ? 64 : 256, S = TPX / N, FS = PW * PW, CS = (S * FS) | 1,
                         GS = TPX + 1, CPT = NF_FI_THREADS / TPX, NU = 32 / CPT;     // staging: thread = (pixel t % TPX, channels t / TPX + CPT u)
};

// Up to NF_FLOWPP_IMG_WGRAD_MAX convolutions of ONE shape per launch: blockIdx.x = layer * n_slabs + slab.  (A launch of one layer is
// a single tile's latency chain on 64 .. 192 workgroups at B = 64; the weight gradients of a pass are independent of each other and
// only the optimizer waits for them: fused_flowpp_img.FlowppImgDefer runs them 16 layers per launch where the pass ends.)
struct NfFiWgMulti { nf_flowpp_img_wgrad_desc d[NF_FLOWPP_IMG_WGRAD_MAX]; };
template <int LGW, int INMODE>
__global__ void __launch_bounds__(NF_FI_THREADS) k_fi_conv_wgrad2(NfFiWgMulti m, int n_slabs, int64_t B, int Ci, int Co, int V) {
    using G = NfFiGeoW<LGW>;
    const int layer = blockIdx.x / n_slabs, slab = blockIdx.x - layer * n_slabs;
    const float* __restrict__ in = m.d[layer].in;
    const float* __restrict__ g = m.d[layer].g_out;
    float* __restrict__ slab_w = m.d[layer].slab_w;
    float* __restrict__ slab_b = m.d[layer].slab_b;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* F = smem;                       // [32][CS]
    float* Gt = F + 32 * G::CS;            // [32][GS]
    float* RED = Gt + 32 * G::GS;          // [4 tap groups][3][1024]: the second half's tiles
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, r32 = lane & 31, hs = lane >> 5;
    const int ph = wid & 1, tg = wid >> 1;
    const int t0 = tg == 0 ? 0 : 2 * tg + 1, nt = tg == 0 ? 3 : 2;         // taps t0 .. t0 + nt - 1
    const int o0 = 32 * blockIdx.y, c0 = 32 * blockIdx.z;
    const int64_t tiles = (B + G::S - 1) / G::S;
    const int Ch = Ci >> 1;
    // staging: this thread's pixel of the tile and its frame position
    const int sp = threadIdx.x % G::TPX, sch = threadIdx.x / G::TPX;
    const int ss = sp >> (2 * LGW), sq = sp & (G::N - 1);
    const int sfpos = ss * G::FS + (((sp >> LGW) & (G::W - 1)) + 1) * G::PW + (sp & (G::W - 1)) + 1;
    const bool slive = nf_fi_live<LGW>(sq, V);              // dead pixels: zero activation AND zero gradient (bias sums included)
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;
    for (int e = threadIdx.x; e < 32 * G::CS; e += NF_FI_THREADS) F[e] = 0.f;
    for (int64_t tile = slab; tile < tiles; tile += n_slabs) {
        const int bs = (int)(tile * G::S) + ss;
        float tf[G::NU], tgv[G::NU];
#pragma unroll
        for (int u = 0; u < G::NU; ++u) {
            const int cc = c0 + sch + G::CPT * u, o = o0 + sch + G::CPT * u;
            const bool okf = bs < B && cc < Ci && slive, okg = bs < B && o < Co && slive;
            const unsigned fi = INMODE == 0 ? (unsigned)((bs * Ci + cc) * G::N + sq) : (unsigned)((bs * Ch + (cc < Ch ? cc : cc - Ch)) * G::N + sq);
            const float vf = in[okf ? fi : 0u], vg = g[okg ? (unsigned)((bs * Co + o) * G::N + sq) : 0u];
            tf[u] = okf ? vf : 0.f;
            tgv[u] = okg ? vg : 0.f;
        }
        __syncthreads();                                     // the previous tile's readers are done (and the halo is zero)
#pragma unroll
        for (int u = 0; u < G::NU; ++u) {
            float v = tf[u];
            if (INMODE == 1) v = nf_fi_elu_fast(c0 + sch + G::CPT * u < Ch ? v : -v);
            F[(sch + G::CPT * u) * G::CS + sfpos] = v;
            Gt[(sch + G::CPT * u) * G::GS + sp] = tgv[u];
        }
        __syncthreads();
#pragma unroll 2
        for (int s2 = 0; s2 < G::TPX / 4; ++s2) {
            const int p = ph * (G::TPX / 2) + 2 * s2 + hs;
            const float a = Gt[r32 * G::GS + p];
            const float* fp = F + r32 * G::CS + (p >> (2 * LGW)) * G::FS + (((p >> LGW) & (G::W - 1)) + 1) * G::PW + (p & (G::W - 1)) + 1;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                if (t < nt) {                                // (wave-uniform)
                    const int tap = t0 + t;
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, fp[(tap / 3 - 1) * G::PW + (tap % 3) - 1], acc[t], 0, 0, 0);
                }
            }
        }
        if (slab_b != nullptr && blockIdx.z == 0 && threadIdx.x < 32)
            for (int p = 0; p < G::TPX; ++p) bsum += Gt[threadIdx.x * G::GS + p];
    }
    if (slab_b != nullptr && blockIdx.z == 0 && threadIdx.x < 32 && o0 + (int)threadIdx.x < Co)
        slab_b[(int64_t)slab * Co + o0 + threadIdx.x] = bsum;
    if (ph == 1) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) RED[((tg * 3 + t) * 16 + r) * 64 + lane] = acc[t][r];
    }
    __syncthreads();
    if (ph == 0) {
        float* sw = slab_w + (int64_t)slab * Co * Ci * 9;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            if (t < nt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = o0 + nf_fi_cd_row(r, hs), c = c0 + r32;
                    if (o < Co && c < Ci) sw[((int64_t)(t0 + t) * Co + o) * Ci + c] = acc[t][r] + RED[((tg * 3 + t) * 16 + r) * 64 + lane];
                }
            }
        }
    }
}

// g_x += elu'(x) * g_cat[:, :C] - elu'(-x) * g_cat[:, C:]   (the concat-ELU in front of the gated convolution)
__global__ void __launch_bounds__(NF_BLOCK) k_fi_celu_bwd(const float* __restrict__ x, const float* __restrict__ gcat,
                                                          float* __restrict__ gx, int64_t B, int CN) {
    const int64_t total = B * CN, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t b = e / CN, r = e - b * CN;
        const float v = x[e];
        gx[e] += nf_fi_elu_grad(v) * gcat[b * 2 * CN + r] - nf_fi_elu_grad(-v) * gcat[b * 2 * CN + CN + r];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// the middle of the conditioner, one sample per workgroup, thread = (head h, position j), 8 channels 8 h .. 8 h + 7 of position j
// ---------------------------------------------------------------------------------------------------------------
struct NfFiMid {
    const float *x, *a, *ln1g, *ln1b, *pos, *w1, *b1, *w2, *b2, *ln2g, *ln2b;
    float* out;
    const float* g_out;
    int g_slabs;
    int64_t g_slab_stride;
    float *g_x, *g_a, *g_ln1g, *g_ln1b, *g_pos, *g_w1, *g_b1, *g_w2, *g_b2, *g_ln2g, *g_ln2b;
    int V;               // side of the image inside the storage map (MK variants; the (32, V, V) parameters keep the reference's layout)
    int per_sample;      // the (32, V, V) parameter gradients (LayerNorm affines, position embedding) are WRITTEN per sample, (B, 32, V, V), for
};                       // nf_slab_sum to fold (64 workgroups x 10 240 same-address atomics were ~9 of the 66 us of an 8 x 8 launch at B = 64)

template <int NT>
__device__ __forceinline__ float nf_fi_block_sum_all(float v, float* scr) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (NT <= 64) return v;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scr[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) r += scr[i];
    return r;
}

__device__ __forceinline__ float nf_fi_dot8(const f32x4& a0, const f32x4& a1, const float* b) {
    return a0[0] * b[0] + a0[1] * b[1] + a0[2] * b[2] + a0[3] * b[3] + a1[0] * b[4] + a1[1] * b[5] + a1[2] * b[6] + a1[3] * b[7];
}

// channel-major plane [32][N] with a per-row rotation: conflict-free both for (fixed c, lanes over p) and (fixed p, lanes over c)
#define NF_FI_IDX(c, p) ((c) * N + (((p) + (c)) & (N - 1)))

// weight / bias gradient of a 1x1 convolution block:  gw[o][c] += sum_p GP[o][p] * T[c][p]  (o, c < 32),  gb[o] += sum_p GP[o][p],
// GP and T rotated planes.  ONE wave on the matrix cores, K = the N positions (the VALU form -- a thread per (o, c) pair walking N
// positions with two LDS reads per product -- was 60 % of the backward at 4 x 4, where the workgroup is a single wave).
template <int N>
__device__ __forceinline__ void nf_fi_pair_mfma(const float* GP, const float* T, float* __restrict__ gw, float* __restrict__ gb) {
    const int lane = threadIdx.x & 63, r32 = lane & 31, hs = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bs = 0.f;
#pragma unroll 4
    for (int s = 0; s < N / 2; ++s) {
        const int p = 2 * s + hs;
        const float a = GP[NF_FI_IDX(r32, p)];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, T[NF_FI_IDX(r32, p)], acc, 0, 0, 0);
        bs += a;
    }
    bs += __shfl_xor(bs, 32, 64);
#pragma unroll
    for (int r = 0; r < 16; ++r) atomicAdd(gw + nf_fi_cd_row(r, hs) * 32 + r32, acc[r]);
    if (hs == 0) atomicAdd(gb + r32, bs);
}

// MK: the image is V x V inside the W x W storage map (V < W).  Thread j < V * V works on image pixel (j / V, j % V) -- the positions are
// COMPACT in the planes, the sweeps over the keys run to V * V -- and the remaining threads carry zeros through every sum and plane.
template <int N, bool BWD, bool MK>
__global__ void __launch_bounds__(4 * N) k_fi_mid(NfFiMid m) {
    constexpr int NT = 4 * N, PL = 32 * N;
    constexpr int WS = N == 256 ? 16 : (N == 64 ? 8 : 4);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* PA = smem;
    float* PB = PA + PL;
    float* PC = PB + PL;
    float* PD = PC + PL;
    float* W1s = PD + PL;          // conv1 weight transposed [32][96]
    float* W2s = W1s + 3072;       // conv2 weight transposed [32][64]; dead after the conv2 backward: CJ[4 N] | DL[4 N] alias it (4 N <= 1024)
    float* Bs = W2s + 2048;        // b1[96] | b2[64]
    float* scr = Bs + 160;         // [16]
    float* CJ = W2s;
    float* DL = W2s + NT;
    const int tid = threadIdx.x, h = tid / N, j = tid & (N - 1);
    const int64_t b = blockIdx.x;
    const int NV = MK ? m.V * m.V : N;                   // live positions = stride of the (32, V, V) parameters
    const bool ok = !MK || j < NV;
    int spx = j;                                         // this thread's pixel of the storage map
    if (MK) {
        const int jy = ok ? j / m.V : 0;
        spx = jy * WS + (ok ? j - jy * m.V : 0);
    }
    const float invn = 1.f / (float)(32 * NV);
    const float scale = 0.35355339059327373f;            // 1 / sqrt(D), D = 8 (flows/modules.py:571)

    for (int e = tid; e < 3072; e += NT) W1s[(e & 31) * 96 + (e >> 5)] = m.w1[e];      // transposed: [c][o], o contiguous
    for (int e = tid; e < 2048; e += NT) W2s[(e & 31) * 64 + (e >> 5)] = m.w2[e];
    for (int e = tid; e < 160; e += NT) Bs[e] = e < 96 ? m.b1[e] : m.b2[e - 96];

    const int64_t base = (b * 32 + 8 * h) * N + spx;       // this thread's elements: base + d * N
    const int pbase = 8 * h * NV + (ok ? j : 0);           // the same inside a (32, H, W) parameter: pbase + d * NV
    float m1, r1;
    // u = x + elu(a) * sigmoid(elu(-a))  (GatedConv2d, flows/modules.py:519-538), LayerNorm 1
    auto ln1 = [&](float* xh1, float* x2) {
        float u[8], s = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const float av = m.a[base + d * N], xv = m.x[base + d * N];
            u[d] = ok ? xv + nf_fi_elu(av) * nf_fi_sigmoid(nf_fi_elu(-av)) : 0.f;
            s += u[d];
        }
        m1 = nf_fi_block_sum_all<NT>(s, scr) * invn;
        float s2 = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) s2 += ok ? (u[d] - m1) * (u[d] - m1) : 0.f;
        r1 = 1.f / sqrtf(nf_fi_block_sum_all<NT>(s2, scr) * invn + NF_FI_LNEPS);
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            xh1[d] = ok ? (u[d] - m1) * r1 : 0.f;
            x2[d] = ok ? xh1[d] * m.ln1g[pbase + d * NV] + m.ln1b[pbase + d * NV] : 0.f;
        }
    };
    float x2[8], v[8], k[8], q[8];
    {
        float xh1[8];
        ln1(xh1, x2);
    }
    // tokens = x2 + pos_emb -> PA ; proj = conv1 (1x1, 32 -> 96): rows 8 h + d of the three groups ("V", "K", "Q" in the reference's naming)
#pragma unroll
    for (int d = 0; d < 8; ++d) PA[NF_FI_IDX(8 * h + d, j)] = ok ? x2[d] + m.pos[pbase + d * NV] : 0.f;
    __syncthreads();
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        v[d] = Bs[8 * h + d];
        k[d] = Bs[32 + 8 * h + d];
        q[d] = Bs[64 + 8 * h + d];
    }
#pragma unroll 2
    for (int c = 0; c < 32; ++c) {
        const float tc = PA[NF_FI_IDX(c, j)];
        const f32x4* wr = reinterpret_cast<const f32x4*>(W1s + c * 96 + 8 * h);
        const f32x4 a0 = wr[0], a1 = wr[1], b0 = wr[8], b1 = wr[9], c0 = wr[16], c1 = wr[17];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] += a0[e] * tc; v[4 + e] += a1[e] * tc;
            k[e] += b0[e] * tc; k[4 + e] += b1[e] * tc;
            q[e] += c0[e] * tc; q[4 + e] += c1[e] * tc;
        }
    }
    // attention: P[i][j] = softmax_i(V_i . K_j / sqrt(D)),  mixed_j = sum_i Q_i P[i][j]   (flows/modules.py:566-574)
    f32x4* PB4 = reinterpret_cast<f32x4*>(PB);
    f32x4* PC4 = reinterpret_cast<f32x4*>(PC);
    PB4[(h * N + j) * 2] = (f32x4){v[0], v[1], v[2], v[3]};
    PB4[(h * N + j) * 2 + 1] = (f32x4){v[4], v[5], v[6], v[7]};
    PC4[(h * N + j) * 2] = (f32x4){q[0], q[1], q[2], q[3]};
    PC4[(h * N + j) * 2 + 1] = (f32x4){q[4], q[5], q[6], q[7]};
    __syncthreads();
    float mx = -INFINITY, l = 0.f, mix[8];
#pragma unroll 2
    for (int i = 0; i < NV; ++i) mx = fmaxf(mx, nf_fi_dot8(PB4[(h * N + i) * 2], PB4[(h * N + i) * 2 + 1], k) * scale);
#pragma unroll
    for (int d = 0; d < 8; ++d) mix[d] = 0.f;
#pragma unroll 2
    for (int i = 0; i < NV; ++i) {
        const float p = __expf(nf_fi_dot8(PB4[(h * N + i) * 2], PB4[(h * N + i) * 2 + 1], k) * scale - mx);
        const f32x4 q0 = PC4[(h * N + i) * 2], q1 = PC4[(h * N + i) * 2 + 1];
        l += p;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mix[e] += p * q0[e];
            mix[4 + e] += p * q1[e];
        }
    }
    {
        const float inv = 1.f / l;
#pragma unroll
        for (int d = 0; d < 8; ++d) mix[d] *= inv;
    }
    __syncthreads();                                     // every thread is done with the tokens and the V / Q planes
#pragma unroll
    for (int d = 0; d < 8; ++d) PA[NF_FI_IDX(8 * h + d, j)] = mix[d];
    __syncthreads();
    // conv2 (1x1, 32 -> 64): y = rows 8 h + d, gate = rows 32 + 8 h + d ;  x3 = x2 + y * sigmoid(gate)
    float y[8], sg[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        y[d] = Bs[96 + 8 * h + d];
        sg[d] = Bs[96 + 32 + 8 * h + d];
    }
#pragma unroll 2
    for (int c = 0; c < 32; ++c) {
        const float tc = PA[NF_FI_IDX(c, j)];
        const f32x4* wr = reinterpret_cast<const f32x4*>(W2s + c * 64 + 8 * h);
        const f32x4 a0 = wr[0], a1 = wr[1], b0 = wr[8], b1 = wr[9];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            y[e] += a0[e] * tc; y[4 + e] += a1[e] * tc;
            sg[e] += b0[e] * tc; sg[4 + e] += b1[e] * tc;
        }
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) sg[d] = nf_fi_sigmoid(sg[d]);
    float xh2[8], m2, r2;
    {
        float x3[8], s = 0.f, s2 = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            x3[d] = ok ? x2[d] + y[d] * sg[d] : 0.f;
            s += x3[d];
        }
        m2 = nf_fi_block_sum_all<NT>(s, scr) * invn;
#pragma unroll
        for (int d = 0; d < 8; ++d) s2 += ok ? (x3[d] - m2) * (x3[d] - m2) : 0.f;
        r2 = 1.f / sqrtf(nf_fi_block_sum_all<NT>(s2, scr) * invn + NF_FI_LNEPS);
#pragma unroll
        for (int d = 0; d < 8; ++d) xh2[d] = ok ? (x3[d] - m2) * r2 : 0.f;
    }
    if (!BWD) {
        if (ok) {
#pragma unroll
            for (int d = 0; d < 8; ++d) m.out[base + d * N] = xh2[d] * m.ln2g[pbase + d * NV] + m.ln2b[pbase + d * NV];
        }
        return;
    }

    // ================================================= backward =================================================
    // parameter gradients are added straight into their buffers, one thread (LayerNorm / position parameters) or one wave (the 1 x 1
    // convolutions' blocks) of the workgroup per address: deterministic mode takes the samples' workgroups one after the other
    NF_DET_ENTER_ALL(nf_fpi);
    float g3[8];                                         // gradient of x3 (= of the attention block's residual input and of y * sg)
    {
        float gh[8], s1 = 0.f, s2 = 0.f;
        // the incoming gradient as the sum of its K-split slabs: slab-outer, so that the eight loads of a slab (and, unrolled, of four
        // slabs) are in flight together -- element-outer it was up to 8 x 42 dependent round trips to memory the previous kernel just wrote
        float g4v[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) g4v[d] = m.g_out[base + d * N];
#pragma unroll 4
        for (int z = 1; z < m.g_slabs; ++z) {
#pragma unroll
            for (int d = 0; d < 8; ++d) g4v[d] += m.g_out[z * m.g_slab_stride + base + d * N];
        }
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const float g4 = ok ? g4v[d] : 0.f;
            if (ok) {
                if (m.per_sample) {
                    m.g_ln2g[b * 32 * NV + pbase + d * NV] = g4 * xh2[d];
                    m.g_ln2b[b * 32 * NV + pbase + d * NV] = g4;
                } else {
                    atomicAdd(m.g_ln2g + pbase + d * NV, g4 * xh2[d]);
                    atomicAdd(m.g_ln2b + pbase + d * NV, g4);
                }
            }
            gh[d] = g4 * m.ln2g[pbase + d * NV];
            s1 += gh[d];
            s2 += gh[d] * xh2[d];
        }
        const float S1 = nf_fi_block_sum_all<NT>(s1, scr) * invn, S2 = nf_fi_block_sum_all<NT>(s2, scr) * invn;
#pragma unroll
        for (int d = 0; d < 8; ++d) g3[d] = ok ? r2 * (gh[d] - S1 - xh2[d] * S2) : 0.f;
    }
    // conv2 backward in two rounds through PB (y rows, then gate rows); PA still holds `mixed`
    float gm[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) gm[d] = 0.f;
#pragma unroll 1
    for (int part = 0; part < 2; ++part) {
        __syncthreads();
#pragma unroll
        for (int d = 0; d < 8; ++d)
            PB[NF_FI_IDX(8 * h + d, j)] = part == 0 ? g3[d] * sg[d] : g3[d] * y[d] * sg[d] * (1.f - sg[d]);
        __syncthreads();
#pragma unroll 1
        for (int o4 = 0; o4 < 8; ++o4) {
            const float e0 = PB[NF_FI_IDX(4 * o4, j)], e1 = PB[NF_FI_IDX(4 * o4 + 1, j)], e2 = PB[NF_FI_IDX(4 * o4 + 2, j)],
                        e3 = PB[NF_FI_IDX(4 * o4 + 3, j)];
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(W2s + (8 * h + d) * 64 + 32 * part + 4 * o4);
                gm[d] += w4[0] * e0 + w4[1] * e1 + w4[2] * e2 + w4[3] * e3;
            }
        }
        if ((tid >> 6) == part % (NT / 64)) nf_fi_pair_mfma<N>(PB, PA, m.g_w2 + 32 * part * 32, m.g_b2 + 32 * part);
    }
    // attention backward.  delta_j = sum_i P[i][j] gP[i][j] = mixed_j . g_mixed_j ;  g_s[i][j] = P[i][j] (gP[i][j] - delta_j)
    __syncthreads();                                     // PA / PB / W2s are free
    float delta = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) delta += mix[d] * gm[d];
    const float cj = mx + __logf(l);                     // P[i][j] = exp(s[i][j] - cj)   (v_exp / v_log: ~1e-7 relative on the probabilities)
    f32x4* PA4 = reinterpret_cast<f32x4*>(PA);
    f32x4* PD4 = reinterpret_cast<f32x4*>(PD);
    PA4[(h * N + j) * 2] = (f32x4){v[0], v[1], v[2], v[3]};
    PA4[(h * N + j) * 2 + 1] = (f32x4){v[4], v[5], v[6], v[7]};
    PB4[(h * N + j) * 2] = (f32x4){q[0], q[1], q[2], q[3]};
    PB4[(h * N + j) * 2 + 1] = (f32x4){q[4], q[5], q[6], q[7]};
    PC4[(h * N + j) * 2] = (f32x4){k[0], k[1], k[2], k[3]};
    PC4[(h * N + j) * 2 + 1] = (f32x4){k[4], k[5], k[6], k[7]};
    PD4[(h * N + j) * 2] = (f32x4){gm[0], gm[1], gm[2], gm[3]};
    PD4[(h * N + j) * 2 + 1] = (f32x4){gm[4], gm[5], gm[6], gm[7]};
    CJ[h * N + j] = cj;
    DL[h * N + j] = delta;
    __syncthreads();
    float gk[8], gv[8], gq[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) gk[d] = gv[d] = gq[d] = 0.f;
#pragma unroll 2
    for (int i = 0; i < NV; ++i) {                       // this thread as the column j: gradient of K_j
        const f32x4 v0 = PA4[(h * N + i) * 2], v1 = PA4[(h * N + i) * 2 + 1];
        const float p = __expf(nf_fi_dot8(v0, v1, k) * scale - cj);
        const float gs = p * (nf_fi_dot8(PB4[(h * N + i) * 2], PB4[(h * N + i) * 2 + 1], gm) - delta);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gk[e] += gs * v0[e];
            gk[4 + e] += gs * v1[e];
        }
    }
#pragma unroll 2
    for (int jj = 0; jj < NV; ++jj) {                    // this thread as the row i = j: gradients of V_i and Q_i
        const f32x4 k0 = PC4[(h * N + jj) * 2], k1 = PC4[(h * N + jj) * 2 + 1];
        const f32x4 g0 = PD4[(h * N + jj) * 2], g1 = PD4[(h * N + jj) * 2 + 1];
        const float p = __expf(nf_fi_dot8(k0, k1, v) * scale - CJ[h * N + jj]);
        const float gs = p * (nf_fi_dot8(g0, g1, q) - DL[h * N + jj]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gv[e] += gs * k0[e];
            gv[4 + e] += gs * k1[e];
            gq[e] += p * g0[e];
            gq[4 + e] += p * g1[e];
        }
    }
    // conv1 backward: the three gradient groups -> PA / PB / PC, the tokens again -> PD
    float xh1[8];
    ln1(xh1, x2);                                        // (recomputed: x2 / xh1 were not kept over the attention; ends in barriers)
    __syncthreads();
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        PA[NF_FI_IDX(8 * h + d, j)] = ok ? gv[d] * scale : 0.f;
        PB[NF_FI_IDX(8 * h + d, j)] = ok ? gk[d] * scale : 0.f;
        PC[NF_FI_IDX(8 * h + d, j)] = ok ? gq[d] : 0.f;
        PD[NF_FI_IDX(8 * h + d, j)] = ok ? x2[d] + m.pos[pbase + d * NV] : 0.f;
    }
    __syncthreads();
    float gt[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) gt[d] = 0.f;
#pragma unroll 1
    for (int part = 0; part < 3; ++part) {
        const float* P = part == 0 ? PA : (part == 1 ? PB : PC);
#pragma unroll 1
        for (int o4 = 0; o4 < 8; ++o4) {
            const float e0 = P[NF_FI_IDX(4 * o4, j)], e1 = P[NF_FI_IDX(4 * o4 + 1, j)], e2 = P[NF_FI_IDX(4 * o4 + 2, j)],
                        e3 = P[NF_FI_IDX(4 * o4 + 3, j)];
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(W1s + (8 * h + d) * 96 + 32 * part + 4 * o4);
                gt[d] += w4[0] * e0 + w4[1] * e1 + w4[2] * e2 + w4[3] * e3;
            }
        }
        if ((tid >> 6) == (part + 2) % (NT / 64)) nf_fi_pair_mfma<N>(P, PD, m.g_w1 + 32 * part * 32, m.g_b1 + 32 * part);
    }
    // LayerNorm 1 and the gate
    {
        float gh[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const float g2_ = g3[d] + gt[d];
            if (ok) {
                if (m.per_sample) {
                    m.g_pos[b * 32 * NV + pbase + d * NV] = gt[d];
                    m.g_ln1g[b * 32 * NV + pbase + d * NV] = g2_ * xh1[d];
                    m.g_ln1b[b * 32 * NV + pbase + d * NV] = g2_;
                } else {
                    atomicAdd(m.g_pos + pbase + d * NV, gt[d]);
                    atomicAdd(m.g_ln1g + pbase + d * NV, g2_ * xh1[d]);
                    atomicAdd(m.g_ln1b + pbase + d * NV, g2_);
                }
            }
            gh[d] = g2_ * m.ln1g[pbase + d * NV];
            s1 += gh[d];
            s2 += gh[d] * xh1[d];
        }
        const float S1 = nf_fi_block_sum_all<NT>(s1, scr) * invn, S2 = nf_fi_block_sum_all<NT>(s2, scr) * invn;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const float gu = r1 * (gh[d] - S1 - xh1[d] * S2);
            const float av = m.a[base + d * N];
            const float e1 = nf_fi_elu(av), s2_ = nf_fi_sigmoid(nf_fi_elu(-av));
            if (ok) {
                m.g_x[base + d * N] = gu;
                m.g_a[base + d * N] = gu * (nf_fi_elu_grad(av) * s2_ - e1 * s2_ * (1.f - s2_) * nf_fi_elu_grad(-av));
            }
        }
    }
    NF_DET_LEAVE_ALL(nf_fpi);
}

// ---------------------------------------------------------------------------------------------------------------
// dynamic LDS above the 64 KB default needs a per-kernel opt-in, once per kernel (keyed by the function's address)
template <typename K>
static inline int nf_fi_optin(K kernel, size_t lds) {
    static std::mutex mu;
    static std::unordered_set<const void*> done;
    if (lds > 160 * 1024) return NF_E_BADARG;
    if (lds <= 64 * 1024) return 0;
    const void* key = reinterpret_cast<const void*>(kernel);
    std::lock_guard<std::mutex> lock(mu);
    if (done.find(key) == done.end()) {
        hipError_t e = hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        done.insert(key);
    }
    return 0;
}

// (H, W) of every entry point = the IMAGE (square, side 1 .. 16); the activation tensors are (B, C, S, S) with S = the storage side:
// the next power of two, at least 4 (nf_flowpp_img_storage).  The (32, H, W) parameters keep the reference's layout.
static inline int nf_fi_lgw(int H, int W) {
    if (H != W || W < 1 || W > 16) return -1;
    return W > 8 ? 4 : (W > 4 ? 3 : 2);
}

extern "C" int nf_flowpp_img_storage(int H, int W) {
    const int lg = nf_fi_lgw(H, W);
    return lg < 0 ? 0 : 1 << lg;
}

extern "C" int nf_flowpp_img_usable(int64_t B, int Ci, int Co, int H, int W) {
    if (B < 1 || Ci < 1 || Co < 1 || nf_fi_lgw(H, W) < 0) return 0;
    const int64_t SN = (int64_t)nf_flowpp_img_storage(H, W) * nf_flowpp_img_storage(H, W);
    const int64_t tiles = (B * SN + 255) / 256;
    if (tiles > 65535 * 32 || (Co + 31) / 32 > 65535 || B * (int64_t)(Ci > Co ? Ci : Co) * SN >= ((int64_t)1 << 31)) return 0;
    return 1;
}

template <int LGW>
static int nf_fi_conv_launch(const float* in, const float* w, const float* bias, float* out, int64_t B, int Ci, int Co, int V, int in_mode,
                             int transposed, int ksplit, hipStream_t st) {
    using G = NfFiGeo<LGW>;
    int rc;
    {
        // 64-pixel tiles while 256-pixel ones hold at most 128 workgroups (measured at B = 64: 129 best, wider launches gain nothing)
        const int t64 = 129;                            // the workgroup count below which the small tiles run
        const int64_t wgs = ((B + G::S - 1) / G::S) * ((Co + 31) / 32) * ksplit;
        if (wgs < t64) {
            using G6 = NfFiGeo64<LGW>;
            const size_t lds6 = (size_t)(32 * G6::CS + 32 * NF_FI_WS + 3 * 2 * 16 * 64) * sizeof(float);
            const dim3 grid6((unsigned)(((B + G6::S - 1) / G6::S) * G6::TPS), (unsigned)((Co + 31) / 32), (unsigned)ksplit);
#define NF_FI_GO6(M_, T_)                                                                                             \
    do {                                                                                                              \
        if ((rc = nf_fi_optin(k_fi_conv64<LGW, M_, T_>, lds6)) != 0) return rc;                                       \
        hipLaunchKernelGGL((k_fi_conv64<LGW, M_, T_>), grid6, dim3(NF_FI_THREADS), lds6, st, in, w, bias, out, B, Ci, Co, V); \
    } while (0)
            if (transposed) NF_FI_GO6(0, true);
            else if (in_mode == 1) NF_FI_GO6(1, false);
            else NF_FI_GO6(0, false);
#undef NF_FI_GO6
            NF_CHECK_LAUNCH();
            return 0;
        }
    }
    const size_t lds = (size_t)(32 * G::CS + 32 * NF_FI_WS) * sizeof(float);
    const dim3 grid((unsigned)((B + G::S - 1) / G::S), (unsigned)((Co + 31) / 32), (unsigned)ksplit);
#define NF_FI_GO(M_, T_)                                                                                            \
    do {                                                                                                            \
        if ((rc = nf_fi_optin(k_fi_conv<LGW, M_, T_>, lds)) != 0) return rc;                                        \
        hipLaunchKernelGGL((k_fi_conv<LGW, M_, T_>), grid, dim3(NF_FI_THREADS), lds, st, in, w, bias, out, B, Ci, Co, V); \
    } while (0)
    if (transposed) NF_FI_GO(0, true);
    else if (in_mode == 1) NF_FI_GO(1, false);
    else NF_FI_GO(0, false);
#undef NF_FI_GO
    NF_CHECK_LAUNCH();
    return 0;
}

// K slabs a launch of nf_flowpp_img_conv may be cut into: enough workgroups for the chip, never more than the 32-channel chunks
extern "C" int nf_flowpp_img_conv_ksplit(int64_t B, int Ci, int Co, int H, int W) {
    if (!nf_flowpp_img_usable(B, Ci, Co, H, W)) return 0;
    const int S = nf_flowpp_img_storage(H, W);
    const int64_t wgs = ((B * S * S + 255) / 256) * ((Co + 31) / 32);
    const int nchunks = (Ci + 31) / 32;
    int64_t k = (256 + wgs - 1) / wgs;
    if (k > nchunks) k = nchunks;
    if (k > NF_FLOWPP_IMG_MAX_KSPLIT) k = NF_FLOWPP_IMG_MAX_KSPLIT;
    return (int)(k < 1 ? 1 : k);
}

extern "C" int nf_flowpp_img_conv(const float* in, const float* weight, const float* bias, float* out, int64_t B, int Ci, int Co,
                                  int H, int W, int in_mode, int transposed, int ksplit, nf_stream_t stream) {
    if (in == nullptr || weight == nullptr || out == nullptr || !nf_flowpp_img_usable(B, Ci, Co, H, W)) return NF_E_BADARG;
    if (in_mode < 0 || in_mode > 1 || (in_mode == 1 && ((Ci & 1) || transposed))) return NF_E_BADARG;
    if (ksplit < 1 || ksplit > NF_FLOWPP_IMG_MAX_KSPLIT || ksplit > (Ci + 31) / 32) return NF_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    switch (nf_fi_lgw(H, W)) {
        case 4: return nf_fi_conv_launch<4>(in, weight, bias, out, B, Ci, Co, W, in_mode, transposed, ksplit, st);
        case 3: return nf_fi_conv_launch<3>(in, weight, bias, out, B, Ci, Co, W, in_mode, transposed, ksplit, st);
        default: return nf_fi_conv_launch<2>(in, weight, bias, out, B, Ci, Co, W, in_mode, transposed, ksplit, st);
    }
}

template <int LGW>
static int nf_fi_wgrad_launch(const NfFiWgMulti& m, int n, int n_slabs, int64_t B, int Ci, int Co, int V, int in_mode, hipStream_t st) {
    using G = NfFiGeoW<LGW>;
    const size_t lds = (size_t)(32 * G::CS + 32 * G::GS + 4 * 3 * 1024) * sizeof(float);
    const dim3 grid((unsigned)(n_slabs * n), (unsigned)((Co + 31) / 32), (unsigned)((Ci + 31) / 32));
    int rc;
    if (in_mode == 1) {
        if ((rc = nf_fi_optin(k_fi_conv_wgrad2<LGW, 1>, lds)) != 0) return rc;
        hipLaunchKernelGGL((k_fi_conv_wgrad2<LGW, 1>), grid, dim3(NF_FI_THREADS), lds, st, m, n_slabs, B, Ci, Co, V);
    } else {
        if ((rc = nf_fi_optin(k_fi_conv_wgrad2<LGW, 0>, lds)) != 0) return rc;
        hipLaunchKernelGGL((k_fi_conv_wgrad2<LGW, 0>), grid, dim3(NF_FI_THREADS), lds, st, m, n_slabs, B, Ci, Co, V);
    }
    NF_CHECK_LAUNCH();
    return 0;
}

// slabs of an nf_flowpp_img_conv_wgrad launch: about one workgroup per compute unit over the (output, input) channel blocks, never more
// than the pixel tiles
extern "C" int nf_flowpp_img_wgrad_slabs(int64_t B, int Ci, int Co, int H, int W) {
    if (!nf_flowpp_img_usable(B, Ci, Co, H, W)) return 0;
    const int SN = nf_flowpp_img_storage(H, W) * nf_flowpp_img_storage(H, W);
    const int tpx = SN <= 64 ? 64 : 256;                     // pixels per tile of k_fi_conv_wgrad2
    const int64_t tiles = (B * SN + tpx - 1) / tpx, blocks = (int64_t)((Co + 31) / 32) * ((Ci + 31) / 32);
    int64_t k = (256 + blocks - 1) / blocks;                 // (512 / 1024 workgroups measured no better)
    if (k > tiles) k = tiles;
    if (k > NF_FLOWPP_IMG_MAX_SLABS) k = NF_FLOWPP_IMG_MAX_SLABS;
    return (int)(k < 1 ? 1 : k);
}

extern "C" int nf_flowpp_img_conv_wgrad_multi(const nf_flowpp_img_wgrad_desc* descs, int n, int n_slabs, int64_t B, int Ci, int Co, int H,
                                              int W, int in_mode, nf_stream_t stream) {
    if (descs == nullptr || n < 1 || n > NF_FLOWPP_IMG_WGRAD_MAX || !nf_flowpp_img_usable(B, Ci, Co, H, W)) return NF_E_BADARG;
    if (in_mode < 0 || in_mode > 1 || (in_mode == 1 && (Ci & 1)) || n_slabs < 1 || n_slabs > NF_FLOWPP_IMG_MAX_SLABS) return NF_E_BADARG;
    NfFiWgMulti m{};
    for (int i = 0; i < n; ++i) {
        if (descs[i].in == nullptr || descs[i].g_out == nullptr || descs[i].slab_w == nullptr) return NF_E_BADARG;
        m.d[i] = descs[i];
    }
    hipStream_t st = (hipStream_t)stream;
    switch (nf_fi_lgw(H, W)) {
        case 4: return nf_fi_wgrad_launch<4>(m, n, n_slabs, B, Ci, Co, W, in_mode, st);
        case 3: return nf_fi_wgrad_launch<3>(m, n, n_slabs, B, Ci, Co, W, in_mode, st);
        default: return nf_fi_wgrad_launch<2>(m, n, n_slabs, B, Ci, Co, W, in_mode, st);
    }
}

extern "C" int nf_flowpp_img_conv_wgrad(const float* in, const float* g_out, float* slab_w, float* slab_b, int n_slabs, int64_t B,
                                        int Ci, int Co, int H, int W, int in_mode, nf_stream_t stream) {
    const nf_flowpp_img_wgrad_desc d = {in, g_out, slab_w, slab_b};
    return nf_flowpp_img_conv_wgrad_multi(&d, 1, n_slabs, B, Ci, Co, H, W, in_mode, stream);
}

extern "C" int nf_flowpp_img_celu_bwd(const float* x, const float* g_cat, float* g_x, int64_t B, int C, int H, int W,
                                      nf_stream_t stream) {
    if (x == nullptr || g_cat == nullptr || g_x == nullptr || B < 0 || C < 1 || H < 1 || W < 1) return NF_E_BADARG;
    if (B == 0) return 0;
    const int CN = C * H * W;
    hipLaunchKernelGGL(k_fi_celu_bwd, dim3(nf_grid_for(B * CN)), dim3(NF_BLOCK), 0, (hipStream_t)stream, x, g_cat, g_x, B, CN);
    NF_CHECK_LAUNCH();
    return 0;
}

template <int N, bool BWD>
static int nf_fi_mid_launch(NfFiMid m, int V, int64_t B, hipStream_t st) {
    const size_t lds = (size_t)(4 * 32 * N + 3072 + 2048 + 160 + 16) * sizeof(float);
    int rc;
    m.V = V;
    if (V * V == N) {
        if ((rc = nf_fi_optin(k_fi_mid<N, BWD, false>, lds)) != 0) return rc;
        hipLaunchKernelGGL((k_fi_mid<N, BWD, false>), dim3((unsigned)B), dim3(4 * N), lds, st, m);
    } else {                                                 // the image inside a larger storage map
        if ((rc = nf_fi_optin(k_fi_mid<N, BWD, true>, lds)) != 0) return rc;
        hipLaunchKernelGGL((k_fi_mid<N, BWD, true>), dim3((unsigned)B), dim3(4 * N), lds, st, m);
    }
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_flowpp_img_mid_fwd(const float* x, const float* a, const float* ln1_g, const float* ln1_b, const float* pos,
                                     const float* conv1_w, const float* conv1_b, const float* conv2_w, const float* conv2_b,
                                     const float* ln2_g, const float* ln2_b, float* out, int64_t B, int H, int W,
                                     nf_stream_t stream) {
    if (x == nullptr || a == nullptr || ln1_g == nullptr || ln1_b == nullptr || pos == nullptr || conv1_w == nullptr ||
        conv1_b == nullptr || conv2_w == nullptr || conv2_b == nullptr || ln2_g == nullptr || ln2_b == nullptr || out == nullptr)
        return NF_E_BADARG;
    if (!nf_flowpp_img_usable(B, 32, 32, H, W) || B > 0x7fffffff) return NF_E_BADARG;
    NfFiMid m = {x, a, ln1_g, ln1_b, pos, conv1_w, conv1_b, conv2_w, conv2_b, ln2_g, ln2_b, out, nullptr, 1, 0, nullptr, nullptr,
                 nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipStream_t st = (hipStream_t)stream;
    switch (nf_fi_lgw(H, W)) {
        case 4: return nf_fi_mid_launch<256, false>(m, W, B, st);
        case 3: return nf_fi_mid_launch<64, false>(m, W, B, st);
        default: return nf_fi_mid_launch<16, false>(m, W, B, st);
    }
}

extern "C" int nf_flowpp_img_mid_bwd(const float* x, const float* a, const float* ln1_g, const float* ln1_b, const float* pos,
                                     const float* conv1_w, const float* conv1_b, const float* conv2_w, const float* conv2_b,
                                     const float* ln2_g, const float* ln2_b, const float* g_out, float* g_x, float* g_a,
                                     float* g_ln1_g, float* g_ln1_b, float* g_pos, float* g_conv1_w, float* g_conv1_b,
                                     float* g_conv2_w, float* g_conv2_b, float* g_ln2_g, float* g_ln2_b, int per_sample, int64_t B, int H,
                                     int W, int g_out_slabs, nf_stream_t stream) {
    if (x == nullptr || a == nullptr || ln1_g == nullptr || ln1_b == nullptr || pos == nullptr || conv1_w == nullptr ||
        conv1_b == nullptr || conv2_w == nullptr || conv2_b == nullptr || ln2_g == nullptr || ln2_b == nullptr || g_out == nullptr ||
        g_x == nullptr || g_a == nullptr || g_ln1_g == nullptr || g_ln1_b == nullptr || g_pos == nullptr || g_conv1_w == nullptr ||
        g_conv1_b == nullptr || g_conv2_w == nullptr || g_conv2_b == nullptr || g_ln2_g == nullptr || g_ln2_b == nullptr)
        return NF_E_BADARG;
    if (!nf_flowpp_img_usable(B, 32, 32, H, W) || B > 0x7fffffff || g_out_slabs < 1 || g_out_slabs > NF_FLOWPP_IMG_MAX_KSPLIT) return NF_E_BADARG;
    NfFiMid m = {x, a, ln1_g, ln1_b, pos, conv1_w, conv1_b, conv2_w, conv2_b, ln2_g, ln2_b, nullptr, g_out, g_out_slabs, B * 32 * (int64_t)nf_flowpp_img_storage(H, W) * nf_flowpp_img_storage(H, W), g_x, g_a, g_ln1_g,
                 g_ln1_b, g_pos, g_conv1_w, g_conv1_b, g_conv2_w, g_conv2_b, g_ln2_g, g_ln2_b};
    m.per_sample = per_sample ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    switch (nf_fi_lgw(H, W)) {
        case 4: return nf_fi_mid_launch<256, true>(m, W, B, st);
        case 3: return nf_fi_mid_launch<64, true>(m, W, B, st);
        default: return nf_fi_mid_launch<16, true>(m, W, B, st);
    }
}
