// Shared tile machinery of the fused convolution kernels (conv_bn.hip: one launch per layer; conv_chain.hip: the whole
// conditioner in one persistent launch): tile geometry, frame decoding, weight staging, the MFMA K loop and the K-split exchange.
#pragma once
#include "nf_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define NF_CV_WAVES 16               // four pixel blocks x four K quarters (3x3) / output-block groups (1x1)
#define NF_CV_THREADS (NF_CV_WAVES * NF_WAVE)
#define NF_CV_PX 128                 // pixels per tile
#define NF_CV_MAX_FRAME 384          // frame positions per channel (all segments of a tile)
#define NF_CV_FJ (NF_CV_MAX_FRAME / NF_WAVE)
#define NF_CV_CU (32 / NF_CV_WAVES)  // channels of a 32-channel chunk a wave stages
#define NF_CV_MAX_I 96
#define NF_CV_MAX_O 192
#define NF_CV_WS 33                  // row stride of the LDS weight tiles of 32 columns (odd: conflict-free both ways)
#define NF_CV_RS (NF_CV_WAVES * 12 * NF_WAVE)   // floats of the K-quarter exchange

struct NfCvGeo {
    int H, W, HW;
    int TH;        // image rows per segment
    int SEG;       // segments (samples, when HW < 128) per tile
    int halo;      // 1 for 3x3, 0 for 1x1
    int FW, FS;    // frame width, frame positions per segment
    int FSZ;       // SEG * FS
    int CS;        // channel stride of a frame in LDS (odd)
    int lgW, lgSP; // log2 of W and of the pixels per segment
    int lgHW;      // log2 of H*W (a power of two)
    int nfj;       // frame positions per lane: ceil(FSZ / 64)
    float invFS, invFW;
    int64_t B;
    int64_t tiles;
    int VH, VW;    // valid extent (<= H, W): pixels beyond it are DEAD -- zero on load like the outside of the image, absent from every
                   // batch sum, never stored.  Maps whose sides are no powers of two (MNIST's 14 x 14, 7 x 7) live in power-of-two
                   // storage (16 x 16, 8 x 8) this way: the tile arithmetic stays shifts, the mask is two compares in nf_cv_decode
    int64_t Nvalid;  // B * VH * VW: the count of a batch statistic
};

static inline int nf_cv_log2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

static inline bool nf_cv_geometry(NfCvGeo& g, int64_t B, int H, int W, int ksize, int PX = NF_CV_PX) {
    if (B < 1 || H < 1 || W < 1 || (ksize != 1 && ksize != 3)) return false;
    g.H = H; g.W = W; g.HW = H * W; g.B = B;
    g.halo = ksize == 3 ? 1 : 0;
    if (g.HW >= PX) {
        if (g.HW % PX != 0 || PX % W != 0 || g.HW > 32768) return false;
        g.TH = PX / W; g.SEG = 1;
    } else {
        if (PX % g.HW != 0) return false;
        g.TH = H; g.SEG = PX / g.HW;
    }
    g.lgW = nf_cv_log2(W);
    g.lgSP = nf_cv_log2(g.TH * W);
    g.lgHW = nf_cv_log2(g.HW);
    if (g.lgW < 0 || g.lgSP < 0 || g.lgHW < 0) return false;
    g.FW = W + 2 * g.halo;
    g.FS = (g.TH + 2 * g.halo) * g.FW;
    g.FSZ = g.SEG * g.FS;
    if (g.FSZ > NF_CV_MAX_FRAME) return false;
    g.CS = g.FSZ | 1;
    g.nfj = (g.FSZ + NF_WAVE - 1) / NF_WAVE;
    g.invFS = 1.f / (float)g.FS;
    g.invFW = 1.f / (float)g.FW;
    g.tiles = (B * g.HW + PX - 1) / PX;
    g.VH = H; g.VW = W; g.Nvalid = B * g.HW;
    return true;
}
static inline bool nf_cv_set_valid(NfCvGeo& g, int vh, int vw) {      // 0 = the whole map
    if (vh < 0 || vh > g.H || vw < 0 || vw > g.W) return false;
    g.VH = vh > 0 ? vh : g.H; g.VW = vw > 0 ? vw : g.W;
    g.Nvalid = g.B * g.VH * g.VW;
    return true;
}
static inline bool nf_cv_masked(const NfCvGeo& g) { return g.VH != g.H || g.VW != g.W; }
// pixel q of a sample (q = y W + x) inside the valid extent
__device__ __forceinline__ bool nf_cv_px_ok(const NfCvGeo& g, int64_t q) { return (int)(q >> g.lgW) < g.VH && (int)(q & (g.W - 1)) < g.VW; }

__device__ __forceinline__ int nf_cv_cd_row(int r, int hs) { return (r & 3) + 8 * (r >> 2) + 4 * hs; }

// frame position of pixel px (0..127) of a tile: shifts only
__device__ __forceinline__ int nf_cv_frame_of(const NfCvGeo& g, int px) {
    const int s = px >> g.lgSP, q = px & ((1 << g.lgSP) - 1);
    return s * g.FS + ((q >> g.lgW) + g.halo) * g.FW + (q & (g.W - 1)) + g.halo;
}

// frame position f -> (owned << 30 | segment << 16 | y*W + x) of the image, or -1 outside the image / batch
#define NF_CV_SEG(t) (((t) >> 16) & 0x3fff)
#define NF_CV_SP(t) ((t) & 0xffff)
__device__ __forceinline__ int nf_cv_decode(const NfCvGeo& g, int64_t b0, int y0, int f) {
    if (f >= g.FSZ) return -1;
    // f / FS and q / FW for f < 384 by reciprocal multiplication (exact: the half-integer offset keeps the quotient
    // 0.5 / FS away from every integer, far beyond fp32 rounding) -- an integer division is ~40 instructions
    const int s = (int)(((float)f + 0.5f) * g.invFS), q = f - s * g.FS;
    const int fy = (int)(((float)q + 0.5f) * g.invFW), fx = q - fy * g.FW;
    const int gy = y0 + fy - g.halo, gx = fx - g.halo;
    const bool ok = gy >= 0 && gy < g.VH && gx >= 0 && gx < g.VW && (b0 + s) < g.B;
    const bool owned = fy >= g.halo && fy < g.TH + g.halo && fx >= g.halo && fx < g.W + g.halo;
    return ok ? (((int)owned << 30) | (s << 16) | (gy * g.W + gx)) : -1;
}
__device__ __forceinline__ void nf_cv_decode_all(int (&t)[NF_CV_FJ], const NfCvGeo& g, int64_t tile, int lane, int64_t& b0) {
    const int64_t P0 = tile * NF_CV_PX;
    b0 = P0 >> g.lgHW;
    const int y0 = g.SEG == 1 ? (int)(P0 & (g.HW - 1)) >> g.lgW : 0;
#pragma unroll
    for (int j = 0; j < NF_CV_FJ; ++j) t[j] = j < g.nfj ? nf_cv_decode(g, b0, y0, lane + NF_WAVE * j) : -1;
}

// sum over the 32 lanes of a wave half
__device__ __forceinline__ float nf_cv_half_sum(float v) {
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, NF_WAVE);
    return v;
}
// sum of 16 per-lane values over the 32 lanes of a wave half in 16 shuffles: each step halves the values a lane carries.
// Returns, in every lane, the total of register index (lane & 31) >> 1.
__device__ __forceinline__ float nf_cv_butterfly16(const float (&s)[16], int c32) {
    float t8[8], t4[4], t2[2];
    const bool b4 = c32 & 16, b3 = c32 & 8, b2 = c32 & 4, b1 = c32 & 2;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float keep = b4 ? s[i + 8] : s[i], send = b4 ? s[i] : s[i + 8];
        t8[i] = keep + __shfl_xor(send, 16, NF_WAVE);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float keep = b3 ? t8[i + 4] : t8[i], send = b3 ? t8[i] : t8[i + 4];
        t4[i] = keep + __shfl_xor(send, 8, NF_WAVE);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float keep = b2 ? t4[i + 2] : t4[i], send = b2 ? t4[i] : t4[i + 2];
        t2[i] = keep + __shfl_xor(send, 4, NF_WAVE);
    }
    const float keep = b1 ? t2[1] : t2[0], send = b1 ? t2[0] : t2[1];
    float v = keep + __shfl_xor(send, 2, NF_WAVE);
    v += __shfl_xor(v, 1, NF_WAVE);
    return v;
}

// ---------------------------------------------------------------------------------------------------------------
// staging helpers.  A CHUNK is up to 32 input channels [i0, i0 + IC), padded to ICP (multiple of 16) rows of zeros.
// ---------------------------------------------------------------------------------------------------------------
// weights of a 3x3 chunk, global (O <= 32, I, 9) -> LDS rows k = tap * ICP + ic, columns oc (TRANSPOSED == false, forward) or
// rows k = tap * OP + oc, columns ic (TRANSPOSED == true, data gradient).  A lane owns the entries r = lane + 64 j of the
// (ic, tap) plane, which are CONTIGUOUS in global memory for every oc; wave w takes oc = w, w + 16.
template <int T>
struct NfCvW {
    static constexpr int NJ = (32 * T + NF_WAVE - 1) / NF_WAVE;
    float v[NJ][NF_CV_CU];
    int rr[NJ], dst[NJ];
};
template <int T, bool TRANSPOSED>
__device__ __forceinline__ void nf_cv_w_load(NfCvW<T>& w, const float* __restrict__ weight, int O, int I, int i0, int IC, int ICP,
                                             int OP, int wid, int lane) {
    const int ostride = I * T;
    const float* src = weight + i0 * T;
#pragma unroll
    for (int j = 0; j < NfCvW<T>::NJ; ++j) {
        const int r = lane + NF_WAVE * j;
        const int ic = r / T, tap = r - ic * T;               // T is a compile-time constant
        w.rr[j] = r < IC * T ? r : -1;
        w.dst[j] = TRANSPOSED ? (tap * OP) * NF_CV_WS + ic : (tap * ICP + ic) * NF_CV_WS;
#pragma unroll
        for (int u = 0; u < NF_CV_CU; ++u) {
            const int oc = wid + u * NF_CV_WAVES;
            w.v[j][u] = (w.rr[j] >= 0 && oc < O) ? src[oc * ostride + r] : 0.f;
        }
    }
}
template <int T, bool TRANSPOSED>
__device__ __forceinline__ void nf_cv_w_store(const NfCvW<T>& w, float* Wl, int O, int wid) {
#pragma unroll
    for (int j = 0; j < NfCvW<T>::NJ; ++j)
#pragma unroll
        for (int u = 0; u < NF_CV_CU; ++u) {
            const int oc = wid + u * NF_CV_WAVES;
            if (w.rr[j] >= 0 && oc < O) Wl[w.dst[j] + (TRANSPOSED ? oc * NF_CV_WS : oc)] = w.v[j][u];
        }
}
// zero rows [first, first + n_pad) of every tap's block of rows_per_tap rows (the K padding)
template <int T>
__device__ __forceinline__ void nf_cv_zero_pad_rows(float* Wl, int rows_per_tap, int first, int n_pad, int wcols) {
    for (int e = threadIdx.x; e < T * n_pad * wcols; e += NF_CV_THREADS) {
        const int tap = e / (n_pad * wcols), q = e - tap * (n_pad * wcols);
        Wl[(tap * rows_per_tap + first) * wcols + q] = 0.f;
    }
}

// activation frame of a chunk: wave w stages channels w, w + 16; lanes = consecutive frame positions (consecutive x);
// ALL loads in flight at once, BatchNorm + ReLU applied on the way to LDS, zeros outside the image / in the padding
struct NfCvA { float v[3][NF_CV_CU]; };            // frame positions are handled three per round (<= 192 of them at once)
template <int JR>
__device__ __forceinline__ void nf_cv_act_load(NfCvA& a, const int (&t)[NF_CV_FJ], const float* __restrict__ in, const NfCvGeo& g,
                                               int I, int i0, int IC, int wid) {
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
        const int tt = t[JR + jj];
        const int base = tt >= 0 ? (NF_CV_SEG(tt) * I + i0) * g.HW + NF_CV_SP(tt) : 0;      // relative to sample b0
#pragma unroll
        for (int u = 0; u < NF_CV_CU; ++u) {
            const int c = wid + u * NF_CV_WAVES;
            a.v[jj][u] = (tt >= 0 && c < IC) ? in[base + c * g.HW] : 0.f;
        }
    }
}
template <int JR>
__device__ __forceinline__ void nf_cv_act_store(const NfCvA& a, float* Al, const int (&t)[NF_CV_FJ], const float* kc,
                                                const NfCvGeo& g, int IC, int ICP, bool has_bn, int wid, int lane) {
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
        const int f = lane + NF_WAVE * (JR + jj);
        const int tt = t[JR + jj];
        if (f < g.FSZ) {
#pragma unroll
            for (int u = 0; u < NF_CV_CU; ++u) {
                const int c = wid + u * NF_CV_WAVES;
                if (c < ICP) {
                    float x = a.v[jj][u];
                    if (has_bn) x = (tt >= 0 && c < IC) ? fmaxf(fmaf(x, kc[c], kc[32 + c]), 0.f) : 0.f;
                    Al[c * g.CS + f] = x;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// the K loop shared by the forward pass and the data gradient: acc[nb] += A[k][32 nb + c32] * B_frame[k][pixel + tap]
// over the groups [g0, g0 + gcount) of four k-pairs (8 rows); a tap has rows / 8 groups, rows a multiple of 16.
//   Wl rows (tap * rows + row), stride wcols; Fl rows `row` (stride CS), column fpos + sign * tap offset.
// The body is branch-free (the one-past-the-end prefetch re-reads a valid group) so that the operand reads of group g+1
// are in flight under the MFMAs of group g with exact wait counts.
// ---------------------------------------------------------------------------------------------------------------
template <int T, int NB>
__device__ __forceinline__ void nf_cv_kloop(f32x16 (&acc)[NB], const float* Wl, const float* Fl, const NfCvGeo& g, int rows,
                                            int wcols, int fpos, int sign, int c32, int hs, int g0, int gcount) {
    const int ngt = rows >> 3;                         // 3x3: rows is 16 or 32 -> ngt is 2 or 4
    const int lgt = ngt == 4 ? 2 : 1;
    const int glast = T * ngt - 1;
    const float* wbase = Wl + hs * wcols + c32;
    const float* fbase = Fl + hs * g.CS + fpos;
    float a0[4][NB], b0[4], a1[4][NB], b1[4];
    int gi = g0;
#define NF_CV_LOAD(A_, B_)                                                                                 \
    do {                                                                                                   \
        const int gg = gi < glast ? gi : glast;                                                            \
        const int tap = T == 1 ? 0 : gg >> lgt, q = gg - tap * ngt;                                        \
        const int dy = T == 9 ? tap / 3 - 1 : 0, dx = T == 9 ? tap - (tap / 3) * 3 - 1 : 0;                \
        const float* fp = fbase + sign * (dy * g.FW + dx) + 8 * q * g.CS;                                  \
        const float* wq = wbase + 8 * gg * wcols;                                                          \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                    \
            B_[u] = fp[2 * u * g.CS];                                                                      \
            _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) A_[u][nb] = wq[2 * u * wcols + 32 * nb];     \
        }                                                                                                  \
        ++gi;                                                                                              \
    } while (0)
#define NF_CV_MFMA(A_, B_)                                                                                 \
    _Pragma("unroll") for (int u = 0; u < 4; ++u)                                                          \
        _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                                  \
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(A_[u][nb], B_[u], acc[nb], 0, 0, 0)
    NF_CV_LOAD(a0, b0);
    for (int i = 0; i < gcount; i += 2) {
        NF_CV_LOAD(a1, b1);
        NF_CV_MFMA(a0, b0);
        NF_CV_LOAD(a0, b0);
        if (i + 1 < gcount) NF_CV_MFMA(a1, b1);
    }
#undef NF_CV_LOAD
#undef NF_CV_MFMA
}

// K-quarter exchange: the four waves (kq = 0..3) of a pixel block each hold a partial 32 x 32 accumulator; wave kq ends up
// with the TOTAL of registers [4 kq, 4 kq + 4) (rows 8 kq + {0..3} + 4 hs).  RS: [pb][owner][slot 0..2][4][64] floats.
// Callers bracket it: __syncthreads() before (RS may alias operand tiles) -- the function syncs between write and read.
__device__ __forceinline__ void nf_cv_quarter_exchange(float (&own)[4], const f32x16& acc, float* RS, int pb, int kq, int lane) {
#pragma unroll
    for (int o = 0; o < 4; ++o)
        if (o != kq) {                                 // wave-uniform
            const int slot = kq < o ? kq : kq - 1;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) RS[(((pb * 4 + o) * 3 + slot) * 4 + rr) * NF_WAVE + lane] = acc[4 * o + rr];
        } else {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) own[rr] = acc[4 * o + rr];
        }
    __syncthreads();
#pragma unroll
    for (int slot = 0; slot < 3; ++slot)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) own[rr] += RS[(((pb * 4 + kq) * 3 + slot) * 4 + rr) * NF_WAVE + lane];
}


// folded constants of the input BatchNorm (threads 0..31): kc[0] scale, kc[1] shift
__device__ __forceinline__ void nf_cv_bn_consts_fwd(float* kc, const nf_conv_desc& d, int I, int64_t Npx, int training,
                                                    float eps, float mom) {
    if (threadIdx.x < 32) {
        const int k = threadIdx.x;
        float sc = 1.f, sh = 0.f;
        if (d.bn_gamma != nullptr && k < I) {
            float mean, invstd;
            if (training) {
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int r = 0; r < NF_STAT_REPL; ++r) { t1 += d.bn_sum[32 * r + k]; t2 += d.bn_sqsum[32 * r + k]; }
                const float invN = 1.f / (float)Npx;
                const float m1 = t1 * invN;
                mean = d.bn_center[k] + m1;
                const float var = fmaxf(t2 * invN - m1 * m1, 0.f);                 // biased, as BatchNorm normalises
                invstd = 1.f / sqrtf(var + eps);
                if (blockIdx.x == 0) {
                    const float rm = d.bn_running_mean[k], rv = d.bn_running_var[k];
                    const float unb = Npx > 1 ? var * ((float)Npx / (float)(Npx - 1)) : var;
                    d.bn_running_mean[k] = (1.f - mom) * rm + mom * mean;
                    d.bn_running_var[k] = (1.f - mom) * rv + mom * unb;
                }
            } else {
                mean = d.bn_running_mean[k];
                invstd = 1.f / sqrtf(d.bn_running_var[k] + eps);
            }
            if (blockIdx.x == 0 && d.bn_save_mean != nullptr) {
                d.bn_save_mean[k] = mean;
                d.bn_save_invstd[k] = invstd;
            }
            sc = d.bn_gamma[k] * invstd;
            sh = d.bn_beta[k] - mean * sc;
        }
        kc[k] = sc;
        kc[32 + k] = sh;
    }
    if (training && d.bn_gamma != nullptr && d.bn_num_batches != nullptr && blockIdx.x == 0 && threadIdx.x == 0)
        d.bn_num_batches[0] += 1;
}

// the large-batch 3x3 kernels (conv_bulk.hip); the plans return 0 when the per-layer kernels of conv_bn.hip serve the launch
int nf_conv_bulk_fwd_plan(const nf_conv_desc* d, int64_t B, int I, int O, int H, int W, int ksize);
int nf_conv_bulk_fwd(const nf_conv_desc* desc, int64_t B, int I, int H, int W, int training, float eps, float mom, hipStream_t st);
int nf_conv_bulk_bwd_plan(const nf_conv_bwd_desc* d, int64_t B, int I, int O, int H, int W, int ksize);
int nf_conv_bulk_bwd(const nf_conv_bwd_desc* desc, int64_t B, int I, int H, int W, hipStream_t st);
int nf_conv_bulk_wgrad_plan(int64_t B, int I, int O, int H, int W, int ksize);
int nf_conv_bulk_wgrad(const nf_conv_bwd_desc* descs, const nf_conv_bwd_desc* tab, int n, int64_t B, int I, int H, int W, int slabs, hipStream_t st);
// the 1x1 output convolution at large batches (conv_bulk.hip: operands straight from global memory, no LDS)
int nf_conv1_bulk_fwd_plan(const nf_conv_desc* d, int64_t B, int I, int O, int H, int W, int ksize);
int nf_conv1_bulk_fwd(const nf_conv_desc* desc, int64_t B, int O, int H, int W, int training, float eps, float mom, hipStream_t st);
int nf_conv1_bulk_bwd_plan(const nf_conv_bwd_desc* d, int64_t B, int I, int O, int H, int W, int ksize);
int nf_conv1_bulk_bwd(const nf_conv_bwd_desc* desc, int64_t B, int O, int H, int W, hipStream_t st);
