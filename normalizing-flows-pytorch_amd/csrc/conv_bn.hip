// Fused 3x3 / 1x1 convolution + BatchNorm2d + ReLU chain on fp32 MFMA -- the image conditioner's building block.
//   ConvNet   : flows/modules.py:416-438   WN(conv3x3) -> [BN -> ReLU -> WN(conv3x3)] x 4 (+ residual) -> BN -> ReLU -> WN(conv1x1)
// The convolutional twin of linear_bn.hip, same contract ("normalise on load, statistics on store", BatchNorm backward
// finished by the producer on load), on NCHW tensors:  rows of the GEMM are pixels (b, y, x), its K axis is (tap, channel).
//
// One workgroup = one TILE of 128 consecutive pixels of the flattened (b, y, x) axis -- whole image rows of one sample
// (H*W >= 128) or whole samples (H*W < 128) -- staged ONCE into LDS as a zero-padded FRAME per channel (BatchNorm + ReLU
// applied while staging, so the nine taps read finished activations); each of the four waves owns 32 pixels.
// The GEMM is computed transposed, out^T[oc][pixel] = sum_k W[oc][k] act[k][pixel] (weights are the A operand): in the
// 32x32 C/D layout a lane then holds ONE pixel and 16 output channels, so every store instruction writes 32 consecutive
// pixels of a channel plane -- the NCHW-coalesced direction -- and the operand loads are conflict-free LDS rows.
// v_mfma_f32_32x32x2_f32 (exact fp32: the 1e-5 parity bar rules out bf16 / xf32).
#include "nf_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define NF_CV_WAVES 4
#define NF_CV_PX 128                 // pixels per tile
#define NF_CV_MAX_FRAME 384          // frame positions per channel (all segments of a tile)
#define NF_CV_MAX_I 64
#define NF_CV_MAX_O 96

struct NfCvGeo {
    int H, W, HW;
    int TH;        // image rows per segment
    int SEG;       // segments (samples, when HW < 128) per tile
    int halo;      // 1 for 3x3, 0 for 1x1
    int FW, FS;    // frame width, frame positions per segment
    int FSZ;       // SEG * FS
    int CS;        // channel stride of a frame in LDS (odd)
    int T;         // taps: 9 or 1
    int64_t B;
    int64_t tiles;
};

static inline bool nf_cv_geometry(NfCvGeo& g, int64_t B, int H, int W, int ksize) {
    if (B < 1 || H < 1 || W < 1 || (ksize != 1 && ksize != 3)) return false;
    g.H = H; g.W = W; g.HW = H * W; g.B = B;
    g.halo = ksize == 3 ? 1 : 0;
    g.T = ksize * ksize;
    if (g.HW >= NF_CV_PX) {
        if (g.HW % NF_CV_PX != 0 || NF_CV_PX % W != 0 || g.HW > 32768) return false;
        g.TH = NF_CV_PX / W; g.SEG = 1;
    } else {
        if (NF_CV_PX % g.HW != 0) return false;
        g.TH = H; g.SEG = NF_CV_PX / g.HW;
    }
    g.FW = W + 2 * g.halo;
    g.FS = (g.TH + 2 * g.halo) * g.FW;
    g.FSZ = g.SEG * g.FS;
    if (g.FSZ > NF_CV_MAX_FRAME) return false;
    g.CS = g.FSZ | 1;
    g.tiles = (B * g.HW + NF_CV_PX - 1) / NF_CV_PX;
    return true;
}

extern "C" int nf_conv_bn_usable(int64_t B, int I, int O, int H, int W, int ksize) {
    NfCvGeo g;
    if (I < 1 || O < 1 || I > NF_CV_MAX_I || O > NF_CV_MAX_O) return 0;
    return nf_cv_geometry(g, B, H, W, ksize) ? 1 : 0;
}

__device__ __forceinline__ float nf_half32_sum_cv(float v) {  // sum over the 32 lanes of this wave half
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, NF_WAVE);
    return v;
}
__device__ __forceinline__ int nf_cv_cd_row(int r, int hs) { return (r & 3) + 8 * (r >> 2) + 4 * hs; }

// frame table: position f of the tile's frame -> (segment << 16 | y*W + x) of the image, or -1 outside the image / batch
__device__ __forceinline__ void nf_cv_frame_table(int* tab, const NfCvGeo& g, int64_t tile) {
    const int64_t P0 = tile * NF_CV_PX;
    const int64_t b0 = P0 / g.HW;
    const int y0 = g.SEG == 1 ? (int)((P0 - b0 * g.HW) / g.W) : 0;
    for (int f = threadIdx.x; f < g.FSZ; f += blockDim.x) {
        const int s = f / g.FS, q = f - s * g.FS;
        const int fy = q / g.FW, fx = q - fy * g.FW;
        const int gy = y0 + fy - g.halo, gx = fx - g.halo;
        const bool ok = gy >= 0 && gy < g.H && gx >= 0 && gx < g.W && (b0 + s) < g.B;
        tab[f] = ok ? ((s << 16) | (gy * g.W + gx)) : -1;
    }
}
// frame position of pixel r (0..127) of a tile
__device__ __forceinline__ int nf_cv_frame_of(const NfCvGeo& g, int r) {
    const int seg_px = g.TH * g.W;
    const int s = r / seg_px, q = r - s * seg_px;
    const int ly = q / g.W, x = q - ly * g.W;
    return s * g.FS + (ly + g.halo) * g.FW + x + g.halo;
}

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
// LDS: Wl[T * IP][32 * OCB]  (k = tap * IP + ic rows, oc contiguous: the A fragment of lane (oc, k) is a conflict-free row)
//      Al[IP][CS]            (finished activations, zero padded; the B fragment of lane (pixel, k) walks a frame row)
//      tab[FSZ], kc[2][64], red[2][4][32]
template <int OCB>
__global__ void __launch_bounds__(NF_CV_WAVES * NF_WAVE) k_conv_bn_fwd(nf_conv_desc d, NfCvGeo g, int I, int O, int IP,
                                                                        int training, float eps, float mom) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int WS = 32 * OCB;
    float* Wl = smem;
    float* Al = Wl + g.T * IP * WS;
    int* tab = reinterpret_cast<int*>(Al + IP * g.CS);
    float* kc = reinterpret_cast<float*>(tab + NF_CV_MAX_FRAME);       // [2][64]
    float* red = kc + 2 * NF_CV_MAX_I;                                 // [2][4][32]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, c32 = lane & 31, hs = lane >> 5;
    const bool has_bn = d.bn_gamma != nullptr;
    const int64_t Npx = g.B * g.HW;
    const float invN = 1.f / (float)Npx;

    // ---- once per workgroup: weights, folded BatchNorm constants ----
    for (int e = threadIdx.x; e < g.T * IP * WS; e += blockDim.x) Wl[e] = 0.f;
    if ((int)threadIdx.x < NF_CV_MAX_I) {
        const int k = threadIdx.x;
        float sc = has_bn ? 0.f : 1.f, sh = 0.f;
        if (has_bn && k < I) {
            float mean, invstd;
            if (training) {
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int r = 0; r < NF_STAT_REPL; ++r) { t1 += d.bn_sum[32 * r + k]; t2 += d.bn_sqsum[32 * r + k]; }
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
        kc[NF_CV_MAX_I + k] = sh;
    }
    if (training && has_bn && d.bn_num_batches != nullptr && blockIdx.x == 0 && threadIdx.x == 0) d.bn_num_batches[0] += 1;
    __syncthreads();
    for (int e = threadIdx.x; e < O * I * g.T; e += blockDim.x) {          // global (O, I, T) -> LDS [tap][ic][oc]
        const int oc = e / (I * g.T), r = e - oc * (I * g.T);
        const int ic = r / g.T, tap = r - ic * g.T;
        Wl[(tap * IP + ic) * WS + oc] = d.weight[e];
    }

    float bias_r[OCB][16];
#pragma unroll
    for (int ob = 0; ob < OCB; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int oc = ob * 32 + nf_cv_cd_row(r, hs);
            bias_r[ob][r] = oc < O ? d.bias[oc] : 0.f;
        }
    const bool want_stats = d.stat_sum != nullptr;      // O <= 32 (OCB == 1) by contract
    const bool has_res = d.residual != nullptr;
    float s1[16], s2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { s1[r] = 0.f; s2[r] = 0.f; }
    const int px = wid * 32 + c32;                      // this lane's pixel of the tile
    const int fpos = nf_cv_frame_of(g, px);

    for (int64_t tile = blockIdx.x; tile < g.tiles; tile += gridDim.x) {
        __syncthreads();                                // previous tile's readers of Al / tab are done
        nf_cv_frame_table(tab, g, tile);
        __syncthreads();
        const int64_t b0 = (tile * NF_CV_PX) / g.HW;
        for (int c = wid; c < IP; c += NF_CV_WAVES) {   // one channel per wave per trip: consecutive lanes, consecutive x
            const float sc = kc[c], sh = kc[NF_CV_MAX_I + c];
            for (int f = lane; f < g.FSZ; f += NF_WAVE) {
                const int t = tab[f];
                float v = 0.f;
                if (t >= 0 && c < I) {
                    const float x = d.in[((b0 + (t >> 16)) * I + c) * g.HW + (t & 0xffff)];
                    v = has_bn ? fmaxf(fmaf(x, sc, sh), 0.f) : x;
                }
                Al[c * g.CS + f] = v;
            }
        }
        __syncthreads();

        f32x16 acc[OCB];
#pragma unroll
        for (int ob = 0; ob < OCB; ++ob)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ob][r] = 0.f;
        for (int tap = 0; tap < g.T; ++tap) {
            const int dy = g.T == 9 ? tap / 3 - 1 : 0, dx = g.T == 9 ? tap - (tap / 3) * 3 - 1 : 0;
            const float* ap = Al + hs * g.CS + fpos + dy * g.FW + dx;
            const float* wp = Wl + (tap * IP + hs) * WS + c32;
#pragma unroll 4
            for (int kk = 0; kk < IP / 2; ++kk) {
                const float bv = ap[2 * kk * g.CS];
#pragma unroll
                for (int ob = 0; ob < OCB; ++ob)
                    acc[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(wp[2 * kk * WS + ob * 32], bv, acc[ob], 0, 0, 0);
            }
        }
        // ---- epilogue: bias, residual, store (32 consecutive pixels per instruction), shifted batch sums ----
        const int64_t P = tile * NF_CV_PX + px;
        const bool pv = P < Npx;
        const int64_t b = pv ? P / g.HW : 0;
        const int64_t q = pv ? P - b * g.HW : 0;
#pragma unroll
        for (int ob = 0; ob < OCB; ++ob)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int oc = ob * 32 + nf_cv_cd_row(r, hs);
                if (pv && oc < O) {
                    const int64_t idx = (b * O + oc) * g.HW + q;
                    float dv = acc[ob][r];
                    if (has_res) dv += d.residual[idx];
                    d.out[idx] = dv + bias_r[ob][r];
                    if (ob == 0) { s1[r] += dv; s2[r] = fmaf(dv, dv, s2[r]); }
                }
            }
    }
    if (want_stats) {                                   // block-uniform branch
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float t1 = nf_half32_sum_cv(s1[r]), t2 = nf_half32_sum_cv(s2[r]);
            if (c32 == 0) {
                const int oc = nf_cv_cd_row(r, hs);
                red[(0 * NF_CV_WAVES + wid) * 32 + oc] = t1;
                red[(1 * NF_CV_WAVES + wid) * 32 + oc] = t2;
            }
        }
        __syncthreads();
        if (wid == 0 && hs == 0 && c32 < O) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < NF_CV_WAVES; ++w) { t1 += red[(0 * NF_CV_WAVES + w) * 32 + c32]; t2 += red[(1 * NF_CV_WAVES + w) * 32 + c32]; }
            const int rep = 32 * (blockIdx.x % NF_STAT_REPL);
            atomicAdd(d.stat_sum + rep + c32, t1);
            atomicAdd(d.stat_sqsum + rep + c32, t2);
        }
    }
}

static inline size_t nf_cv_fwd_lds(const NfCvGeo& g, int IP, int OCB) {
    return sizeof(float) * ((size_t)g.T * IP * 32 * OCB + (size_t)IP * g.CS + NF_CV_MAX_FRAME + 2 * NF_CV_MAX_I + 2 * NF_CV_WAVES * 32);
}

// dynamic LDS above the 64 KB default needs a per-kernel opt-in (160 KB per CU on gfx950); once per instantiation
template <typename K>
static inline int nf_cv_optin(K kernel, size_t lds) {
    static bool done = false;          // one static per kernel type
    if (lds > 160 * 1024) return NF_E_BADARG;
    if (!done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        done = true;
    }
    return 0;
}

extern "C" int nf_conv_bn_fwd(const nf_conv_desc* desc, int64_t B, int I, int O, int H, int W, int ksize, int training,
                              float bn_eps, float bn_momentum, nf_stream_t stream) {
    NfCvGeo g;
    if (desc == nullptr || I < 1 || O < 1 || I > NF_CV_MAX_I || O > NF_CV_MAX_O) return NF_E_BADARG;
    if (B == 0) return 0;
    if (!nf_cv_geometry(g, B, H, W, ksize)) return NF_E_BADARG;
    if (desc->bn_gamma != nullptr && I > 32) return NF_E_BADARG;          // statistics vectors are 32 wide
    if (desc->stat_sum != nullptr && O > 32) return NF_E_BADARG;
    const int IP = (I + 1) & ~1;
    const int OCB = (O + 31) / 32;
    const size_t lds = nf_cv_fwd_lds(g, IP, OCB);
    unsigned grid = (unsigned)(g.tiles < 1024 ? g.tiles : 1024);
    hipStream_t st = (hipStream_t)stream;
    int rc;
#define NF_LAUNCH(N_)                                                                                                  \
    rc = nf_cv_optin(k_conv_bn_fwd<N_>, lds);                                                                          \
    if (rc) return rc;                                                                                                 \
    hipLaunchKernelGGL(k_conv_bn_fwd<N_>, dim3(grid), dim3(NF_CV_WAVES * NF_WAVE), lds, st, *desc, g, I, O, IP, training, \
                       bn_eps, bn_momentum)
    if (OCB == 1) { NF_LAUNCH(1); }
    else if (OCB == 2) { NF_LAUNCH(2); }
    else { NF_LAUNCH(3); }
#undef NF_LAUNCH
    NF_CHECK_LAUNCH();
    return 0;
}
