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
//
// These launches are LATENCY-bound (a 64 x 32 x 16 x 16 layer is 0.3 GFLOP over 128 workgroups, one wave per SIMD): what
// matters is the length of the serial chain inside a workgroup.  Hence: every global load of a phase is issued before the
// first dependent store (one memory latency per phase, not one per loop trip); no integer division on the per-element
// paths (the spatial extents are powers of two, the weight decode is done once per lane); the MFMA operands are read from
// LDS one group ahead of the MFMAs that consume them; the per-channel batch sums use a halving butterfly (16 shuffles per
// statistic instead of 80).
#include <mutex>
#include <unordered_set>
#include <vector>

#include "nf_common.h"
#include "nf_det.h"

NF_DET_STATE(nf_cvb)
NF_DET_HOST_API(nf_cvb)

#include "nf_conv_core.h"

// phase stamps of workgroup 0 (tools/probes/conv_prof.py builds this file with -DNF_CV_PROF=1; 100 MHz wall clock)
#ifdef NF_CV_PROF
__device__ long long nf_cv_prof[32];
#define NF_CV_STAMP(i)                                                                 \
    do {                                                                               \
        if (blockIdx.x == 0 && threadIdx.x == 0) nf_cv_prof[i] = wall_clock64();       \
    } while (0)
extern "C" int nf_cv_prof_read(long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(nf_cv_prof), sizeof(long long) * 32);
}
#else
#define NF_CV_STAMP(i)
#endif

extern "C" int nf_conv_bn_usable(int64_t B, int I, int O, int H, int W, int ksize) {
    NfCvGeo g;
    if (I < 1 || O < 1 || I > NF_CV_MAX_I || O > NF_CV_MAX_O) return 0;
    if (ksize == 3 && O > 32) return 0;                  // the 3x3 layers of the reference produce 32 channels
    if (ksize == 1 && I > 32) return 0;
    return nf_cv_geometry(g, B, H, W, ksize) ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------------
// forward.  Sixteen waves: wave w works on pixel block pb = w & 3 (32 pixels) and
//   3x3 (O <= 32): K quarter kq = w >> 2 of the (tap, channel) axis; the quarters are exchanged through LDS and every wave
//                  finishes FOUR output channels x two halves of its pixel block (bias, residual, store, batch sums);
//   1x1 (I <= 32): output blocks ob = (w >> 2), (w >> 2) + 4 with the whole K (no exchange).
// The staging work (weights, activation frame) is spread over all sixteen waves: four waves per SIMD hide each other's
// issue and memory latency, which is what bounds these small launches.
// ---------------------------------------------------------------------------------------------------------------
// LDS: Wl[T * 32][WCOLS]  (one chunk)     Al[32][CS]     kc[2][32], red[2][4][32];   RS aliases Wl | Al
template <int T, int OCB>
__global__ void __launch_bounds__(NF_CV_THREADS) k_conv_bn_fwd(nf_conv_desc d, NfCvGeo g, int I, int O, int training, float eps,
                                                               float mom) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int WCOLS = 32 * OCB + 1;
    constexpr int NOB = T == 9 ? 1 : (OCB + 3) / 4;    // output blocks per wave
    float* Wl = smem;
    float* Al = Wl + T * 32 * WCOLS;
    float* tail = smem + (T * 32 * WCOLS + 32 * g.CS > NF_CV_RS ? T * 32 * WCOLS + 32 * g.CS : NF_CV_RS);
    float* kc = tail;                                   // [2][32]
    float* red = kc + 64;                               // [2][4][32]
    float* RS = smem;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, c32 = lane & 31, hs = lane >> 5;
    const int pb = wid & 3, kq = wid >> 2;
    const bool has_bn = d.bn_gamma != nullptr;
    const int64_t Npx = g.B * g.HW;
    const int nchunks = (I + 31) / 32;

    NF_CV_STAMP(0);
    nf_cv_bn_consts_fwd(kc, d, I, g.Nvalid, training, eps, mom);
    const bool want_stats = d.stat_sum != nullptr;      // O <= 32 by contract
    const bool has_res = d.residual != nullptr;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};      // 3x3: this wave's four channels
    const int px = pb * 32 + c32;                       // this lane's pixel of the tile
    const int fpos = nf_cv_frame_of(g, px);
    NF_CV_STAMP(1);

    for (int64_t tile = blockIdx.x; tile < g.tiles; tile += gridDim.x) {
        const int64_t P0t = tile * NF_CV_PX;
        const int64_t b0 = P0t >> g.lgHW;
        const int y0t = g.SEG == 1 ? (int)(P0t & (g.HW - 1)) >> g.lgW : 0;
        NF_CV_STAMP(20);
        const float* in0 = d.in + b0 * I * g.HW;
        const int64_t P = tile * NF_CV_PX + px;
        const bool pv = P < Npx && nf_cv_px_ok(g, P & (g.HW - 1));       // (dead pixels of a masked map: not stored, not summed)
        const int64_t b = pv ? P >> g.lgHW : 0;
        const int64_t q = pv ? P & (g.HW - 1) : 0;
        float rres[4], bias_r[4];                      // 3x3 epilogue operands, fetched under the staging and the K loop
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int oc = 8 * kq + rr + 4 * hs;
            bias_r[rr] = (T == 9 && oc < O) ? d.bias[oc] : 0.f;
            rres[rr] = (T == 9 && has_res && pv && oc < O) ? d.residual[(b * O + oc) * g.HW + q] : 0.f;
        }
        f32x16 acc[NOB];
#pragma unroll
        for (int n = 0; n < NOB; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
        for (int ch = 0; ch < nchunks; ++ch) {
            const int i0 = 32 * ch, IC = min(32, I - i0), ICP = (IC + 15) & ~15;
            // staging, one frame position per lane and trip (see the backward kernel): the weights ride the first trip
            NfCvW<T> wv;
            if (T == 9) nf_cv_w_load<T, false>(wv, d.weight, O, I, i0, IC, ICP, 0, wid, lane);
            NF_CV_STAMP(21);
#pragma unroll 1
            for (int jj = 0; jj < g.nfj; ++jj) {
                const int f = lane + NF_WAVE * jj;
                const int t = nf_cv_decode(g, b0, y0t, f);
                const int sp = t >= 0 ? NF_CV_SP(t) : 0, sg_ = t >= 0 ? NF_CV_SEG(t) : 0;
                float xa[NF_CV_CU];
#pragma unroll
                for (int u = 0; u < NF_CV_CU; ++u) {
                    const int c = wid + u * NF_CV_WAVES;
                    xa[u] = (t >= 0 && c < IC) ? in0[(sg_ * I + i0 + c) * g.HW + sp] : 0.f;
                }
                if (jj == 0) {
                    __syncthreads();                    // previous readers of Wl / Al / RS are done; kc is written
                    NF_CV_STAMP(22);
                    if (T == 9) {
                        nf_cv_w_store<T, false>(wv, Wl, O, wid);
                    } else {                            // 1x1, I <= 32: row ic, column oc of the wide tile
                        for (int e = threadIdx.x; e < O * IC; e += NF_CV_THREADS) {
                            const int oc = e / IC, ic = e - oc * IC;
                            Wl[ic * WCOLS + oc] = d.weight[oc * I + i0 + ic];
                        }
                    }
                    if (ICP > IC) nf_cv_zero_pad_rows<T>(Wl, ICP, IC, ICP - IC, WCOLS);
                    NF_CV_STAMP(23);
                }
                if (f < g.FSZ) {
#pragma unroll
                    for (int u = 0; u < NF_CV_CU; ++u) {
                        const int c = wid + u * NF_CV_WAVES;
                        if (c < ICP) {
                            float x = xa[u];
                            if (has_bn) x = (t >= 0 && c < IC) ? fmaxf(fmaf(x, kc[c], kc[32 + c]), 0.f) : 0.f;
                            Al[c * g.CS + f] = x;
                        }
                    }
                }
            }
            NF_CV_STAMP(24);
            __syncthreads();
            NF_CV_STAMP(2);
            if (T == 9) {
                const int ng = T * (ICP >> 3);
                const int g0 = (kq * ng) >> 2, g1 = ((kq + 1) * ng) >> 2;
                nf_cv_kloop<T, 1>(reinterpret_cast<f32x16(&)[1]>(acc[0]), Wl, Al, g, ICP, WCOLS, fpos, +1, c32, hs, g0, g1 - g0);
            } else {
#pragma unroll
                for (int n = 0; n < NOB; ++n) {
                    const int ob = kq + 4 * n;
                    if (ob < OCB)                       // wave-uniform
                        nf_cv_kloop<T, 1>(reinterpret_cast<f32x16(&)[1]>(acc[n]), Wl + 32 * ob, Al, g, ICP, WCOLS, fpos, +1, c32,
                                          hs, 0, ICP >> 3);
                }
            }
        }
        NF_CV_STAMP(3);
        if (T == 9) {
            // ---- exchange the K quarters, then finish four channels per half: bias, residual, store, shifted batch sums ----
            __syncthreads();                            // every wave is done with Wl / Al: RS may overwrite them
            float own[4];
            nf_cv_quarter_exchange(own, acc[0], RS, pb, kq, lane);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int oc = 8 * kq + rr + 4 * hs;
                if (pv && oc < O) {
                    const float dv = own[rr] + rres[rr];
                    d.out[(b * O + oc) * g.HW + q] = dv + bias_r[rr];
                    s1[rr] += dv;
                    s2[rr] = fmaf(dv, dv, s2[rr]);
                }
            }
        } else {
#pragma unroll
            for (int n = 0; n < NOB; ++n) {
                const int ob = kq + 4 * n;
                if (ob < OCB) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int oc = ob * 32 + nf_cv_cd_row(r, hs);
                        if (pv && oc < O) d.out[(b * O + oc) * g.HW + q] = acc[n][r] + d.bias[oc];
                    }
                }
            }
        }
    }
    NF_CV_STAMP(4);
    if (T == 9 && want_stats) {                         // block-uniform branch
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const float t1 = nf_cv_half_sum(s1[rr]), t2 = nf_cv_half_sum(s2[rr]);
            if (c32 == 0) {
                const int oc = 8 * kq + rr + 4 * hs;
                red[(0 * 4 + pb) * 32 + oc] = t1;
                red[(1 * 4 + pb) * 32 + oc] = t2;
            }
        }
        __syncthreads();
        if (threadIdx.x < 64) {                         // wave 0 adds for the workgroup; half 0: sums, half 1: squares
            NF_DET_REPL_CHAIN(NF_STAT_REPL);            // (deterministic mode: the workgroups of a replica in block order, the replicas side by side)
            NF_DET_ENTER_WAVE_K(nf_cvb);
            if (c32 < O) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) t += red[(hs * 4 + w) * 32 + c32];
                const int rep = 32 * (blockIdx.x % NF_STAT_REPL);
                atomicAdd((hs == 0 ? d.stat_sum : d.stat_sqsum) + rep + c32, t);
            }
            NF_DET_LEAVE_WAVE_K(nf_cvb);
        }
    }
    NF_CV_STAMP(5);
}

// dynamic LDS above the 64 KB default needs a per-kernel opt-in (160 KB per CU on gfx950), once per kernel (keyed by the
// function's address: the instantiations of one template share their pointer TYPE)
template <typename K>
static inline int nf_cv_optin(K kernel, size_t lds) {
    static std::mutex mu;
    static std::unordered_set<const void*> done;
    if (lds > 160 * 1024) return NF_E_BADARG;
    const void* key = reinterpret_cast<const void*>(kernel);
    std::lock_guard<std::mutex> lock(mu);
    if (done.find(key) == done.end()) {
        hipError_t e = hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        done.insert(key);
    }
    return 0;
}

static inline bool nf_cv_fits_int32(int64_t B, int C, int HW) { return B * C * HW < (int64_t)1 << 31; }

extern "C" int nf_conv_bn_fwd(const nf_conv_desc* desc, int64_t B, int I, int O, int H, int W, int ksize, int training,
                              float bn_eps, float bn_momentum, nf_stream_t stream) {
    NfCvGeo g;
    if (desc == nullptr || !nf_conv_bn_usable(B > 0 ? B : 1, I, O, H, W, ksize)) return NF_E_BADARG;
    if (B == 0) return 0;
    if (!nf_cv_geometry(g, B, H, W, ksize)) return NF_E_BADARG;
    if (desc->bn_gamma != nullptr && I > 32) return NF_E_BADARG;          // statistics vectors are 32 wide
    if (desc->stat_sum != nullptr && (O > 32 || ksize != 3)) return NF_E_BADARG;
    if (desc->residual != nullptr && ksize != 3) return NF_E_BADARG;
    if (!nf_cv_set_valid(g, desc->valid_h, desc->valid_w)) return NF_E_BADARG;
    const bool masked = nf_cv_masked(g);               // a map in power-of-two storage: the per-layer kernels below only
    if (!masked && nf_conv_bulk_fwd_plan(desc, B, I, O, H, W, ksize))          // large batches: independent waves on the bf16 matrix pipe (conv_bulk.hip)
        return nf_conv_bulk_fwd(desc, B, I, H, W, training, bn_eps, bn_momentum, (hipStream_t)stream);
    if (!masked && nf_conv1_bulk_fwd_plan(desc, B, I, O, H, W, ksize))         // ... and the 1x1 output convolution: operands straight from global memory
        return nf_conv1_bulk_fwd(desc, B, O, H, W, training, bn_eps, bn_momentum, (hipStream_t)stream);
    const int OCB = (O + 31) / 32;
    const int T = ksize * ksize;
    size_t tiles_f = (size_t)T * 32 * (32 * OCB + 1) + (size_t)32 * g.CS;
    if (tiles_f < NF_CV_RS) tiles_f = NF_CV_RS;
    if (!nf_cv_fits_int32(B, I > O ? I : O, H * W)) return NF_E_BADARG;
    const size_t lds = sizeof(float) * (tiles_f + 64 + 2 * 4 * 32);
    unsigned grid = (unsigned)(g.tiles < 1024 ? g.tiles : 1024);
    hipStream_t st = (hipStream_t)stream;
    int rc;
#define NF_LAUNCH(T_, N_)                                                                                              \
    do {                                                                                                               \
        rc = nf_cv_optin(k_conv_bn_fwd<T_, N_>, lds);                                                                  \
        if (rc) return rc;                                                                                             \
        hipLaunchKernelGGL((k_conv_bn_fwd<T_, N_>), dim3(grid), dim3(NF_CV_THREADS), lds, st, *desc, g, I, O, training, bn_eps, \
                           bn_momentum);                                                                               \
    } while (0)
    if (T == 9) NF_LAUNCH(9, 1);
    else
        switch (OCB) {
            case 1: NF_LAUNCH(1, 1); break;
            case 2: NF_LAUNCH(1, 2); break;
            case 3: NF_LAUNCH(1, 3); break;
            case 4: NF_LAUNCH(1, 4); break;
            case 5: NF_LAUNCH(1, 5); break;
            default: NF_LAUNCH(1, 6); break;
        }
#undef NF_LAUNCH
    NF_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// backward (training mode): the gradient G of `out` is assembled on load exactly as nf_linear_bn_bwd does
//     G = g_direct + g_skip + BNbwd(gn_src)          (frame with halo: the data gradient reads G at the nine neighbours)
//     gn_out[b,i,y,x] = ( sum_{o,tap} W[o,i,tap] G[b,o,(y,x) - tap] ) * [act > 0]     + its two batch sums
//     g_weff[o,i,tap] = sum_px G[o,px] act[i,px + tap]                                 (K = the tile's 128 pixels)
// Three GEMM families on the same LDS frames: data gradient (A = W^T rows ic, B = G frame rows oc), weight gradient
// (A = G frame rows oc walked along pixels, B = act frame rows ic, shifted by the tap) -- the frames have an odd channel
// stride, so lanes that walk channels (weight gradient) and lanes that walk pixels (data gradient) are both conflict-free.
// Sixteen waves.  Weight gradient: the 32 x 32 tiles -- 9 taps (3x3) or OCB output blocks (1x1) -- go to waves 0..8, which
// walk the tile's 128 pixels; the workgroup writes ONE slab (summed by nf_slab_sum: no atomics, deterministic).  Data
// gradient: pixel block x K quarter like the forward pass; the waves without a weight-gradient tile start it early.
// ---------------------------------------------------------------------------------------------------------------
// LDS: Wd[T * 32 OCB][33]  (one input chunk; rows k = tap * OP + oc, columns ic)     Al[32][CS]     Gl[32 OCB][CS]
//      cb[5][32] (consumer BatchNorm constants), kc[4][32] (input BatchNorm), red[2][4][32];
//      the K-quarter exchange RS aliases the start of the tiles once their last reader is done
// MODE 0: data and weight gradient (a stand-alone layer).  A conditioner backward inside a trainer step splits them: MODE 1 =
// the data gradient only -- the chain the next layer waits for: no activation frame, no weight-gradient tiles, no slab, no
// bias sums --, MODE 2 = the weight gradient only -- nothing but Adam waits for it, so the MODE 2 passes of every layer of
// the model are deferred and run sixteen layers per launch (blockIdx.y = layer) after the last layer's data gradient, where
// they fill the machine instead of sitting on the latency chain of 8 .. 128 workgroups (nf_conv_bn_wgrad_multi): no weights,
// no K loop, no exchange, no stores of g_store / gn_out, no BatchNorm sums.
#define NF_CV_WG_MAX NF_CONV_WGRAD_MAX
struct NfCvBwdMulti { nf_conv_bwd_desc d[NF_CV_WG_MAX]; };

template <int T, int ICB, int OCB, int MODE>
__device__ __forceinline__ void nf_cv_bwd_body(const nf_conv_bwd_desc& d, const NfCvGeo& g, int I, int O, int iters) {
    constexpr bool DG = MODE != 2, WG = MODE != 1;
    static_assert(T == 1 || OCB == 1, "3x3 layers produce <= 32 channels");
    static_assert(T == 9 || ICB == 1, "the 1x1 layer consumes <= 32 channels");
    constexpr int NTC = T == 9 ? 9 : OCB;              // weight-gradient tiles per input chunk
    constexpr int NU = NTC;
    constexpr int NUW = (NU + NF_CV_WAVES - 1) / NF_CV_WAVES;
    constexpr int OPmax = 32 * OCB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int OP = (O + 15) & ~15;                     // K rows of the data gradient, zero padded
    float* Wd = smem;
    float* Al = Wd + (DG ? T * OPmax * NF_CV_WS : 0);  // (the weight-gradient pass stages no weights)
    float* Gl = Al + 32 * g.CS;
    float* endf = Gl + OPmax * g.CS;
    float* X = smem;                                   // K-quarter exchange: over Wd | Al (3x3; never reaches Gl) or all tiles (1x1)
    if (DG && endf < smem + NF_CV_RS) endf = smem + NF_CV_RS;
    float* cb = endf;                                  // [5][32]
    float* kc = cb + 5 * 32;                           // [4][32]
    float* red = kc + 4 * 32;                          // [2][4][32]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, c32 = lane & 31, hs = lane >> 5;
    const int pb = wid & 3, kq = wid >> 2;
    const bool has_bn = d.bn_gamma != nullptr;         // => ICB == 1
    const bool has_src = d.gn_src != nullptr;          // => OCB == 1
    const int64_t Npx = g.B * g.HW;
    const float invN = 1.f / (float)g.Nvalid;           // (a masked map: the valid pixels only)

    NF_CV_STAMP(8);
    if (threadIdx.x < 32) {
        const int oo = threadIdx.x;
        float c1 = 0.f, mean = 0.f, invstd = 0.f, mg = 0.f, mgx = 0.f;
        if (has_src && oo < O) {
            invstd = d.cbn_save_invstd[oo];
            mean = d.cbn_save_mean[oo];
            c1 = d.cbn_gamma[oo] * invstd;
            if (d.cbn_sum_g != nullptr) {
#pragma unroll
                for (int r = 0; r < NF_STAT_REPL; ++r) { mg += d.cbn_sum_g[32 * r + oo]; mgx += d.cbn_sum_gx[32 * r + oo]; }
                mg *= invN;
                mgx *= invN;
            }
        }
        cb[oo] = c1; cb[32 + oo] = mean; cb[64 + oo] = invstd; cb[96 + oo] = mg; cb[128 + oo] = mgx;
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
    // weight-gradient tile of this wave: tl = wid + 16 v -> tap (3x3) or output block (1x1)
    int aoff[NUW], boff[NUW];
    bool uval[NUW];
#pragma unroll
    for (int v = 0; v < NUW; ++v) {
        const int unit = wid + NF_CV_WAVES * v;
        uval[v] = unit < NU;
        const int tl = uval[v] ? unit : 0;
        const int tap = T == 9 ? tl : 0, ob = T == 9 ? 0 : tl;
        const int dy = T == 9 ? tap / 3 - 1 : 0, dx = T == 9 ? tap - (tap / 3) * 3 - 1 : 0;
        aoff[v] = (ob * 32 + c32) * g.CS;
        boff[v] = c32 * g.CS + dy * g.FW + dx;
    }
    float sg[4] = {0.f, 0.f, 0.f, 0.f}, sgx[4] = {0.f, 0.f, 0.f, 0.f};
    float gbw[NUW];
#pragma unroll
    for (int v = 0; v < NUW; ++v) gbw[v] = 0.f;
    const int px = pb * 32 + c32;
    const int fpos = nf_cv_frame_of(g, px);
    float* slab = d.g_weff + (int64_t)blockIdx.x * O * I * T;
    NF_CV_STAMP(9);

    for (int it = 0; it < iters; ++it) {
        const int64_t tile = (int64_t)it * gridDim.x + blockIdx.x;
        if (tile >= g.tiles) break;                    // block-uniform
        const int64_t b0 = (tile * NF_CV_PX) >> g.lgHW;
        const float* in0 = d.in + b0 * I * g.HW;
        const int64_t go0 = b0 * O * g.HW;             // sample b0 of the (B, O, H, W) tensors
        const int64_t P = tile * NF_CV_PX + px;
        const bool pv = P < Npx && nf_cv_px_ok(g, P & (g.HW - 1));
        const int64_t pb64 = pv ? P >> g.lgHW : 0;
        const int64_t pq = pv ? P & (g.HW - 1) : 0;
#pragma unroll
        for (int ib = 0; ib < ICB; ++ib) {
            const int i0 = 32 * ib, IC = min(32, I - i0), ICP = (IC + 15) & ~15;
            // staging, one frame position per lane and trip (three trips at the 16 x 16 level): the loads of a trip -- two
            // activation channels, two channels of each tensor G is assembled from, and in the first trip the weights -- are in
            // flight together.  Everything at once needs ~70 live registers on top of the kernel's own state and spilled (120
            // scratch operations, 25 us); three short round trips are cheaper than that.
            NfCvW<T> wv;
            if (DG && T == 9) nf_cv_w_load<T, true>(wv, d.weight, O, I, i0, IC, ICP, OP, wid, lane);
            const int64_t P0t = tile * NF_CV_PX;
            const int y0t = g.SEG == 1 ? (int)(P0t & (g.HW - 1)) >> g.lgW : 0;
#pragma unroll 1
            for (int jj = 0; jj < g.nfj; ++jj) {
                const int f = lane + NF_WAVE * jj;
                const int t = nf_cv_decode(g, b0, y0t, f);
                const int sp = t >= 0 ? NF_CV_SP(t) : 0, sg_ = t >= 0 ? NF_CV_SEG(t) : 0;
                float xa[NF_CV_CU];
#pragma unroll
                for (int u = 0; u < NF_CV_CU; ++u) {
                    const int c = wid + u * NF_CV_WAVES;
                    xa[u] = (WG && t >= 0 && c < IC) ? in0[(sg_ * I + i0 + c) * g.HW + sp] : 0.f;
                }
                if (jj == 0) __syncthreads();          // previous readers of the frames / Wd / exchange are done; cb, kc
                if (ib == 0) {
                    // ---- G frame, assembled; g_store for the pixels this tile owns ----
                    for (int c0 = wid; c0 < OPmax; c0 += NF_CV_CU * NF_CV_WAVES) {
                        float w1[NF_CV_CU], w2[NF_CV_CU], w3[NF_CV_CU], w4[NF_CV_CU];
#pragma unroll
                        for (int u = 0; u < NF_CV_CU; ++u) {
                            const int c = c0 + u * NF_CV_WAVES;
                            const bool ok = t >= 0 && c < O;
                            const int64_t idx = go0 + (sg_ * O + c) * g.HW + sp;
                            w1[u] = (ok && d.g_direct != nullptr) ? d.g_direct[idx] : 0.f;
                            w2[u] = (ok && d.g_skip != nullptr) ? d.g_skip[idx] : 0.f;
                            w3[u] = (ok && has_src) ? d.gn_src[idx] : 0.f;
                            w4[u] = (ok && has_src) ? d.out[idx] : 0.f;
                        }
                        if (f < g.FSZ) {
#pragma unroll
                            for (int u = 0; u < NF_CV_CU; ++u) {
                                const int c = c0 + u * NF_CV_WAVES;
                                if (c < OPmax) {
                                    float v = 0.f;
                                    if (t >= 0 && c < O) {
                                        v = w1[u] + w2[u];
                                        if (has_src) {                          // BatchNorm backward on load
                                            const float xh = (w4[u] - cb[32 + c]) * cb[64 + c];
                                            v += cb[c] * (w3[u] - cb[96 + c] - xh * cb[128 + c]);
                                        }
                                        if (DG && d.g_store != nullptr && (t >> 30)) d.g_store[go0 + (sg_ * O + c) * g.HW + sp] = v;
                                    }
                                    Gl[c * g.CS + f] = v;
                                }
                            }
                        }
                    }
                }
                if (WG && f < g.FSZ) {
#pragma unroll
                    for (int u = 0; u < NF_CV_CU; ++u) {
                        const int c = wid + u * NF_CV_WAVES;
                        if (c < ICP) {
                            float x = xa[u];
                            if (has_bn) x = (t >= 0 && c < IC) ? fmaxf(fmaf(x, kc[c], kc[32 + c]), 0.f) : 0.f;
                            Al[c * g.CS + f] = x;
                        }
                    }
                }
                if (DG && jj == 0) {
                    if (T == 9) {
                        nf_cv_w_store<T, true>(wv, Wd, O, wid);
                    } else {                           // 1x1: row oc, column ic
                        for (int e = threadIdx.x; e < O * IC; e += NF_CV_THREADS) {
                            const int oc = e / IC, ic = e - oc * IC;
                            Wd[oc * NF_CV_WS + ic] = d.weight[oc * I + i0 + ic];
                        }
                    }
                    if (OP > O) nf_cv_zero_pad_rows<T>(Wd, OP, O, OP - O, NF_CV_WS);
                }
            }
            __syncthreads();
            NF_CV_STAMP(10);

            // ---- weight gradient of this chunk: a tile walks the 128 pixels two at a time (64 MFMAs), operands of the next
            //      four steps in flight (branch-free body, the one-past-the-end prefetch wraps around) ----
            f32x16 accW[NUW];
#pragma unroll
            for (int v = 0; v < NUW; ++v)
#pragma unroll
                for (int r = 0; r < 16; ++r) accW[v][r] = 0.f;
            if (WG) {
                float av0[4][NUW], bv0[4][NUW], av1[4][NUW], bv1[4][NUW];
#define NF_CV_WLOAD(A_, B_, J_)                                                                            \
    _Pragma("unroll") for (int jj = 0; jj < 4; ++jj)                                                       \
        _Pragma("unroll") for (int v = 0; v < NUW; ++v) {                                                  \
            const int fp = nf_cv_frame_of(g, (2 * ((J_) + jj) + hs) & (NF_CV_PX - 1));                     \
            A_[jj][v] = Gl[aoff[v] + fp];                                                                  \
            B_[jj][v] = Al[boff[v] + fp];                                                                  \
        }
#define NF_CV_WMFMA(A_, B_)                                                                                \
    _Pragma("unroll") for (int jj = 0; jj < 4; ++jj)                                                       \
        _Pragma("unroll") for (int v = 0; v < NUW; ++v)                                                    \
            if (uval[v]) {                                                                                 \
                if (ib == 0) gbw[v] += A_[jj][v];                                                          \
                accW[v] = __builtin_amdgcn_mfma_f32_32x32x2f32(A_[jj][v], B_[jj][v], accW[v], 0, 0, 0);    \
            }
                NF_CV_WLOAD(av0, bv0, 0);
                for (int j0 = 0; j0 < NF_CV_PX / 2; j0 += 8) {
                    NF_CV_WLOAD(av1, bv1, j0 + 4);
                    NF_CV_WMFMA(av0, bv0);
                    NF_CV_WLOAD(av0, bv0, j0 + 8);
                    NF_CV_WMFMA(av1, bv1);
                }
#undef NF_CV_WLOAD
#undef NF_CV_WMFMA
            }
            NF_CV_STAMP(11);
            // ---- data gradient of this input chunk: rows ic, columns pixel, K = (tap, oc) split in quarters; taps mirrored ----
            f32x16 accD[1];
#pragma unroll
            for (int r = 0; r < 16; ++r) accD[0][r] = 0.f;
            float xin[4];
            if (DG && d.gn_out != nullptr) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int ic = 8 * kq + rr + 4 * hs;
                    xin[rr] = (has_bn && pv && ic < IC) ? d.in[(pb64 * I + i0 + ic) * g.HW + pq] : 0.f;
                }
                const int ng = T * (OP >> 3);
                const int g0 = (kq * ng) >> 2, g1 = ((kq + 1) * ng) >> 2;
                nf_cv_kloop<T, 1>(accD, Wd, Gl, g, OP, NF_CV_WS, fpos, -1, c32, hs, g0, g1 - g0);
            }
            NF_CV_STAMP(12);
            // ---- this workgroup's slab of g_weff, in (T, O, I) order; first tile stores, later tiles add ----
#pragma unroll
            for (int v = 0; v < NUW; ++v)
                if (WG && uval[v]) {
                    const int tl = wid + NF_CV_WAVES * v;
                    const int tap = T == 9 ? tl : 0, ob = T == 9 ? 0 : tl;
                    const int ic = i0 + c32;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int oc = ob * 32 + nf_cv_cd_row(r, hs);
                        if (oc < O && c32 < IC) {
                            float* p = slab + (tap * O + oc) * I + ic;         // (T, O, I): consecutive lanes, consecutive floats
                            if (it == 0) *p = accW[v][r];
                            else *p += accW[v][r];
                        }
                    }
                }
            float own[4] = {0.f, 0.f, 0.f, 0.f};
            if (DG && d.gn_out != nullptr) {           // block-uniform
                __syncthreads();                       // every wave is done with Wd / Al / Gl reads of this chunk
                nf_cv_quarter_exchange(own, accD[0], X, pb, kq, lane);
            }
            if (DG && d.gn_out != nullptr) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int ic = 8 * kq + rr + 4 * hs;
                    if (pv && ic < IC) {
                        float gn = own[rr];
                        if (has_bn) {
                            const float x = xin[rr];
                            gn = fmaf(x, kc[ic], kc[32 + ic]) > 0.f ? gn : 0.f;
                            sg[rr] += gn;
                            sgx[rr] = fmaf(gn, (x - kc[64 + ic]) * kc[96 + ic], sgx[rr]);
                        }
                        d.gn_out[(pb64 * I + i0 + ic) * g.HW + pq] = gn;
                    }
                }
            }
        }
    }
    NF_CV_STAMP(13);
    // ---- bias and BatchNorm sums ----
    NF_DET_REPL_CHAIN(NF_STAT_REPL);                   // (blockIdx.y = the layer of a multi-launch: chains per layer and replica)
    NF_DET_ENTER_ALL_K(nf_cvb);                        // (no two threads of the workgroup add to one address; the turn covers both groups)
    if (WG && d.g_bias != nullptr) {
#pragma unroll
        for (int v = 0; v < NUW; ++v) {
            const int tl = wid + NF_CV_WAVES * v;
            if (uval[v] && (T == 1 || tl == 0)) {      // 3x3: the tile of tap 0; 1x1: every output block
                const int ob = T == 9 ? 0 : tl;
                const float t = gbw[v] + __shfl_xor(gbw[v], 32, NF_WAVE);
                const int oc = ob * 32 + c32;
                if (hs == 0 && oc < O) atomicAdd(d.g_bias + 256 * (blockIdx.x % NF_STAT_REPL) + oc, t);
            }
        }
    }
    if (DG && has_bn && d.sum_g != nullptr) {          // block-uniform
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const float t1 = nf_cv_half_sum(sg[rr]), t2 = nf_cv_half_sum(sgx[rr]);
            if (c32 == 0) {
                const int ic = 8 * kq + rr + 4 * hs;
                red[(0 * 4 + pb) * 32 + ic] = t1;
                red[(1 * 4 + pb) * 32 + ic] = t2;
            }
        }
        __syncthreads();
        if (threadIdx.x < 64 && c32 < I) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) t += red[(hs * 4 + w) * 32 + c32];
            const int rep = 32 * (blockIdx.x % NF_STAT_REPL);
            atomicAdd((hs == 0 ? d.sum_g : d.sum_gx) + rep + c32, t);
        }
    }
    NF_DET_LEAVE_ALL_K(nf_cvb);
    NF_CV_STAMP(14);
}


// ---------------------------------------------------------------------------------------------------------------
// The deferred weight-gradient pass of a 3x3 layer with <= 32 input channels (every hidden layer of the conditioner): MODE 2 above,
// restated around what bounded it.  Measured on MODE 2 (16 x 16, phase stamps): a workgroup was ~20 us old per tile for 3.8 us
// of MFMA work -- three frame trips of ten loads, one memory round trip each, behind the constants' own round trip; then nine of the
// sixteen waves walked one tap each (three of them on one SIMD: 5.1 us of matrix pipe there, the other seven waves idle).  Here
//   * everything a tile needs is requested in ONE batch -- BatchNorm constants (threads < 64, once), the activation frame
//     (<= 12 values per thread), and per OWNED pixel (the tap tiles read G at owned pixels only) four channels of each tensor G
//     is assembled from --, and the batch of the NEXT tile is issued in front of this tile's MFMA walk;
//   * the walk is cut into 144 units (tap, 8 pixels) and every wave takes nine: 36 MFMAs each, four waves per SIMD.  A wave's
//     units touch at most two taps -> two accumulators, kept over all tiles of the workgroup;
//   * at the end the partial tiles meet in LDS in a FIXED order (a tap has <= 3 pieces: piece 0 stores, 1 and 2 add, a barrier in
//     between: deterministic) and leave as the workgroup's slab in whole 128-byte rows.
// ---------------------------------------------------------------------------------------------------------------
// base + 32-bit BYTE offset (an element index is scaled in 64 bits and costs a VGPR pair per load; the host routes tensors beyond
// 2^30 elements to the MODE 2 body)
__device__ __forceinline__ float nf_cv_ld32(const float* __restrict__ base, unsigned byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
struct NfCvWgTile {                                    // one tile's operands in flight
    float w1[4], w2[4], w3[4], w4[4];
    float xa[NF_CV_FJ][NF_CV_CU];
    unsigned okm;                                      // bit 2 jj + u: xa[jj][u] is inside the image and the layer's channels; bit 31: G pixel valid
};
// What of a tile's addresses does not depend on the tile.  Decoding six frame positions and scaling 28 element indices per tile was
// ~3 us of VALU time per wave -- four waves per SIMD: more than the tile's MFMAs; per tile what is left is a scalar base per tensor,
// one compare per frame position and the loads.
struct NfCvWgInv {
    unsigned ginv;                                     // G: byte offset of (sample-in-tile, channel gq, pixel) from the tile's base
    unsigned xinv[NF_CV_FJ];                           // frame position jj, channel wid: byte offset from (tile base - one row - 1)
    unsigned xcls;                                     // 2 bits per jj: 0 never inside, 1 inside, 2 top halo row, 3 bottom halo row
    unsigned xs;                                       // 5 bits per jj: sample of the tile (tiles of several samples)
};
__device__ __forceinline__ void nf_cv_wg_invariants(NfCvWgInv& v, const NfCvGeo& g, int I, int O, int gp, int gq, int wid, int lane) {
    const bool multi = g.HW < NF_CV_PX;                // several whole samples per tile
    const int sgp = multi ? gp >> g.lgHW : 0, qgp = multi ? gp & (g.HW - 1) : gp;
    v.ginv = 4u * (unsigned)((sgp * O + gq) * g.HW + qgp);
    v.xcls = 0u; v.xs = 0u;
#pragma unroll
    for (int jj = 0; jj < NF_CV_FJ; ++jj) {
        const int f = lane + NF_WAVE * jj;
        unsigned off = 0u, cls = 0u, sj = 0u;
        if (jj < g.nfj && f < g.FSZ) {
            const int sm = (int)(((float)f + 0.5f) * g.invFS), q = f - sm * g.FS;            // (as nf_cv_decode)
            const int fy = (int)(((float)q + 0.5f) * g.invFW), fx = q - fy * g.FW;
            const int gx = fx - g.halo, fyh = fy - g.halo;
            if (gx >= 0 && gx < g.W) {
                cls = fyh < 0 ? 2u : (fyh >= g.TH ? 3u : 1u);
                off = 4u * (unsigned)((sm * I + wid) * g.HW + (fyh + 1) * g.W + gx + 1);
                sj = (unsigned)sm;
            }
        }
        v.xinv[jj] = off;
        v.xcls |= cls << (2 * jj);
        v.xs |= sj << (5 * jj);
    }
}
__device__ __forceinline__ void nf_cv_wg_issue(NfCvWgTile& t, const NfCvWgInv& v, const nf_conv_bwd_desc& d, const NfCvGeo& g, int I, int O,
                                               int64_t tile, int gp, int gq, int wid) {
    const int64_t P0t = tile * NF_CV_PX;
    const int64_t b0 = P0t >> g.lgHW;
    const int q0 = (int)(P0t & (g.HW - 1));            // first pixel of the tile in its sample (0 for tiles of whole samples)
    const int y0t = g.SEG == 1 ? q0 >> g.lgW : 0;
    const bool pv = P0t + gp < g.B * g.HW;
    // (unconditional loads from clamped offsets, selected afterwards: a predicated load is a branch, and a branch ends the batch)
    const int64_t gbase = b0 * O * g.HW + q0;
    const unsigned cstep = 4u * 8u * (unsigned)g.HW;   // channels gq + 8 k
    unsigned gi[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) gi[k] = (pv && gq + 8 * k < O) ? v.ginv + (unsigned)k * cstep : 0u;
    if (d.g_direct != nullptr) {                       // (uniform)
        const float* bp = d.g_direct + gbase;
#pragma unroll
        for (int k = 0; k < 4; ++k) t.w1[k] = nf_cv_ld32(bp, gi[k]);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) t.w1[k] = 0.f;
    }
    if (d.g_skip != nullptr) {
        const float* bp = d.g_skip + gbase;
#pragma unroll
        for (int k = 0; k < 4; ++k) t.w2[k] = nf_cv_ld32(bp, gi[k]);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) t.w2[k] = 0.f;
    }
    if (d.gn_src != nullptr) {
        const float* bp3 = d.gn_src + gbase;
        const float* bp4 = d.out + gbase;
#pragma unroll
        for (int k = 0; k < 4; ++k) { t.w3[k] = nf_cv_ld32(bp3, gi[k]); t.w4[k] = nf_cv_ld32(bp4, gi[k]); }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) { t.w3[k] = 0.f; t.w4[k] = 0.f; }
    }
    // activation frame: one position per lane and trip, every trip in flight
    const float* xb = d.in + b0 * I * g.HW + (y0t - 1) * g.W - 1;             // (offsets carry + one row + 1: never negative)
    const unsigned safe = 4u * (unsigned)(g.W + 1);    // -> element 0 of the tile's first sample
    const bool top = y0t > 0, bot = y0t + g.TH < g.H;
    const unsigned ustep = 4u * (unsigned)NF_CV_WAVES * (unsigned)g.HW;
    unsigned okm = pv ? 1u << 31 : 0u;
#pragma unroll
    for (int jj = 0; jj < NF_CV_FJ; ++jj) {
        const unsigned cls = (v.xcls >> (2 * jj)) & 3u, sj = (v.xs >> (5 * jj)) & 31u;
        const bool in = (cls == 1u || (cls == 2u && top) || (cls == 3u && bot)) && b0 + (int64_t)sj < g.B;
#pragma unroll
        for (int u = 0; u < NF_CV_CU; ++u) {
            const bool ok = in && wid + u * NF_CV_WAVES < I;
            okm |= ok ? 1u << (2 * jj + u) : 0u;
            t.xa[jj][u] = nf_cv_ld32(xb, ok ? v.xinv[jj] + (unsigned)u * ustep : safe);
        }
    }
    t.okm = okm;
}
// four steps (8 pixels) of a tap tile: A = G rows (oc) at the pixels, B = activation rows (ic) at the pixels shifted by the tap
template <bool BIAS>
__device__ __forceinline__ void nf_cv_wg_unit(f32x16& acc, float& gsum, const float* Gl, const float* Al, const NfCvGeo& g, int aoff,
                                              int boff, int sgrp, int hs) {
    float av[4], bv[4];
    if (g.W >= 8) {                                    // the unit's 8 pixels share a row: one frame position, immediates (uniform)
        const int fp = nf_cv_frame_of(g, 8 * sgrp) + hs;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            av[jj] = Gl[aoff + fp + 2 * jj];
            bv[jj] = Al[boff + fp + 2 * jj];
        }
    } else {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int fp = nf_cv_frame_of(g, 2 * (4 * sgrp + jj) + hs);
            av[jj] = Gl[aoff + fp];
            bv[jj] = Al[boff + fp];
        }
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        if (BIAS) gsum += av[jj];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj], bv[jj], acc, 0, 0, 0);
    }
}
__device__ __forceinline__ float nf_cv_uniform(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
// T = 9: the 3x3 hidden layers.  T = 1: a 1x1 layer of <= 32 output channels (the output convolution of the 16 x 16 and 8 x 8 levels):
// one tile, sixteen units, one per wave -- sixteen pieces meet at the end.
template <int T>
__device__ __forceinline__ void nf_cv_wgrad3_body(const nf_conv_bwd_desc& d, const NfCvGeo& g, int I, int O, int iters) {
    static_assert(T == 9 || T == 1, "taps");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // two (activation, G) frame pairs: pair it & 1 is walked while pair (it + 1) & 1 is filled -- ONE barrier per tile, and no barrier
    // between a wave's transforms and its walk: half of a SIMD's waves walk first and fill afterwards, so that their VALU work
    // runs under the other half's MFMAs (phase-aligned behind two barriers per tile the walk was 6.5 of a tile's 10.4 us)
    const int FP = 64 * g.CS;                          // floats per pair: Al [32][CS] | Gl [32][CS]
    float* cb = smem + 2 * FP;                         // [5][32]
    float* kc = cb + 5 * 32;                           // [2][32]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, c32 = lane & 31, hs = lane >> 5;
    const bool has_bn = d.bn_gamma != nullptr;
    const bool has_src = d.gn_src != nullptr;
    const float invN = 1.f / (float)(g.B * g.HW);
    NF_CV_STAMP(8);

    // constants: requested now, written to LDS in front of the first barrier (behind the first tile's loads in program order)
    float k0 = 0.f, k1 = 0.f, k2 = 0.f, k3 = 0.f, k4 = 0.f;
    if (threadIdx.x < 32) {
        const int oo = threadIdx.x;
        if (has_src && oo < O) {
            k2 = d.cbn_save_invstd[oo];
            k1 = d.cbn_save_mean[oo];
            k0 = d.cbn_gamma[oo];
            if (d.cbn_sum_g != nullptr) {
#pragma unroll
                for (int r = 0; r < NF_STAT_REPL; ++r) { k3 += d.cbn_sum_g[32 * r + oo]; k4 += d.cbn_sum_gx[32 * r + oo]; }
            }
        }
    } else if (threadIdx.x < 64) {
        const int k = threadIdx.x - 32;
        if (has_bn && k < I) {
            k2 = d.bn_save_mean[k];
            k3 = d.bn_save_invstd[k];
            k0 = d.bn_gamma[k];
            k1 = d.bn_beta[k];
        }
    }
    // the wave's nine units 9 wid .. 9 wid + 8 of (tap = unit >> 4, pixel group = unit & 15): nA of them in tap tA, the rest in tA + 1
    const int u0 = T * wid, tA = u0 >> 4, sA = u0 & 15;
    const int nA = min(T, 16 - sA), nB = T - nA;
    const int dyA = T == 9 ? tA / 3 - 1 : 0, dxA = T == 9 ? tA - (tA / 3) * 3 - 1 : 0;
    const int tB = tA + 1 < T ? tA + 1 : tA;           // (wave 15 ends on the last unit of the last tap: nB = 0)
    const int dyB = T == 9 ? tB / 3 - 1 : 0, dxB = T == 9 ? tB - (tB / 3) * 3 - 1 : 0;
    const int aoff = c32 * g.CS;
    const int boffA = c32 * g.CS + dyA * g.FW + dxA, boffB = c32 * g.CS + dyB * g.FW + dxB;
    f32x16 accA, accB;
#pragma unroll
    for (int r = 0; r < 16; ++r) { accA[r] = 0.f; accB[r] = 0.f; }
    float gbw = 0.f;                                   // bias gradient: G summed over the pixels, by the waves that walk tap 0
    const int gp = threadIdx.x & (NF_CV_PX - 1), gq = threadIdx.x >> 7;        // G: pixel gp, channels gq + 8 k
    const int gfpos = nf_cv_frame_of(g, gp);

    NfCvWgInv inv;
    nf_cv_wg_invariants(inv, g, I, O, gp, gq, wid, lane);
    NfCvWgTile tl;
    nf_cv_wg_issue(tl, inv, d, g, I, O, blockIdx.x, gp, gq, wid);               // (grid <= tiles: the first tile exists)
    if (threadIdx.x < 32) {
        const int oo = threadIdx.x;
        cb[oo] = k0 * k2; cb[32 + oo] = k1; cb[64 + oo] = k2; cb[96 + oo] = k3 * invN; cb[128 + oo] = k4 * invN;
    } else if (threadIdx.x < 64) {
        const int k = threadIdx.x - 32;
        const float sc = has_bn ? k0 * k3 : 1.f;
        kc[k] = sc; kc[32 + k] = has_bn ? k1 - k2 * sc : 0.f;
    }
    __syncthreads();
    // a wave's channels are the same for all its lanes (G: gq + 8 k; activations: wid + 16 u): their constants live in scalar
    // registers for the whole launch (from LDS per tile they were 104 reads per thread)
    float c1[4], cmean[4], cinv[4], cmg[4], cmgx[4], xsc[NF_CV_CU], xsh[NF_CV_CU];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = gq + 8 * k;
        c1[k] = nf_cv_uniform(cb[c]); cmean[k] = nf_cv_uniform(cb[32 + c]); cinv[k] = nf_cv_uniform(cb[64 + c]);
        cmg[k] = nf_cv_uniform(cb[96 + c]); cmgx[k] = nf_cv_uniform(cb[128 + c]);
    }
#pragma unroll
    for (int u = 0; u < NF_CV_CU; ++u) {
        const int c = wid + u * NF_CV_WAVES;
        xsc[u] = nf_cv_uniform(kc[c]); xsh[u] = nf_cv_uniform(kc[32 + c]);
    }
    // G of the owned pixels and the activation frame of the tile in flight -> frame pair pr
    auto fill = [&](int pr) {
        float* Alp = smem + pr * FP;
        float* Glp = Alp + 32 * g.CS;
        const bool pv = (tl.okm >> 31) != 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = gq + 8 * k;
            float v = 0.f;
            if (pv && c < O) {
                v = tl.w1[k] + tl.w2[k];
                if (has_src) {                         // BatchNorm backward on load
                    const float xh = (tl.w4[k] - cmean[k]) * cinv[k];
                    v += c1[k] * (tl.w3[k] - cmg[k] - xh * cmgx[k]);
                }
            }
            Glp[c * g.CS + gfpos] = v;
        }
        const int ICP = (min(32, I) + 15) & ~15;
#pragma unroll
        for (int jj = 0; jj < NF_CV_FJ; ++jj) {
            const int f = lane + NF_WAVE * jj;
            if (jj < g.nfj && f < g.FSZ) {
#pragma unroll
                for (int u = 0; u < NF_CV_CU; ++u) {
                    const int c = wid + u * NF_CV_WAVES;
                    if (c < ICP) {
                        float x = tl.xa[jj][u];
                        if (!((tl.okm >> (2 * jj + u)) & 1u)) x = 0.f;         // outside the image / beyond the layer's channels
                        else if (has_bn) x = fmaxf(fmaf(x, xsc[u], xsh[u]), 0.f);
                        Alp[c * g.CS + f] = x;
                    }
                }
            }
        }
    };
    auto walk = [&](int pr) {
        const float* Alp = smem + pr * FP;
        const float* Glp = Alp + 32 * g.CS;
        for (int un = 0; un < nA; ++un) {
            if (tA == 0) nf_cv_wg_unit<true>(accA, gbw, Glp, Alp, g, aoff, boffA, sA + un, hs);
            else nf_cv_wg_unit<false>(accA, gbw, Glp, Alp, g, aoff, boffA, sA + un, hs);
        }
        for (int un = 0; un < nB; ++un) nf_cv_wg_unit<false>(accB, gbw, Glp, Alp, g, aoff, boffB, un, hs);
    };
    const int64_t tile0 = blockIdx.x, tstep = gridDim.x;
    fill(0);
    if (1 < iters && tile0 + tstep < g.tiles) nf_cv_wg_issue(tl, inv, d, g, I, O, tile0 + tstep, gp, gq, wid);
    const bool fill_first = ((wid >> 2) & 1) == 0;     // two of a SIMD's four waves (wave w sits on SIMD w & 3)
    for (int it = 0; it < iters; ++it) {
        if (it == 1) NF_CV_STAMP(15);
        if (it == 2) NF_CV_STAMP(14);
        const int64_t tile = (int64_t)it * tstep + tile0;
        if (tile >= g.tiles) break;                    // block-uniform
        __syncthreads();                               // pair it & 1 is complete; the walks of the other pair are done
        const bool next = it + 1 < iters && tile + tstep < g.tiles;              // its batch is in flight
        const bool next2 = it + 2 < iters && tile + 2 * tstep < g.tiles;
        if (fill_first) {
            if (next) fill((it + 1) & 1);
            if (next2) nf_cv_wg_issue(tl, inv, d, g, I, O, tile + 2 * tstep, gp, gq, wid);
            walk(it & 1);
        } else {
            walk(it & 1);
            if (next) fill((it + 1) & 1);
            if (next2) nf_cv_wg_issue(tl, inv, d, g, I, O, tile + 2 * tstep, gp, gq, wid);
        }
    }
    // ---- the partial tiles meet: red[tap][oc][ic] over the frames, pieces in fixed order ----
    float* red = smem;                                 // 9 x 1024 floats <= 64 CS (host-checked)
    const int pieceA = wid - (16 * tA) / T, pieceB = wid - (16 * tB) / T;     // waves before this one that walk the same tap
    constexpr int NPASS = T == 9 ? 3 : 16;
#pragma unroll 1
    for (int pass = 0; pass < NPASS; ++pass) {
        __syncthreads();                               // (first: the last walk is done with the frames)
        if (pieceA == pass) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* q = red + tA * 1024 + nf_cv_cd_row(r, hs) * 32 + c32;
                *q = pass == 0 ? accA[r] : *q + accA[r];
            }
        }
        if (nB > 0 && pieceB == pass) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* q = red + tB * 1024 + nf_cv_cd_row(r, hs) * 32 + c32;
                *q = pass == 0 ? accB[r] : *q + accB[r];
            }
        }
    }
    __syncthreads();
    float* slab = d.g_weff + (int64_t)blockIdx.x * O * I * T;                  // (T, O, I): whole rows, consecutive lanes
    for (int e = threadIdx.x; e < T * 1024; e += NF_CV_THREADS) {
        const int tap = e >> 10, oc = (e >> 5) & 31, ic = e & 31;
        if (oc < O && ic < I) slab[(tap * O + oc) * I + ic] = red[e];
    }
    {
        NF_DET_REPL_CHAIN(NF_STAT_REPL);
        NF_DET_ENTER_ALL_K(nf_cvb);                    // waves 0 and 1 walked tap 0 and add to the same addresses: in wave order in the mode
        const float t = gbw + __shfl_xor(gbw, 32, NF_WAVE);
        nf_det_waves(nf_det_, NF_CV_THREADS / NF_WAVE, wid, [&] {
            if (d.g_bias != nullptr && tA == 0 && hs == 0 && c32 < O) atomicAdd(d.g_bias + 256 * (blockIdx.x % NF_STAT_REPL) + c32, t);
        });
        NF_DET_LEAVE_ALL_K(nf_cvb);
    }
}

template <int T, int ICB, int OCB, int MODE>
__global__ void __launch_bounds__(NF_CV_THREADS) k_conv_bn_bwd(nf_conv_bwd_desc d, NfCvGeo g, int I, int O, int iters) {
    nf_cv_bwd_body<T, ICB, OCB, MODE>(d, g, I, O, iters);
}
template <int T, int ICB, int OCB, bool LEAN>
__global__ void __launch_bounds__(NF_CV_THREADS) k_conv_bn_wgrad_multi(NfCvBwdMulti m, const nf_conv_bwd_desc* __restrict__ tab, NfCvGeo g,
                                                                       int I, int O, int iters) {
    // tab != NULL: the layers' descriptors are a TABLE in device memory (nf_conv_bn_wgrad_table: any number of layers per launch)
    const nf_conv_bwd_desc& d = tab != nullptr ? tab[blockIdx.y] : m.d[blockIdx.y];
    if constexpr (LEAN) nf_cv_wgrad3_body<T>(d, g, I, O, iters);
    else nf_cv_bwd_body<T, ICB, OCB, 2>(d, g, I, O, iters);
}

#define NF_CV_BWD_MAX_SLABS 128
extern "C" int nf_conv_bwd_slabs(int64_t B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const int64_t tiles = (B * H * W + NF_CV_PX - 1) / NF_CV_PX;
    return (int)(tiles < NF_CV_BWD_MAX_SLABS ? tiles : NF_CV_BWD_MAX_SLABS);
}

// Slabs per layer of a deferred weight-gradient launch of n_layers layers: one workgroup per compute unit over the whole launch
// (a workgroup keeps its tap tiles in registers over its tiles and pays launch, constants and the slab once; more, shorter-lived
// workgroups measured slower: 16 layers at 16 x 16, 128 / 32 / 16 / 8 slabs: 123 / 94 / 81 / 198 us).  (The overrides of rounds 2 - 4, NF_CONV_WGRAD_SLABS / _BLOCKS, are gone: no setting beat this rule.)
extern "C" int nf_conv_wgrad_slabs(int64_t B, int H, int W, int n_layers) {
    if (B <= 0 || H <= 0 || W <= 0 || n_layers < 1) return 0;
    const int64_t tiles = (B * H * W + NF_CV_PX - 1) / NF_CV_PX;
    const int cap = 256;                               // workgroups of the whole launch: one per compute unit
    int64_t want = cap / n_layers;
    if (want < 1) want = 1;
    if (want > NF_CV_BWD_MAX_SLABS) want = NF_CV_BWD_MAX_SLABS;
    return (int)(tiles < want ? tiles : want);
}

extern "C" int nf_conv_bn_bwd(const nf_conv_bwd_desc* desc, int64_t B, int I, int O, int H, int W, int ksize,
                              nf_stream_t stream) {
    NfCvGeo g;
    if (desc == nullptr || !nf_conv_bn_usable(B > 0 ? B : 1, I, O, H, W, ksize)) return NF_E_BADARG;
    if (desc->g_weff == nullptr && desc->g_bias != nullptr) return NF_E_BADARG;   // the bias sums belong to the weight-gradient pass
    if (B == 0) return 0;
    if (!nf_cv_geometry(g, B, H, W, ksize)) return NF_E_BADARG;
    if (desc->bn_gamma != nullptr && I > 32) return NF_E_BADARG;
    if (desc->gn_src != nullptr && O > 32) return NF_E_BADARG;            // consumer BatchNorm sums are 32 wide
    if (!nf_cv_fits_int32(B, I > O ? I : O, H * W)) return NF_E_BADARG;
    if (!nf_cv_set_valid(g, desc->valid_h, desc->valid_w)) return NF_E_BADARG;
    const bool masked = nf_cv_masked(g);
    if (!masked && nf_conv_bulk_bwd_plan(desc, B, I, O, H, W, ksize))          // large batches, data pass: conv_bulk.hip
        return nf_conv_bulk_bwd(desc, B, I, H, W, (hipStream_t)stream);
    if (!masked && nf_conv1_bulk_bwd_plan(desc, B, I, O, H, W, ksize))
        return nf_conv1_bulk_bwd(desc, B, O, H, W, (hipStream_t)stream);
    const int ICB = (I + 31) / 32, OCB = (O + 31) / 32;
    const int T = ksize * ksize;
    size_t body = (size_t)T * 32 * OCB * NF_CV_WS + (size_t)32 * g.CS + (size_t)32 * OCB * g.CS;
    if (body < NF_CV_RS) body = NF_CV_RS;             // the K-quarter exchange aliases the start of the tiles
    const size_t lds = sizeof(float) * (body + 5 * 32 + 4 * 32 + 2 * 4 * 32);
    const unsigned grid = (unsigned)nf_conv_bwd_slabs(B, H, W);
    const int iters = (int)((g.tiles + grid - 1) / grid);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    const bool dgrad_only = desc->g_weff == nullptr;  // MODE 1: the weight gradient follows in nf_conv_bn_wgrad_multi
#define NF_LAUNCH(T_, IB_, OB_)                                                                                        \
    do {                                                                                                               \
        if (dgrad_only) {                                                                                              \
            rc = nf_cv_optin(k_conv_bn_bwd<T_, IB_, OB_, 1>, lds);                                                     \
            if (rc) return rc;                                                                                         \
            hipLaunchKernelGGL((k_conv_bn_bwd<T_, IB_, OB_, 1>), dim3(grid), dim3(NF_CV_THREADS), lds, st, *desc, g, I, O, iters); \
        } else {                                                                                                       \
            rc = nf_cv_optin(k_conv_bn_bwd<T_, IB_, OB_, 0>, lds);                                                     \
            if (rc) return rc;                                                                                         \
            hipLaunchKernelGGL((k_conv_bn_bwd<T_, IB_, OB_, 0>), dim3(grid), dim3(NF_CV_THREADS), lds, st, *desc, g, I, O, iters); \
        }                                                                                                              \
    } while (0)
    if (T == 9) {
        if (ICB == 1) NF_LAUNCH(9, 1, 1);
        else if (ICB == 2) NF_LAUNCH(9, 2, 1);
        else NF_LAUNCH(9, 3, 1);
    } else {
        switch (OCB) {
            case 1: NF_LAUNCH(1, 1, 1); break;
            case 2: NF_LAUNCH(1, 1, 2); break;
            case 3: NF_LAUNCH(1, 1, 3); break;
            case 4: NF_LAUNCH(1, 1, 4); break;
            case 5: NF_LAUNCH(1, 1, 5); break;
            default: NF_LAUNCH(1, 1, 6); break;
        }
    }
#undef NF_LAUNCH
    NF_CHECK_LAUNCH();
    return 0;
}

// a chunk of descriptors from the kernel arguments into a device table (nf_conv_bn_wgrad_table): by-value arguments are part of a captured
// launch, so a hipGraph replays the table's construction without re-reading host memory
#define NF_CV_DESC_WRITE 320                            // descriptors per writer launch: 320 x 184 B = 58 KB of kernel arguments (64 KB are taken)
struct NfCvDescChunk { nf_conv_bwd_desc d[NF_CV_DESC_WRITE]; };
static_assert(sizeof(NfCvDescChunk) + 16 <= 60 * 1024, "kernel-argument segment");
__global__ void k_conv_desc_write(NfCvDescChunk m, nf_conv_bwd_desc* __restrict__ dst, int cnt) {
    const unsigned* src = reinterpret_cast<const unsigned*>(&m);
    unsigned* out = reinterpret_cast<unsigned*>(dst);
    const int words = cnt * (int)(sizeof(nf_conv_bwd_desc) / sizeof(unsigned));
    for (int i = threadIdx.x; i < words; i += blockDim.x) out[i] = src[i];
}
static_assert(sizeof(nf_conv_bwd_desc) % sizeof(unsigned) == 0, "descriptor: whole words");

// descs: n descriptors on the host (sanitised in place).  tab == NULL: n <= NF_CV_WG_MAX, they travel in the kernel arguments.  tab != NULL:
// the table form -- written to `tab` sixteen per tiny launch in front of the pass, then read by the workgroups.
static int nf_conv_bn_wgrad_launch(nf_conv_bwd_desc* descs, const nf_conv_bwd_desc* tab, int n, int slabs, int64_t B, int I, int O, int H,
                                   int W, int ksize, nf_stream_t stream) {
    NfCvGeo g;
    if (descs == nullptr || n < 1 || (tab == nullptr && n > NF_CV_WG_MAX) || n > NF_CONV_WGRAD_TABLE_MAX ||
        !nf_conv_bn_usable(B > 0 ? B : 1, I, O, H, W, ksize))
        return NF_E_BADARG;
    if (B == 0) return 0;
    if (!nf_cv_geometry(g, B, H, W, ksize)) return NF_E_BADARG;
    if (!nf_cv_fits_int32(B, I > O ? I : O, H * W)) return NF_E_BADARG;
    if (!nf_cv_set_valid(g, descs[0].valid_h, descs[0].valid_w)) return NF_E_BADARG;      // (one shape per launch: layer 0's extent)
    const bool masked = nf_cv_masked(g);
    bool one_plain = true;                             // (conv_bulk.hip keeps ONE plain gradient tensor per layer in flight)
    for (int k = 0; k < n; ++k) {
        if (descs[k].valid_h != descs[0].valid_h || descs[k].valid_w != descs[0].valid_w) return NF_E_BADARG;
        if (descs[k].g_weff == nullptr || descs[k].in == nullptr) return NF_E_BADARG;
        if (descs[k].bn_gamma != nullptr && I > 32) return NF_E_BADARG;
        if (descs[k].gn_src != nullptr && O > 32) return NF_E_BADARG;
        descs[k].g_store = nullptr; descs[k].gn_out = nullptr; descs[k].sum_g = nullptr; descs[k].sum_gx = nullptr;   // the data pass did those
        one_plain = one_plain && !(descs[k].g_direct != nullptr && descs[k].g_skip != nullptr);
    }
    NfCvBwdMulti m{};
    hipStream_t st = (hipStream_t)stream;
    if (tab == nullptr) {
        for (int k = 0; k < n; ++k) m.d[k] = descs[k];
    } else {
        for (int k0 = 0; k0 < n; k0 += NF_CV_DESC_WRITE) {
            const int cnt = n - k0 < NF_CV_DESC_WRITE ? n - k0 : NF_CV_DESC_WRITE;
            NfCvDescChunk w{};
            for (int k = 0; k < cnt; ++k) w.d[k] = descs[k0 + k];
            hipLaunchKernelGGL(k_conv_desc_write, dim3(1), dim3(256), 0, st, w, const_cast<nf_conv_bwd_desc*>(tab) + k0, cnt);
        }
        m = NfCvBwdMulti{};
    }
    const int ICB = (I + 31) / 32, OCB = (O + 31) / 32;
    const int T = ksize * ksize;
    const size_t lds = sizeof(float) * ((size_t)32 * g.CS + (size_t)32 * OCB * g.CS + 5 * 32 + 4 * 32 + 2 * 4 * 32);
    if (slabs < 1 || slabs > NF_CV_BWD_MAX_SLABS || slabs > g.tiles) return NF_E_BADARG;
    const unsigned grid = (unsigned)slabs;
    if (!masked && one_plain && nf_conv_bulk_wgrad_plan(B, I, O, H, W, ksize))  // large batches: the pixel-contraction kernel of conv_bulk.hip (bf16 matrix pipe)
        return nf_conv_bulk_wgrad(tab == nullptr ? m.d : nullptr, tab, n, B, I, H, W, (int)grid, st);
    const int iters = (int)((g.tiles + grid - 1) / grid);
    int rc;
    // the lean body addresses with 32-bit byte offsets (3x3, one input chunk: every hidden layer)
    const size_t lds_lean = sizeof(float) * ((size_t)2 * 64 * g.CS + 5 * 32 + 2 * 32);       // two frame pairs
    const bool lean = !masked && ICB == 1 && OCB == 1 && B * (int64_t)(I > O ? I : O) * H * W < ((int64_t)1 << 30) && 64 * g.CS >= T * 1024 &&
                      lds_lean <= 160 * 1024;      // (the lean body classifies frame rows per tile, not per pixel: masked maps take the general one)
#define NF_LAUNCH(T_, IB_, OB_, LEAN_)                                                                                 \
    do {                                                                                                               \
        const size_t lds_ = LEAN_ ? lds_lean : lds;                                                                    \
        rc = nf_cv_optin(k_conv_bn_wgrad_multi<T_, IB_, OB_, LEAN_>, lds_);                                            \
        if (rc) return rc;                                                                                             \
        hipLaunchKernelGGL((k_conv_bn_wgrad_multi<T_, IB_, OB_, LEAN_>), dim3(grid, (unsigned)n), dim3(NF_CV_THREADS), lds_, st, m, \
                           tab, g, I, O, iters);                                                                       \
    } while (0)
    if (T == 9) {
        if (ICB == 1 && lean) NF_LAUNCH(9, 1, 1, true);
        else if (ICB == 1) NF_LAUNCH(9, 1, 1, false);
        else if (ICB == 2) NF_LAUNCH(9, 2, 1, false);
        else NF_LAUNCH(9, 3, 1, false);
    } else {
        switch (OCB) {
            case 1:
                if (lean) NF_LAUNCH(1, 1, 1, true);
                else NF_LAUNCH(1, 1, 1, false);
                break;
            case 2: NF_LAUNCH(1, 1, 2, false); break;
            case 3: NF_LAUNCH(1, 1, 3, false); break;
            case 4: NF_LAUNCH(1, 1, 4, false); break;
            case 5: NF_LAUNCH(1, 1, 5, false); break;
            default: NF_LAUNCH(1, 1, 6, false); break;
        }
    }
#undef NF_LAUNCH
    NF_CHECK_LAUNCH();
    return 0;
}
extern "C" int nf_conv_bn_wgrad_multi(const nf_conv_bwd_desc* descs, int n, int64_t B, int I, int O, int H, int W, int ksize,
                                      nf_stream_t stream) {
    if (descs == nullptr || n < 1 || n > NF_CV_WG_MAX) return NF_E_BADARG;
    nf_conv_bwd_desc local[NF_CV_WG_MAX];
    for (int k = 0; k < n; ++k) local[k] = descs[k];
    const int slabs = nf_conv_wgrad_slabs(B, H, W, n);
    return nf_conv_bn_wgrad_launch(local, nullptr, n, slabs > 0 ? slabs : 1, B, I, O, H, W, ksize, stream);
}
extern "C" int nf_conv_bn_wgrad_table(const nf_conv_bwd_desc* descs, nf_conv_bwd_desc* table_dev, int n, int slabs, int64_t B, int I, int O,
                                      int H, int W, int ksize, nf_stream_t stream) {
    if (table_dev == nullptr || descs == nullptr || n < 1 || n > NF_CONV_WGRAD_TABLE_MAX) return NF_E_BADARG;
    static thread_local std::vector<nf_conv_bwd_desc> local;                 // (the launcher sanitises its copy, not the caller's array)
    local.assign(descs, descs + n);
    return nf_conv_bn_wgrad_launch(local.data(), table_dev, n, slabs, B, I, O, H, W, ksize, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// dst[e] (+)= sum_s src[s * stride + e]: the slab / replica sums of one conditioner backward in one launch
// ---------------------------------------------------------------------------------------------------------------
// The grid is ONE dimension cut into per-job ranges (first[j] .. first[j + 1]): a job gets the workgroups ITS size asks for.  (As a
// (max blocks, jobs) grid a launch that mixed a 387 k-element weight gradient with 79 small jobs started 82 k workgroups, 80 k of them
// with nothing to do: 43 us per launch in the image Flow++ step.)
struct NfSlabArgs { nf_slab_sum_desc d[NF_SLAB_SUM_MAX]; unsigned first[NF_SLAB_SUM_MAX + 1]; int n_jobs; };
static_assert(sizeof(NfSlabArgs) < 60 * 1024, "kernel-argument segment");
__global__ void __launch_bounds__(NF_BLOCK) k_slab_sum(NfSlabArgs args) {
    int job = 0;
    for (int step = 1024; step > 0; step >>= 1)             // the last job whose first workgroup is <= blockIdx.x
        if (job + step < args.n_jobs && args.first[job + step] <= blockIdx.x) job += step;
    const nf_slab_sum_desc& d = args.d[job];
    const unsigned local = blockIdx.x - args.first[job], nblk = args.first[job + 1] - args.first[job];
    if (d.n <= 4 && d.taps <= 1 && d.n_slabs >= 64) {
        // a few elements over MANY slabs (per-workgroup partial sums of a scalar gradient: csrc/mixlog.hip): a wave per element, lane l
        // takes the slabs l, l + 64, ... and the lanes meet in a fixed order -- one thread walking 384 slabs was 24 us of dependent loads
        if (local != 0) return;
        const int lane = threadIdx.x & (NF_WAVE - 1);
        for (int64_t e = threadIdx.x >> 6; e < d.n; e += blockDim.x >> 6) {
            float t = 0.f;
            for (int sl = lane; sl < d.n_slabs; sl += NF_WAVE) t += d.src[(int64_t)sl * d.stride + e];
            t = nf_wave_sum(t);
            if (lane == 0) d.dst[e] = d.accumulate ? d.dst[e] + t : t;
        }
        return;
    }
    for (int64_t e = (int64_t)local * blockDim.x + threadIdx.x; e < d.n; e += (int64_t)nblk * blockDim.x) {
        int64_t se = e;                                 // dst is (O, I, T), the slabs are (T, O, I) when taps > 1
        if (d.taps > 1) {
            const int64_t oi = e / d.taps;
            const int tap = (int)(e - oi * d.taps);
            se = (int64_t)tap * (d.n / d.taps) + oi;
        }
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        int s = 0;
        for (; s + 8 <= d.n_slabs; s += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = d.src[(int64_t)(s + u) * d.stride + se];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u & 3] += v[u];
        }
        for (; s < d.n_slabs; ++s) t[0] += d.src[(int64_t)s * d.stride + se];
        const float v = (t[0] + t[1]) + (t[2] + t[3]);
        d.dst[e] = d.accumulate ? d.dst[e] + v : v;
    }
}

extern "C" int nf_slab_sum(const nf_slab_sum_desc* descs, int n_jobs, nf_stream_t stream) {
    if (descs == nullptr || n_jobs < 1 || n_jobs > NF_SLAB_SUM_MAX) return NF_E_BADARG;
    NfSlabArgs args;
    unsigned total = 0;
    for (int i = 0; i < n_jobs; ++i) {
        if (descs[i].src == nullptr || descs[i].dst == nullptr || descs[i].n < 0 || descs[i].n_slabs < 0) return NF_E_BADARG;
        args.d[i] = descs[i];
        args.first[i] = total;
        unsigned nb = nf_grid_for(descs[i].n > 0 ? descs[i].n : 1);
        if (nb > 512) nb = 512;                         // (a large job walks with a stride)
        total += nb;
    }
    args.first[n_jobs] = total;
    args.n_jobs = n_jobs;
    hipLaunchKernelGGL(k_slab_sum, dim3(total), dim3(NF_BLOCK), 0, (hipStream_t)stream, args);
    NF_CHECK_LAUNCH();
    return 0;
}
