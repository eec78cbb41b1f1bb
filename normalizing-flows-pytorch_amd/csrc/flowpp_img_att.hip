// The middle of the image Flow++ conditioner (gate -> LayerNorm -> GatedAttn -> LayerNorm, flows/modules.py:519-578) cut BY ATTENTION HEAD,
// for batches that do not fill the chip with one workgroup per sample (csrc/flowpp_img.hip: k_fi_mid holds B of 256 compute units, and its
// softmax sweeps -- 4 heads x N x N scores of 8-deep dot products on the vector ALUs -- are 80 % of its time at N = 256):
//     forward   k_fi_att_fwd  (B x 4 workgroups of 4 N threads)  gate, LayerNorm 1, this head's 24 rows of conv1, the two sweeps -> mixed, c_j
//               k_fi_post_fwd (B workgroups of 4 N threads)    conv2, gate, LayerNorm 2 -> out
//     backward  k_fi_post_bwd (B)      LayerNorm 2 and conv2 backward -> g3 (gradient of the attention block's residual input), g_mixed
//               k_fi_att_bwd  (B x 4)  the softmax backward sweeps, this head's rows of conv1's weight gradient, its part of g_tokens
//               k_fi_pre_bwd  (B)      sum of the four parts, position embedding, LayerNorm 1 and gate backward -> g_x, g_a
// Same arithmetic per element as k_fi_mid (the per-sample LayerNorm statistics are summed in a different order: rounding only); every
// kernel recomputes what it needs from x and a, the only tensors carried from the forward to the backward are `mixed` and c_j.
#include "nf_common.h"
#include "nf_det.h"

NF_DET_STATE(nf_fpa)
NF_DET_HOST_API(nf_fpa)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define NF_FA_LNEPS 1.0e-5f
#define NF_FA_SCALE 0.35355339059327373f                 // 1 / sqrt(D), D = 8 (flows/modules.py:571)
#define NF_FA_IDX(c, p) ((c) * N + (((p) + (c)) & (N - 1)))   // channel-major plane with a per-row rotation (see flowpp_img.hip)

__device__ __forceinline__ int nf_fa_cd_row(int r, int hs) { return (r & 3) + 8 * (r >> 2) + 4 * hs; }
__device__ __forceinline__ float nf_fa_elu(float v) { return v > 0.f ? v : expm1f(v); }
__device__ __forceinline__ float nf_fa_elu_grad(float v) { return v > 0.f ? 1.f : expf(v); }
__device__ __forceinline__ float nf_fa_sigmoid(float v) { return 1.f / (1.f + expf(-v)); }
__device__ __forceinline__ float nf_fa_dot8(const f32x4& a0, const f32x4& a1, const float* b) {
    return a0[0] * b[0] + a0[1] * b[1] + a0[2] * b[2] + a0[3] * b[3] + a1[0] * b[4] + a1[1] * b[5] + a1[2] * b[6] + a1[3] * b[7];
}

template <int NT>
__device__ __forceinline__ float nf_fa_block_sum_all(float v, float* scr) {
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

struct NfFiAtt {
    const float *x, *a, *ln1g, *ln1b, *pos, *w1, *b1, *w2, *b2, *ln2g, *ln2b;
    const float *mixed_in, *cj_in, *g_out, *g3_in, *g_mixed_in, *gt_in;
    float *mixed, *cj, *out, *g3, *g_mixed, *gt_part;
    float *g_x, *g_a, *g_ln1g, *g_ln1b, *g_pos, *g_w1, *g_b1, *g_w2, *g_b2, *g_ln2g, *g_ln2b;
    int g_slabs;
    int64_t g_slab_stride;
    int per_sample;      // the (32, H, W) parameter gradients (LayerNorm affines, position embedding) are WRITTEN per sample, (B, 32, H, W),
};                   // for nf_slab_sum to fold: as 1.5 M same-address atomics they were ~20 us of these 30-50 us kernels at B = 64

// ---- per (sample, head): 4 N threads, thread = (position j = tid >> 2, quarter qz = tid & 3) ----------------------------------------------
// The four quarter-threads of a position are adjacent lanes: they share the position's column of the score matrix -- each walks every
// fourth key -- and meet through two lane exchanges; in the prologue each owns 8 of the 32 token channels and 6 of the head's 24 conv1
// rows.  (With one thread per position the workgroup was 4 waves, one per SIMD, and the sweeps ran at the latency of their own
// LDS -> FMA chains: 45 / 76 us forward / backward at N = 256.)
// Wh[24][32]: rows 0..7 = "V" rows 8 h + r of conv1, 8..15 = "K" rows 32 + 8 h + r, 16..23 = "Q" rows 64 + 8 h + r;  Bh[24] their biases
template <int NT>
__device__ __forceinline__ void nf_fa_stage_head(float* Wh, float* Bh, const NfFiAtt& m, int h) {
    for (int e = threadIdx.x; e < 768; e += NT) {
        const int r = e >> 5, c = e & 31;
        Wh[e] = m.w1[((r >> 3) * 32 + 8 * h + (r & 7)) * 32 + c];
    }
    for (int e = threadIdx.x; e < 24; e += NT) Bh[e] = m.b1[(e >> 3) * 32 + 8 * h + (e & 7)];
}

__device__ __forceinline__ float nf_fa_quad_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    return v + __shfl_xor(v, 2, 64);
}

// tokens of position j into the rotated plane TP (this thread: channels 8 qz .. 8 qz + 7), then rows 6 qz .. 6 qz + 5 of the head's conv1
// into the [position][8] planes Vs / Ks / Qs.  Ends with a barrier.
template <int N>
__device__ __forceinline__ void nf_fa_prologue(const NfFiAtt& m, int64_t b, int j, int qz, float* TP, float* Vs, float* Ks, float* Qs,
                                               const float* Wh, const float* Bh, float* scr) {
    constexpr int NT = 4 * N;
    const float invn = 1.f / (float)(32 * N);
    const int64_t base = (b * 32 + 8 * qz) * N + j;
    const int pbase = 8 * qz * N + j;
    float u[8], s = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        const float av = m.a[base + d * N];
        u[d] = m.x[base + d * N] + nf_fa_elu(av) * nf_fa_sigmoid(nf_fa_elu(-av));
        s += u[d];
    }
    const float m1 = nf_fa_block_sum_all<NT>(s, scr) * invn;
    float s2 = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) s2 += (u[d] - m1) * (u[d] - m1);
    const float r1 = 1.f / sqrtf(nf_fa_block_sum_all<NT>(s2, scr) * invn + NF_FA_LNEPS);
#pragma unroll
    for (int d = 0; d < 8; ++d)
        TP[NF_FA_IDX(8 * qz + d, j)] = (u[d] - m1) * r1 * m.ln1g[pbase + d * N] + m.ln1b[pbase + d * N] + m.pos[pbase + d * N];
    __syncthreads();                                     // (Wh / Bh are staged as well)
    float t[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) t[c] = TP[NF_FA_IDX(c, j)];
#pragma unroll
    for (int rr = 0; rr < 6; ++rr) {
        const int r = 6 * qz + rr;
        float acc = Bh[r];
        const f32x4* w = reinterpret_cast<const f32x4*>(Wh + r * 32);
#pragma unroll
        for (int c4 = 0; c4 < 8; ++c4) {
            const f32x4 wv = w[c4];
            acc += wv[0] * t[4 * c4] + wv[1] * t[4 * c4 + 1] + wv[2] * t[4 * c4 + 2] + wv[3] * t[4 * c4 + 3];
        }
        float* dst = r < 8 ? Vs : (r < 16 ? Ks : Qs);
        dst[j * 8 + (r & 7)] = acc;
    }
    __syncthreads();
}

template <int N>
__global__ void __launch_bounds__(4 * N) k_fi_att_fwd(NfFiAtt m) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* TP = smem;               // tokens, rotated plane [32][N]
    float* Vs = TP + 32 * N;        // [N][8] x 3: V | K | Q
    float* Ks = Vs + 8 * N;
    float* Qs = Ks + 8 * N;
    float* Wh = Qs + 8 * N;         // [24][32]
    float* Bh = Wh + 768;           // [24]
    float* scr = Bh + 24;           // [16]
    const int j = threadIdx.x >> 2, qz = threadIdx.x & 3, h = blockIdx.y;
    const int64_t b = blockIdx.x;
    nf_fa_stage_head<4 * N>(Wh, Bh, m, h);
    nf_fa_prologue<N>(m, b, j, qz, TP, Vs, Ks, Qs, Wh, Bh, scr);
    const f32x4* V4 = reinterpret_cast<const f32x4*>(Vs);
    const f32x4* K4 = reinterpret_cast<const f32x4*>(Ks);
    const f32x4* Q4 = reinterpret_cast<const f32x4*>(Qs);
    float k[8];
    {
        const f32x4 k0 = K4[2 * j], k1 = K4[2 * j + 1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            k[e] = k0[e];
            k[4 + e] = k1[e];
        }
    }
    float mx = -INFINITY, l = 0.f, mix[8];
#pragma unroll 4
    for (int i = qz; i < N; i += 4) mx = fmaxf(mx, nf_fa_dot8(V4[2 * i], V4[2 * i + 1], k) * NF_FA_SCALE);
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
#pragma unroll
    for (int d = 0; d < 8; ++d) mix[d] = 0.f;
#pragma unroll 2
    for (int i = qz; i < N; i += 4) {
        const float pr = __expf(nf_fa_dot8(V4[2 * i], V4[2 * i + 1], k) * NF_FA_SCALE - mx);
        const f32x4 q0 = Q4[2 * i], q1 = Q4[2 * i + 1];
        l += pr;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mix[e] += pr * q0[e];
            mix[4 + e] += pr * q1[e];
        }
    }
    l = nf_fa_quad_sum(l);
    const float inv = 1.f / l;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        const float v = nf_fa_quad_sum(mix[d]) * inv;
        if ((d >> 1) == qz) m.mixed[(b * 32 + 8 * h + d) * N + j] = v;
    }
    if (qz == 0) m.cj[(b * 4 + h) * N + j] = mx + __logf(l);
}

// one wave: gw[o(row)][c] += sum_p GP[row][p] * T[c][p] for the 24 rows of a head (rows 24..31 of GP are zero), gb[o(row)] += sum_p GP[row][p]
template <int N>
__device__ __forceinline__ void nf_fa_pair_head(const float* GP, const float* T, float* __restrict__ gw, float* __restrict__ gb, int h) {
    const int lane = threadIdx.x & 63, r32 = lane & 31, hs = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bs = 0.f;
#pragma unroll 4
    for (int s = 0; s < N / 2; ++s) {
        const int p = 2 * s + hs;
        const float a = GP[NF_FA_IDX(r32, p)];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, T[NF_FA_IDX(r32, p)], acc, 0, 0, 0);
        bs += a;
    }
    bs += __shfl_xor(bs, 32, 64);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = nf_fa_cd_row(r, hs);
        if (row < 24) atomicAdd(gw + ((row >> 3) * 32 + 8 * h + (row & 7)) * 32 + r32, acc[r]);
    }
    if (hs == 0 && r32 < 24) atomicAdd(gb + (r32 >> 3) * 32 + 8 * h + (r32 & 7), bs);
}

template <int N>
__global__ void __launch_bounds__(4 * N) k_fi_att_bwd(NfFiAtt m) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* TP = smem;               // tokens, rotated plane [32][N]
    float* GP = TP + 32 * N;        // gradient of this head's 24 conv1 rows, rotated plane [32][N] (rows 24..31 zero)
    float* Vs = GP + 32 * N;        // [N][8] x 4: V | K | Q | g_mixed
    float* Ks = Vs + 8 * N;
    float* Qs = Ks + 8 * N;
    float* Gs = Qs + 8 * N;
    float* CJ = Gs + 8 * N;         // [N]
    float* DL = CJ + N;             // [N]
    float* Wh = DL + N;             // [24][32]
    float* Bh = Wh + 768;           // [24]
    float* scr = Bh + 24;           // [16]
    const int j = threadIdx.x >> 2, qz = threadIdx.x & 3, h = blockIdx.y;
    const int64_t b = blockIdx.x;
    nf_fa_stage_head<4 * N>(Wh, Bh, m, h);
    // this thread's two channels of g_mixed / mixed at position j (d = 2 qz, 2 qz + 1): delta_j = mixed_j . g_mixed_j over the quad
    {
        float dl = 0.f;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int d = 2 * qz + e;
            const float g = m.g_mixed_in[(b * 32 + 8 * h + d) * N + j];
            Gs[j * 8 + d] = g;
            dl += m.mixed_in[(b * 32 + 8 * h + d) * N + j] * g;
        }
        dl = nf_fa_quad_sum(dl);
        if (qz == 0) {
            DL[j] = dl;
            CJ[j] = m.cj_in[(b * 4 + h) * N + j];
        }
    }
    nf_fa_prologue<N>(m, b, j, qz, TP, Vs, Ks, Qs, Wh, Bh, scr);      // (its barriers cover Gs / CJ / DL)
    const f32x4* V4 = reinterpret_cast<const f32x4*>(Vs);
    const f32x4* K4 = reinterpret_cast<const f32x4*>(Ks);
    const f32x4* Q4 = reinterpret_cast<const f32x4*>(Qs);
    const f32x4* G4 = reinterpret_cast<const f32x4*>(Gs);
    float v[8], k[8], q[8], gm[8];
    {
        const f32x4 a0 = V4[2 * j], a1 = V4[2 * j + 1], b0 = K4[2 * j], b1 = K4[2 * j + 1], c0 = Q4[2 * j], c1 = Q4[2 * j + 1],
                    d0 = G4[2 * j], d1 = G4[2 * j + 1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = a0[e]; v[4 + e] = a1[e];
            k[e] = b0[e]; k[4 + e] = b1[e];
            q[e] = c0[e]; q[4 + e] = c1[e];
            gm[e] = d0[e]; gm[4 + e] = d1[e];
        }
    }
    const float cj = CJ[j], delta = DL[j];
    // g_s[i][j] = P[i][j] (gP[i][j] - delta_j), P[i][j] = exp(s[i][j] - c_j), gP[i][j] = Q_i . g_mixed_j
    float gp[24];                                        // gradient of the head's conv1 rows at position j: V rows | K rows | Q rows
#pragma unroll
    for (int r = 0; r < 24; ++r) gp[r] = 0.f;
#pragma unroll 2
    for (int i = qz; i < N; i += 4) {                    // the column j: gradient of K_j
        const f32x4 v0 = V4[2 * i], v1 = V4[2 * i + 1];
        const float gs = __expf(nf_fa_dot8(v0, v1, k) * NF_FA_SCALE - cj) * (nf_fa_dot8(Q4[2 * i], Q4[2 * i + 1], gm) - delta);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gp[8 + e] += gs * v0[e];
            gp[12 + e] += gs * v1[e];
        }
    }
#pragma unroll 2
    for (int jj = qz; jj < N; jj += 4) {                 // the row i = j: gradients of V_i and Q_i
        const f32x4 k0 = K4[2 * jj], k1 = K4[2 * jj + 1];
        const f32x4 g0 = G4[2 * jj], g1 = G4[2 * jj + 1];
        const float pr = __expf(nf_fa_dot8(k0, k1, v) * NF_FA_SCALE - CJ[jj]);
        const float gs = pr * (nf_fa_dot8(g0, g1, q) - DL[jj]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gp[e] += gs * k0[e];
            gp[4 + e] += gs * k1[e];
            gp[16 + e] += pr * g0[e];
            gp[20 + e] += pr * g1[e];
        }
    }
#pragma unroll
    for (int r = 0; r < 24; ++r) gp[r] = nf_fa_quad_sum(gp[r]) * (r < 16 ? NF_FA_SCALE : 1.f);
    // rows 6 qz .. 6 qz + 5 (and two of the zero rows) of the gradient plane; this quarter's 8 channels of the tokens' gradient
#pragma unroll
    for (int r = 0; r < 24; ++r)
        if (r / 6 == qz) GP[NF_FA_IDX(r, j)] = gp[r];
    GP[NF_FA_IDX(24 + 2 * qz, j)] = 0.f;
    GP[NF_FA_IDX(25 + 2 * qz, j)] = 0.f;
    {
        float gt[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) gt[d] = 0.f;
#pragma unroll
        for (int r = 0; r < 24; ++r) {
            const f32x4* w = reinterpret_cast<const f32x4*>(Wh + r * 32 + 8 * qz);
            const f32x4 w0 = w[0], w1 = w[1];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                gt[e] += w0[e] * gp[r];
                gt[4 + e] += w1[e] * gp[r];
            }
        }
#pragma unroll
        for (int d = 0; d < 8; ++d) m.gt_part[((b * 4 + h) * 32 + 8 * qz + d) * N + j] = gt[d];
    }
    __syncthreads();
    if (threadIdx.x < 64) {                // wave 0 adds this (sample, head)'s block: the workgroups in block order in deterministic mode
        NF_DET_ENTER_WAVE(nf_fpa);
        nf_fa_pair_head<N>(GP, TP, m.g_w1, m.g_b1, h);
        NF_DET_LEAVE_WAVE(nf_fpa);
    }
}

// ---- per sample, thread = (head h, position j) owning channels 8 h .. 8 h + 7 of position j ----------------------------------------------
template <int N>
struct NfFaS {       // the gate + LayerNorm 1 of this thread's eight elements
    float m1, r1;
    template <typename M>
    __device__ __forceinline__ void run(const M& m, int64_t base, int pbase, float* scr, float (&xh1)[8], float (&x2)[8]) {
        constexpr int NT = 4 * N;
        const float invn = 1.f / (float)(32 * N);
        float u[8], s = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const float av = m.a[base + d * N];
            u[d] = m.x[base + d * N] + nf_fa_elu(av) * nf_fa_sigmoid(nf_fa_elu(-av));
            s += u[d];
        }
        m1 = nf_fa_block_sum_all<NT>(s, scr) * invn;
        float s2 = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) s2 += (u[d] - m1) * (u[d] - m1);
        r1 = 1.f / sqrtf(nf_fa_block_sum_all<NT>(s2, scr) * invn + NF_FA_LNEPS);
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            xh1[d] = (u[d] - m1) * r1;
            x2[d] = xh1[d] * m.ln1g[pbase + d * N] + m.ln1b[pbase + d * N];
        }
    }
};

template <int N, bool BWD>
__global__ void __launch_bounds__(4 * N) k_fi_post(NfFiAtt m) {
    constexpr int NT = 4 * N, PL = 32 * N;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* PA = smem;               // mixed, rotated plane
    float* PB = PA + PL;            // backward: the two halves of conv2's output gradient, one after the other
    float* W2s = PB + (BWD ? PL : 0);     // conv2 weight transposed [32][64]
    float* Bs = W2s + 2048;         // b2[64]
    float* scr = Bs + 64;           // [16]
    const int tid = threadIdx.x, h = tid / N, j = tid & (N - 1);
    const int64_t b = blockIdx.x;
    const float invn = 1.f / (float)(32 * N);
    for (int e = tid; e < 2048; e += NT) W2s[(e & 31) * 64 + (e >> 5)] = m.w2[e];
    for (int e = tid; e < 64; e += NT) Bs[e] = m.b2[e];
    const int64_t base = (b * 32 + 8 * h) * N + j;
    const int pbase = 8 * h * N + j;
    float x2[8];
    {
        float xh1[8];
        NfFaS<N> st;
        st.run(m, base, pbase, scr, xh1, x2);
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) PA[NF_FA_IDX(8 * h + d, j)] = m.mixed_in[base + d * N];
    __syncthreads();
    // conv2 (1x1, 32 -> 64): y = rows 8 h + d, gate = rows 32 + 8 h + d ;  x3 = x2 + y * sigmoid(gate)
    float y[8], sg[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        y[d] = Bs[8 * h + d];
        sg[d] = Bs[32 + 8 * h + d];
    }
#pragma unroll 2
    for (int c = 0; c < 32; ++c) {
        const float tc = PA[NF_FA_IDX(c, j)];
        const f32x4* wr = reinterpret_cast<const f32x4*>(W2s + c * 64 + 8 * h);
        const f32x4 a0 = wr[0], a1 = wr[1], b0 = wr[8], b1 = wr[9];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            y[e] += a0[e] * tc; y[4 + e] += a1[e] * tc;
            sg[e] += b0[e] * tc; sg[4 + e] += b1[e] * tc;
        }
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) sg[d] = nf_fa_sigmoid(sg[d]);
    float xh2[8], r2;
    {
        float x3[8], s = 0.f, s2 = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            x3[d] = x2[d] + y[d] * sg[d];
            s += x3[d];
        }
        const float m2 = nf_fa_block_sum_all<NT>(s, scr) * invn;
#pragma unroll
        for (int d = 0; d < 8; ++d) s2 += (x3[d] - m2) * (x3[d] - m2);
        r2 = 1.f / sqrtf(nf_fa_block_sum_all<NT>(s2, scr) * invn + NF_FA_LNEPS);
#pragma unroll
        for (int d = 0; d < 8; ++d) xh2[d] = (x3[d] - m2) * r2;
    }
    if (!BWD) {
#pragma unroll
        for (int d = 0; d < 8; ++d) m.out[base + d * N] = xh2[d] * m.ln2g[pbase + d * N] + m.ln2b[pbase + d * N];
        return;
    }
    NF_DET_ENTER_ALL(nf_fpa);              // (one thread / one wave of the workgroup per address; the samples' workgroups in block order)
    float g3[8];
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
            const float g4 = g4v[d];
            if (m.per_sample) {
                m.g_ln2g[base + d * N] = g4 * xh2[d];
                m.g_ln2b[base + d * N] = g4;
            } else {
                atomicAdd(m.g_ln2g + pbase + d * N, g4 * xh2[d]);
                atomicAdd(m.g_ln2b + pbase + d * N, g4);
            }
            gh[d] = g4 * m.ln2g[pbase + d * N];
            s1 += gh[d];
            s2 += gh[d] * xh2[d];
        }
        const float S1 = nf_fa_block_sum_all<NT>(s1, scr) * invn, S2 = nf_fa_block_sum_all<NT>(s2, scr) * invn;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            g3[d] = r2 * (gh[d] - S1 - xh2[d] * S2);
            m.g3[base + d * N] = g3[d];
        }
    }
    float gm[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) gm[d] = 0.f;
#pragma unroll 1
    for (int part = 0; part < 2; ++part) {
        __syncthreads();
#pragma unroll
        for (int d = 0; d < 8; ++d)
            PB[NF_FA_IDX(8 * h + d, j)] = part == 0 ? g3[d] * sg[d] : g3[d] * y[d] * sg[d] * (1.f - sg[d]);
        __syncthreads();
#pragma unroll 1
        for (int o4 = 0; o4 < 8; ++o4) {
            const float e0 = PB[NF_FA_IDX(4 * o4, j)], e1 = PB[NF_FA_IDX(4 * o4 + 1, j)], e2 = PB[NF_FA_IDX(4 * o4 + 2, j)],
                        e3 = PB[NF_FA_IDX(4 * o4 + 3, j)];
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(W2s + (8 * h + d) * 64 + 32 * part + 4 * o4);
                gm[d] += w4[0] * e0 + w4[1] * e1 + w4[2] * e2 + w4[3] * e3;
            }
        }
        if ((tid >> 6) == part % (NT / 64)) {            // conv2's weight gradient: one wave, matrix cores, K = the N positions
            const int lane = tid & 63, r32 = lane & 31, hs = lane >> 5;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            float bs = 0.f;
#pragma unroll 4
            for (int s = 0; s < N / 2; ++s) {
                const int p = 2 * s + hs;
                const float a = PB[NF_FA_IDX(r32, p)];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, PA[NF_FA_IDX(r32, p)], acc, 0, 0, 0);
                bs += a;
            }
            bs += __shfl_xor(bs, 32, 64);
#pragma unroll
            for (int r = 0; r < 16; ++r) atomicAdd(m.g_w2 + (32 * part + nf_fa_cd_row(r, hs)) * 32 + r32, acc[r]);
            if (hs == 0) atomicAdd(m.g_b2 + 32 * part + r32, bs);
        }
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) m.g_mixed[base + d * N] = gm[d];
    NF_DET_LEAVE_ALL(nf_fpa);
}

template <int N>
__global__ void __launch_bounds__(4 * N) k_fi_pre_bwd(NfFiAtt m) {
    constexpr int NT = 4 * N;
    __shared__ float scr[16];
    const int tid = threadIdx.x, h = tid / N, j = tid & (N - 1);
    const int64_t b = blockIdx.x;
    const float invn = 1.f / (float)(32 * N);
    const int64_t base = (b * 32 + 8 * h) * N + j;
    const int pbase = 8 * h * N + j;
    float xh1[8], x2[8];
    NfFaS<N> st;
    st.run(m, base, pbase, scr, xh1, x2);
    float gh[8], s1 = 0.f, s2 = 0.f;
    NF_DET_ENTER_ALL(nf_fpa);
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        float gt = 0.f;
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) gt += m.gt_in[((b * 4 + hh) * 32 + 8 * h + d) * N + j];
        const float g2 = m.g3_in[base + d * N] + gt;
        if (m.per_sample) {
            m.g_pos[base + d * N] = gt;
            m.g_ln1g[base + d * N] = g2 * xh1[d];
            m.g_ln1b[base + d * N] = g2;
        } else {
            atomicAdd(m.g_pos + pbase + d * N, gt);
            atomicAdd(m.g_ln1g + pbase + d * N, g2 * xh1[d]);
            atomicAdd(m.g_ln1b + pbase + d * N, g2);
        }
        gh[d] = g2 * m.ln1g[pbase + d * N];
        s1 += gh[d];
        s2 += gh[d] * xh1[d];
    }
    const float S1 = nf_fa_block_sum_all<NT>(s1, scr) * invn, S2 = nf_fa_block_sum_all<NT>(s2, scr) * invn;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        const float gu = st.r1 * (gh[d] - S1 - xh1[d] * S2);
        const float av = m.a[base + d * N];
        const float e1 = nf_fa_elu(av), s2_ = nf_fa_sigmoid(nf_fa_elu(-av));
        m.g_x[base + d * N] = gu;
        m.g_a[base + d * N] = gu * (nf_fa_elu_grad(av) * s2_ - e1 * s2_ * (1.f - s2_) * nf_fa_elu_grad(-av));
    }
    NF_DET_LEAVE_ALL(nf_fpa);
}

// ---------------------------------------------------------------------------------------------------------------
#include <mutex>
#include <unordered_set>
template <typename K>
static inline int nf_fa_optin(K kernel, size_t lds) {
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

static inline int nf_fa_n(int64_t B, int H, int W) {
    if (B < 1 || B > 0x7fffffff || H != W) return 0;
    return W == 16 ? 256 : 0;        // (8 x 8 maps measured slower cut by head than one workgroup per sample: not offered)
}

extern "C" int nf_flowpp_img_att_usable(int64_t B, int H, int W) { return nf_fa_n(B, H, W) != 0 && B * 32 * H * W < ((int64_t)1 << 31) ? 1 : 0; }

#define NF_FA_NONNULL(...)                                   \
    do {                                                     \
        const void* ps_[] = {__VA_ARGS__};                   \
        for (const void* p_ : ps_)                           \
            if (p_ == nullptr) return NF_E_BADARG;           \
    } while (0)

extern "C" int nf_flowpp_img_att_fwd(const float* x, const float* a, const float* ln1_g, const float* ln1_b, const float* pos,
                                     const float* conv1_w, const float* conv1_b, float* mixed, float* cj, int64_t B, int H, int W,
                                     nf_stream_t stream) {
    NF_FA_NONNULL(x, a, ln1_g, ln1_b, pos, conv1_w, conv1_b, mixed, cj);
    if (!nf_flowpp_img_att_usable(B, H, W)) return NF_E_BADARG;
    NfFiAtt m = {};
    m.x = x; m.a = a; m.ln1g = ln1_g; m.ln1b = ln1_b; m.pos = pos; m.w1 = conv1_w; m.b1 = conv1_b; m.mixed = mixed; m.cj = cj;
    constexpr int N = 256;
    const size_t lds = (size_t)(32 * N + 3 * 8 * N + 768 + 24 + 16) * sizeof(float);
    int rc;
    if ((rc = nf_fa_optin(k_fi_att_fwd<N>, lds)) != 0) return rc;
    hipLaunchKernelGGL(k_fi_att_fwd<N>, dim3((unsigned)B, 4), dim3(4 * N), lds, (hipStream_t)stream, m);
    NF_CHECK_LAUNCH();
    return 0;
}

template <int N, bool BWD>
static int nf_fa_post_launch(const NfFiAtt& m, int64_t B, hipStream_t st) {
    const size_t lds = (size_t)((BWD ? 2 : 1) * 32 * N + 2048 + 64 + 16) * sizeof(float);
    int rc;
    if ((rc = nf_fa_optin(k_fi_post<N, BWD>, lds)) != 0) return rc;
    hipLaunchKernelGGL((k_fi_post<N, BWD>), dim3((unsigned)B), dim3(4 * N), lds, st, m);
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_flowpp_img_post_fwd(const float* x, const float* a, const float* ln1_g, const float* ln1_b, const float* mixed,
                                      const float* conv2_w, const float* conv2_b, const float* ln2_g, const float* ln2_b, float* out,
                                      int64_t B, int H, int W, nf_stream_t stream) {
    NF_FA_NONNULL(x, a, ln1_g, ln1_b, mixed, conv2_w, conv2_b, ln2_g, ln2_b, out);
    if (!nf_flowpp_img_att_usable(B, H, W)) return NF_E_BADARG;
    NfFiAtt m = {};
    m.x = x; m.a = a; m.ln1g = ln1_g; m.ln1b = ln1_b; m.mixed_in = mixed; m.w2 = conv2_w; m.b2 = conv2_b; m.ln2g = ln2_g; m.ln2b = ln2_b;
    m.out = out;
    return nf_fa_post_launch<256, false>(m, B, (hipStream_t)stream);
}

extern "C" int nf_flowpp_img_post_bwd(const float* x, const float* a, const float* ln1_g, const float* ln1_b, const float* mixed,
                                      const float* conv2_w, const float* conv2_b, const float* ln2_g, const float* ln2_b,
                                      const float* g_out, int g_out_slabs, float* g3, float* g_mixed, float* g_conv2_w, float* g_conv2_b,
                                      float* g_ln2_g, float* g_ln2_b, int per_sample, int64_t B, int H, int W, nf_stream_t stream) {
    NF_FA_NONNULL(x, a, ln1_g, ln1_b, mixed, conv2_w, conv2_b, ln2_g, ln2_b, g_out, g3, g_mixed, g_conv2_w, g_conv2_b, g_ln2_g, g_ln2_b);
    if (!nf_flowpp_img_att_usable(B, H, W) || g_out_slabs < 1 || g_out_slabs > NF_FLOWPP_IMG_MAX_KSPLIT) return NF_E_BADARG;
    NfFiAtt m = {};
    m.x = x; m.a = a; m.ln1g = ln1_g; m.ln1b = ln1_b; m.mixed_in = mixed; m.w2 = conv2_w; m.b2 = conv2_b; m.ln2g = ln2_g; m.ln2b = ln2_b;
    m.g_out = g_out; m.g_slabs = g_out_slabs; m.g_slab_stride = B * 32 * H * W; m.g3 = g3; m.g_mixed = g_mixed; m.g_w2 = g_conv2_w;
    m.g_b2 = g_conv2_b; m.g_ln2g = g_ln2_g; m.g_ln2b = g_ln2_b; m.per_sample = per_sample;
    return nf_fa_post_launch<256, true>(m, B, (hipStream_t)stream);
}

template <int N>
static int nf_fa_att_bwd_launch(const NfFiAtt& m, int64_t B, hipStream_t st) {
    const size_t lds = (size_t)(4 * 8 * N + 2 * N + 2 * 32 * N + 768 + 24 + 16) * sizeof(float);
    int rc;
    if ((rc = nf_fa_optin(k_fi_att_bwd<N>, lds)) != 0) return rc;
    hipLaunchKernelGGL(k_fi_att_bwd<N>, dim3((unsigned)B, 4), dim3(4 * N), lds, st, m);
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_flowpp_img_att_bwd(const float* x, const float* a, const float* ln1_g, const float* ln1_b, const float* pos,
                                     const float* conv1_w, const float* conv1_b, const float* mixed, const float* cj,
                                     const float* g_mixed, float* gt_part, float* g_conv1_w, float* g_conv1_b, int64_t B, int H, int W,
                                     nf_stream_t stream) {
    NF_FA_NONNULL(x, a, ln1_g, ln1_b, pos, conv1_w, conv1_b, mixed, cj, g_mixed, gt_part, g_conv1_w, g_conv1_b);
    if (!nf_flowpp_img_att_usable(B, H, W)) return NF_E_BADARG;
    NfFiAtt m = {};
    m.x = x; m.a = a; m.ln1g = ln1_g; m.ln1b = ln1_b; m.pos = pos; m.w1 = conv1_w; m.b1 = conv1_b; m.mixed_in = mixed; m.cj_in = cj;
    m.g_mixed_in = g_mixed; m.gt_part = gt_part; m.g_w1 = g_conv1_w; m.g_b1 = g_conv1_b;
    return nf_fa_att_bwd_launch<256>(m, B, (hipStream_t)stream);
}

extern "C" int nf_flowpp_img_pre_bwd(const float* x, const float* a, const float* ln1_g, const float* ln1_b, const float* g3,
                                     const float* gt_part, float* g_x, float* g_a, float* g_ln1_g, float* g_ln1_b, float* g_pos,
                                     int per_sample, int64_t B, int H, int W, nf_stream_t stream) {
    NF_FA_NONNULL(x, a, ln1_g, ln1_b, g3, gt_part, g_x, g_a, g_ln1_g, g_ln1_b, g_pos);
    if (!nf_flowpp_img_att_usable(B, H, W)) return NF_E_BADARG;
    NfFiAtt m = {};
    m.x = x; m.a = a; m.ln1g = ln1_g; m.ln1b = ln1_b; m.g3_in = g3; m.gt_in = gt_part; m.g_x = g_x; m.g_a = g_a; m.g_ln1g = g_ln1_g;
    m.g_ln1b = g_ln1_b; m.g_pos = g_pos; m.per_sample = per_sample;
    hipLaunchKernelGGL(k_fi_pre_bwd<256>, dim3((unsigned)B), dim3(1024), 0, (hipStream_t)stream, m);
    NF_CHECK_LAUNCH();
    return 0;
}
