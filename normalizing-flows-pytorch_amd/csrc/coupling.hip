// Fused affine coupling: split gather -> tanh*scale+bias -> z0*exp(s)+t -> merge scatter -> per-sample sum(s) into ld.
// Reference: flows/coupling.py:32-43 (split/merge wrapper), :104-122 (transform / inverse), flows/maf.py:101-107.
// HBM-bound: reads z (4 B/elem) + params (4 B/elem of z) and writes y (4 B/elem) = 12 B per element of z.
//
// Two launch shapes:
//   rows : n_half <= NF_ROWS_MAX (2-D data: one transformed feature per sample) -> one thread per sample,
//          no cross-lane reduction for ld at all.
//   slab : images -> grid (sample, slab); 256 threads stride over a slab of the sample's half tensor; the
//          per-sample sum(s) is a wave64 shuffle + LDS reduction and ONE write (or one atomic when slabs > 1).
#include "nf_common.h"
#include "nf_det.h"

NF_DET_STATE(nf_cpl)
NF_DET_HOST_API(nf_cpl)

#define NF_ROWS_MAX 16
#define NF_SLAB 2048
#define NF_ATOMIC_GRID 512  // kernels ending in same-address atomics: <= 2 blocks per CU (an atomic costs ~23 ns serialised)

__device__ __forceinline__ float nf_scale_of(float s_raw, float a, float c) { return tanhf(s_raw) * a + c; }

// ---------------------------------------------------------------------------------------------------------------
template <bool INVERSE>
__global__ void __launch_bounds__(NF_BLOCK) k_affine_rows_fwd(const float* __restrict__ z, const float* __restrict__ tp,
                                                              const float* __restrict__ sp, int64_t pbs,
                                                              const float* __restrict__ p_a, const float* __restrict__ p_c,
                                                              float* __restrict__ y, float* __restrict__ ld, NfSplit s,
                                                              int64_t B) {
    const float a = p_a[0], c = p_c[0];
    const bool has_pass = s.mode != NF_SPLIT_NONE;
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += (int64_t)gridDim.x * blockDim.x) {
        const float* zb = z + b * s.n_full;
        float* yb = y + b * s.n_full;
        float acc = 0.f;
        for (int e = 0; e < s.n_half; ++e) {
            const int o0 = nf_half_to_full(s, 0, e);
            const float sv = nf_scale_of(sp[b * pbs + e], a, c);
            const float t = tp[b * pbs + e];
            const float z0 = zb[o0];
            yb[o0] = INVERSE ? expf(-sv) * (z0 - t) : z0 * expf(sv) + t;
            acc += sv;
            if (has_pass) {
                const int o1 = nf_half_to_full(s, 1, e);
                yb[o1] = zb[o1];
            }
        }
        ld[b] += INVERSE ? -acc : acc;
    }
}

template <bool INVERSE>
__global__ void __launch_bounds__(NF_BLOCK) k_affine_slab_fwd(const float* __restrict__ z, const float* __restrict__ tp,
                                                              const float* __restrict__ sp, int64_t pbs,
                                                              const float* __restrict__ p_a, const float* __restrict__ p_c,
                                                              float* __restrict__ y, float* __restrict__ ld, NfSplit s) {
    __shared__ float scratch[NF_BLOCK / NF_WAVE];
    const float a = p_a[0], c = p_c[0];
    const bool has_pass = s.mode != NF_SPLIT_NONE;
    const int64_t b = blockIdx.x;
    const int e0 = blockIdx.y * NF_SLAB;
    const int e1 = min(e0 + NF_SLAB, s.n_half);
    const float* zb = z + b * s.n_full;
    float* yb = y + b * s.n_full;
    const float* tb = tp + b * pbs;
    const float* sb = sp + b * pbs;
    float acc = 0.f;
    for (int e = e0 + threadIdx.x; e < e1; e += NF_BLOCK) {
        const int o0 = nf_half_to_full(s, 0, e);
        const float sv = nf_scale_of(sb[e], a, c);
        const float t = tb[e];
        const float z0 = zb[o0];
        yb[o0] = INVERSE ? expf(-sv) * (z0 - t) : z0 * expf(sv) + t;
        acc += sv;
        if (has_pass) {
            const int o1 = nf_half_to_full(s, 1, e);
            yb[o1] = zb[o1];
        }
    }
    const float tot = nf_block_sum(acc, scratch);
    if (threadIdx.x == 0) {
        const float d = INVERSE ? -tot : tot;
        if (gridDim.y == 1) ld[b] += d;
        else { NF_DET_ENTER_COL(nf_cpl); atomicAdd(ld + b, d); NF_DET_LEAVE_COL(nf_cpl); }      // (grid (sample, slab): a chain per sample)
    }
}

// ---------------------------------------------------------------------------------------------------------------
// autograd of the forward direction
__device__ __forceinline__ void nf_affine_bwd_elem(float gy0, float gld, float z0, float sr, float a, float c,
                                                   float& g_z0, float& g_sr, float& acc_a, float& acc_c) {
    const float th = tanhf(sr);
    const float es = expf(th * a + c);
    g_z0 = gy0 * es;
    const float gs = gy0 * z0 * es + gld;
    g_sr = gs * a * (1.f - th * th);
    acc_a += gs * th;
    acc_c += gs;
}

__global__ void __launch_bounds__(NF_BLOCK) k_affine_rows_bwd(const float* __restrict__ gy, const float* __restrict__ gld,
                                                              const float* __restrict__ z, const float* __restrict__ sp,
                                                              int64_t pbs, const float* __restrict__ p_a,
                                                              const float* __restrict__ p_c, float* __restrict__ gz,
                                                              float* __restrict__ gt, float* __restrict__ gs,
                                                              float* __restrict__ g_scale, float* __restrict__ g_bias,
                                                              NfSplit s, int64_t B) {
    __shared__ float scratch[NF_BLOCK / NF_WAVE];
    const float a = p_a[0], c = p_c[0];
    const bool has_pass = s.mode != NF_SPLIT_NONE;
    float acc_a = 0.f, acc_c = 0.f;
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += (int64_t)gridDim.x * blockDim.x) {
        const int64_t fb = b * s.n_full;
        const float gl = gld[b];
        for (int e = 0; e < s.n_half; ++e) {
            const int o0 = nf_half_to_full(s, 0, e);
            const float g0 = gy[fb + o0];
            float g_z0, g_sr;
            nf_affine_bwd_elem(g0, gl, z[fb + o0], sp[b * pbs + e], a, c, g_z0, g_sr, acc_a, acc_c);
            gz[fb + o0] = g_z0;
            gt[b * pbs + e] = g0;
            gs[b * pbs + e] = g_sr;
            if (has_pass) {
                const int o1 = nf_half_to_full(s, 1, e);
                gz[fb + o1] = gy[fb + o1];
            }
        }
    }
    const float ta = nf_block_sum(acc_a, scratch);
    const float tc = nf_block_sum(acc_c, scratch);
    if (threadIdx.x == 0) {
        NF_DET_ADD2(nf_cpl, g_scale, ta, g_bias, tc);
    }
}

__global__ void __launch_bounds__(NF_BLOCK) k_affine_slab_bwd(const float* __restrict__ gy, const float* __restrict__ gld,
                                                              const float* __restrict__ z, const float* __restrict__ sp,
                                                              int64_t pbs, const float* __restrict__ p_a,
                                                              const float* __restrict__ p_c, float* __restrict__ gz,
                                                              float* __restrict__ gt, float* __restrict__ gs,
                                                              float* __restrict__ g_scale, float* __restrict__ g_bias,
                                                              NfSplit s, int64_t B, int slabs) {
    __shared__ float scratch[NF_BLOCK / NF_WAVE];
    const float a = p_a[0], c = p_c[0];
    const bool has_pass = s.mode != NF_SPLIT_NONE;
    float acc_a = 0.f, acc_c = 0.f;
    const int64_t items = B * slabs;                      // persistent blocks: ONE pair of atomics per block
    for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
        const int64_t b = item / slabs;
        const int e0 = (int)(item - b * slabs) * NF_SLAB;
        const int e1 = min(e0 + NF_SLAB, s.n_half);
        const int64_t fb = b * s.n_full;
        const float gl = gld[b];
        for (int e = e0 + threadIdx.x; e < e1; e += NF_BLOCK) {
            const int o0 = nf_half_to_full(s, 0, e);
            const float g0 = gy[fb + o0];
            float g_z0, g_sr;
            nf_affine_bwd_elem(g0, gl, z[fb + o0], sp[b * pbs + e], a, c, g_z0, g_sr, acc_a, acc_c);
            gz[fb + o0] = g_z0;
            gt[b * pbs + e] = g0;
            gs[b * pbs + e] = g_sr;
            if (has_pass) {
                const int o1 = nf_half_to_full(s, 1, e);
                gz[fb + o1] = gy[fb + o1];
            }
        }
    }
    const float ta = nf_block_sum(acc_a, scratch);
    const float tc = nf_block_sum(acc_c, scratch);
    if (threadIdx.x == 0) {
        NF_DET_ADD2(nf_cpl, g_scale, ta, g_bias, tc);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Vectorised image kernels (W % 4 == 0): a thread owns ONE float4 of the FULL tensor's row (c, y, 4 x4 .. 4 x4 + 3), so the read
// of z and the write of y are full 16-byte coalesced lines whatever the split map; the map only decides which of the vector's
// components are transformed and where their two parameters live:
//   channel split : the whole vector belongs to one half; parameters are the float4 at the same offset inside the half
//   checker split : even x (dx = 0) belong to squeezed channel k0 = 4 c + 2 (y & 1), odd x to k0 + 1 (squeeze.py:36-41), each
//                   with its own half / half channel m; components (x, z) = dx 0, (y, w) = dx 1, half columns j = 2 x4, 2 x4 + 1:
//                   the parameters of one parity are ONE float2 (consecutive j), coalesced along the half's rows.
// 32-bit index arithmetic, three small divisions per VECTOR (the scalar kernels do five per element).
// Grid (sample, slab of NF_VSLAB vectors); per-sample sum(s) by block reduction -> one add / atomic.
// ---------------------------------------------------------------------------------------------------------------
#define NF_VSLAB 1024     // float4 per block and trip of the grid's y axis (four per thread)

struct NfVecSel { int which, e; };      // half (0 = transformed) and offset inside the half of the first of a parity's two elements
__device__ __forceinline__ NfVecSel nf_checker_sel(const NfSplit& s, int k, int i, int j0) {
    const int q = k / s.C;
    const int sel = (q == 1 || q == 2) ? 1 : 0;
    const int m = sel ? k - s.C : (q == 0 ? k : k - 2 * s.C);
    NfVecSel r;
    r.which = sel ^ s.odd;
    r.e = (m * s.h + i) * s.w + j0;
    return r;
}

template <bool INVERSE, bool CHECKER>
__global__ void __launch_bounds__(NF_BLOCK) k_affine_img_fwd(const float* __restrict__ z, const float* __restrict__ tp,
                                                             const float* __restrict__ sp, int64_t pbs, const float* __restrict__ p_a,
                                                             const float* __restrict__ p_c, float* __restrict__ y, float* __restrict__ ld,
                                                             NfSplit s) {
    __shared__ float scratch[NF_BLOCK / NF_WAVE];
    const float a = p_a[0], c = p_c[0];
    const int64_t b = blockIdx.x;
    const int n4 = s.n_full >> 2, W4 = s.W >> 2, h4 = s.n_half >> 2;
    const float4* zb = reinterpret_cast<const float4*>(z + b * s.n_full);
    float4* yb = reinterpret_cast<float4*>(y + b * s.n_full);
    const float* tb = tp + b * pbs;
    const float* sb = sp + b * pbs;
    float acc = 0.f;
    const int v1 = min((int)(blockIdx.y + 1) * NF_VSLAB, n4);
    for (int v = blockIdx.y * NF_VSLAB + threadIdx.x; v < v1; v += NF_BLOCK) {
        float4 zv = zb[v];
        if (!CHECKER) {
            const int sel = v >= h4 ? 1 : 0;
            if ((sel ^ s.odd) == 0) {
                const int e4 = v - sel * h4;
                const float4 t = reinterpret_cast<const float4*>(tb)[e4], sr = reinterpret_cast<const float4*>(sb)[e4];
                const float s0 = nf_scale_of(sr.x, a, c), s1 = nf_scale_of(sr.y, a, c), s2 = nf_scale_of(sr.z, a, c), s3 = nf_scale_of(sr.w, a, c);
                zv.x = INVERSE ? expf(-s0) * (zv.x - t.x) : zv.x * expf(s0) + t.x;
                zv.y = INVERSE ? expf(-s1) * (zv.y - t.y) : zv.y * expf(s1) + t.y;
                zv.z = INVERSE ? expf(-s2) * (zv.z - t.z) : zv.z * expf(s2) + t.z;
                zv.w = INVERSE ? expf(-s3) * (zv.w - t.w) : zv.w * expf(s3) + t.w;
                acc += (s0 + s1) + (s2 + s3);
            }
        } else {
            const int r = v / W4, x4 = v - r * W4;
            const int cc = r / s.H, yy = r - cc * s.H;
            const int k0 = 4 * cc + 2 * (yy & 1);
            const NfVecSel q0 = nf_checker_sel(s, k0, yy >> 1, 2 * x4), q1 = nf_checker_sel(s, k0 + 1, yy >> 1, 2 * x4);
            if (q0.which == 0) {
                const float2 t = *reinterpret_cast<const float2*>(tb + q0.e), sr = *reinterpret_cast<const float2*>(sb + q0.e);
                const float s0 = nf_scale_of(sr.x, a, c), s1 = nf_scale_of(sr.y, a, c);
                zv.x = INVERSE ? expf(-s0) * (zv.x - t.x) : zv.x * expf(s0) + t.x;
                zv.z = INVERSE ? expf(-s1) * (zv.z - t.y) : zv.z * expf(s1) + t.y;
                acc += s0 + s1;
            }
            if (q1.which == 0) {
                const float2 t = *reinterpret_cast<const float2*>(tb + q1.e), sr = *reinterpret_cast<const float2*>(sb + q1.e);
                const float s0 = nf_scale_of(sr.x, a, c), s1 = nf_scale_of(sr.y, a, c);
                zv.y = INVERSE ? expf(-s0) * (zv.y - t.x) : zv.y * expf(s0) + t.x;
                zv.w = INVERSE ? expf(-s1) * (zv.w - t.y) : zv.w * expf(s1) + t.y;
                acc += s0 + s1;
            }
        }
        yb[v] = zv;
    }
    const float tot = nf_block_sum(acc, scratch);
    if (threadIdx.x == 0) {
        const float dd = INVERSE ? -tot : tot;
        if (gridDim.y == 1) ld[b] += dd;
        else { NF_DET_ENTER_COL(nf_cpl); atomicAdd(ld + b, dd); NF_DET_LEAVE_COL(nf_cpl); }
    }
}

#define NF_BIG 1024       // threads of the kernels that END in same-address atomics: 256 blocks (one per CU, 16 waves) instead of
                          // 1024 -- an atomic costs ~23 ns serialised at the L2, 1024 of them were a 23 us tail on a ~80 us kernel
template <bool CHECKER>
__global__ void __launch_bounds__(NF_BIG) k_affine_img_bwd(const float* __restrict__ gy, const float* __restrict__ gld,
                                                             const float* __restrict__ z, const float* __restrict__ sp, int64_t pbs,
                                                             const float* __restrict__ p_a, const float* __restrict__ p_c,
                                                             float* __restrict__ gz, float* __restrict__ gt, float* __restrict__ gs,
                                                             float* __restrict__ g_scale, float* __restrict__ g_bias, NfSplit s, int64_t B,
                                                             int slabs) {
    __shared__ float scratch[NF_BIG / NF_WAVE];
    const float a = p_a[0], c = p_c[0];
    const int n4 = s.n_full >> 2, W4 = s.W >> 2, h4 = s.n_half >> 2;
    float acc_a = 0.f, acc_c = 0.f;
    // persistent blocks over the vectors of ALL samples (a (3,32,32) sample has only 768: per-sample slabs would idle a quarter
    // of a 1024-thread block): ONE pair of atomics per block
    const unsigned total = (unsigned)B * (unsigned)n4;
    (void)slabs;
    for (unsigned gv = blockIdx.x * NF_BIG + threadIdx.x; gv < total; gv += gridDim.x * NF_BIG) {
        const unsigned b = gv / (unsigned)n4;
        const int v = (int)(gv - b * (unsigned)n4);
        const float4* gyb = reinterpret_cast<const float4*>(gy + (int64_t)b * s.n_full);
        const float4* zb = reinterpret_cast<const float4*>(z + (int64_t)b * s.n_full);
        float4* gzb = reinterpret_cast<float4*>(gz + (int64_t)b * s.n_full);
        const float* sb = sp + (int64_t)b * pbs;
        float* gtb = gt + (int64_t)b * pbs;
        float* gsb = gs + (int64_t)b * pbs;
        const float gl = gld[b];
        {
            float4 g = gyb[v];
            if (!CHECKER) {
                const int sel = v >= h4 ? 1 : 0;
                if ((sel ^ s.odd) == 0) {
                    const int e4 = v - sel * h4;
                    const float4 zv = zb[v], sr = reinterpret_cast<const float4*>(sb)[e4];
                    float4 o, os;
                    nf_affine_bwd_elem(g.x, gl, zv.x, sr.x, a, c, o.x, os.x, acc_a, acc_c);
                    nf_affine_bwd_elem(g.y, gl, zv.y, sr.y, a, c, o.y, os.y, acc_a, acc_c);
                    nf_affine_bwd_elem(g.z, gl, zv.z, sr.z, a, c, o.z, os.z, acc_a, acc_c);
                    nf_affine_bwd_elem(g.w, gl, zv.w, sr.w, a, c, o.w, os.w, acc_a, acc_c);
                    reinterpret_cast<float4*>(gtb)[e4] = g;
                    reinterpret_cast<float4*>(gsb)[e4] = os;
                    g = o;
                }
            } else {
                const int r = v / W4, x4 = v - r * W4;
                const int cc = r / s.H, yy = r - cc * s.H;
                const int k0 = 4 * cc + 2 * (yy & 1);
                const NfVecSel q0 = nf_checker_sel(s, k0, yy >> 1, 2 * x4), q1 = nf_checker_sel(s, k0 + 1, yy >> 1, 2 * x4);
                if (q0.which == 0 || q1.which == 0) {
                    const float4 zv = zb[v];
                    if (q0.which == 0) {
                        const float2 sr = *reinterpret_cast<const float2*>(sb + q0.e);
                        float2 os;
                        *reinterpret_cast<float2*>(gtb + q0.e) = make_float2(g.x, g.z);
                        nf_affine_bwd_elem(g.x, gl, zv.x, sr.x, a, c, g.x, os.x, acc_a, acc_c);
                        nf_affine_bwd_elem(g.z, gl, zv.z, sr.y, a, c, g.z, os.y, acc_a, acc_c);
                        *reinterpret_cast<float2*>(gsb + q0.e) = os;
                    }
                    if (q1.which == 0) {
                        const float2 sr = *reinterpret_cast<const float2*>(sb + q1.e);
                        float2 os;
                        *reinterpret_cast<float2*>(gtb + q1.e) = make_float2(g.y, g.w);
                        nf_affine_bwd_elem(g.y, gl, zv.y, sr.x, a, c, g.y, os.x, acc_a, acc_c);
                        nf_affine_bwd_elem(g.w, gl, zv.w, sr.y, a, c, g.w, os.y, acc_a, acc_c);
                        *reinterpret_cast<float2*>(gsb + q1.e) = os;
                    }
                }
            }
            gzb[v] = g;
        }
    }
    const float ta = nf_block_sum(acc_a, scratch);
    const float tc = nf_block_sum(acc_c, scratch);
    if (threadIdx.x == 0) {
        NF_DET_ADD2(nf_cpl, g_scale, ta, g_bias, tc);
    }
}

// 2-D data, D = 2, packed parameters (t, s_raw interleaved per sample: the conditioner output (B, 2)): two samples per thread,
// every access one 16-byte (8 for ld) vector.  tr = index (0 / 1) of the transformed feature.
template <bool INVERSE>
__global__ void __launch_bounds__(NF_BLOCK) k_affine_d2_fwd(const float4* __restrict__ z, const float4* __restrict__ prm,
                                                            const float* __restrict__ p_a, const float* __restrict__ p_c,
                                                            float4* __restrict__ y, float2* __restrict__ ld, int tr, int64_t B2) {
    const float a = p_a[0], c = p_c[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B2; i += (int64_t)gridDim.x * blockDim.x) {
        float4 zv = z[i];
        const float4 p = prm[i];                          // t0, s0, t1, s1
        float2 l = ld[i];
        const float s0 = nf_scale_of(p.y, a, c), s1 = nf_scale_of(p.w, a, c);
        const float z0 = tr ? zv.y : zv.x, z1 = tr ? zv.w : zv.z;
        const float y0 = INVERSE ? expf(-s0) * (z0 - p.x) : z0 * expf(s0) + p.x;
        const float y1 = INVERSE ? expf(-s1) * (z1 - p.z) : z1 * expf(s1) + p.z;
        if (tr) { zv.y = y0; zv.w = y1; } else { zv.x = y0; zv.z = y1; }
        l.x += INVERSE ? -s0 : s0;
        l.y += INVERSE ? -s1 : s1;
        y[i] = zv;
        ld[i] = l;
    }
}
__global__ void __launch_bounds__(NF_BIG) k_affine_d2_bwd(const float4* __restrict__ gy, const float2* __restrict__ gld,
                                                            const float4* __restrict__ z, const float4* __restrict__ prm,
                                                            const float* __restrict__ p_a, const float* __restrict__ p_c,
                                                            float4* __restrict__ gz, float4* __restrict__ gprm,
                                                            float* __restrict__ g_scale, float* __restrict__ g_bias, int tr, int64_t B2) {
    __shared__ float scratch[NF_BIG / NF_WAVE];
    const float a = p_a[0], c = p_c[0];
    float acc_a = 0.f, acc_c = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B2; i += (int64_t)gridDim.x * blockDim.x) {
        float4 g = gy[i];
        const float4 zv = z[i], p = prm[i];
        const float2 gl = gld[i];
        float4 gp;
        float o0, o1;
        const float g0 = tr ? g.y : g.x, g1 = tr ? g.w : g.z;
        nf_affine_bwd_elem(g0, gl.x, tr ? zv.y : zv.x, p.y, a, c, o0, gp.y, acc_a, acc_c);
        nf_affine_bwd_elem(g1, gl.y, tr ? zv.w : zv.z, p.w, a, c, o1, gp.w, acc_a, acc_c);
        gp.x = g0;
        gp.z = g1;
        if (tr) { g.y = o0; g.w = o1; } else { g.x = o0; g.z = o1; }
        gz[i] = g;
        gprm[i] = gp;
    }
    const float ta = nf_block_sum(acc_a, scratch);
    const float tc = nf_block_sum(acc_c, scratch);
    if (threadIdx.x == 0) {
        NF_DET_ADD2(nf_cpl, g_scale, ta, g_bias, tc);
    }
}

static inline int nf_half_to_full_host(const NfSplit& s, int which, int e) { return 2 * e + (which ^ s.odd); }   // NF_SPLIT_1D
static inline bool nf_al(const void* p, unsigned m) { return ((uintptr_t)p & (m - 1)) == 0; }
// image vector path: channel / checker split, W % 4 == 0, every base 16-byte aligned, parameter rows 16-byte aligned
static inline bool nf_affine_vec_ok(const NfSplit& s, const void* z, const void* y, const void* t, const void* sp, const void* t2,
                                    const void* s2, int64_t pbs) {
    if (s.mode != NF_SPLIT_CHANNEL && s.mode != NF_SPLIT_CHECKER) return false;
    if (s.W % 4 != 0 || s.n_half % 4 != 0 || pbs % 4 != 0 || s.n_full >= (1 << 30)) return false;
    return nf_al(z, 16) && nf_al(y, 16) && nf_al(t, 16) && nf_al(sp, 16) && nf_al(t2, 16) && nf_al(s2, 16);
}

// ---------------------------------------------------------------------------------------------------------------
extern "C" int nf_affine_coupling_fwd(const float* z, const float* t_ptr, const float* s_ptr, int64_t param_bstride,
                                      const float* s_log_scale, const float* s_bias, float* y, float* ld, int mode,
                                      int odd, int inverse, int64_t B, int C, int H, int W, nf_stream_t stream) {
    NfSplit s;
    if (!nf_make_split(s, mode, odd, C, H, W)) return NF_E_BADARG;
    if (B == 0 || s.n_half == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (s.mode == NF_SPLIT_1D && C == 2 && s_ptr == t_ptr + 1 && param_bstride == 2 && (B & 1) == 0 && nf_al(z, 16) && nf_al(y, 16) &&
        nf_al(t_ptr, 16) && nf_al(ld, 8)) {
        const int tr = nf_half_to_full_host(s, 0, 0);
        dim3 grid(nf_grid_for(B / 2)), block(NF_BLOCK);
        if (inverse) hipLaunchKernelGGL(k_affine_d2_fwd<true>, grid, block, 0, st, (const float4*)z, (const float4*)t_ptr, s_log_scale, s_bias, (float4*)y, (float2*)ld, tr, B / 2);
        else hipLaunchKernelGGL(k_affine_d2_fwd<false>, grid, block, 0, st, (const float4*)z, (const float4*)t_ptr, s_log_scale, s_bias, (float4*)y, (float2*)ld, tr, B / 2);
        NF_CHECK_LAUNCH();
        return 0;
    }
    if (s.n_half > NF_ROWS_MAX && B <= 0x7fffffffLL && nf_affine_vec_ok(s, z, y, t_ptr, s_ptr, nullptr, nullptr, param_bstride)) {
        dim3 grid((unsigned)B, (unsigned)((s.n_full / 4 + NF_VSLAB - 1) / NF_VSLAB)), block(NF_BLOCK);
        const bool ck = s.mode == NF_SPLIT_CHECKER;
#define NF_L(INV_, CK_) hipLaunchKernelGGL((k_affine_img_fwd<INV_, CK_>), grid, block, 0, st, z, t_ptr, s_ptr, param_bstride, s_log_scale, s_bias, y, ld, s)
        if (inverse) { if (ck) NF_L(true, true); else NF_L(true, false); }
        else { if (ck) NF_L(false, true); else NF_L(false, false); }
#undef NF_L
        NF_CHECK_LAUNCH();
        return 0;
    }
    if (s.n_half <= NF_ROWS_MAX) {
        dim3 grid(nf_grid_for(B)), block(NF_BLOCK);
        if (inverse) hipLaunchKernelGGL(k_affine_rows_fwd<true>, grid, block, 0, st, z, t_ptr, s_ptr, param_bstride, s_log_scale, s_bias, y, ld, s, B);
        else hipLaunchKernelGGL(k_affine_rows_fwd<false>, grid, block, 0, st, z, t_ptr, s_ptr, param_bstride, s_log_scale, s_bias, y, ld, s, B);
    } else {
        if (B > 0x7fffffffLL) return NF_E_BADARG;
        dim3 grid((unsigned)B, (unsigned)((s.n_half + NF_SLAB - 1) / NF_SLAB)), block(NF_BLOCK);
        if (inverse) hipLaunchKernelGGL(k_affine_slab_fwd<true>, grid, block, 0, st, z, t_ptr, s_ptr, param_bstride, s_log_scale, s_bias, y, ld, s);
        else hipLaunchKernelGGL(k_affine_slab_fwd<false>, grid, block, 0, st, z, t_ptr, s_ptr, param_bstride, s_log_scale, s_bias, y, ld, s);
    }
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_affine_coupling_bwd(const float* g_y, const float* g_ld, const float* z, const float* t_ptr,
                                      const float* s_ptr, int64_t param_bstride, const float* s_log_scale,
                                      const float* s_bias, float* g_z, float* g_t, float* g_s, float* g_scale,
                                      float* g_bias, int mode, int odd, int64_t B, int C, int H, int W,
                                      nf_stream_t stream) {
    (void)t_ptr;  // the shift does not enter any gradient
    NfSplit s;
    if (!nf_make_split(s, mode, odd, C, H, W)) return NF_E_BADARG;
    if (B == 0 || s.n_half == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (s.mode == NF_SPLIT_1D && C == 2 && s_ptr == t_ptr + 1 && g_s == g_t + 1 && param_bstride == 2 && (B & 1) == 0 && nf_al(z, 16) &&
        nf_al(g_y, 16) && nf_al(g_z, 16) && nf_al(t_ptr, 16) && nf_al(g_t, 16) && nf_al(g_ld, 8)) {
        const int tr = nf_half_to_full_host(s, 0, 0);
        unsigned g = nf_grid_for(B / 2, NF_BIG);
        if (g > 256) g = 256;
        hipLaunchKernelGGL(k_affine_d2_bwd, dim3(g), dim3(NF_BIG), 0, st, (const float4*)g_y, (const float2*)g_ld, (const float4*)z,
                           (const float4*)t_ptr, s_log_scale, s_bias, (float4*)g_z, (float4*)g_t, g_scale, g_bias, tr, B / 2);
        NF_CHECK_LAUNCH();
        return 0;
    }
    if (s.n_half > NF_ROWS_MAX && B * (int64_t)(s.n_full / 4) < ((int64_t)1 << 31) &&
        nf_affine_vec_ok(s, z, g_z, g_y, s_ptr, g_t, g_s, param_bstride)) {
        const int slabs = (s.n_full / 4 + NF_VSLAB - 1) / NF_VSLAB;
        const int64_t items = B * slabs;
        const unsigned g = (unsigned)(items < 256 ? items : 256);
        if (s.mode == NF_SPLIT_CHECKER)
            hipLaunchKernelGGL(k_affine_img_bwd<true>, dim3(g), dim3(NF_BIG), 0, st, g_y, g_ld, z, s_ptr, param_bstride, s_log_scale,
                               s_bias, g_z, g_t, g_s, g_scale, g_bias, s, B, slabs);
        else
            hipLaunchKernelGGL(k_affine_img_bwd<false>, dim3(g), dim3(NF_BIG), 0, st, g_y, g_ld, z, s_ptr, param_bstride, s_log_scale,
                               s_bias, g_z, g_t, g_s, g_scale, g_bias, s, B, slabs);
        NF_CHECK_LAUNCH();
        return 0;
    }
    if (s.n_half <= NF_ROWS_MAX) {
        unsigned g = nf_grid_for(B);
        if (g > NF_ATOMIC_GRID) g = NF_ATOMIC_GRID;
        hipLaunchKernelGGL(k_affine_rows_bwd, dim3(g), dim3(NF_BLOCK), 0, st, g_y, g_ld, z, s_ptr, param_bstride,
                           s_log_scale, s_bias, g_z, g_t, g_s, g_scale, g_bias, s, B);
    } else {
        const int slabs = (s.n_half + NF_SLAB - 1) / NF_SLAB;
        const int64_t items = B * slabs;
        const unsigned g = (unsigned)(items < 2 * NF_ATOMIC_GRID ? items : 2 * NF_ATOMIC_GRID);
        hipLaunchKernelGGL(k_affine_slab_bwd, dim3(g), dim3(NF_BLOCK), 0, st, g_y, g_ld, z, s_ptr, param_bstride,
                           s_log_scale, s_bias, g_z, g_t, g_s, g_scale, g_bias, s, B, slabs);
    }
    NF_CHECK_LAUNCH();
    return 0;
}
