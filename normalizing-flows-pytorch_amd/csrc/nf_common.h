// Shared device helpers for libnfhip (gfx950 / CDNA4 only: wave64, 256 CUs in 8 XCDs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nfhip.h"

#define NF_WAVE 64
#define NF_BLOCK 256
#define NF_MAX_GRID 4096  // >> 256 CUs x 8 resident blocks; grid-stride beyond (guide: guideline 11)

#define NF_CHECK_LAUNCH()                 \
    do {                                  \
        hipError_t e_ = hipGetLastError(); \
        if (e_ != hipSuccess) return (int)e_; \
    } while (0)

static inline unsigned nf_grid_for(int64_t work_items, int per_block = NF_BLOCK) {
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > NF_MAX_GRID) g = NF_MAX_GRID;
    return (unsigned)g;
}

// ---------------------------------------------------------------------------------------------------------------
// split maps (reference flows/squeeze.py).  All per-sample offsets fit int32.
// ---------------------------------------------------------------------------------------------------------------
struct NfSplit {
    int mode, odd;
    int C, H, W;     // full tensor (per sample)
    int Ch, h, w;    // half tensor (per sample)
    int n_half;      // Ch*h*w
    int n_full;      // C*H*W
};

static inline bool nf_make_split(NfSplit& s, int mode, int odd, int C, int H, int W) {
    s.mode = mode; s.odd = odd ? 1 : 0; s.C = C; s.H = H; s.W = W;
    switch (mode) {
        case NF_SPLIT_1D:
            if (H != 1 || W != 1 || (C & 1)) return false;
            s.Ch = C / 2; s.h = 1; s.w = 1; break;
        case NF_SPLIT_CHECKER:
            if ((H & 1) || (W & 1)) return false;
            s.Ch = 2 * C; s.h = H / 2; s.w = W / 2; break;
        case NF_SPLIT_CHANNEL:
            if (C & 1) return false;
            s.Ch = C / 2; s.h = H; s.w = W; break;
        case NF_SPLIT_NONE:
            s.Ch = C; s.h = H; s.w = W; break;
        default: return false;
    }
    s.n_half = s.Ch * s.h * s.w;
    s.n_full = C * H * W;
    return true;
}

// offset (inside one sample of the FULL tensor) of element e = (m*h + i)*w + j of half `which` (0 = z0, 1 = z1)
__device__ __forceinline__ int nf_half_to_full(const NfSplit& s, int which, int e) {
    const int sel = which ^ s.odd;  // 0: the reference's first-returned half before the odd swap
    switch (s.mode) {
        case NF_SPLIT_1D:  // squeeze.py:68-69: z.view(B, C/2, 2)[:, :, sel]
            return 2 * e + sel;
        case NF_SPLIT_CHANNEL:  // squeeze.py:7: halves of dim 1
            return e + sel * s.n_half;
        case NF_SPLIT_CHECKER: {  // squeeze.py:36-41: squeezed channel k = 4c + 2dy + dx, chunks [a b c d]
            const int hw = s.h * s.w;
            const int m = e / hw, r = e - m * hw;
            const int i = r / s.w, j = r - i * s.w;
            const int k = sel == 0 ? (m < s.C ? m : m + 2 * s.C) : (m + s.C);
            const int c = k >> 2, dy = (k >> 1) & 1, dx = k & 1;
            return (c * s.H + 2 * i + dy) * s.W + 2 * j + dx;
        }
        default:  // NF_SPLIT_NONE
            return e;
    }
}

// full-tensor element (c, pixel p = y*W + x) -> (which half, index inside the half)
__device__ __forceinline__ void nf_full_to_half(const NfSplit& s, int c, int p, int& which, int& e) {
    switch (s.mode) {
        case NF_SPLIT_1D: { const int sel = c & 1; which = sel ^ s.odd; e = c >> 1; return; }
        case NF_SPLIT_CHANNEL: {
            const int hc = s.C >> 1, sel = c >= hc ? 1 : 0;
            which = sel ^ s.odd; e = (c - sel * hc) * (s.H * s.W) + p; return;
        }
        default: {  // NF_SPLIT_CHECKER
            const int y = p / s.W, x = p - y * s.W;
            const int k = 4 * c + 2 * (y & 1) + (x & 1), q = k / s.C;
            const int sel = (q == 1 || q == 2) ? 1 : 0;
            const int m = sel ? k - s.C : (q == 0 ? k : k - 2 * s.C);
            which = sel ^ s.odd; e = (m * s.h + (y >> 1)) * s.w + (x >> 1); return;
        }
    }
}

// squeezed (space-to-depth) element e = (k*h + i)*w + j  ->  offset in the (C,H,W) sample   squeeze.py:90-92
__device__ __forceinline__ int nf_squeezed_to_full(int e, int H, int W) {
    const int h = H >> 1, w = W >> 1, hw = h * w;
    const int k = e / hw, r = e - k * hw;
    const int i = r / w, j = r - i * w;
    const int c = k >> 2, dy = (k >> 1) & 1, dx = k & 1;
    return (c * H + 2 * i + dy) * W + 2 * j + dx;
}

// ---------------------------------------------------------------------------------------------------------------
// reductions: wave64 shuffle -> LDS -> one value per block
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float nf_wave_sum(float v) {
#pragma unroll
    for (int off = NF_WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, NF_WAVE);
    return v;
}

// sum over the block; result valid in thread 0.  `scratch` holds >= blockDim.x/64 floats.
__device__ __forceinline__ float nf_block_sum(float v, float* scratch) {
    const int lane = threadIdx.x & (NF_WAVE - 1), wid = threadIdx.x >> 6;
    v = nf_wave_sum(v);
    __syncthreads();  // protect scratch reuse across consecutive calls
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + NF_WAVE - 1) >> 6;
        for (int i = 0; i < nw; ++i) r += scratch[i];
    }
    return r;
}

// ---------------------------------------------------------------------------------------------------------------
// numerics matching the torch primitives the reference relies on
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float nf_softplus(float x) {  // F.softplus: beta 1, threshold 20
    return x > 20.f ? x : log1pf(expf(x));
}
__device__ __forceinline__ float nf_logsigmoid(float x) {  // min(x,0) - log1p(exp(-|x|))
    return fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
}

// ---- persistent kernels: bounded spin loops with a LOUD failure ---------------------------------------------------------------------
// Every translation unit with software grid exchanges declares its own state (device globals are per TU without -fgpu-rdc):
//   <p>_timeouts    count of spin loops that gave up
//   <p>_spin_limit  poll budget of one spin loop (default 2^22 polls ~ seconds; nf_persistent_config lowers it for the tests)
//   <p>_host_flag   device pointer of ONE pinned, host-mapped word shared by all TUs: a loop that gives up stores 1 there with
//                   system scope, so the host sees the failure WITHOUT synchronising (checked by _native.call / FlowTrainer)
#define NF_PERSIST_STATE(p)                              \
    __device__ unsigned p##_timeouts;                    \
    __device__ unsigned p##_spin_limit = (1u << 22);     \
    __device__ unsigned* p##_host_flag;
#define NF_PERSIST_GIVE_UP(p)                                                                         \
    do {                                                                                              \
        atomicAdd(&p##_timeouts, 1u);                                                                 \
        unsigned* hf_ = p##_host_flag;                                                                \
        if (hf_) __hip_atomic_store(hf_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);            \
    } while (0)
#define NF_PERSIST_HOST_API(p)                                                                        \
    __attribute__((visibility("hidden"))) int p##_persist_read(unsigned* v) {                         \
        return (int)hipMemcpyFromSymbol(v, HIP_SYMBOL(p##_timeouts), sizeof(unsigned));               \
    }                                                                                                 \
    __attribute__((visibility("hidden"))) int p##_persist_set(unsigned limit, unsigned* flag_dev, int reset) { \
        hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(p##_spin_limit), &limit, sizeof(unsigned));       \
        if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(p##_host_flag), &flag_dev, sizeof(unsigned*)); \
        if (e == hipSuccess && reset) {                                                               \
            const unsigned zero = 0;                                                                  \
            e = hipMemcpyToSymbol(HIP_SYMBOL(p##_timeouts), &zero, sizeof(unsigned));                 \
        }                                                                                             \
        return (int)e;                                                                                \
    }
