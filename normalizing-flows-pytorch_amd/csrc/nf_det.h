// Deterministic mode (nf_deterministic(1) / NF_DETERMINISTIC=1): batch sums that meet at one address through float atomics are
// added in a FIXED order -- workgroup by workgroup in linear block order -- so that two runs of a launch from identical inputs are
// bit-identical (the reference's CPU path reproduces itself; main.py:308-311 has a determinism switch).  Off (the default) the
// atomics race as before.
//
// Mechanism: an ordered TURNSTILE per translation unit.  A workgroup's atomics are issued by designated threads; before its first
// atomic such a thread waits until the turn word equals the workgroup's linear index, after its last one it fences (the atomics are
// performed at the device-coherent level before the turn moves) and passes the turn on; the last workgroup of the grid resets it to 0
// for the next launch.  Workgroups are dispatched in linear order, a workgroup holding the turn waits for nobody, hence no deadlock:
// every workgroup with a smaller index is resident or finished.  Launches of one translation unit share the turn word: they are
// serialised by the stream they run on (the engine launches on ONE stream; the side stream of the capture warm-up is joined before
// anything else runs).  Within a workgroup the order is fixed by construction at every site (one thread per address, or waves taking
// turns); each site says which.  The cost is the serialised tail (~1 us per workgroup): a verification mode, not the fast path.
#pragma once
#include <hip/hip_runtime.h>

// per translation unit: [0] mode on / off, [1] the turn, [2] waits that gave up (a workgroup that never passed: a bug -- the launch then
// finishes unordered instead of hanging the device, and nf_deterministic_timeouts() reports it)
// Round 6: the turnstile serialises a whole grid (C4: 655 ms per step against 23 racing -- sixteen-layer weight-gradient launches of 4 096
// workgroups took 12 ms each).  Two cheaper forms take the sites that dominated (profiles/r06_deterministic_cost.txt):
//   * KEYED turnstiles (NF_DET_*_K): only workgroups that add to the SAME addresses need an order.  A site names its chain -- key (which
//     chain: the layer / head of a multi-launch, the replica of a replicated sum), rank (this workgroup's place among the workgroups
//     of the chain in linear block order) and count -- and waits on the chain's own turn word: 128 chains of 32 instead of one of 4 096;
//   * the LAST-WORKGROUP FOLD (nf_det_fold_add): scalar sums of a grid (one thread per workgroup adds NV values) are written to a slab,
//     the workgroup that arrives last adds the slab in block order: no workgroup waits for another.
#define NF_DET_KEYS 4096            /* turn words of the keyed chains of one translation unit */
#define NF_DET_FOLD_MAX 2048        /* workgroups of a grid the last-workgroup fold takes (x NF_DET_FOLD_NV values) */
#define NF_DET_FOLD_NV 2
#define NF_DET_STATE(p)             \
    __device__ unsigned p##_det[3]; \
    __device__ unsigned p##_det_turns[NF_DET_KEYS]; \
    __device__ float p##_det_slab[NF_DET_FOLD_MAX * NF_DET_FOLD_NV]; \
    __device__ unsigned p##_det_cnt[1]; \
    static int p##_det_host = 0;    /* host mirror of [0]: launchers that change their launch shape in the mode read it */
#define NF_DET_HOST_API(p)                                                                       \
    __attribute__((visibility("hidden"))) int p##_det_set(int on) {                              \
        const unsigned w[3] = {on ? 1u : 0u, 0u, 0u};                                            \
        p##_det_host = on ? 1 : 0;                                                               \
        void* tp = nullptr;                                                                      \
        if (hipGetSymbolAddress(&tp, HIP_SYMBOL(p##_det_turns)) == hipSuccess) (void)hipMemset(tp, 0, sizeof(unsigned) * NF_DET_KEYS); \
        if (hipGetSymbolAddress(&tp, HIP_SYMBOL(p##_det_cnt)) == hipSuccess) (void)hipMemset(tp, 0, sizeof(unsigned)); \
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(p##_det), w, sizeof(w));                        \
    }                                                                                            \
    __attribute__((visibility("hidden"))) int p##_det_timeouts(unsigned* out) {                  \
        unsigned w[3];                                                                           \
        const hipError_t e = hipMemcpyFromSymbol(w, HIP_SYMBOL(p##_det), sizeof(w));             \
        *out = w[2];                                                                             \
        return (int)e;                                                                           \
    }
#define NF_DET_SPIN_LIMIT (1u << 22)

__device__ __forceinline__ unsigned nf_det_block() { return blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); }
__device__ __forceinline__ unsigned nf_det_blocks() { return gridDim.x * gridDim.y * gridDim.z; }
__device__ __forceinline__ bool nf_det_on(const unsigned* w) { return __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u; }
// the calling thread waits for its workgroup's turn
__device__ __forceinline__ void nf_det_wait(unsigned* w) {
    const unsigned me = nf_det_block();
    unsigned spins = 0;
    while (__hip_atomic_load(w + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != me) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > NF_DET_SPIN_LIMIT) { atomicAdd(w + 2, 1u); break; }
    }
}
// ... and passes it on behind its atomics
__device__ __forceinline__ void nf_det_pass(unsigned* w) {
    __threadfence();
    const unsigned me = nf_det_block(), n = nf_det_blocks();
    __hip_atomic_store(w + 1, me + 1 == n ? 0u : me + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- site forms --------------------------------------------------------------------------------------------------------------------
// (1) ONE thread of the workgroup issues all of the workgroup's atomics (every workgroup of the grid passes here exactly once):
//         NF_DET_ENTER(p);  atomicAdd(...); ...  NF_DET_LEAVE(p);
#define NF_DET_ENTER(p)                           \
    const bool nf_det_ = nf_det_on(p##_det);      \
    if (nf_det_) nf_det_wait(p##_det)
#define NF_DET_LEAVE(p) \
    if (nf_det_) nf_det_pass(p##_det)
// (2) SEVERAL threads of the workgroup issue atomics (to addresses no two of them share), reached by ALL threads of the workgroup in
//     uniform control flow:   NF_DET_ENTER_ALL(p);  if (mine) atomicAdd(...);  NF_DET_LEAVE_ALL(p);
#define NF_DET_ENTER_ALL(p)                                   \
    const bool nf_det_ = nf_det_on(p##_det);                  \
    if (nf_det_) {                                            \
        if (threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) nf_det_wait(p##_det); \
        __syncthreads();                                      \
    }
#define NF_DET_LEAVE_ALL(p)                                   \
    if (nf_det_) {                                            \
        __threadfence();                                      \
        __syncthreads();                                      \
        if (threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) nf_det_pass(p##_det); \
    }
// (3) inside (2): waves of the workgroup that add to the SAME addresses take turns in wave order (uniform control flow, `nw` waves):
//         nf_det_waves(nf_det_, nw, wid, [&] { atomics of this wave });
template <class F>
__device__ __forceinline__ void nf_det_waves(bool det, int nw, int wid, F&& f) {
    if (det) {
        for (int w_ = 0; w_ < nw; ++w_) {
            if (wid == w_) { f(); __threadfence(); }
            __syncthreads();
        }
    } else {
        f();
    }
}
// (1b) ONE WAVE of the workgroup issues all of the workgroup's atomics (lanes to addresses no two of them share); reached by the whole
//      wave (lane 0 included) in wave-uniform control flow:   NF_DET_ENTER_WAVE(p);  if (mine) atomicAdd(...);  NF_DET_LEAVE_WAVE(p);
#define NF_DET_ENTER_WAVE(p)                      \
    const bool nf_det_ = nf_det_on(p##_det);      \
    if (nf_det_ && (threadIdx.x & 63) == 0) nf_det_wait(p##_det)
#define NF_DET_LEAVE_WAVE(p)                                     \
    if (nf_det_) {                                               \
        __threadfence();                                         \
        if ((threadIdx.x & 63) == 0) nf_det_pass(p##_det);       \
    }

// ---- keyed chains (round 6) ---------------------------------------------------------------------------------------------------------
// chain `key` < NF_DET_KEYS; the workgroup is number `rank` of `count` in it (linear block order: a workgroup only ever waits for one with a
// smaller linear index, which is resident or finished -- the no-deadlock argument of the plain turnstile).  A key out of range is a bug
// of the site: counted like a wait that gave up, the adds then go unordered.
__device__ __forceinline__ void nf_det_wait_k(unsigned* w, unsigned* turns, unsigned key, unsigned rank) {
    if (key >= NF_DET_KEYS) { atomicAdd(w + 2, 1u); return; }
    unsigned spins = 0;
    while (__hip_atomic_load(turns + key, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != rank) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > NF_DET_SPIN_LIMIT) { atomicAdd(w + 2, 1u); break; }
    }
}
__device__ __forceinline__ void nf_det_pass_k(unsigned* turns, unsigned key, unsigned rank, unsigned count) {
    if (key >= NF_DET_KEYS) return;
    __threadfence();
    __hip_atomic_store(turns + key, rank + 1 >= count ? 0u : rank + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// the chain of a REPLICATED sum (include/nfhip.h NF_STAT_REPL: workgroup b of a launch adds into replica b % R; blockIdx.y = the layer of
// a multi-launch, whose layers have tensors of their own): key (blockIdx.y, replica), rank blockIdx.x / R
#define NF_DET_REPL_CHAIN(R)                                                                                  \
    const unsigned nf_det_key_ = blockIdx.y * (R) + blockIdx.x % (R), nf_det_rank_ = blockIdx.x / (R),        \
                   nf_det_cnt_ = (gridDim.x - blockIdx.x % (R) + (R) - 1) / (R)
// the chain of an UNREPLICATED sum of a (workgroups, layers) launch: key blockIdx.y, rank blockIdx.x
#define NF_DET_ROW_CHAIN()                                                                                    \
    const unsigned nf_det_key_ = blockIdx.y, nf_det_rank_ = blockIdx.x, nf_det_cnt_ = gridDim.x
// the chain of a sum whose addresses are picked by blockIdx.x (and blockIdx.z), met by the blockIdx.y workgroups of a (x, y, z) launch:
// key (blockIdx.z, blockIdx.x), rank blockIdx.y.  More chains than turn words: key = NF_DET_KEYS (out of range: counted, unordered) -- the
// launchers of such sites keep gridDim.x * gridDim.z <= NF_DET_KEYS or use the grid-wide form.
#define NF_DET_COL_CHAIN()                                                                                    \
    const unsigned nf_det_key_ = blockIdx.z * gridDim.x + blockIdx.x, nf_det_rank_ = blockIdx.y, nf_det_cnt_ = gridDim.y
// forms (1), (2), (1b) on the chain declared just before (NF_DET_REPL_CHAIN / NF_DET_ROW_CHAIN / NF_DET_COL_CHAIN)
#define NF_DET_ENTER_K(p)                         \
    const bool nf_det_ = nf_det_on(p##_det);      \
    if (nf_det_) nf_det_wait_k(p##_det, p##_det_turns, nf_det_key_, nf_det_rank_)
#define NF_DET_LEAVE_K(p) \
    if (nf_det_) nf_det_pass_k(p##_det_turns, nf_det_key_, nf_det_rank_, nf_det_cnt_)
#define NF_DET_ENTER_ALL_K(p)                                 \
    const bool nf_det_ = nf_det_on(p##_det);                  \
    if (nf_det_) {                                            \
        if (threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) nf_det_wait_k(p##_det, p##_det_turns, nf_det_key_, nf_det_rank_); \
        __syncthreads();                                      \
    }
#define NF_DET_LEAVE_ALL_K(p)                                 \
    if (nf_det_) {                                            \
        __threadfence();                                      \
        __syncthreads();                                      \
        if (threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) nf_det_pass_k(p##_det_turns, nf_det_key_, nf_det_rank_, nf_det_cnt_); \
    }
#define NF_DET_ENTER_WAVE_K(p)                    \
    const bool nf_det_ = nf_det_on(p##_det);      \
    if (nf_det_ && (threadIdx.x & 63) == 0) nf_det_wait_k(p##_det, p##_det_turns, nf_det_key_, nf_det_rank_)
#define NF_DET_LEAVE_WAVE_K(p)                                   \
    if (nf_det_) {                                               \
        __threadfence();                                         \
        if ((threadIdx.x & 63) == 0) nf_det_pass_k(p##_det_turns, nf_det_key_, nf_det_rank_, nf_det_cnt_); \
    }

// ---- last-workgroup fold (round 6) ----------------------------------------------------------------------------------------------------
// ONE thread of EVERY workgroup of the grid calls this exactly once with the workgroup's NV <= NF_DET_FOLD_NV partial sums: they go to a
// slab, the caller that arrives last adds the slab's rows in block order and adds the totals to dst[i].  Returns false (nothing done)
// when the grid is larger than the slab: the caller then takes the turnstile.  Launches of a translation unit are stream-ordered.
template <int NV>
__device__ __forceinline__ bool nf_det_fold_add(float* slab, unsigned* cnt, const float (&v)[NV], float* const (&dst)[NV]) {
    static_assert(NV <= NF_DET_FOLD_NV, "slab row");
    const unsigned me = nf_det_block(), n = nf_det_blocks();
    if (n > NF_DET_FOLD_MAX) return false;
#pragma unroll
    for (int i = 0; i < NV; ++i) __hip_atomic_store(slab + me * NF_DET_FOLD_NV + i, v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == n) {
        __threadfence();
        float t[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) t[i] = 0.f;
        for (unsigned b = 0; b < n; ++b)
#pragma unroll
            for (int i = 0; i < NV; ++i) t[i] += __hip_atomic_load(slab + b * NF_DET_FOLD_NV + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int i = 0; i < NV; ++i) atomicAdd(dst[i], t[i]);
        __hip_atomic_store(cnt, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    return true;
}

// the commonest site: ONE thread of every workgroup of the grid adds the workgroup's pair of partial sums to two scalars (a coupling's
// scale / shift gradients).  Racing: two atomics.  Ordered: the last-workgroup fold, the grid-wide turnstile beyond its slab.
#define NF_DET_ADD2(p, d0, v0, d1, v1)                                                         \
    do {                                                                                       \
        if (nf_det_on(p##_det)) {                                                              \
            const float nf_v_[2] = {(v0), (v1)};                                               \
            float* const nf_d_[2] = {(d0), (d1)};                                              \
            if (!nf_det_fold_add<2>(p##_det_slab, p##_det_cnt, nf_v_, nf_d_)) {                \
                nf_det_wait(p##_det);                                                          \
                atomicAdd((d0), (v0));                                                         \
                atomicAdd((d1), (v1));                                                         \
                nf_det_pass(p##_det);                                                          \
            }                                                                                  \
        } else {                                                                               \
            atomicAdd((d0), (v0));                                                             \
            atomicAdd((d1), (v1));                                                             \
        }                                                                                      \
    } while (0)

// form (1) on a COLUMN chain with the grid-wide turnstile as the fallback when the launch has more columns than turn words: ONE thread of
// every workgroup of an (x, y, z) launch adds to addresses picked by (blockIdx.x, blockIdx.z) -- a sample's log-det, a channel's sums
#define NF_DET_ENTER_COL(p)                                                                                   \
    const bool nf_det_ = nf_det_on(p##_det);                                                                  \
    const bool nf_det_col_ = gridDim.x * gridDim.z <= NF_DET_KEYS;                                            \
    if (nf_det_) {                                                                                            \
        if (nf_det_col_) nf_det_wait_k(p##_det, p##_det_turns, blockIdx.z * gridDim.x + blockIdx.x, blockIdx.y); \
        else nf_det_wait(p##_det);                                                                            \
    }
#define NF_DET_LEAVE_COL(p)                                                                                   \
    if (nf_det_) {                                                                                            \
        if (nf_det_col_) nf_det_pass_k(p##_det_turns, blockIdx.z * gridDim.x + blockIdx.x, blockIdx.y, gridDim.y); \
        else nf_det_pass(p##_det);                                                                            \
    }
// ... and ONE scalar of a whole grid (the loss): the last-workgroup fold, the turnstile beyond its slab
#define NF_DET_ADD1(p, d0, v0)                                                                 \
    do {                                                                                       \
        if (nf_det_on(p##_det)) {                                                              \
            const float nf_v_[1] = {(v0)};                                                     \
            float* const nf_d_[1] = {(d0)};                                                    \
            if (!nf_det_fold_add<1>(p##_det_slab, p##_det_cnt, nf_v_, nf_d_)) {                \
                nf_det_wait(p##_det);                                                          \
                atomicAdd((d0), (v0));                                                         \
                nf_det_pass(p##_det);                                                          \
            }                                                                                  \
        } else {                                                                               \
            atomicAdd((d0), (v0));                                                             \
        }                                                                                      \
    } while (0)
