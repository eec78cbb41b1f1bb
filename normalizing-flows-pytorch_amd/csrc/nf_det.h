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
#define NF_DET_STATE(p)             \
    __device__ unsigned p##_det[3]; \
    static int p##_det_host = 0;    /* host mirror of [0]: launchers that change their launch shape in the mode read it */
#define NF_DET_HOST_API(p)                                                                       \
    __attribute__((visibility("hidden"))) int p##_det_set(int on) {                              \
        const unsigned w[3] = {on ? 1u : 0u, 0u, 0u};                                            \
        p##_det_host = on ? 1 : 0;                                                               \
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
