// Latency of an all-to-all slot exchange (the statistics exchange of csrc/conv_chain.hip in miniature) among G workgroups
//   (a) dealt over all XCDs (blocks 0 .. G-1), agent-scope relaxed atomics: what the chain kernels do today;
//   (b) all on ONE XCD (blocks 8 k of a grid of 8 G), agent-scope relaxed atomics;
//   (c) all on ONE XCD, workgroup-scope atomics (sc0: served by the XCD's L2, no trip to the memory side).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/xcd_probe tools/probes/xcd_exchange_probe.hip && /tmp/xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define XCC_REG ((3 << 11) | 20)
#define SPIN_LIMIT (1u << 13)

template <int SCOPE>
__device__ __forceinline__ void put(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, SCOPE); }
template <int SCOPE>
__device__ __forceinline__ unsigned long long get(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, SCOPE); }

// slots[round & 1][G][64]; every workgroup publishes 64 values per round and gathers all G x 64
template <int SCOPE>
__global__ void __launch_bounds__(256) k_exchange(unsigned long long* slots, int G, int stride, int rounds, long long* out, unsigned* bad, float* sink) {
    if (blockIdx.x % stride != 0) return;
    const int me = blockIdx.x / stride;
    if (me >= G) return;
    __shared__ float xs[128 * 65];
    __shared__ unsigned sflag;
    const unsigned xcc = __builtin_amdgcn_s_getreg(XCC_REG) & 15u;
    if (threadIdx.x == 0) out[8 + me] = xcc;
    float acc = 0.f;
    long long t0 = 0;
    for (int r = 0; r < rounds + 8; ++r) {
        if (r == 8) { __syncthreads(); t0 = wall_clock64(); }
        if ((r & 15) == 15) {                       // somebody gave up: everybody leaves (uniform per workgroup)
            if (threadIdx.x == 0) sflag = __hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (sflag != 0u) break;
        }
        unsigned long long* rs = slots + (size_t)(r & 3) * 128 * 64;
        const unsigned gen = (unsigned)(r + 1);
        if (threadIdx.x < 64) put<SCOPE>(rs + me * 64 + threadIdx.x, ((unsigned long long)gen << 32) | (unsigned)__float_as_uint((float)(me + r)));
        for (int e = threadIdx.x; e < G * 64; e += 256) {
            unsigned long long v;
            unsigned spins = 0;
            while ((unsigned)((v = get<SCOPE>(rs + e)) >> 32) != gen) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LIMIT) { atomicAdd(bad, 1u); break; }
            }
            xs[(e >> 6) * 65 + (e & 63)] = __uint_as_float((unsigned)v);
        }
        __syncthreads();
        for (int b = threadIdx.x & 31; b < G; b += 32) acc += xs[b * 65 + (threadIdx.x >> 2)];
        __syncthreads();
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0 && me == 0) { out[0] = t1 - t0; }
    if (acc == 123.456f) sink[0] = acc;
}

int main() {
    unsigned long long* slots; long long* out; unsigned* bad; float* sink;
    hipMalloc(&slots, 4 * 128 * 64 * 8); hipMalloc(&out, 256 * 8); hipMalloc(&bad, 4); hipMalloc(&sink, 4);
    const int rounds = 400;
    auto run = [&](const char* name, int scope, int G, int stride) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(slots, 0, 4 * 128 * 64 * 8); hipMemset(out, 0, 256 * 8); hipMemset(bad, 0, 4);
            if (scope == 0) hipLaunchKernelGGL(k_exchange<__HIP_MEMORY_SCOPE_AGENT>, dim3(G * stride), dim3(256), 0, 0, slots, G, stride, rounds, out, bad, sink);
            else hipLaunchKernelGGL(k_exchange<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(G * stride), dim3(256), 0, 0, slots, G, stride, rounds, out, bad, sink);
            hipError_t e = hipDeviceSynchronize();
            std::vector<long long> h(256); unsigned hb = 0;
            hipMemcpy(h.data(), out, 256 * 8, hipMemcpyDeviceToHost); hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
            unsigned mask = 0;
            for (int i = 0; i < G; ++i) mask |= 1u << h[8 + i];
            if (rep == 2) printf("%-58s G %3d: %.2f us per exchange, XCD mask 0x%02x, gave up %u, %s\n", name, G, h[0] / 100.0 / rounds, mask, hb, hipGetErrorString(e));
        }
    };
    for (int G : {2, 8, 16, 32}) {
        run("(a) over all XCDs, agent-scope atomics", 0, G, 1);
        run("(b) one XCD, agent-scope atomics", 0, G, 8);
        run("(c) one XCD, workgroup-scope atomics (sc0, L2 of the XCD)", 1, G, 8);
    }
    run("(a) over all XCDs, agent-scope atomics", 0, 64, 1);
    run("(a) over all XCDs, agent-scope atomics", 0, 128, 1);
    return 0;
}
