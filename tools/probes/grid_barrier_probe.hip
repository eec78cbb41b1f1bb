// measures the latency of a software grid barrier (global atomic counter + spin) on MI355X, with and without a
// stat-accumulate (64 float atomics into 8 replicas) + read-back per round: the cost model of a persistent BatchNorm chain.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/grid_barrier_probe tools/probes/grid_barrier_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(counter, 1u);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 24)) break;     // never hang the box
        }
        __threadfence();
    }
    __syncthreads();
}

__global__ void k_barrier(unsigned* counter, int rounds) {
    for (int k = 0; k < rounds; ++k) grid_barrier(counter, (unsigned)(k + 1) * gridDim.x);
}

__global__ void k_barrier_stats(unsigned* counter, float* stats, float* sink, int rounds) {
    float acc = 0.f;
    for (int k = 0; k < rounds; ++k) {
        float* st = stats + (size_t)k * 8 * 64;
        if (threadIdx.x < 64) atomicAdd(st + (blockIdx.x & 7) * 64 + threadIdx.x, 1.0f + threadIdx.x);
        grid_barrier(counter, (unsigned)(k + 1) * gridDim.x);
        if (threadIdx.x < 64) {
            float s = 0.f;
            for (int r = 0; r < 8; ++r) s += __hip_atomic_load(st + r * 64 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc += s;
        }
    }
    if (threadIdx.x < 64) sink[blockIdx.x * 64 + threadIdx.x] = acc;
}

int main() {
    const int rounds = 200;
    unsigned* counter; float *stats, *sink;
    hipMalloc(&counter, 4); hipMalloc(&stats, (size_t)rounds * 8 * 64 * 4); hipMalloc(&sink, 1024 * 64 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int grids[] = {1, 8, 16, 32, 64, 128, 256, 512};
    for (int mode = 0; mode < 2; ++mode)
        for (int g : grids) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipMemset(counter, 0, 4); hipMemset(stats, 0, (size_t)rounds * 8 * 64 * 4);
                hipDeviceSynchronize();
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(k_barrier, dim3(g), dim3(256), 0, 0, counter, rounds);
                else hipLaunchKernelGGL(k_barrier_stats, dim3(g), dim3(256), 0, 0, counter, stats, sink, rounds);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            std::vector<float> h(64);
            hipMemcpy(h.data(), sink, 64 * 4, hipMemcpyDeviceToHost);
            printf("%s grid %4d: %.2f us per round   (check %.0f)\n", mode ? "barrier+stats" : "barrier      ", g,
                   best * 1e3f / rounds, mode ? h[0] : 0.f);
        }
    return 0;
}
