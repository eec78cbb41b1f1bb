// cost of folding per-workgroup weight-gradient partials with float atomics instead of a grid barrier + cross-workgroup sum:
// G workgroups of 512 threads each add NPER values per thread into the SAME n addresses (n = 512 * NPER), one kernel.
//   hipcc --offload-arch=gfx950 -O3 -o normalizing-flows-pytorch_amd/build/atomic_fold_probe tools/probes/atomic_fold_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NPER>
__global__ void __launch_bounds__(512) k_fold(float* dst, const float* src, long long* stamps) {
    const long long t0 = wall_clock64();
    float v[NPER];
#pragma unroll
    for (int u = 0; u < NPER; ++u) v[u] = src[(blockIdx.x * NPER + u) * 512 + threadIdx.x];
#pragma unroll
    for (int u = 0; u < NPER; ++u) atomicAdd(dst + u * 512 + threadIdx.x, v[u]);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[0] = wall_clock64() - t0;
}

int main() {
    float *dst, *src; long long* stamps;
    hipMalloc(&dst, 512 * 16 * 4); hipMalloc(&src, 256 * 16 * 512 * 4); hipMalloc(&stamps, 64);
    hipMemset(dst, 0, 512 * 16 * 4); hipMemset(src, 0, 256 * 16 * 512 * 4);
    for (int G : {2, 8, 32, 128}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_fold<13>, dim3(G), dim3(512), 0, 0, dst, src, stamps);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(k_fold<13>, dim3(G), dim3(512), 0, 0, dst, src, stamps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h; hipMemcpy(&h, stamps, 8, hipMemcpyDeviceToHost);
        printf("G = %3d: kernel %.2f us per launch (incl. launch), block 0 in-kernel %.2f us for 13 atomics/thread (6656 addresses)\n", G,
               ms * 1000.f / 20, h / 100.0);
    }
    return 0;
}
