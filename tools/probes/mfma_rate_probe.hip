// v_mfma_f32_32x32x2_f32 issue rate of ONE wave per SIMD in a short kernel (what the latency-bound conditioner launches see):
// a dependent chain on one accumulator vs. two / four independent accumulators; s_memtime (100 MHz) and s_memrealtime.
//   hipcc --offload-arch=gfx950 -O3 -o normalizing-flows-pytorch_amd/build/mfma_rate_probe tools/probes/mfma_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) k_chain(float* out, long long* stamps, int n) {
    f32x16 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float x = threadIdx.x * 0.001f, y = 1.f + threadIdx.x * 0.002f;
    const long long t0 = wall_clock64();
    const long long c0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
    }
    const long long c1 = clock64();
    const long long t1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = c1 - c0; }
}

template <int NACC>
static void run(int blocks, int n, float* out, long long* stamps) {
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_chain<NACC>, dim3(blocks), dim3(256), 0, 0, out, stamps, n);
        hipDeviceSynchronize();
    }
    long long h[2];
    hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost);
    const double ns = h[0] * 10.0;
    printf("blocks %4d  acc %d  mfma/wave %5d : %8.1f ns  -> %6.1f ns per MFMA ; clock64 ticks per MFMA %.1f\n", blocks, NACC,
           n * NACC, ns, ns / (n * NACC), (double)h[1] / (n * NACC));
}

int main() {
    float* out; long long* stamps;
    hipMalloc(&out, 1024 * 256 * sizeof(float));
    hipMalloc(&stamps, 64);
    for (int blocks : {32, 128, 1024}) {
        run<1>(blocks, 144, out, stamps);
        run<2>(blocks, 72, out, stamps);
        run<4>(blocks, 36, out, stamps);
        run<1>(blocks, 4000, out, stamps);
        run<4>(blocks, 1000, out, stamps);
    }
    return 0;
}
