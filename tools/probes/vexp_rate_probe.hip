// Chip-wide issue rate of the transcendental VALU instructions the mixture-of-logistics kernels are made of (v_exp_f32, v_log_f32,
// v_rcp_f32) next to v_fma_f32, eight independent chains per lane, every SIMD loaded with eight waves: the "v_exp roof" that
// bench.py / tools/kernel_sweep.py quote for those kernels (NF_TRANS_PEAK) is the number this prints on the box.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/vexp_rate_probe tools/probes/vexp_rate_probe.hip && /tmp/vexp_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ void __launch_bounds__(256) k_rate(float* out, int n) {
    float v[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) v[a] = 0.5f + 0.001f * (threadIdx.x + a);
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            if (OP == 0) v[a] = __builtin_amdgcn_exp2f(v[a]) * 0.25f;          // (the multiply keeps the chain bounded; counted below)
            if (OP == 1) v[a] = __builtin_amdgcn_logf(v[a]) + 3.0f;
            if (OP == 2) v[a] = __builtin_amdgcn_rcpf(v[a]) + 0.5f;
            if (OP == 3) v[a] = __builtin_fmaf(v[a], 0.999f, 0.001f);
            if (OP == 4) v[a] = v[a] * 0.25f + 0.0f * v[a];                     // the companion op alone (two plain VALU)
        }
    }
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 8; ++a) s += v[a];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
static double run(const char* name, float* out, int blocks, int n) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, out, n);
    hipEventRecord(a, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, out, n);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    const double ops = 5.0 * blocks * 256.0 * 8.0 * n;
    const double rate = ops / (ms * 1e-3);
    printf("%-28s blocks %5d : %8.3f ms  -> %8.2f T lane-iterations / s chip-wide (%5.2f per clock per SIMD at 2.4 GHz)\n", name, blocks,
           ms / 5, rate * 1e-12, rate / (1024.0 * 2.4e9));
    return rate;
}

int main() {
    float* out;
    hipMalloc(&out, 8192 * 256 * sizeof(float));
    const int n = 4096;
    for (int blocks : {2048, 8192}) {
        run<3>("v_fma_f32", out, blocks, n);
        run<4>("v_mul + v_fma (companion)", out, blocks, n);
        run<0>("v_exp_f32 + v_mul", out, blocks, n);
        run<1>("v_log_f32 + v_add", out, blocks, n);
        run<2>("v_rcp_f32 + v_add", out, blocks, n);
    }
    return 0;
}
