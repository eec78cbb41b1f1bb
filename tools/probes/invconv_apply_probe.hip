// Where does the 64-pixel-block invertible-1x1 apply (C = 48, P = 64) lose its time?  Same kernel body as
// k_invconv_apply_mfma4 with the pieces switched off one at a time, over grid sizes.
//   MODE 0: full   1: no matrix instructions (stores the loaded values)   2: every 4th matrix instruction
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/_bin/invconv_apply_probe tools/probes/invconv_apply_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define RT 3
#define KQ 12
template <int MODE, int OCC>
__global__ void __launch_bounds__(256, OCC) k(const float* __restrict__ z, const float* __restrict__ M, float* __restrict__ y, int64_t B) {
    const int C = 48, P = 64;
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    float a[RT][KQ];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int q = 0; q < KQ; ++q) a[rt][q] = M[(16 * rt + li) * C + 4 * q + lk];
    const int64_t nblk = B;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    f32x4 bv[KQ], nx[KQ];
    auto fetch = [&](int64_t blk, f32x4* dst) {
        const bool ok = blk < nblk;
        const float* zb = z + (ok ? blk : 0) * C * P + 4 * li;
#pragma unroll
        for (int q = 0; q < KQ; ++q) dst[q] = ok ? *reinterpret_cast<const f32x4*>(zb + (int64_t)(4 * q + lk) * P) : (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    fetch(wave, nx);
    for (int64_t blk = wave; blk < nblk; blk += nwaves) {
        float* yb = y + blk * C * P + 4 * li;
#pragma unroll
        for (int q = 0; q < KQ; ++q) bv[q] = nx[q];
        fetch(blk + nwaves, nx);
        if (MODE == 1) {
#pragma unroll
            for (int q = 0; q < KQ; ++q) *reinterpret_cast<f32x4*>(yb + (int64_t)(4 * q + lk) * P) = bv[q] * a[0][q];
            continue;
        }
        f32x4 acc[4][RT];
#pragma unroll
        for (int pb = 0; pb < 4; ++pb)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[pb][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < KQ; ++q)
#pragma unroll
            for (int pb = 0; pb < 4; ++pb)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    if (MODE == 2 && (q & 3)) { acc[pb][rt][0] += a[rt][q] * bv[q][pb]; continue; }
                    acc[pb][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][q], bv[q][pb], acc[pb][rt], 0, 0, 0);
                }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<f32x4*>(yb + (int64_t)(16 * rt + 4 * lk + j) * P) = (f32x4){acc[0][rt][j], acc[1][rt][j], acc[2][rt][j], acc[3][rt][j]};
    }
}
template <int MODE, int OCC>
static void run(const char* what, const float* z, const float* M, float* y, int64_t B, int grid) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<MODE, OCC>), dim3(grid), dim3(256), 0, 0, z, M, y, B);
    hipEventRecord(e0);
    const int R = 20;
    for (int i = 0; i < R; ++i) hipLaunchKernelGGL((k<MODE, OCC>), dim3(grid), dim3(256), 0, 0, z, M, y, B);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1000.0 / R, gb = (double)B * 48 * 64 * 8 / 1e9;
    printf("%-28s grid %5d  %7.1f us  %7.1f GB/s\n", what, grid, us, gb / (us * 1e-6));
}
int main() {
    const int64_t B = 8192;
    float *z, *y, *M;
    hipMalloc(&z, B * 48 * 64 * 4); hipMalloc(&y, B * 48 * 64 * 4); hipMalloc(&M, 48 * 48 * 4);
    hipMemset(z, 0, B * 48 * 64 * 4); hipMemset(M, 0, 48 * 48 * 4);
    for (int grid : {256, 512, 1024, 2048}) {
        run<0, 1>("full occ1", z, M, y, B, grid);
        run<0, 2>("full occ2", z, M, y, B, grid);
        run<1, 2>("no mfma occ2", z, M, y, B, grid);
        run<1, 4>("no mfma occ4", z, M, y, B, grid);
        run<2, 2>("1/4 mfma occ2", z, M, y, B, grid);
    }
    return 0;
}
