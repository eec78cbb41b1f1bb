// Probe for the 3-way bf16 split of fp32 GEMM operands on the gfx950 matrix cores:
//   1. operand / result layout of v_mfma_f32_32x32x16_bf16 (checked against a float64 host product),
//   2. accuracy of the six-product form  ah*bh + (ah*bm + am*bh) + (ah*bl + am*bm + al*bh)  against the exact fp32 MFMA
//      (v_mfma_f32_32x32x2_f32) and against float64, on K = 288 dot products of conditioner-like operands,
//   3. time of one 32 -> 32 channel 3 x 3 layer's K loop per 128-pixel workgroup tile (16 waves = 4 pixel blocks x 4 K quarters,
//      operands from LDS as ds_read_b128), fp32 MFMA against the split form.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/bf16x3_probe tools/probes/bf16x3_probe.hip && /tmp/bf16x3_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    l = (__bf16)r2;
}

// D[32][32] = A[32][K] * B[K][32], K = 288, one wave; mode 0: fp32 MFMA, mode 1: bf16 x 3 (six products)
__global__ void k_gemm(const float* A, const float* B, float* D, int K, int mode) {
    const int lane = threadIdx.x, c32 = lane & 31, hs = lane >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (mode == 0) {
        for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[c32 * K + k + hs], B[(k + hs) * 32 + c32], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 16) {
            bf16x8 ah, am, al, bh, bm, bl;
            for (int j = 0; j < 8; ++j) {
                __bf16 h, m, l;
                split3(A[c32 * K + k + 8 * hs + j], h, m, l);
                ah[j] = h; am[j] = m; al[j] = l;
                split3(B[(k + 8 * hs + j) * 32 + c32], h, m, l);
                bh[j] = h; bm[j] = m; bl[j] = l;
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        }
    }
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * hs) * 32 + c32] = acc[r];
}

// timing: 1024 threads, wave (pb, kq); fp32: 9 groups x (2 ds_read_b128 + 4 MFMA 32x32x2); split: 5 steps x (6 ds_read_b128 + 6 MFMA 32x32x16)
template <int MODE>
__global__ void __launch_bounds__(1024) k_time(float* out, int layers) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, c32 = lane & 31, hs = lane >> 5, pb = wid & 3, kq = wid >> 2;
    for (int e = threadIdx.x; e < 30000; e += 1024) sm[e] = 1e-3f * (float)(e & 255);
    __syncthreads();
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int l = 0; l < layers; ++l) {
        if (MODE == 0) {
            const float* wb = sm + 16000 + hs * 132 + 4 * c32 + kq * 9 * 2 * 132;
            const float* fb = sm + hs * 804 + 4 * (pb * 40 + c32 + 11);
#pragma unroll
            for (int g = 0; g < 9; ++g) {
                const float4 a = *(const float4*)(wb + g * 264), b = *(const float4*)(fb + 4 * (g - 4) + (g & 3) * 1608);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
            }
        } else {
            // planes of 4 octets x CS x 8 bf16 (16 B per position): plane stride 4 * 201 * 4 floats
            const float* wb = sm + 16000 + hs * 128 + 4 * c32 + kq * 5 * 2 * 128;
            const float* fb = sm + hs * 804 + 4 * (pb * 40 + c32 + 11);
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                bf16x8 ah = *(const bf16x8*)(wb + s * 256), am = *(const bf16x8*)(wb + s * 256 + 3000), al = *(const bf16x8*)(wb + s * 256 + 6000);
                bf16x8 bh = *(const bf16x8*)(fb + 4 * (s - 2)), bm = *(const bf16x8*)(fb + 4 * (s - 2) + 3216), bl = *(const bf16x8*)(fb + 4 * (s - 2) + 6432);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
            }
        }
        __syncthreads();
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

int main() {
    const int K = 288;
    std::vector<float> A(32 * K), B(K * 32), D0(1024), D1(1024);
    srand(7);
    for (auto& v : A) v = 0.06f * ((float)rand() / RAND_MAX * 2.f - 1.f);
    for (auto& v : B) { float u = (float)rand() / RAND_MAX * 3.f - 1.f; v = u > 0.f ? u : 0.f; }     // ReLU-like activations
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k_gemm, dim3(1), dim3(64), 0, 0, dA, dB, dD, K, mode);
        hipMemcpy(mode ? D1.data() : D0.data(), dD, 4096, hipMemcpyDeviceToHost);
    }
    double e0 = 0, e1 = 0, e01 = 0, scale = 0, sabs = 0;
    for (int m = 0; m < 32; ++m)
        for (int n = 0; n < 32; ++n) {
            double ref = 0, ab = 0;
            for (int k = 0; k < K; ++k) { ref += (double)A[m * K + k] * B[k * 32 + n]; ab += fabs((double)A[m * K + k] * B[k * 32 + n]); }
            e0 = fmax(e0, fabs(D0[m * 32 + n] - ref)); e1 = fmax(e1, fabs(D1[m * 32 + n] - ref));
            e01 = fmax(e01, fabs((double)D0[m * 32 + n] - D1[m * 32 + n]));
            scale = fmax(scale, fabs(ref)); sabs = fmax(sabs, ab);
        }
    printf("K = %d: max |fp32 MFMA - f64| %.3e   max |bf16x3 - f64| %.3e   max |fp32 MFMA - bf16x3| %.3e   (max |D| %.3e, max sum|a||b| %.3e)\n",
           K, e0, e1, e01, scale, sabs);
    printf("layout %s\n", e1 < 1e-4 * scale ? "OK (A: row l%%32, k = 8 (l/32) + j; B: col l%%32; D as 32x32x2f32)" : "MISMATCH");
    float* dOut; hipMalloc(&dOut, 4096);
    hipFuncSetAttribute((const void*)k_time<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
    hipFuncSetAttribute((const void*)k_time<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 2; ++mode)
        for (int grid : {32, 128, 256}) {
            const int layers = 400;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a, 0);
                if (mode == 0) hipLaunchKernelGGL(k_time<0>, dim3(grid), dim3(1024), 150000, 0, dOut, layers);
                else hipLaunchKernelGGL(k_time<1>, dim3(grid), dim3(1024), 150000, 0, dOut, layers);
                hipEventRecord(b, 0); hipEventSynchronize(b);
            }
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("%s, %3d workgroups: %.3f us per layer K loop (+ barrier)\n", mode ? "bf16 x 3 (30 MFMA 32x32x16 per wave)" : "fp32      (36 MFMA 32x32x2  per wave)", grid, ms * 1e3 / layers);
        }
    return 0;
}
